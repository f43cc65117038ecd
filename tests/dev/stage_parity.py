#!/usr/bin/env python3
"""Per-stage parity of the fused kernel against the CPU oracle (debug tool; runs on the GPU box).

Uses the library's dump entry point (yf_network_run_device_dump) and compares every fused stage's tensor with the
matching tflite op output of the oracle.  Prints the first failing stage."""
import importlib
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402
from oracle.np_restatement import load_yfm  # noqa: E402

yf = importlib.import_module("stm32h7-yolo_amd")

STAGES = [("T1", 2), ("T2", 4), ("T3", 5), ("T4", 7), ("Q21", 21), ("T6", 11), ("T7", 12), ("T8", 14), ("T9", 16),
          ("T11", 18), ("T14", 22), ("T15", 24), ("Q45", 45), ("T17", 28), ("T18", 29), ("T19", 31), ("T20", 33),
          ("T22", 35), ("T23", 37), ("T24", 39), ("T26", 41), ("T30", 46), ("T31", 48), ("T32", 50), ("T33", 52)]


def op_offsets():
    m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    offs, sizes, off = [], [], 0
    for o in m["ops"]:
        s = m["tensors"][o["out"]]["shape"]
        n = s[1] * s[2] * s[3]
        offs.append(off); sizes.append(n); off += n
    return offs, sizes


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    rng = np.random.default_rng(0)
    x = rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
    orc = Oracle()
    head_ref, dump_ref = orc.run(x, dump=True)
    net = yf.Network().init()
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    db = net.dump_bytes()
    d_dump = torch.zeros((n, db), dtype=torch.int8, device="cuda")
    net.run_device(d_in.data_ptr(), d_out.data_ptr(), n, None, d_dump.data_ptr())
    torch.cuda.synchronize()
    dump = d_dump.cpu().numpy()
    head = d_out.cpu().numpy()
    offs, sizes = op_offsets()
    off = 0
    ok = True
    for name, op in STAGES:
        sz = sizes[op]
        got = dump[:, off:off + sz]
        ref = dump_ref[:, offs[op]:offs[op] + sz]
        bad = int((got != ref).sum())
        if bad:
            ok = False
            f, i = np.argwhere(got != ref)[0]
            print(f"{name:4s} (tfl op {op:2d}) MISMATCH {bad}/{got.size}  first: frame {f} elem {i} got {got[f, i]} ref {ref[f, i]}; "
                  f"max|d| {np.abs(got.astype(int) - ref.astype(int)).max()}")
        else:
            print(f"{name:4s} (tfl op {op:2d}) ok ({sz} B/frame)")
        off += sz
    bad = int((head != head_ref).sum())
    print("HEAD", "ok" if not bad else f"MISMATCH {bad}/{head.size}")
    # non-dump variants
    for f, w in ((1, 4), (2, 4), (2, 8), (4, 8), (202, 8)):
        net.configure(f, w)
        d_out.zero_()
        net.run_device(d_in.data_ptr(), d_out.data_ptr(), n)
        torch.cuda.synchronize()
        b = int((d_out.cpu().numpy() != head_ref).sum())
        print(net.kernel_name, "head", "ok" if not b else f"MISMATCH {b}")
        ok &= not b
    sys.exit(0 if ok and not bad else 1)


if __name__ == "__main__":
    main()
