#!/usr/bin/env python3
"""Per-stage parity of the fused kernel against the CPU oracle (debug tool; runs on the GPU box).

Uses the library's dump entry point (yf_network_run_device_dump) and compares every fused stage's tensor with the
matching tflite op output of the oracle.  Prints the first failing stage.

    stage_parity.py [n]                       the product library's debug build (staged order: what the per-node observer runs)
    stage_parity.py --prod-order [--ties-up] n [n ...]    (--ties-up: the second kernel set, against the oracle's variant of that rounding)
                                              the laboratory library's dump build that KEEPS THE PRODUCTION STAGE ORDER (yf_fused56.hip.h, YF_PDUMP:
                                              pools beside the branch on 3 + 5 waves, conv2d_10's output on concat_22's bytes, the 7x7 tail once per pair of
                                              groups on four frames through the HBM park), on a grid of ONE workgroup so that consecutive groups pair up:
                                              n = 5 -> a pair and an unpaired last group holding one frame; n = 8 -> two pairs; then once more on the full
                                              grid.  The 25 fused-stage tensors of every frame must equal the oracle's ops."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PROD_ORDER = "--prod-order" in sys.argv
VARIANT = 0                                     # oracle variant the dumps are compared with (--ties-up: rounding "ties upward on the dense convs" = the second kernel set)
if "--ties-up" in sys.argv:
    sys.argv.remove("--ties-up")
    os.environ["YF_REQUANT_ROUNDING"] = "ties_up"
    VARIANT = 1
if PROD_ORDER:                                  # before the library is loaded: the laboratory build and its switches (yf_engine.hip, YF_LAB)
    sys.argv.remove("--prod-order")
    os.environ["YF_LIB_PATH"] = os.path.join(ROOT, "stm32h7-yolo_amd", "lib_lab", "libyf_network.so")
    os.environ["YF_LAB_DUMP_PROD_ORDER"] = "1"
import numpy as np      # noqa: E402
import torch            # noqa: E402

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


def prod_order(sizes_n):
    """the laboratory's production-order dump build; one fresh engine per grid size (YF_LAB_GRID_DIV is read at ai_network_init)"""
    offs, sizes = op_offsets()
    orc = Oracle()
    ok = True
    for grid_div, label in ((1 << 20, "one workgroup (groups pair up)"), (1, "full grid")):
        os.environ["YF_LAB_GRID_DIV"] = str(grid_div)
        net = yf.Network().init()
        for n in sizes_n:
            x = np.random.default_rng(100 + n).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
            head_ref, dump_ref = orc.run(x, dump=True, threads=8, variant=VARIANT)
            d_in = torch.from_numpy(x).cuda()
            d_out = torch.full((n + 1, 7, 7, 18), 77, dtype=torch.int8, device="cuda")
            d_dump = torch.full((n + 1, net.dump_bytes()), 77, dtype=torch.int8, device="cuda")
            net.run_device(d_in.data_ptr(), d_out.data_ptr(), n, None, d_dump.data_ptr())
            torch.cuda.synchronize()
            dump, head = d_dump.cpu().numpy(), d_out.cpu().numpy()
            off, bad_stages = 0, []
            for name, op in STAGES:
                if not np.array_equal(dump[:n, off:off + sizes[op]], dump_ref[:, offs[op]:offs[op] + sizes[op]]):
                    bad_stages.append(name)
                off += sizes[op]
            good = not bad_stages and np.array_equal(head[:n], head_ref) and (head[n] == 77).all() and (dump[n] == 77).all()
            print(f"production order, {net.kernel_name}, {label}, n = {n}: {'25 stage tensors + head ok' if good else 'MISMATCH ' + str(bad_stages)}")
            ok &= bool(good)
        net.destroy()
    print("stage parity (production order) ok" if ok else "stage parity (production order) FAILED")
    sys.exit(0 if ok else 1)


def main():
    if PROD_ORDER:
        prod_order([int(a) for a in sys.argv[1:]] or [5, 3, 8])
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
