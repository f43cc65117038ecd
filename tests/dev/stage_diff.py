#!/usr/bin/env python3
"""Which stage tensors of the debug (dump) build differ from the oracle's per-op outputs?  Prints every stage, does not stop at the first.  DEV TOOL."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
yf = importlib.import_module("stm32h7-yolo_amd")
from oracle.np_restatement import load_yfm
from oracle.oracle import Oracle
from test_gpu_parity import STAGES
m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
sizes = [int(np.prod(m["tensors"][o["out"]]["shape"][1:])) for o in m["ops"]]
shapes = [m["tensors"][o["out"]]["shape"][1:] for o in m["ops"]]
offs = np.concatenate([[0], np.cumsum(sizes)])
n = 6
x = np.random.default_rng(42).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
orc = Oracle()
head_ref, dump_ref = orc.run(x, dump=True)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
d_dump = torch.zeros((n, net.dump_bytes()), dtype=torch.int8, device="cuda")
net.run_device(d_in.data_ptr(), d_out.data_ptr(), n, None, d_dump.data_ptr()); torch.cuda.synchronize()
dump = d_dump.cpu().numpy(); off = 0
for name, op in STAGES:
    got = dump[:, off:off + sizes[op]].reshape([n] + list(shapes[op])); ref = dump_ref[:, offs[op]:offs[op] + sizes[op]].reshape(got.shape)
    bad = got != ref
    msg = "ok" if not bad.any() else f"{bad.sum()} of {bad.size} differ; per frame {bad.reshape(n, -1).sum(axis=1)}; rows {bad.sum(axis=(0, 2, 3))}; cols {bad.sum(axis=(0, 1, 3))}; channels {bad.sum(axis=(0, 1, 2))}"
    print(f"{name:8s} op {op:2d} {msg}")
    off += sizes[op]
print("heads equal:", np.array_equal(d_out.cpu().numpy(), head_ref))
