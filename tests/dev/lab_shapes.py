#!/usr/bin/env python3
"""The fused kernel shapes that only the lab library carries (make lab -> lib_lab/: <1,4>, <2,4>, <4,8>) against the CPU oracle: ragged batch sizes
around the resident grid (workgroups with 1, 2, 3 ... groups; paired and unpaired tails).  Test helper: tests/test_gpu_parity.py runs it in a fresh
process with YF_LIB_PATH pointing at the lab library."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
yf = importlib.import_module("stm32h7-yolo_amd")
block = np.random.default_rng(31).integers(-128, 128, (256, 56, 56, 3), dtype=np.int8)
ref = Oracle().run(block, threads=16)
net = yf.Network(device=0).init()
rng = np.random.default_rng(32)
bad = 0
for shape in ((1, 4), (2, 4), (4, 8)):
    net.configure(*shape)
    for n in (1, 7, 130, 511, 1025, 2049, 3073, 5001):
        pick = rng.integers(0, 256, n)
        d_in = torch.from_numpy(block[pick]).cuda()
        d_out = torch.full((n + 1, 7, 7, 18), 55, dtype=torch.int8, device="cuda")
        torch.cuda.synchronize()
        net.run_device(d_in.data_ptr(), d_out.data_ptr(), n)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        ok = np.array_equal(got[:n], ref[pick]) and (got[n] == 55).all()
        bad += not ok
        print(net.kernel_name, "n", n, "ok" if ok else "MISMATCH", flush=True)
print("lab shapes ok" if not bad else f"lab shapes: {bad} mismatches")
sys.exit(1 if bad else 0)
