#!/usr/bin/env python3
"""160x160 parity of the banded path -- or, with the lab library (YF_LIB_PATH=.../lib_lab/libyf_network.so) and YF_160_LAYERWISE=1, of the layer-by-layer
   form -- against the CPU oracle.  Test helper (tests/test_gpu_parity.py runs it in a fresh process); runs on the GPU box."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
yf = importlib.import_module("stm32h7-yolo_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
x = np.random.default_rng(4).integers(-128, 128, (n, 160, 160, 3), dtype=np.int8)
x[1] = -128; x[2] = 127
ref = Oracle().run(x, threads=8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 20, 20, 18), dtype=torch.int8, device="cuda")
net.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), n)
torch.cuda.synchronize()
got = d_out.cpu().numpy()
bad = int((got != ref).sum())
print("160x160 head", "ok" if not bad else f"MISMATCH {bad}/{got.size}; first frame with mismatch {np.argwhere(got != ref)[0]}")
if not bad:
    nb = 1024
    xb = np.tile(x, (nb // n + 1, 1, 1, 1))[:nb]
    d_in = torch.from_numpy(xb).cuda(); d_out = torch.zeros((nb, 20, 20, 18), dtype=torch.int8, device="cuda")
    net.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), nb); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): net.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), nb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"batch {nb}: {dt*1e3:.2f} ms -> {nb/dt:.0f} frames/s; all copies equal: {bool((d_out.cpu().numpy()[:n] == ref).all())}")
sys.exit(1 if bad else 0)
