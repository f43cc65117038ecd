#!/usr/bin/env python3
"""Does the kernel time depend on the input distribution (LDS LUT bank conflicts are data dependent)?  DEV TOOL."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
yf = importlib.import_module("stm32h7-yolo_amd")
n = 4096
rng = np.random.default_rng(1)
real = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
sets = {
    "uniform int8 noise": rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8),
    "real frames (27 tiled)": real[np.arange(n) % real.shape[0]],
    "constant 0": np.zeros((n, 56, 56, 3), np.int8),
    "gaussian sigma 20": np.clip(rng.normal(0, 20, (n, 56, 56, 3)), -128, 127).astype(np.int8),
    "smooth noise (8x8 blocks)": np.repeat(np.repeat(rng.integers(-128, 128, (n, 7, 7, 3), dtype=np.int8), 8, 1), 8, 2),
}
net = yf.Network().init()
d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
for name, x in sets.items():
    d_in = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 3)
    ms = min(net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 20) for _ in range(3))
    print(f"{name:28s} {ms*1e3:7.1f} us  {n/ms*1e3/1e6:6.2f} M frames/s")
