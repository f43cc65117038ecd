#!/usr/bin/env python3
"""Kernel time of tiny batches per compiled kernel shape (device buffers, events around 50 launches).  DEV TOOL."""
import importlib, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
net = yf.Network().init()
x = np.random.default_rng(1).integers(-128, 128, (64, 56, 56, 3), dtype=np.int8)
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((64, 7, 7, 18), dtype=torch.int8, device="cuda")
for f, w in ((1, 4), (2, 4), (2, 8), (4, 8), (1, 8)):
    try: net.configure(f, w)
    except Exception as e: print(f, w, "not compiled"); continue
    for n in (1, 2, 4, 16, 64):
        net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 20)
        ms = min(net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 50) for _ in range(5))
        print(f"{net.kernel_name}  n={n:3d}  {ms*1e3:7.1f} us per launch")
