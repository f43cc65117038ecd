#!/usr/bin/env python3
"""Kernel times of the 160x160 path (1024 frames), events around 20 back-to-back batches after a clock settle.  DEV TOOL."""
import importlib, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
net = yf.Network().init()
n = 1024
d_in = torch.from_numpy(np.random.default_rng(4).integers(-128, 128, (n, 160, 160, 3), dtype=np.int8)).cuda()
d_out = torch.zeros((n, 20, 20, 18), dtype=torch.int8, device="cuda")
run = lambda: net.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.08:
    for _ in range(8): run()
    torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 20)
print(f"160x160: {best*1e3:.1f} us per {n} frames -> {n/best/1e3:.3f} M frames/s")
