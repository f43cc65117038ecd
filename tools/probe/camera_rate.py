#!/usr/bin/env python3
"""Rate of the camera path (112x112 RGB565 frames -> heads + boxes in one launch), events around 50 launches, eight rotating batches.  DEV TOOL."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
n, cap = 4096, 4
net = yf.Network().init()
rng = np.random.default_rng(5)
ins = [torch.from_numpy(rng.integers(0, 256, (n, 112 * 112 * 2), dtype=np.uint8)).cuda() for _ in range(8)]
d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
d_dets = torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda"); d_cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
run = lambda k: net.run_camera_device(ins[k % 8].data_ptr(), d_out.data_ptr(), n, d_dets.data_ptr(), d_cnt.data_ptr(), cap)
for k in range(400): run(k)
torch.cuda.synchronize()
res = []
for rnd in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(50): run(k)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50 * 1e3)
print(f"camera path: {np.median(res):.1f} us per {n} frames -> {n / np.median(res):.2f} M frames/s")
