#!/usr/bin/env python3
"""Rate of the fused int8 kernel (network + decode, one launch) against the batch size per launch: 4096 / 8192 / 16384 / 32768 frames through
run_decode_device, inputs rotating over > 256 MB (HBM-resident, as in bench.py), HIP events around K launches after bench.py's clock settle.
A straight line t(n) = a + b n through the four points puts a number on the per-launch FIXED cost a (prologue, ramp of the first groups, drain of the
last ones) -- VERDICT round 4, item 3(a).  DEV TOOL.   usage: batch_rate.py [rounds]"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cap = 4
net = yf.Network().init()
rng = np.random.default_rng(7)
pool = torch.from_numpy(rng.integers(-128, 128, (65536, 56, 56, 3), dtype=np.int8)).cuda()      # 617 MB: every size rotates through all of it
res = {}
for n in (4096, 8192, 16384, 32768):
    nb = 65536 // n
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    d_dets = torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda"); d_cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
    k = 0
    def run():
        global k
        x = pool[(k % nb) * n:(k % nb + 1) * n]; k += 1
        net.run_decode_device(x.data_ptr(), d_out.data_ptr(), n, d_dets.data_ptr(), d_cnt.data_ptr(), cap)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < 60:
        for _ in range(4): run()
        torch.cuda.synchronize()
    iters = max(8, 200 * 4096 // n)
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    res[n] = float(np.median(ts))
    print(f"n {n:6d}: {res[n]:9.2f} us per launch  = {res[n] * 4096 / n:7.2f} us per 4096 frames  ({n / res[n]:.2f} M images/s)   {[round(t, 1) for t in ts]}", flush=True)
ns = np.array(sorted(res)); t = np.array([res[n] for n in ns])
b, a = np.polyfit(ns, t, 1)
print(f"fit t(n) = {a:.2f} us + {b * 4096:.2f} us per 4096 frames: fixed cost per launch {a:.2f} us = {100 * a / res[4096]:.1f} % of a 4096-frame launch")
