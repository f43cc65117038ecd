#!/usr/bin/env python3
"""What the fused box decode costs: network-only launches against network + decode launches, interleaved, events around 50 launches,
   eight rotating input batches (HBM-resident inputs as in bench.py).  DEV TOOL."""
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
n, cap = 4096, 4
net = yf.Network().init()
rng = np.random.default_rng(5)
ins = [torch.from_numpy(rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)).cuda() for _ in range(8)]
d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
d_dets = torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda"); d_cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
def run(decode, k):
    x = ins[k % 8]
    if decode: net.run_decode_device(x.data_ptr(), d_out.data_ptr(), n, d_dets.data_ptr(), d_cnt.data_ptr(), cap)
    else: net.run_device(x.data_ptr(), d_out.data_ptr(), n)
import time
t0, k = time.perf_counter(), 0      # the same clock settle as bench.py's: untimed launches for 60 ms
while (time.perf_counter() - t0) * 1e3 < float(os.environ.get("YF_SETTLE_MS", "60")):
    for _ in range(8): run(k & 1, k); k += 1
    torch.cuda.synchronize()
res = {0: [], 1: []}
for rnd in range(10):
    for dec in (0, 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(50): run(dec, k)
        e1.record(); torch.cuda.synchronize()
        res[dec].append(e0.elapsed_time(e1) / 50 * 1e3)
a, b = np.median(res[0]), np.median(res[1])
print(f"network only {a:.1f} us per launch, network + fused decode {b:.1f} us  (+{100 * (b / a - 1):.1f} %)")
