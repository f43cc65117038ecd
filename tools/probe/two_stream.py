#!/usr/bin/env python3
"""What launching consecutive 4096-frame batches on TWO alternating streams gives against one stream (network + decode, rotating HBM-resident inputs,
HIP events around K launches, bench.py's settle): on one stream a launch's ramp, drain and dispatch gap are serial (6.6 us of fixed cost per launch,
tools/probe/batch_rate.py); on two, the next batch's workgroups may start on CUs the previous launch has left.  DEV TOOL (a launch-policy what-if,
not a kernel change)."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
n, cap = 4096, 4
NS = int(os.environ.get("YF_PROBE_STREAMS", "2"))
net = yf.Network().init()
rng = np.random.default_rng(8)
ins = [torch.from_numpy(rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)).cuda() for _ in range(8)]
outs = [(torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda"), torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda"),
         torch.zeros((n,), dtype=torch.int32, device="cuda")) for _ in range(max(2, NS))]
streams = [torch.cuda.Stream() for _ in range(max(2, NS))]
torch.cuda.synchronize()
def run(k, ns):
    o = outs[k % len(outs)]; s = streams[k % ns]
    net.run_decode_device(ins[k % 8].data_ptr(), o[0].data_ptr(), n, o[1].data_ptr(), o[2].data_ptr(), cap, stream=s.cuda_stream)
def region(ns, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for s in streams: s.synchronize()
    t0 = time.perf_counter()
    e0.record(streams[0])
    for k in range(iters): run(k, ns)
    for s in streams: s.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6
t0 = time.perf_counter(); k = 0
while (time.perf_counter() - t0) * 1e3 < 80:
    for _ in range(8): run(k, 1); k += 1
    torch.cuda.synchronize()
res = {1: [], NS: []}
for rnd in range(6):
    for ns in (1, NS):
        res[ns].append(region(ns, 400))
a, b = float(np.median(res[1])), float(np.median(res[NS]))
print(f"one stream {a:.2f} us per 4096-frame step (wall clock over 400 launches), {NS} alternating streams {b:.2f} us ({100 * (b / a - 1):+.1f} %)   grid divisor {os.environ.get('YF_LAB_GRID_DIV', '1')}   {[round(x, 1) for x in res[1]]} {[round(x, 1) for x in res[NS]]}")
