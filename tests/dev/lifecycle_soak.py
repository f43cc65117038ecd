#!/usr/bin/env python3
"""create -> init -> run (device path on a side stream, host path, fp16, 160x160) -> destroy, a hundred times in one process: the reference's firmware calls
aiInit once, a host application may not.  Device memory must come back (free memory after cycle k within 64 MiB of free memory after cycle 3), host memory must not grow (resident set within 64 MiB of
cycle 10's), every head
must stay equal to the oracle's, ai_network_init on a live network (re-initialisation, network.c:3385-3399 allows it) must not leak either, and ai_network_destroy with a launch still in
flight on a caller's stream must let that launch finish (its heads are checked).
Test helper: tests/test_gpu_parity.py runs it in a fresh process."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
yf = importlib.import_module("stm32h7-yolo_amd")
x = np.random.default_rng(71).integers(-128, 128, (700, 56, 56, 3), dtype=np.int8)
ref = Oracle().run(x, threads=16)
x160 = np.random.default_rng(72).integers(-128, 128, (8, 160, 160, 3), dtype=np.int8)
d_in = torch.from_numpy(x).cuda()
d_160 = torch.from_numpy(x160).cuda()
d_f16 = torch.from_numpy((np.random.default_rng(73).integers(0, 256, (16, 56, 56, 3)) / 255.0).astype(np.float16)).cuda()
d_out = torch.zeros((700, 7, 7, 18), dtype=torch.int8, device="cuda")
d_o160 = torch.zeros((8, 20, 20, 18), dtype=torch.int8, device="cuda")
d_of = torch.zeros((16, 7, 7, 18), dtype=torch.float32, device="cuda")
side = torch.cuda.Stream()
torch.cuda.synchronize()
import psutil
proc = psutil.Process()
free, rss, bad = [], [], 0
for cycle in range(100):
    net = yf.Network(device=0).init()
    if cycle % 3 == 0:
        net.init()                                        # re-initialise a live network
    net.configure(2, 8)
    net.run_device(d_in.data_ptr(), d_out.data_ptr(), 700, side.cuda_stream)
    side.synchronize()
    bad += not np.array_equal(d_out.cpu().numpy(), ref)
    bad += not np.array_equal(net.run(x[:300]), ref[:300])         # host path, mid size
    bad += not np.array_equal(net.run(x[:3]), ref[:3])             # host path, zero-copy
    if cycle % 5 == 0:
        bad += not np.array_equal(net.run(np.tile(x, (4, 1, 1, 1))[:2500])[-700:], np.tile(ref, (4, 1, 1, 1))[:2500][-700:])   # host path, pipelined (starts the download thread)
    net.fp16_init()
    net.fp16_run_device(d_f16.data_ptr(), d_of.data_ptr(), 16)
    net.run_device_hw(160, 160, d_160.data_ptr(), d_o160.data_ptr(), 8)
    torch.cuda.synchronize()
    if cycle % 7 == 0:                                    # destroy with a launch still in flight: ai_network_destroy waits for the device before it frees anything
        d_out.zero_()
        net.run_device(d_in.data_ptr(), d_out.data_ptr(), 700, side.cuda_stream)
        net.destroy()
        side.synchronize()
        bad += not np.array_equal(d_out.cpu().numpy(), ref)
    else:
        net.destroy()
    free.append(torch.cuda.mem_get_info()[0])
    rss.append(proc.memory_info().rss)
drift = (free[3] - min(free[3:])) / 2 ** 20
print(f"100 cycles: {bad} mismatches; free device memory after cycle 3: {free[3] / 2**20:.0f} MiB, lowest afterwards {min(free[3:]) / 2**20:.0f} MiB (drift {drift:.1f} MiB)")
host = (max(rss[10:]) - rss[10]) / 2 ** 20
print(f"host resident memory after cycle 10: {rss[10] / 2**20:.0f} MiB, highest afterwards {max(rss[10:]) / 2**20:.0f} MiB (growth {host:.1f} MiB)")
ok = bad == 0 and drift <= 64 and host <= 64
print("lifecycle soak ok" if ok else "lifecycle soak FAILED")
sys.exit(0 if ok else 1)
