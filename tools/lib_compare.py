#!/usr/bin/env python3
"""Kernel time of differently compiled libraries on the same box: each library in its own process, rounds interleaved.
   usage: lib_compare.py <lib dir> [<lib dir> ...]   (directories under stm32h7-yolo_amd/)   DEV TOOL."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, %r)
yf = importlib.import_module("stm32h7-yolo_amd")
n = 4096
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
cap = 16
d_d = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda"); d_c = torch.zeros((n,), dtype=torch.int32, device="cuda")
for _ in range(30): net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 10)
t_plain = min(net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 10) for _ in range(20)) * 1e3
def with_decode(k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): net.run_decode_device(d_in.data_ptr(), d_out.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
with_decode(50)
print(t_plain, min(with_decode(20) for _ in range(10)))
''' % ROOT
res = {d: [] for d in sys.argv[1:]}
for rnd in range(3):
    for d in sys.argv[1:]:
        env = dict(os.environ, YF_LIB_PATH=os.path.join(ROOT, "stm32h7-yolo_amd", d, "libyf_network.so"))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        res[d].append(tuple(float(v) for v in out.stdout.strip().split("\n")[-1].split()) if out.returncode == 0 else (float("nan"),) * 2)
for d, v in res.items():
    print(f"{d:28s} us per launch, network only / network + fused decode, per round: " + "  ".join(f"{a:6.1f}/{b:6.1f}" for a, b in v))
