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
for _ in range(30): net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 10)
print(min(net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 10) for _ in range(20)) * 1e3)
''' % ROOT
res = {d: [] for d in sys.argv[1:]}
for rnd in range(3):
    for d in sys.argv[1:]:
        env = dict(os.environ, YF_LIB_PATH=os.path.join(ROOT, "stm32h7-yolo_amd", d, "libyf_network.so"))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        res[d].append(float(out.stdout.strip().split("\n")[-1]) if out.returncode == 0 else float("nan"))
for d, v in res.items():
    print(f"{d:28s} best-of-20 per round (us): " + " ".join(f"{t:7.1f}" for t in v))
