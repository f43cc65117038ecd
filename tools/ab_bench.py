#!/usr/bin/env python3
"""Interleaved A/B timing of the shipped kernel vs an experimental (YF_EXP) build, one process, one device.
   usage: ab_bench.py [F NW [FX NWX]]   shipped shape F,NW (default 2 8) vs experimental shape FX,NWX (default = F,NW)"""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
a = [int(v) for v in sys.argv[1:]]
f, w = (a[0], a[1]) if len(a) >= 2 else (2, 8)
fx, wx = (a[2], a[3]) if len(a) >= 4 else (f, w)
n = int(os.environ.get("YF_N", "4096"))
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_a = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda"); d_b = torch.zeros_like(d_a)
res = {0: [], 200: []}
for rnd in range(12):
    for off in (0, 200):
        net.configure((f if off == 0 else fx) + off, w if off == 0 else wx)
        ms = net.time_device(d_in.data_ptr(), (d_a if off == 0 else d_b).data_ptr(), n, 10)
        if rnd >= 2: res[off].append(ms)
torch.cuda.synchronize()
same = bool(torch.equal(d_a, d_b))
a, b = np.array(res[0]), np.array(res[200])
print(f"shipped      F={f} NW={w}  median {np.median(a)*1e3:7.1f} us  min {a.min()*1e3:7.1f}")
print(f"experimental F={fx} NW={wx} median {np.median(b)*1e3:7.1f} us  min {b.min()*1e3:7.1f}   ratio exp/shipped {np.median(b)/np.median(a):.4f}  outputs identical: {same}")
