#!/usr/bin/env python3
"""Where do the heads of the experimental (YF_EXP) build differ from the shipped kernel's?  DEV TOOL.
   usage: YF_LIB_PATH=.../lib_x<mask>/libyf_network.so python3 tools/exp_diff.py [n]"""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda()
outs = []
for off in (0, 200):
    net.configure(2 + off, 8)
    d = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    net.run_device(d_in.data_ptr(), d.data_ptr(), n)
    torch.cuda.synchronize()
    outs.append(d.cpu().numpy())
a, b = outs
bad = a != b
print(f"{n} frames: {bad.sum()} of {bad.size} head bytes differ; frames with a difference: {bad.reshape(n, -1).any(axis=1).sum()}")
if bad.any():
    fr = np.nonzero(bad.reshape(n, -1).any(axis=1))[0]
    print("first differing frames:", fr[:16], " parity of frame index among differing:", np.bincount(fr % 2, minlength=2))
    print("differences per head row   :", bad.sum(axis=(0, 2, 3)))
    print("differences per head column:", bad.sum(axis=(0, 1, 3)))
    print("differences per channel    :", bad.sum(axis=(0, 1, 2)))
    d = (a.astype(int) - b.astype(int))[bad]
    print("magnitude of differences: mean |d| %.2f, max %d" % (np.abs(d).mean(), np.abs(d).max()))
