#!/usr/bin/env python3
"""Where do the waves wait?  Needs a library built with -DYF_BARPROF (make ... HIPFLAGS+=' -DYF_BARPROF'): the production
kernel then records, per wave, the cycles spent inside every __syncthreads() of the group loop.  DEV TOOL."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
from tools.stage_profile_names import NAMES
n = int(os.environ.get("YF_N", "4096"))
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
wgs, nw = min(512, n // 2), 8
d_prof = torch.zeros((wgs * nw * 41,), dtype=torch.int64, device="cuda")
for _ in range(3):
    net.run_device(d_in.data_ptr(), d_out.data_ptr(), n, None, d_prof.data_ptr())
torch.cuda.synchronize()
p = d_prof.cpu().numpy().reshape(wgs, nw, 41).astype(np.float64)
tot = p[:, :, 0]
waits = p[:, :, 1:]
print(f"{wgs} workgroups x {nw} waves, {n} frames: loop cycles per wave mean {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f})")
print(f"time inside barriers: {100 * waits.sum(axis=2).mean() / tot.mean():.1f}% of the loop (mean over waves)")
labels = ["top of loop (arena free)"] + [f"after {nm}" for nm in NAMES[:6]] + ["after pool_8 v", "after conv2d_10 (dw)"] + [f"after {nm}" for nm in NAMES[7:]]
for i in range(40):
    w = waits[:, :, i]
    if w.sum() == 0: continue
    print(f"  barrier {i:2d} {labels[i] if i < len(labels) else '':34s} mean {100 * w.mean() / tot.mean():5.2f}%   per wave: " + " ".join(f"{100 * w[:, k].mean() / tot.mean():4.1f}" for k in range(nw)))
