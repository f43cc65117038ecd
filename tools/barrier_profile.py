#!/usr/bin/env python3
"""Stage timeline of the fused kernel.  Needs a library built with -DYF_BARPROF
(make -C stm32h7-yolo_amd/csrc OUT=../lib_prof EXTRA_HIPFLAGS=-DYF_BARPROF; YF_LIB_PATH=.../lib_prof/libyf_network.so):
every wave then stores the cycle counter on arrival at and on release from each __syncthreads() of its workgroup's
second group (steady state).  Prints, per barrier interval, how long the waves worked (mean / slowest wave = the
interval's critical path) and how long they waited.  DEV TOOL."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
from tools.stage_profile_names import NAMES
n = int(os.environ.get("YF_N", "4096"))
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
if os.environ.get("YF_CFG"): net.configure(*[int(v) for v in os.environ["YF_CFG"].split(",")])
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
wgs, nw = 512, 8
d_prof = torch.zeros((wgs * nw * 80,), dtype=torch.int64, device="cuda")
for _ in range(3):
    net.run_device(d_in.data_ptr(), d_out.data_ptr(), n, None, d_prof.data_ptr())
torch.cuda.synchronize()
p = d_prof.cpu().numpy().reshape(wgs, nw, 40, 2).astype(np.float64)
nb = int((p[0, 0, :, 0] > 0).sum())
p = p[:, :, :nb, :]
arrive, leave = p[..., 0], p[..., 1]
body = np.empty_like(arrive); body[:, :, 0] = 0; body[:, :, 1:] = arrive[:, :, 1:] - leave[:, :, :-1]
wait = leave - arrive
span = leave[:, :, -1] - leave[:, :, 0]
print(f"{wgs} workgroups x {nw} waves, {nb} barriers per group; cycles from the first to the last barrier of a group: "
      f"mean {span.mean():.0f} (min {span.min():.0f} max {span.max():.0f})")
print(f"working {100 * body.sum(axis=2).mean() / span.mean():.1f}%  waiting in barriers {100 * wait[:, :, 1:].sum(axis=2).mean() / span.mean():.1f}%")
# the production kernel (round 4) runs pool_8's passes beside conv2d_10 / conv2d_13: 26 barriers per group; a -DYF_POOL_MERGE=0 build has the staged order (28)
if nb == 26:
    labels = ["top of loop"] + list(NAMES[:5]) + ["pool_8 h | conv2d_10 (dw)", "conv2d_12", "pool_8 v | conv2d_13"] + list(NAMES[9:])
else:
    labels = ["top of loop"] + [f"{nm}" for nm in NAMES[:6]] + ["pool_8 v", "conv2d_10 (dw)"] + [f"{nm}" for nm in NAMES[7:]]
# tail batching: the profiled (second) group of a workgroup fetches the parked group's T15 behind conv2d_23 and then runs the
# tail for four frames; its interval times are per PAIR of groups
k23 = next(i for i, nm in enumerate(labels) if nm.startswith("conv2d_23"))
labels = labels[:k23 + 1] + [nm + "   [4 frames]" for nm in labels[k23 + 1:]]    # the parked T15 returns by LDS-DMA during conv2d_23
print(f"{'interval ending at barrier':44s} {'work mean':>10s} {'slowest':>9s} {'wait mean':>10s}   work per wave")
for i in range(1, nb):
    b, w = body[:, :, i], wait[:, :, i]
    print(f"  {i:2d} {labels[i] if i < len(labels) else '':40s} {b.mean():10.0f} {b.max(axis=1).mean():9.0f} {w.mean():10.0f}   "
          + " ".join(f"{b[:, k].mean():6.0f}" for k in range(nw)))
