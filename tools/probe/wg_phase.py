#!/usr/bin/env python3
"""Are the two workgroups of a CU in phase?  Needs a -DYF_BARPROF library (YF_LIB_PATH): the barrier stamps of the profiled group are absolute
counter values, so the arrival times of workgroup i and of workgroup j at the same barrier can be compared.  Prints, for j = i + 256 (the
dispatcher's second round: the likely CU mate) and for j = i + 1, the spread of the time offsets at the first and the middle barrier.  DEV TOOL."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
n = 4096
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
wgs, nw = 512, 8
d_prof = torch.zeros((wgs * nw * 80,), dtype=torch.int64, device="cuda")
for _ in range(3): net.run_device(d_in.data_ptr(), d_out.data_ptr(), n, None, d_prof.data_ptr())
torch.cuda.synchronize()
p = d_prof.cpu().numpy().reshape(wgs, nw, 40, 2).astype(np.float64)
nb = int((p[0, 0, :, 0] > 0).sum())
arr = p[:, 0, :nb, 0]                       # wave 0's arrival at each barrier
span = (arr[:, -1] - arr[:, 0]).mean()
print(f"{nb} barriers, group span {span:.0f} cycles")
for name, shift in (("i + 256", 256), ("i + 1", 1), ("i + 8", 8), ("i + 64", 64)):
    a, b = arr[:wgs - shift], arr[shift:]
    for k in (0, nb // 2):
        d = b[:, k] - a[:, k]
        print(f"  partner {name:8s} barrier {k:2d}: offset mean {d.mean():9.0f}  median {np.median(d):9.0f}  |offset| median {np.median(np.abs(d)):8.0f}  as a fraction of the span {np.median(np.abs(d)) / span:.2f}")
