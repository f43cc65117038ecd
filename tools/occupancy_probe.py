#!/usr/bin/env python3
"""How does launch time scale with workgroups per CU?  n frames -> n/F groups; 256 CUs.  DEV TOOL."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
f, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 8)
N = 8192
x = np.random.default_rng(1).integers(-128, 128, (N, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
net.configure(f, w)
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((N, 7, 7, 18), dtype=torch.int8, device="cuda")
print("kernel", net.kernel_name)
for groups in (64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 4096):
    n = groups * f
    if n > N: break
    net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 3)
    ms = net.time_device(d_in.data_ptr(), d_out.data_ptr(), n, 20)
    print(f"groups {groups:5d} ({groups/256:5.2f} per CU)  {ms*1e3:8.1f} us   {n/ms*1e3/1e6:6.2f} M frames/s")
