#!/usr/bin/env python3
"""Per-stage time profile of the fused kernel (debug build; YF_N=frames, e.g. 512 = one workgroup per CU): time(stop=k) - time(stop=k-1)."""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
NAMES = ["input staging", "conv2d_1", "conv2d_3 (dw)", "conv2d_5", "conv2d_6", "pool_8 h", "pool_8 v + conv2d_10 (dw)", "conv2d_12",
         "conv2d_13", "conv2d_15 (dw)", "conv2d_17+add", "conv2d_19", "conv2d_23", "pool_25 + conv2d_27 (dw)", "conv2d_29",
         "conv2d_30", "conv2d_32 (dw)", "conv2d_34+add", "conv2d_36", "conv2d_38 (dw)", "conv2d_40+add", "conv2d_42", "conv2d_47",
         "conv2d_49 (dw)", "conv2d_51", "conv2d_53 + store"]
n = int(os.environ.get("YF_N", "4096"))
x = np.random.default_rng(1).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
net = yf.Network().init()
if len(sys.argv) > 2: net.configure(int(sys.argv[1]), int(sys.argv[2]))
print("profiling debug build of", net.kernel_name)
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
net.time_stages(d_in.data_ptr(), d_out.data_ptr(), n, 3, 0)
prev = 0.0
tot = net.time_stages(d_in.data_ptr(), d_out.data_ptr(), n, 10, 0)
for k in range(1, 27):
    ms = net.time_stages(d_in.data_ptr(), d_out.data_ptr(), n, 10, k if k < 26 else 0)
    print(f"{k:2d} {NAMES[k-1]:28s} cum {ms*1e3:8.1f} us   stage {1e3*(ms-prev):7.1f} us  {100*(ms-prev)/tot:5.1f}%")
    prev = ms
