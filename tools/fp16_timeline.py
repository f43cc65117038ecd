#!/usr/bin/env python3
"""Stage timeline of the fused fp16 kernel.  Needs a library built with -DYF16_BARPROF
(make -C stm32h7-yolo_amd/csrc OUT=../lib_f16prof EXTRA_FP16FLAGS=-DYF16_BARPROF; YF_LIB_PATH=.../lib_f16prof/libyf_network.so):
in its second frame (the one that closes a pair and runs the 7x7 tail on two frames) every wave of a workgroup stores the cycle counter
on arrival at and on release from each barrier.  Prints, per barrier interval, the waves' working time (mean / slowest) and their
wait.  DEV TOOL."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
n = 4096
x = (np.random.default_rng(3).integers(0, 256, (n, 56, 56, 3)).astype(np.float32) / 255.0).astype(np.float16)
net = yf.Network().init(); net.fp16_init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.float32, device="cuda")
for _ in range(20): net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n)
torch.cuda.synchronize()
path = "/tmp/yf16_prof.bin"
os.environ["YF16_PROF_OUT"] = path
net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n); torch.cuda.synchronize()
del os.environ["YF16_PROF_OUT"]
wgs, nw = (256 if os.environ.get('YF16_ONE_WG_PER_CU') else 512), 8
p = np.fromfile(path, np.int64).reshape(wgs, nw, 40, 2).astype(np.float64)
arrive, leave = p[..., 0], p[..., 1]
# entries 0..11: the barriers of the workgroup's second frame (front stages); 20, 21: the two barriers of the tail phase of its first batch
names = ["input staging + halo fills", "conv2d_1", "conv2d_3 (dw)", "conv2d_5 -> conv2d_6", "pool_8 h | conv2d_10 (dw)",
         "pool_8 v | conv2d_12", "conv2d_13", "conv2d_15 (dw)", "conv2d_17+add", "conv2d_19", "conv2d_23", "pool_25 | conv2d_27 (dw) -> park slot"]
NF = len(names)
front = leave[:, :, NF - 1] - leave[:, :, 0]
print(f"{wgs} workgroups x {nw} waves; front stages of one frame (first to last barrier, without the staging stage): mean {front.mean():.0f} cycles "
      f"(min {front.min():.0f} max {front.max():.0f})")
print(f"{'interval ending at barrier':44s} {'work mean':>10s} {'slowest':>9s} {'wait mean':>10s}   work per wave")
tw = tb = 0.0
for i in range(1, NF):
    b, w = arrive[:, :, i] - leave[:, :, i - 1], leave[:, :, i] - arrive[:, :, i]
    tw += b.mean(); tb += w.mean()
    print(f"  {i:2d} {names[i]:40s} {b.mean():10.0f} {b.max(axis=1).mean():9.0f} {w.mean():10.0f}   " + " ".join(f"{b[:, k].mean():6.0f}" for k in range(nw)))
print(f"front: working {100 * tw / (tw + tb):.1f}%  waiting in barriers {100 * tb / (tw + tb):.1f}%")
b, w = arrive[:, :, 21] - leave[:, :, 20], leave[:, :, 21] - arrive[:, :, 21]
print(f"tail phase (one wave per frame, {nw} frames): chain mean {b.mean():.0f} slowest {b.max(axis=1).mean():.0f} wait {w.mean():.0f} cycles;  per wave " + " ".join(f"{b[:, k].mean():6.0f}" for k in range(nw)))
