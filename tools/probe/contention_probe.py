#!/usr/bin/env python3
"""What a small kernel on ANOTHER stream costs the fused int8 kernel (4096 frames per launch, back to back, rotating inputs): the stand-in for the RCCL
all-gather of a multi-GPU step, which cannot run with more than one rank on a one-GPU box.  Per step one spin kernel (tools/probe/contend.so) of W workgroups
x 512 threads for T microseconds is issued on a second stream.  DEV TOOL.   usage (through gpurun): python3 tools/probe/contention_probe.py"""
import ctypes, importlib, os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
yf = importlib.import_module("stm32h7-yolo_amd")
spin = ctypes.CDLL(os.path.join(root, "tools", "probe", "contend.so"))
spin.spin_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n, cap = 4096, 4
net = yf.Network().init()
rng = np.random.default_rng(5)
ins = [torch.from_numpy(rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)).cuda() for _ in range(8)]
d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
d_dets = torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda"); d_cnt = torch.zeros((n,), dtype=torch.int32, device="cuda")
main, side = torch.cuda.current_stream(), torch.cuda.Stream()
def steps(k, w, t, lds):
    for i in range(k):
        net.run_decode_device(ins[i % 8].data_ptr(), d_out.data_ptr(), n, d_dets.data_ptr(), d_cnt.data_ptr(), cap)
        if w: spin.spin_launch(w, 512, lds, t, side.cuda_stream)
def timed(w, t, lds, k=200):
    steps(50, w, t, lds); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main); steps(k, w, t, lds); e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
steps(400, 0, 0, 0); torch.cuda.synchronize()
base = timed(0, 0, 0)
print(f"no neighbour: {base:.1f} us per step")
for w, t, lds in ((1, 25, 0), (4, 25, 0), (8, 25, 0), (16, 25, 0), (32, 25, 0), (4, 25, 32768), (4, 100, 0), (16, 100, 0)):
    v = timed(w, t, lds)
    print(f"neighbour of {w:2d} workgroups x 512 threads, {lds // 1024:2d} KB LDS each, {t:3d} us per step: {v:.1f} us per step (+{100 * (v / base - 1):.1f} %)")
print(f"no neighbour again: {timed(0, 0, 0):.1f} us per step")
