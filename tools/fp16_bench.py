#!/usr/bin/env python3
"""Throughput of the fp16 side configuration (BASELINE configs[3]): 4096 frames through the fused fp16 kernel.  DEV TOOL.
   usage: fp16_bench.py [--dump out.npy]     (--dump: the logits of the first 512 frames and of a ragged 1027-frame run go to a file, for tools/fp16_ab.py)"""
import importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
n = 4096
x = (np.random.default_rng(3).integers(0, 256, (n, 56, 56, 3)).astype(np.float32) / 255.0).astype(np.float16)
net = yf.Network().init()
net.fp16_init()
d_in = torch.from_numpy(x).cuda(); d_out = torch.zeros((n, 7, 7, 18), dtype=torch.float32, device="cuda")
# the same clock settle as bench.py's (an idle GPU ramps its clock over ~10 ms: 60 launches = 8 ms would leave the timed loop inside the ramp)
SETTLE_MS = float(os.environ.get("YF_SETTLE_MS", "60"))
t0 = time.perf_counter()
while (time.perf_counter() - t0) * 1e3 < SETTLE_MS:
    for _ in range(8): net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n)
    torch.cuda.synchronize()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"fp16 configuration: {dt*1e3:.4f} ms per {n} frames -> {n/dt/1e6:.2f} M frames/s", flush=True)
if "--dump" in sys.argv:
    path = sys.argv[sys.argv.index("--dump") + 1]
    full = d_out.cpu().numpy()
    d_r = torch.zeros((1027, 7, 7, 18), dtype=torch.float32, device="cuda")
    net.fp16_run_device(d_in.data_ptr(), d_r.data_ptr(), 1027)
    torch.cuda.synchronize()
    rag = d_r.cpu().numpy()
    np.save(path, np.concatenate([full[:512], full[-256:], rag[-259:]]))
    print("ragged run equals the full one:", bool(np.array_equal(rag, full[:1027])), " finite:", bool(np.isfinite(full).all()), flush=True)
