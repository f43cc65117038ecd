#!/usr/bin/env python3
"""ai_network_run on host arrays: rate per batch size (best of 7 after one untimed call).  DEV TOOL."""
import importlib, sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
yf = importlib.import_module("stm32h7-yolo_amd")
net = yf.Network().init()
rng = np.random.default_rng(1)
for n in [int(v) for v in os.environ.get("YF_NS", "1,8,64,512,2048,4096,8192,65535").split(",")]:
    x = rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
    out = net.run(x)
    best = float("inf")
    for _ in range(7):
        t0 = time.perf_counter(); net.run(x, out=out); best = min(best, time.perf_counter() - t0)
    print(f"n={n:6d}  {best*1e6:10.1f} us  {n/best/1e6:7.3f} M images/s")
