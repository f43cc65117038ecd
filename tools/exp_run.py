#!/usr/bin/env python3
"""Launch ONE build of the fused kernel (shipped, or the experimental namespace with YF_EXPERIMENTAL=1) on eight rotating
input batches -- the thing to put behind `rocprofv3 --pmc ...` when an experimental build's counters are wanted.  DEV TOOL."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
yf = importlib.import_module("stm32h7-yolo_amd")
n, nb = 4096, 8
exp = os.environ.get("YF_EXPERIMENTAL", "0") == "1"
net = yf.Network().init()
net.configure(2 + (200 if exp else 0), 8)
rng = np.random.default_rng(1)
ins = [torch.from_numpy(rng.integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)).cuda() for _ in range(nb)]
out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for it in range(int(os.environ.get("YF_ITERS", "40"))):
    net.run_device(ins[it % nb].data_ptr(), out.data_ptr(), n, s)
torch.cuda.synchronize()
print(net.kernel_name, "done")
