#!/usr/bin/env python3
"""A launch that FAILS between the scratch map's get() and mark() must not keep its region (csrc/yf_stream_scratch.h, Lease): with the lab library
and YF_LAB_FAIL_LAUNCHES=9 the first nine fused launches get an invalid grid (hipErrorInvalidConfiguration, nothing runs) -- nine distinct streams,
one more than the map's eight regions.  Then sixteen good launches on nine OTHER streams must all get a region and give the oracle's heads (round 4:
every get() from a new stream returned hipErrorNotReady from the ninth failure on).  Test helper: tests/test_gpu_parity.py runs it in a fresh process."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
yf = importlib.import_module("stm32h7-yolo_amd")
n_fail = int(os.environ["YF_LAB_FAIL_LAUNCHES"])
block = np.random.default_rng(51).integers(-128, 128, (1024, 56, 56, 3), dtype=np.int8)
ref = Oracle().run(block, threads=16)
net = yf.Network(device=0).init()
net.configure(2, 8)                                  # the batched shape: it parks T15 tensors in the stream's region
d_in = torch.from_numpy(block).cuda()
torch.cuda.synchronize()
dead, failures = [], 0
for k in range(n_fail):
    st = torch.cuda.Stream()
    o = torch.empty((1024, 7, 7, 18), dtype=torch.int8, device="cuda")
    try:
        net.run_device(d_in.data_ptr(), o.data_ptr(), 1024, st.cuda_stream)
    except Exception as e:                           # noqa: BLE001
        failures += 1
        last = str(e)
    dead.append(st)
print(f"{failures} of {n_fail} launches failed as arranged ({last.splitlines()[0][:100]})", flush=True)
good, outs = [torch.cuda.Stream() for _ in range(9)], []
for k in range(16):
    st = good[k % 9]
    o = torch.empty((1024, 7, 7, 18), dtype=torch.int8, device="cuda")
    net.run_device(d_in.data_ptr(), o.data_ptr(), 1024, st.cuda_stream)
    outs.append(o)
torch.cuda.synchronize()
bad = sum(not np.array_equal(o.cpu().numpy(), ref) for o in outs)
held = net.scratch_bytes()
print(f"16 good launches on 9 streams: {bad} mismatches, scratch held {held / 2**20:.1f} MiB", flush=True)
ok = failures == n_fail and bad == 0 and 0 < held <= 8 * 48 * 2 ** 20
print("failed-launch rehearsal ok" if ok else "failed-launch rehearsal FAILED")
sys.exit(0 if ok else 1)
