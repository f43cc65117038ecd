#!/usr/bin/env python3
"""Device-resident batches whose byte offsets pass 2^31 AND 2^32: 481 280 int8 frames (4.5 GB of input, 424 MB of heads) through yf_network_run_decode_device
in ONE launch, 240 640 fp16 frames (4.5 GB) through the fp16 kernel, 60 160 frames of 160x160 (4.6 GB) through the banded kernels (59 chunks of the 1024-frame
arena).  The inputs are a 512- / 256- / 16-frame block repeated, so every copy's result must equal the block's (the int8 ones also the oracle's): a 32-bit byte offset
(signed or unsigned) anywhere in the kernels or the launch code would show at the far end.  Test helper: tests/test_gpu_parity.py runs it in a fresh process."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
yf = importlib.import_module("stm32h7-yolo_amd")
net = yf.Network(device=0).init()
orc = Oracle()
ok = True
# ---- int8 56x56: 470 copies of a 512-frame block
block = np.random.default_rng(81).integers(-128, 128, (512, 56, 56, 3), dtype=np.int8)
ref = orc.run(block, threads=16)
reps, n = 940, 940 * 512
d_in = torch.from_numpy(block).cuda().repeat(reps, 1, 1, 1)
assert d_in.numel() > 2 ** 32
cap = 2
d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
d_d = torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda")
d_c = torch.zeros((n,), dtype=torch.int32, device="cuda")
net.run_decode_device(d_in.data_ptr(), d_out.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap)
torch.cuda.synchronize()
want = torch.from_numpy(ref).cuda()
same = bool((d_out.view(reps, 512, 7, 7, 18) == want[None]).all())
counts = d_c.view(reps, 512)
frames_field = d_d.view(n, cap, 28)[:, 0, :4].contiguous().view(torch.int32).view(-1)
fired = d_c > 0
idx_ok = bool((frames_field[fired].long() == torch.arange(n, device="cuda")[fired]).all())       # the record of frame f says f, also beyond 2^31 bytes of input
print(f"int8: {n} frames, {d_in.numel() / 2**30:.2f} GiB of input: heads equal the oracle's in every copy: {same}; counts periodic: {bool((counts == counts[0][None]).all())}; frame indices: {idx_ok}", flush=True)
ok = ok and same and bool((counts == counts[0][None]).all()) and idx_ok and int(fired.sum()) > 0
del d_in, d_out, d_d, d_c
# ---- fp16 56x56: 470 copies of a 256-frame block
net.fp16_init()
b16 = (np.random.default_rng(82).integers(0, 256, (256, 56, 56, 3)) / 255.0).astype(np.float16)
n = 940 * 256
d_in = torch.from_numpy(b16).cuda().repeat(940, 1, 1, 1)
assert d_in.numel() * 2 > 2 ** 32
d_out = torch.zeros((n, 7, 7, 18), dtype=torch.float32, device="cuda")
d_one = torch.zeros((256, 7, 7, 18), dtype=torch.float32, device="cuda")
net.fp16_run_device(d_in.data_ptr(), d_one.data_ptr(), 256)
net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n)
torch.cuda.synchronize()
same = bool((d_out.view(940, 256, 7, 7, 18) == d_one[None]).all())
print(f"fp16: {n} frames, {d_in.numel() * 2 / 2**30:.2f} GiB of input: every copy bit for bit the first block: {same}", flush=True)
ok = ok and same
del d_in, d_out
# ---- int8 160x160: 1880 copies of a 16-frame block (30 chunks of the 1024-frame arena)
b160 = np.random.default_rng(83).integers(-128, 128, (16, 160, 160, 3), dtype=np.int8)
r160 = orc.run(b160, threads=16)
n = 3760 * 16
d_in = torch.from_numpy(b160).cuda().repeat(3760, 1, 1, 1)
assert d_in.numel() > 2 ** 32
d_out = torch.zeros((n, 20, 20, 18), dtype=torch.int8, device="cuda")
net.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), n)
torch.cuda.synchronize()
same = bool((d_out.view(3760, 16, 20, 20, 18) == torch.from_numpy(r160).cuda()[None]).all())
print(f"160x160: {n} frames, {d_in.numel() / 2**30:.2f} GiB of input: heads equal the oracle's in every copy: {same}", flush=True)
ok = ok and same
print("huge batches ok" if ok else "huge batches FAILED")
sys.exit(0 if ok else 1)
