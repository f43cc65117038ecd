#!/usr/bin/env python3
"""Writes tests/golden/fp16_logits_8.npy: the fp32 head logits of the fused fp16 kernel (BASELINE configs[3]) on the eight frames of
tests/test_gpu_parity.py::test_baseline_config4_fp16_tolerance (six golden frames as pixel / 255 + two seeded ones).  GPU ONLY (run through gpurun).

What the fixture is and is not: it is THIS kernel's output on gfx950 (MFMA accumulation order included), committed so that a refactor of the fp16
kernel -- a changed arena plan, an earlier DMA -- is pinned bit for bit by the GPU suite instead of by an A/B tool run by hand.  It is NOT a
reference value: the parity claim of this configuration is the tolerance check against the fp32 numpy evaluation of the reference's ONNX graph
in the same test.  Regenerate it (and say so in the commit) when the kernel's arithmetic order is changed on purpose."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
yf = importlib.import_module("stm32h7-yolo_amd")
gold = np.fromfile(os.path.join(ROOT, "tests", "golden", "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)
u8 = np.random.default_rng(3).integers(0, 256, (8, 56, 56, 3), dtype=np.uint8)
u8[:6] = (gold.astype(np.int16) + 128).astype(np.uint8)
x16 = (u8.astype(np.float32) / 255).astype(np.float16)
net = yf.Network().init()
net.fp16_init()
d_in = torch.from_numpy(x16).cuda()
d_out = torch.zeros((8, 7, 7, 18), dtype=torch.float32, device="cuda")
net.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), 8)
torch.cuda.synchronize()
out = d_out.cpu().numpy()
assert np.isfinite(out).all()
dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "fp16_logits_8.npy")
np.save(dst, out)
print("wrote", dst, "build", net.build_id, "sum", float(out.sum()))
