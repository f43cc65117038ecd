#!/usr/bin/env python3
"""Builds ptq_float_convs.npz: the float32 weights and biases of the 24 convolutions of the reference's FLOAT model
(yoloface/tflite/yoloface.tflite, a DATA file) in graph order: w0..w23 (tflite layout: OHWI for CONV_2D, 1HWC for
DEPTHWISE_CONV_2D), b0..b23, dw (1 = depthwise).  Container-only (needs /root/reference).  Used by tests/test_ptq.py to
check that the int8 model's weights and biases are exactly the TFLite post-training quantisation of these."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "tools"))
from tflite_reader import read_tflite  # noqa: E402

m = read_tflite("/root/reference/yoloface/tflite/yoloface.tflite")
out, k = {}, 0
for o in m["ops"]:
    if o["op"] in ("CONV_2D", "DEPTHWISE_CONV_2D"):
        out[f"w{k}"] = m["tensors"][o["inputs"][1]]["data"].astype(np.float32)
        out[f"b{k}"] = m["tensors"][o["inputs"][2]]["data"].astype(np.float32)
        out[f"dw{k}"] = np.int32(o["op"] == "DEPTHWISE_CONV_2D")
        k += 1
assert k == 24
np.savez_compressed(os.path.join(HERE, "ptq_float_convs.npz"), **out)
print("24 convolutions,", sum(v.size for n, v in out.items() if n[0] == "w"), "weights")
