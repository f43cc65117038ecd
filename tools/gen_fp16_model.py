#!/usr/bin/env python3
"""Extract the fp32 weights of the reference's ONNX export for the fp16 configuration -- DEV TOOL, container only.

Source: /root/reference/yoloface/pytorch/yoloface-50k.onnx (the only weight-bearing file of yoloface/pytorch/;
the Darknet checkpoint named at yoloface.py:424 is not in the repository -- SURVEY.md Appendix B).
Output: stm32h7-yolo_amd/model/yoloface_fp32.yfw
  'YFW1', u32 n_conv, then per conv (graph order == tflite conv order):
  u32 depthwise, cin, cout, k, stride, n_weights ; f32 weights (dense: OHWI, depthwise: HWC) ; f32 bias[cout]
"""
import os
import struct
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from onnx_reader import read_onnx  # noqa: E402


def main():
    m = read_onnx("/root/reference/yoloface/pytorch/yoloface-50k.onnx")
    init = m["initializers"]
    convs = [n for n in m["nodes"] if n["op"] == "Conv"]
    assert len(convs) == 24
    out = [b"YFW1", struct.pack("<I", len(convs))]
    shapes = []
    for n in convs:
        w, b = init[n["inputs"][1]], init[n["inputs"][2]]
        a = n["attrs"]
        cout, cin_g, k, _ = w.shape
        dw = a["group"] > 1
        assert a["strides"][0] == a["strides"][1] and a["dilations"] == [1, 1]
        assert (a["pads"] == [1, 1, 1, 1]) == (k == 3) and (not dw or (cin_g == 1 and a["group"] == cout))
        if dw:
            wl = np.ascontiguousarray(w[:, 0].transpose(1, 2, 0))          # [kh][kw][c]
            cin = cout
        else:
            wl = np.ascontiguousarray(w.transpose(0, 2, 3, 1))             # OHWI
            cin = cin_g
        out.append(struct.pack("<6I", int(dw), cin, cout, k, a["strides"][0], wl.size))
        out.append(wl.astype("<f4").tobytes())
        out.append(b.astype("<f4").tobytes())
        shapes.append((int(dw), cin, cout, k, a["strides"][0]))
    expect = [(0, 3, 8, 3, 2), (1, 8, 8, 3, 1), (0, 8, 4, 1, 1), (0, 4, 18, 1, 1), (1, 18, 18, 3, 2), (0, 18, 6, 1, 1),
              (0, 6, 36, 1, 1), (1, 36, 36, 3, 1), (0, 36, 6, 1, 1), (0, 6, 18, 1, 1), (0, 36, 24, 1, 1), (1, 24, 24, 3, 2),
              (0, 24, 8, 1, 1), (0, 8, 40, 1, 1), (1, 40, 40, 3, 1), (0, 40, 8, 1, 1), (0, 8, 40, 1, 1), (1, 40, 40, 3, 1),
              (0, 40, 8, 1, 1), (0, 8, 24, 1, 1), (0, 48, 40, 1, 1), (1, 40, 40, 3, 1), (0, 40, 32, 1, 1), (0, 32, 18, 1, 1)]
    assert shapes == expect, "ONNX conv order does not match the tflite graph"
    alphas = {n["attrs"]["alpha"] for n in m["nodes"] if n["op"] == "LeakyRelu"}
    assert len(alphas) == 1 and abs(alphas.pop() - 0.1) < 1e-7
    path = os.path.join(os.path.dirname(HERE), "stm32h7-yolo_amd", "model", "yoloface_fp32.yfw")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "wb").write(b"".join(out))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
