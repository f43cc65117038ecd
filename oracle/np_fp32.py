"""fp32 evaluation of the reference's ONNX graph (yoloface/pytorch/yoloface-50k.onnx) in numpy -- TEST INFRASTRUCTURE.

The fp16 configuration (BASELINE configs[3]) is checked against this with a tolerance on the head logits
(atol 2e-2, rtol 2e-2); it is NOT comparable with the int8 head (PTQ of this narrow net is lossy).
Graph: the same 24 convs / 17 LeakyReLU(0.1) / 2 max-pools / 3 adds / 2 concats as the tflite model
(reference yoloface/pytorch/yoloface.py:83-119); input NHWC float in [0,1] (uint8/255), output [7,7,18] logits.
"""
import struct
import numpy as np


def load_yfw(path):
    b = open(path, "rb").read()
    assert b[:4] == b"YFW1"
    n = struct.unpack_from("<I", b, 4)[0]
    off, convs = 8, []
    for _ in range(n):
        dw, cin, cout, k, stride, nw = struct.unpack_from("<6I", b, off); off += 24
        w = np.frombuffer(b, "<f4", nw, off).copy(); off += 4 * nw
        bias = np.frombuffer(b, "<f4", cout, off).copy(); off += 4 * cout
        w = w.reshape((k, k, cout)) if dw else w.reshape((cout, k, k, cin))
        convs.append(dict(dw=bool(dw), cin=cin, cout=cout, k=k, stride=stride, w=w, b=bias))
    return convs


def _conv(x, c):
    k, s = c["k"], c["stride"]
    h, w, _ = x.shape
    if k == 1:
        return (x.reshape(-1, c["cin"]) @ c["w"].reshape(c["cout"], c["cin"]).T + c["b"]).reshape(h, w, c["cout"]).astype(np.float32)
    oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    xp = np.zeros((h + 2, w + 2, x.shape[2]), np.float32)
    xp[1:h + 1, 1:w + 1] = x
    y = np.zeros((oh, ow, c["cout"]), np.float32)
    for ky in range(3):
        for kx in range(3):
            patch = xp[ky:ky + (oh - 1) * s + 1:s, kx:kx + (ow - 1) * s + 1:s]
            y += patch * c["w"][ky, kx] if c["dw"] else patch @ c["w"][:, ky, kx, :].T
    return (y + c["b"]).astype(np.float32)


def _leaky(x):
    return np.where(x > 0, x, np.float32(0.1) * x).astype(np.float32)


def _pool(x, k, pad):
    h, w, c = x.shape
    oh, ow = (h + 2 * pad - k) // 2 + 1, (w + 2 * pad - k) // 2 + 1
    y = np.empty((oh, ow, c), np.float32)
    for oy in range(oh):
        for ox in range(ow):
            y0, x0 = 2 * oy - pad, 2 * ox - pad
            y[oy, ox] = x[max(y0, 0):min(y0 + k, h), max(x0, 0):min(x0 + k, w)].reshape(-1, c).max(axis=0)
    return y


def run_fp32(convs, frame):
    """frame float32 [H,W,3] in [0,1] -> logits float32 [H/8, W/8, 18]"""
    c = iter(convs)
    x = _leaky(_conv(np.asarray(frame, np.float32), next(c)))        # conv1
    x = _leaky(_conv(x, next(c)))                                     # dw3
    x = _conv(x, next(c))                                             # c5
    t4 = _leaky(_conv(x, next(c)))                                    # c6
    x = _leaky(_conv(t4, next(c)))                                    # dw10
    t7 = _conv(x, next(c))                                            # c12
    x = _leaky(_conv(t7, next(c)))                                    # c13
    x = _leaky(_conv(x, next(c)))                                     # dw15
    x = _conv(x, next(c)) + t7                                        # c17 + add
    x = _leaky(_conv(x, next(c)))                                     # c19
    x = np.concatenate([_pool(t4, 8, 3), x], axis=2)                  # concat (pool first)
    t15 = _leaky(_conv(x, next(c)))                                   # c23
    x = _leaky(_conv(t15, next(c)))                                   # dw27
    t18 = _conv(x, next(c))                                           # c29
    x = _leaky(_conv(t18, next(c)))                                   # c30
    x = _leaky(_conv(x, next(c)))                                     # dw32
    t22 = _conv(x, next(c)) + t18                                     # c34 + add
    x = _leaky(_conv(t22, next(c)))                                   # c36
    x = _leaky(_conv(x, next(c)))                                     # dw38
    x = _conv(x, next(c)) + t22                                       # c40 + add
    x = _leaky(_conv(x, next(c)))                                     # c42
    x = np.concatenate([_pool(t15, 4, 1), x], axis=2)
    x = _leaky(_conv(x, next(c)))                                     # c47
    x = _leaky(_conv(x, next(c)))                                     # dw49
    x = _leaky(_conv(x, next(c)))                                     # c51
    return _conv(x, next(c))                                          # head
