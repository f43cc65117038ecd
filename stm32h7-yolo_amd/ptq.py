"""Post-training quantisation arithmetic -- host-side mirror of the reference's `yoloface/tflite/tflite_quantize.py:29-96`.

The reference produces `yoloface_int8.tflite` with the TFLite converter (`tf.lite.TFLiteConverter`, DEFAULT optimisation,
int8 builtins, representative dataset = `small_dataset/*.jpg` resized to 56x56 and divided by 255).  TensorFlow is not
available here; this module restates the converter's published quantisation rules so the constants the engine consumes
(`yf_host_prep.c`) can be regenerated from a float model and a calibration set:

  * weights      per output channel, symmetric, narrow range:  s_c = max|w_c| / 127,  q = clip(round(w / s_c), -127, 127)
  * bias         int32,  q = round(b / (s_in * s_c))            (scale s_in * s_c, zero point 0)
  * activations  asymmetric int8 over the calibrated [min, max], range extended to contain 0:
                 s = (max - min) / 255,  zp = clip(round(-128 - min / s), -128, 127)
  * MAX_POOL_2D keeps the parameters of its input; CONCATENATION inputs are requantised to the output's parameters
    (the QUANTIZE ops of the int8 graph)

`tests/test_ptq.py` checks the first two rules EXACTLY against the reference's two model files (every int8 weight and every
int32 bias of `yoloface_int8.tflite` is reproduced from the float weights of `yoloface.tflite`), and the third by calibrating
on the reference's representative dataset prepared as its script prepares it (`resize_linear_u8` restates OpenCV's
INTER_LINEAR resize, which is what makes the difference: PIL's antialiased resize left the first layers 10-27 % off): every
activation zero point and 42 of 46 scales are reproduced (float32 rounding noise), four scales within 0.8 %.
This is an offline tool; it is not on the inference path.
"""
import numpy as np


def quantize_conv_weights(w, channel_axis):
    """float weights -> (int8 weights, float32 per-channel scales).  channel_axis: 0 for CONV_2D (OHWI), 3 for
    DEPTHWISE_CONV_2D (1HWC)."""
    w = np.asarray(w, np.float32)
    axes = tuple(k for k in range(w.ndim) if k != channel_axis)
    scale = (np.abs(w).max(axis=axes) / np.float32(127.0)).astype(np.float32)
    shape = [1] * w.ndim
    shape[channel_axis] = -1
    safe = np.where(scale == 0, np.float32(1), scale).reshape(shape)
    q = np.clip(np.round(w / safe), -127, 127).astype(np.int8)
    return q, scale


def quantize_bias(b, s_in, s_w):
    """float bias -> int32 with scale s_in * s_w[c] (evaluated in double, as the converter does)."""
    scale = np.float64(np.float32(s_in)) * np.asarray(s_w, np.float32).astype(np.float64)
    return np.round(np.asarray(b, np.float32).astype(np.float64) / scale).astype(np.int64).astype(np.int32)


def activation_qparams(rmin, rmax):
    """calibrated range -> (float32 scale, int zero point) for an int8 activation tensor."""
    rmin, rmax = min(float(rmin), 0.0), max(float(rmax), 0.0)
    if rmax == rmin:
        return np.float32(1.0), 0
    scale = (rmax - rmin) / 255.0
    zp = int(np.clip(np.round(-128.0 - rmin / scale), -128, 127))
    return np.float32(scale), zp


class Calibrator:
    """Running min/max per named tensor over a representative dataset (TFLite calibration keeps the extremes over all
    samples)."""

    def __init__(self):
        self.ranges = {}

    def observe(self, name, value):
        v = np.asarray(value)
        lo, hi = float(v.min()), float(v.max())
        if name in self.ranges:
            a, b = self.ranges[name]
            lo, hi = min(lo, a), max(hi, b)
        self.ranges[name] = (lo, hi)

    def qparams(self, name):
        return activation_qparams(*self.ranges[name])


def resize_linear_u8(img, out_w, out_h):
    """OpenCV's `cv2.resize(img, (out_w, out_h))` (INTER_LINEAR, the default) for uint8 images [H, W, C], restated from the
    published algorithm (imgproc/resize.cpp: pixel centres aligned -- src = (dst + 0.5) * scale - 0.5 --, NO antialiasing,
    fixed-point weights of 11 bits per axis):
        horizontal pass  row[x] = S[sx] * a0 + S[sx + 1] * a1              a0 + a1 = 2048 (int16 weights, round half even)
        vertical pass    dst = ( ((b0 * (row0 >> 4)) >> 16) + ((b1 * (row1 >> 4)) >> 16) + 2 ) >> 2
    The reference's calibration images go through exactly this call (tflite_quantize.py:45-52)."""
    src = np.asarray(img, np.uint8)
    h, w = src.shape[:2]
    src = src.reshape(h, w, -1).astype(np.int64)

    def axis(n_out, n_in):
        scale = np.float64(n_in) / n_out
        idx, c0, c1 = np.zeros(n_out, np.int64), np.zeros(n_out, np.int64), np.zeros(n_out, np.int64)
        for d in range(n_out):
            f = np.float32((d + 0.5) * scale - 0.5)                        # OpenCV computes this in float
            s = int(np.floor(f))
            f = np.float32(f - s)
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= n_in - 1:
                s, f = n_in - 1, np.float32(0)
            # saturate_cast<short>(float) = cvRound: round half to even
            c0[d] = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))
            c1[d] = int(np.rint(np.float32(f * np.float32(2048))))
            idx[d] = s
        return idx, c0, c1

    xi, xa0, xa1 = axis(out_w, w)
    yi, yb0, yb1 = axis(out_h, h)
    xi1 = np.minimum(xi + 1, w - 1)
    rows = src[:, xi, :] * xa0[None, :, None] + src[:, xi1, :] * xa1[None, :, None]          # [H, out_w, C], int32 range
    yi1 = np.minimum(yi + 1, h - 1)
    r0, r1 = rows[yi], rows[yi1]
    out = (((yb0[:, None, None] * (r0 >> 4)) >> 16) + ((yb1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)
