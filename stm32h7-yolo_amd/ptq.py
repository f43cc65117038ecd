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
int32 bias of `yoloface_int8.tflite` is reproduced from the float weights of `yoloface.tflite`) and the third within a
tolerance (the reference resizes its calibration images with OpenCV, which is not available here).
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
