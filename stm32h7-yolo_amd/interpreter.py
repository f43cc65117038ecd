"""Host-side mirror of the reference's Python path (yoloface/tflite/tflite_prediction.py:23-63).

The reference drives `tf.lite.Interpreter(model_path='yoloface_int8.tflite')` with
allocate_tensors / get_input_details / set_tensor / invoke / get_tensor and then decodes boxes in numpy.
`Interpreter` offers the same calls over the MI355X engine (through the C-ABI); `decode_boxes` is the
reference's post-processing, kept in numpy float32 exactly as written there except that sigmoid/exp come
from the committed 256-entry float32 tables (every argument is a function of one int8 value).
"""
import os

import numpy as np

from .binding import Network, IN_H, IN_W, IN_C, OUT_H, OUT_W, OUT_C

INPUT_SCALE, INPUT_ZERO_POINT = 0.003921568859368563, -128      # tflite tensor #0 (network.c:665-669)
OUTPUT_SCALE, OUTPUT_ZERO_POINT = 0.14218327403068542, -15      # tflite tensor #100 (network.c:882-886)
ANCHORS = np.array([[9, 14], [12, 17], [22, 21]], dtype=np.float32)   # tflite_prediction.py:44-47

_TABLES = None


def decode_tables():
    """(sigmoid, exp) float32[256] indexed by q+128; baked in the library as gen/yf_decode_tables_gen.h."""
    global _TABLES
    if _TABLES is None:
        import re
        p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "gen", "yf_decode_tables_gen.h")
        src = open(p).read()
        out = []
        for name in ("yf_sigmoid_bits", "yf_exp_bits"):
            body = src[src.index(name):]
            body = body[body.index("{") + 1: body.index("}")]
            bits = np.array([int(v.rstrip("u"), 16) for v in re.findall(r"0x[0-9a-f]+u", body)], dtype=np.uint32)
            assert bits.size == 256
            out.append(bits.view(np.float32).copy())
        _TABLES = tuple(out)
    return _TABLES


class Interpreter:
    """tf.lite.Interpreter look-alike for yoloface_int8 (tflite_prediction.py:23-41)."""

    # tf.lite.experimental.OpResolverType by name -> the library's rounding (yf_network_set_requant_rounding).  BUILTIN_REF is what the oracle restates and what
    # SURVEY 8(c).3 fixes as "the tflite int8 reference"; BUILTIN / BUILTIN_WITHOUT_DEFAULT_DELEGATES are the optimized kernels, whose dense convolutions go
    # through ruy (ties upward).  AUTO -- what the reference's script gets by passing nothing (tflite_prediction.py:23) -- maps to the library's DEFAULT,
    # which is the reference rounding by the project's contract although TensorFlow's AUTO is most likely the optimized set.
    _RESOLVER_ROUNDING = {"AUTO": None, "BUILTIN_REF": 0, "BUILTIN": 1, "BUILTIN_WITHOUT_DEFAULT_DELEGATES": 1}

    def __init__(self, model_path="yoloface_int8.tflite", device=0, experimental_op_resolver_type="AUTO", requant_rounding=None, **_ignored):
        # the model is baked into the library; model_path is accepted for call-site compatibility
        self.model_path = model_path
        self._net = Network(device=device)
        name = getattr(experimental_op_resolver_type, "name", experimental_op_resolver_type)      # an enum member or its name
        if name not in self._RESOLVER_ROUNDING:
            raise ValueError(f"experimental_op_resolver_type: one of {sorted(self._RESOLVER_ROUNDING)}")
        rounding = requant_rounding if requant_rounding is not None else self._RESOLVER_ROUNDING[name]
        if rounding is not None:
            self._net.set_requant_rounding(rounding)
        self._in = None
        self._out = None
        self._allocated = False

    def allocate_tensors(self):
        self._net.init()
        self._in = np.zeros((1, IN_H, IN_W, IN_C), np.int8)
        self._out = np.zeros((1, OUT_H, OUT_W, OUT_C), np.int8)
        self._allocated = True

    def get_input_details(self):
        return [dict(name="Input", index=0, shape=np.array([1, IN_H, IN_W, IN_C], np.int32), dtype=np.int8,
                     quantization=(INPUT_SCALE, INPUT_ZERO_POINT))]

    def get_output_details(self):
        return [dict(name="Identity", index=100, shape=np.array([1, OUT_H, OUT_W, OUT_C], np.int32), dtype=np.int8,
                     quantization=(OUTPUT_SCALE, OUTPUT_ZERO_POINT))]

    def resize_tensor_input(self, index, shape):
        """Batch dimension only (the engine runs any number of frames per invoke)."""
        if index != 0 or list(shape[1:]) != [IN_H, IN_W, IN_C]:
            raise ValueError("only the batch dimension of input 0 can change")
        self._in = np.zeros((int(shape[0]), IN_H, IN_W, IN_C), np.int8)

    def set_tensor(self, index, value):
        if not self._allocated:
            raise RuntimeError("allocate_tensors() first")
        v = np.asarray(value)
        if index != 0 or v.dtype != np.int8 or tuple(v.shape[1:]) != (IN_H, IN_W, IN_C):
            raise ValueError("input 0 expects int8 [n,56,56,3]")
        self._in = np.ascontiguousarray(v)

    def invoke(self):
        if not self._allocated:
            raise RuntimeError("allocate_tensors() first")
        self._out = self._net.run(self._in)

    def get_tensor(self, index):
        if index != 100:
            raise ValueError("only the output tensor (index 100) is readable")
        return self._out.copy()


def decode_boxes(head, conf_thres=0.7, w_scale=1.0, h_scale=1.0):
    """tflite_prediction.py:42-63 for one int8 head [7,7,18]: int32 boxes [k,4] in (a,row,col) order."""
    sig, ex = decode_tables()
    idx = np.asarray(head, np.int8).astype(np.int32) + 128
    nx, ny = idx.shape[0], idx.shape[1]
    idx = idx.reshape((nx, ny, 3, 6)).transpose([2, 0, 1, 3])
    output = np.zeros(idx.shape, np.float32)
    yv, xv = np.meshgrid(np.arange(ny), np.arange(nx))
    grid = np.stack((yv, xv), 2).reshape((1, ny, nx, 2)).astype(np.float32)
    output[..., 0:2] = (sig[idx[..., 0:2]] + grid) * 8
    output[..., 2:4] = ex[idx[..., 2:4]] * ANCHORS.reshape(3, 1, 1, 2)
    output[..., 4:] = sig[idx[..., 4:]]
    output = output.reshape((-1, 6))
    x = output[output[..., 4] > np.float32(conf_thres)]
    if not x.shape[0]:
        return np.zeros((0, 4), np.int32)
    half = x[:, 2:4] / 2                                      # centre +- half the size (float32 throughout, as the reference computes it)
    boxes = np.concatenate([x[:, 0:2] - half, x[:, 0:2] + half], axis=1)
    boxes *= np.array([w_scale, h_scale, w_scale, h_scale], np.float32)
    return boxes.astype(np.int32)


def format_uart(frame_no, dets_fw, count=None):
    """The firmware's UART text of one frame, byte for byte (stm32/User/main.c:46,53: frame banner, 40 dashes, the
    total line; yoloface.c:148: one line per face; CR LF line ends), parsed unchanged by the reference monitor's
    regexes (上位机/IAP/main.py:325-363).  Pure-Python mirror of the library's yf_network_format_uart."""
    dashes = "-" * 40
    n = len(dets_fw) if count is None else int(count)
    lines = ["=== Frame %d ===\r\n%s\r\n" % (frame_no, dashes)]
    for k, d in enumerate(dets_fw, 1):
        lines.append("[Face %d] BBox: [%d, %d, %d, %d], Conf: %.2f\r\n" % (k & 255, d["x1"], d["y1"], d["x2"], d["y2"], d["conf"]))
    lines.append("%s\r\n[INFO] Total faces detected: %d\r\n" % (dashes, n & 255))
    return "".join(lines)
