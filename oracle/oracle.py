"""ctypes front-end of the CPU oracle (oracle/yf_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED: see oracle/yf_oracle.h.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libyf_oracle.so")
MODEL_PATH = os.path.join(_HERE, "model", "yoloface_int8.yfm")


# rounding variants of the requantisation step (oracle/yf_oracle.h, YFO_RV_*)
RV_REF, RV_UP_DENSE, RV_UP_ALL, RV_FP32, RV_SINGLE_DENSE = range(5)
VARIANTS = {"R": RV_REF, "U": RV_UP_DENSE, "U-all": RV_UP_ALL, "X": RV_FP32, "S": RV_SINGLE_DENSE}


class Det(ctypes.Structure):
    _fields_ = [("frame", ctypes.c_int32), ("anchor", ctypes.c_uint8), ("row", ctypes.c_uint8),
                ("col", ctypes.c_uint8), ("q_conf", ctypes.c_int8), ("conf", ctypes.c_float),
                ("x1", ctypes.c_int32), ("y1", ctypes.c_int32), ("x2", ctypes.c_int32), ("y2", ctypes.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "yf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libyf_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    build()
    lib = ctypes.CDLL(_LIB_PATH)
    lib.yfo_load.restype = ctypes.c_void_p
    lib.yfo_load.argtypes = [ctypes.c_char_p]
    lib.yfo_free.argtypes = [ctypes.c_void_p]
    lib.yfo_num_ops.argtypes = [ctypes.c_void_p]
    lib.yfo_dump_bytes.restype = ctypes.c_long
    lib.yfo_dump_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.yfo_out_shape.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 3
    lib.yfo_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.yfo_leaky_lut.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.yfo_run_variant.argtypes = lib.yfo_run.argtypes + [ctypes.c_int]
    lib.yfo_leaky_lut_variant.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.yfo_mbqm_mode.restype = ctypes.c_int32
    lib.yfo_mbqm_mode.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int, ctypes.c_int]
    lib.yfo_decode_py.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.POINTER(Det), ctypes.c_int]
    lib.yfo_decode_c.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.POINTER(Det), ctypes.c_int, ctypes.c_int]
    lib.yfo_prepare_rgb565.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.yfo_quantize_multiplier.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int)]
    lib.yfo_mbqm.restype = ctypes.c_int32
    lib.yfo_mbqm.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int]
    lib.yfo_srdhm.restype = ctypes.c_int32
    lib.yfo_srdhm.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.yfo_rdivpot.restype = ctypes.c_int32
    lib.yfo_rdivpot.argtypes = [ctypes.c_int32, ctypes.c_int]
    return lib


def decode_tables():
    """(sigmoid, exp) float32[256] -- the committed tables (tests/golden/decode_tables_f32.bin)."""
    p = os.path.join(os.path.dirname(_HERE), "tests", "golden", "decode_tables_f32.bin")
    t = np.fromfile(p, dtype="<f4").reshape(2, 256)
    return t[0].copy(), t[1].copy()


class Oracle:
    """CPU restatement of the int8 yoloface graph (TFLite reference-kernel semantics)."""

    def __init__(self, model_path=MODEL_PATH):
        self.lib = _load()
        self.h = self.lib.yfo_load(model_path.encode())
        if not self.h:
            raise RuntimeError(f"cannot load {model_path}")
        self.sig, self.ex = decode_tables()

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.yfo_free(self.h)
            self.h = None

    @property
    def num_ops(self):
        return self.lib.yfo_num_ops(self.h)

    def out_shape(self, h=56, w=56):
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.lib.yfo_out_shape(self.h, h, w, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def dump_bytes(self, h=56, w=56):
        return self.lib.yfo_dump_bytes(self.h, h, w)

    def run(self, frames, dump=False, threads=1, variant=RV_REF):
        """frames: int8 [n,h,w,3] -> head int8 [n,oh,ow,18] (and [n,dump_bytes] when dump).
        variant: RV_* -- another published rounding of the requantisation step (default: TFLite's reference kernels)."""
        x = np.ascontiguousarray(frames, dtype=np.int8)
        assert x.ndim == 4 and x.shape[3] == 3
        n, h, w, _ = x.shape
        oh, ow, oc = self.out_shape(h, w)
        out = np.empty((n, oh, ow, oc), np.int8)
        d = np.empty((n, self.dump_bytes(h, w)), np.int8) if dump else None
        rc = self.lib.yfo_run_variant(self.h, x.ctypes.data, n, h, w, out.ctypes.data,
                                      d.ctypes.data if dump else None, threads, int(variant))
        if rc != n:
            raise RuntimeError(f"yfo_run failed rc={rc}")
        return (out, d) if dump else out

    def leaky_lut(self, op_index, variant=RV_REF):
        lut = np.empty(256, np.int8)
        if self.lib.yfo_leaky_lut_variant(self.h, op_index, lut.ctypes.data, int(variant)) != 0:
            raise ValueError("not a LEAKY_RELU op")
        return lut

    def decode_py(self, head, frame=0, w_scale=1.0, h_scale=1.0, max_dets=147):
        hd = np.ascontiguousarray(head, dtype=np.int8)
        gh, gw = hd.shape[-3], hd.shape[-2]
        buf = (Det * max_dets)()
        n = self.lib.yfo_decode_py(hd.ctypes.data, gh, gw, frame, self.sig.ctypes.data, self.ex.ctypes.data,
                                   w_scale, h_scale, buf, max_dets)
        return [(d.frame, d.anchor, d.row, d.col, d.q_conf, d.conf, d.x1, d.y1, d.x2, d.y2) for d in buf[:min(n, max_dets)]]

    def decode_c(self, head, frame=0, max_dets=147, host_x86=False):
        """yoloface.c:98-152; host_x86: float -> int as an x86-64 build of that file converts (the MCU saturates)."""
        hd = np.ascontiguousarray(head, dtype=np.int8)
        buf = (Det * max_dets)()
        n = self.lib.yfo_decode_c(hd.ctypes.data, frame, self.sig.ctypes.data, self.ex.ctypes.data, buf, max_dets, int(host_x86))
        return [(d.frame, d.anchor, d.row, d.col, d.q_conf, d.conf, d.x1, d.y1, d.x2, d.y2) for d in buf[:min(n, max_dets)]]

    def prepare_rgb565(self, rgb565_112):
        src = np.ascontiguousarray(rgb565_112, dtype=np.uint8).reshape(112 * 112 * 2)
        out = np.empty((56, 56, 3), np.int8)
        self.lib.yfo_prepare_rgb565(src.ctypes.data, out.ctypes.data)
        return out
