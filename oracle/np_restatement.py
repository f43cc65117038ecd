"""Second, independent restatement of the int8 graph in numpy -- TEST INFRASTRUCTURE ONLY, never shipped.

Written separately from oracle/yf_oracle.c (vectorised int64 arithmetic, own .yfm parser, own
QuantizeMultiplier) so that a slip in either implementation shows up as a mismatch in
tests/test_oracle.py.  Same semantics source: TFLite 2.10 reference kernels.
PARITY UNPINNED against the real interpreter (see oracle/yf_oracle.h).
"""
import math
import struct
import numpy as np

ADD, CONCAT, CONV, DWCONV, MAXPOOL, PAD, LEAKY, QUANTIZE = 0, 2, 3, 4, 17, 34, 98, 114


def load_yfm(path):
    b = open(path, "rb").read()
    assert b[:4] == b"YFM1"
    nt, no, tin, tout, nd = struct.unpack_from("<5I", b, 4)
    off = 24
    tensors = []
    for _ in range(nt):
        s0, s1, s2, s3, ty, zp, ns, soff, qdim, doff, dbytes = struct.unpack_from("<4iIiIIiII", b, off)
        off += 44
        tensors.append(dict(shape=[s0, s1, s2, s3], type=ty, zp=zp, ns=ns, soff=soff, qdim=qdim, doff=doff, dbytes=dbytes))
    ops = []
    for _ in range(no):
        v = struct.unpack_from("<I3ii7iI", b, off)
        off += 52
        ops.append(dict(op=v[0], ins=list(v[1:4]), out=v[4], padding=v[5], sw=v[6], sh=v[7], fw=v[8], fh=v[9],
                        dm=v[10], axis=v[11], alpha=struct.unpack("<f", struct.pack("<I", v[12]))[0]))
    data = b[off:off + nd]
    for t in tensors:
        t["scale"] = np.frombuffer(data, "<f4", t["ns"], t["soff"]).copy() if t["ns"] else np.zeros(0, np.float32)
        if t["doff"] != 0xFFFFFFFF:
            dt = np.int8 if t["type"] == 0 else np.dtype("<i4")
            n = t["dbytes"] // np.dtype(dt).itemsize
            t["data"] = np.frombuffer(data, dt, n, t["doff"]).copy()
        else:
            t["data"] = None
    return dict(tensors=tensors, ops=ops, input=tin, output=tout)


def quantize_multiplier(d):
    if d == 0.0:
        return 0, 0
    q, shift = math.frexp(d)
    qf = q * float(1 << 31)
    qi = int(math.floor(abs(qf) + 0.5)) * (1 if qf >= 0 else -1)      # C round(): half away from zero
    if qi == (1 << 31):
        qi //= 2
        shift += 1
    if shift < -31:
        return 0, 0
    return qi, shift


def srdhm(a, b):
    """round-half-up of a*b/2^31 (== gemmlowp's nudge + truncating division), int64 numpy."""
    ab = a.astype(np.int64) * np.int64(b)
    return ((ab + (1 << 30)) >> 31).astype(np.int64)


def rdivpot(x, e):
    """round-half-away-from-zero of x/2^e."""
    if e == 0:
        return x
    half = np.int64(1) << (e - 1)
    return np.where(x >= 0, (x + half) >> e, -((-x + half) >> e))


def mbqm(x, m, shift, mode=0):
    """mode 0: TFLite reference (ties away from zero); 1: right shift ties upward (ruy vector kernels / ARM srshl);
    2: single rounding (ruy standard C++).  The variants mirror oracle/yf_oracle.h YFO_RV_*."""
    ls, rs = (shift, 0) if shift > 0 else (0, -shift)
    if mode == 2:
        total = 31 - shift
        return (x.astype(np.int64) * np.int64(m) + (np.int64(1) << (total - 1))) >> total
    s = srdhm(x * (1 << ls), m)
    if mode == 1:
        return (s + (np.int64(1) << (rs - 1))) >> rs if rs else s
    return rdivpot(s, rs)


# variant -> (dense conv, depthwise conv, element-wise) form; "f" = float32 requantisation (XNNPACK qs8)
VARIANT_MODES = {0: (0, 0, 0), 1: (1, 0, 0), 2: (1, 1, 1), 3: ("f", "f", 0), 4: (2, 0, 0)}


def clamp8(x):
    return np.clip(x, -128, 127).astype(np.int8)


def _same_valid(padding, n, k, s):
    if padding == 0:
        out = -(-n // s)
        return out, max(0, (out - 1) * s + k - n) // 2
    return (n - k) // s + 1, 0


class NpModel:
    def __init__(self, path):
        self.m = load_yfm(path)

    def run(self, frame, dump=False, variant=0):
        """frame int8 [h,w,3] -> head int8 [oh,ow,18] (+ list of every op output)."""
        T, ops = self.m["tensors"], self.m["ops"]
        m_dense, m_dw, m_elt = VARIANT_MODES[variant]
        f32 = np.float32
        val = {self.m["input"]: np.asarray(frame, np.int8)}
        outs = []
        for o in ops:
            x = val[o["ins"][0]]
            ti, to = T[o["ins"][0]], T[o["out"]]
            s_in = f32(ti["scale"][0]) if ti["ns"] else None
            s_out = f32(to["scale"][0])
            zi, zo = ti["zp"], to["zp"]
            if o["op"] == PAD:
                p = T[o["ins"][1]]["data"].reshape(4, 2)
                y = np.pad(x, ((p[1][0], p[1][1]), (p[2][0], p[2][1]), (0, 0)), constant_values=zo)
            elif o["op"] in (CONV, DWCONV):
                wt = T[o["ins"][1]]
                w = wt["data"].reshape(wt["shape"]).astype(np.int64)
                bias = T[o["ins"][2]]["data"].astype(np.int64)
                kh, kw = wt["shape"][1], wt["shape"][2]
                oh, ph = _same_valid(o["padding"], x.shape[0], kh, o["sh"])
                ow, pw = _same_valid(o["padding"], x.shape[1], kw, o["sw"])
                xc = x.astype(np.int64) - zi                      # padded taps contribute 0
                need_h = (oh - 1) * o["sh"] + kh
                need_w = (ow - 1) * o["sw"] + kw
                xp = np.zeros((need_h, need_w, x.shape[2]), np.int64)
                hh = min(x.shape[0], need_h - ph)
                ww = min(x.shape[1], need_w - pw)
                xp[ph:ph + hh, pw:pw + ww] = xc[:hh, :ww]
                cout = wt["shape"][0] if o["op"] == CONV else wt["shape"][3]
                acc = np.zeros((oh, ow, cout), np.int64)
                for fy in range(kh):
                    for fx in range(kw):
                        patch = xp[fy:fy + (oh - 1) * o["sh"] + 1:o["sh"], fx:fx + (ow - 1) * o["sw"] + 1:o["sw"]]
                        if o["op"] == CONV:
                            acc += patch @ w[:, fy, fx, :].T
                        else:
                            acc += patch * w[0, fy, fx, :]
                acc += bias
                y = np.empty(acc.shape, np.int64)
                mode = m_dense if o["op"] == CONV else m_dw
                for c in range(cout):
                    if mode == "f":
                        fs = f32(f32(s_in * f32(wt["scale"][c])) / s_out)             # float32 arithmetic
                        y[..., c] = np.rint(acc[..., c].astype(np.float32) * fs).astype(np.int64)   # ties to even
                        continue
                    eff = float(s_in) * float(f32(wt["scale"][c])) / float(s_out)
                    m, sh = quantize_multiplier(eff)
                    y[..., c] = mbqm(acc[..., c], m, sh, mode)
                y = clamp8(y + zo)
            elif o["op"] == LEAKY:
                alpha = f32(o["alpha"])
                ma, sa = quantize_multiplier(float(f32(s_in * alpha / s_out)))
                mi, si = quantize_multiplier(float(f32(s_in / s_out)))
                v = x.astype(np.int64) - zi
                y = clamp8(zo + np.where(v >= 0, mbqm(v, mi, si, m_elt), mbqm(v, ma, sa, m_elt)))
            elif o["op"] == MAXPOOL:
                oh, ph = _same_valid(o["padding"], x.shape[0], o["fh"], o["sh"])
                ow, pw = _same_valid(o["padding"], x.shape[1], o["fw"], o["sw"])
                y = np.empty((oh, ow, x.shape[2]), np.int8)
                for oy in range(oh):
                    for ox in range(ow):
                        y0, x0 = oy * o["sh"] - ph, ox * o["sw"] - pw
                        win = x[max(y0, 0):min(y0 + o["fh"], x.shape[0]), max(x0, 0):min(x0 + o["fw"], x.shape[1])]
                        y[oy, ox] = win.reshape(-1, x.shape[2]).max(axis=0)
            elif o["op"] == ADD:
                x2, t2 = val[o["ins"][1]], T[o["ins"][1]]
                s1, s2 = f32(ti["scale"][0]), f32(t2["scale"][0])
                twice = float(f32(2) * max(s1, s2))
                m1, h1 = quantize_multiplier(float(s1) / twice)
                m2, h2 = quantize_multiplier(float(s2) / twice)
                mo, ho = quantize_multiplier(twice / float(f32(1 << 20) * s_out))
                a = (x.astype(np.int64) - zi) << 20
                b = (x2.astype(np.int64) - t2["zp"]) << 20
                y = clamp8(mbqm(mbqm(a, m1, h1, m_elt) + mbqm(b, m2, h2, m_elt), mo, ho, m_elt) + zo)
            elif o["op"] == QUANTIZE:
                m, sh = quantize_multiplier(float(s_in) / float(s_out))
                y = clamp8(mbqm(x.astype(np.int64) - zi, m, sh, m_elt) + zo)
            elif o["op"] == CONCAT:
                y = np.concatenate([x, val[o["ins"][1]]], axis=2)
            else:
                raise NotImplementedError(o["op"])
            val[o["out"]] = y
            outs.append(y)
        head = val[self.m["output"]]
        return (head, outs) if dump else head

    def run_float(self, frame, float_convs=None, observe=None):
        """float64 evaluation of the dequantised graph (structural sanity).
        float_convs: optional list of (weights in tflite layout, bias) replacing the dequantised int8 constants, in
        conv order (PTQ calibration, tests/test_ptq.py); observe(tensor_index, value) sees every op output."""
        T, ops = self.m["tensors"], self.m["ops"]
        t0 = T[self.m["input"]]
        val = {self.m["input"]: (np.asarray(frame, np.float64) - t0["zp"]) * float(t0["scale"][0])}
        if observe is not None:
            observe(self.m["input"], val[self.m["input"]])
        n_conv = 0
        for o in ops:
            x = val[o["ins"][0]]
            if o["op"] == PAD:
                p = T[o["ins"][1]]["data"].reshape(4, 2)
                y = np.pad(x, ((p[1][0], p[1][1]), (p[2][0], p[2][1]), (0, 0)))
            elif o["op"] in (CONV, DWCONV):
                wt, ti = T[o["ins"][1]], T[o["ins"][0]]
                wq = wt["data"].reshape(wt["shape"]).astype(np.float64)
                ws = wt["scale"].astype(np.float64)
                w = wq * (ws[:, None, None, None] if o["op"] == CONV else ws[None, None, None, :])
                bias = T[o["ins"][2]]["data"].astype(np.float64) * float(ti["scale"][0]) * ws
                if float_convs is not None:
                    w = np.asarray(float_convs[n_conv][0], np.float64).reshape(wt["shape"])
                    bias = np.asarray(float_convs[n_conv][1], np.float64)
                n_conv += 1
                kh, kw = wt["shape"][1], wt["shape"][2]
                oh, ph = _same_valid(o["padding"], x.shape[0], kh, o["sh"])
                ow, pw = _same_valid(o["padding"], x.shape[1], kw, o["sw"])
                need_h, need_w = (oh - 1) * o["sh"] + kh, (ow - 1) * o["sw"] + kw
                xp = np.zeros((need_h, need_w, x.shape[2]))
                hh, ww = min(x.shape[0], need_h - ph), min(x.shape[1], need_w - pw)
                xp[ph:ph + hh, pw:pw + ww] = x[:hh, :ww]
                cout = wt["shape"][0] if o["op"] == CONV else wt["shape"][3]
                y = np.zeros((oh, ow, cout))
                for fy in range(kh):
                    for fx in range(kw):
                        patch = xp[fy:fy + (oh - 1) * o["sh"] + 1:o["sh"], fx:fx + (ow - 1) * o["sw"] + 1:o["sw"]]
                        y += patch @ w[:, fy, fx, :].T if o["op"] == CONV else patch * w[0, fy, fx, :]
                y += bias
            elif o["op"] == LEAKY:
                y = np.where(x >= 0, x, x * float(np.float32(o["alpha"])))
            elif o["op"] == MAXPOOL:
                oh, ph = _same_valid(o["padding"], x.shape[0], o["fh"], o["sh"])
                ow, pw = _same_valid(o["padding"], x.shape[1], o["fw"], o["sw"])
                y = np.empty((oh, ow, x.shape[2]))
                for oy in range(oh):
                    for ox in range(ow):
                        y0, x0 = oy * o["sh"] - ph, ox * o["sw"] - pw
                        win = x[max(y0, 0):min(y0 + o["fh"], x.shape[0]), max(x0, 0):min(x0 + o["fw"], x.shape[1])]
                        y[oy, ox] = win.reshape(-1, x.shape[2]).max(axis=0)
            elif o["op"] == ADD:
                y = x + val[o["ins"][1]]
            elif o["op"] == QUANTIZE:
                y = x
            elif o["op"] == CONCAT:
                y = np.concatenate([x, val[o["ins"][1]]], axis=2)
            val[o["out"]] = y
            if observe is not None:
                observe(o["out"], y)
        return val[self.m["output"]]
