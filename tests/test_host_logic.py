"""CPU tests of the product's host logic (stm32h7-yolo_amd/csrc/yf_host_prep.c): the device tables are checked
against the oracle's independent fixed-point statement.  No GPU, no compute calls."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "stm32h7-yolo_amd")
N_DENSE, N_DW, N_ADD, N_LUT, N_CS = 17, 7, 3, 19, 24
# const-stages of the fused kernel in execution order: ("d", dense index) / ("w", depthwise index), residual-add index or None
CONST_STAGES = [("d", 0, None), ("w", 0, None), ("d", 1, None), ("d", 2, None), ("w", 1, None), ("d", 3, None), ("d", 4, None), ("w", 2, None),
                ("d", 5, 0), ("d", 6, None), ("d", 7, None), ("w", 3, None), ("d", 8, None), ("d", 9, None), ("w", 4, None), ("d", 10, 1),
                ("d", 11, None), ("w", 5, None), ("d", 12, 2), ("d", 13, None), ("d", 14, None), ("w", 6, None), ("d", 15, None), ("d", 16, None)]
DENSE_OPS = [1, 5, 6, 12, 13, 17, 19, 23, 29, 30, 34, 36, 40, 42, 47, 51, 53]
DW_OPS = [3, 10, 15, 27, 32, 38, 49]
LEAKY_LUT_IDS = {2: 0, 4: 1, 7: 2, 11: 4, 14: 5, 16: 6, 20: 7, 24: 8, 28: 10, 31: 11, 33: 12, 37: 13, 39: 14, 48: 16, 50: 17, 52: 18}


class Dense(ctypes.Structure):
    _fields_ = [("w_off", ctypes.c_uint32), ("c_off", ctypes.c_uint32), ("cout", ctypes.c_uint16),
                ("cout_pad4", ctypes.c_uint16), ("k", ctypes.c_uint16), ("krow", ctypes.c_uint16)]


class Dw(ctypes.Structure):
    _fields_ = [("g_off", ctypes.c_uint32), ("c", ctypes.c_uint16), ("ngroups", ctypes.c_uint16)]


class Add(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("zp1", "zp2", "zpo", "m1", "s1", "m2", "s2", "mo", "so", "kco", "rso")] + \
               [("mo2", ctypes.c_uint32), ("zro", ctypes.c_uint32), ("c64o", ctypes.c_uint32 * 2)]


class Index(ctypes.Structure):
    _fields_ = [("dense", Dense * N_DENSE), ("dw", Dw * N_DW), ("add", Add * N_ADD), ("lut_off", ctypes.c_uint32),
                ("total_bytes", ctypes.c_uint32), ("in_zp", ctypes.c_int32), ("halo_zp", ctypes.c_int32 * N_DW),
                ("cs_v_off", ctypes.c_uint32 * N_CS), ("cs_v_bytes", ctypes.c_uint32 * N_CS), ("cs_s_off", ctypes.c_uint32 * N_CS)]


@pytest.fixture(scope="module")
def prep():
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "../lib/libyf_hostprep.so"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(PKG, "lib", "libyf_hostprep.so"))
    lib.yf_prepare_tables.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(Index)]
    lib.yf_quantize_multiplier.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int)]
    lib.yf_mbqm.restype = ctypes.c_int32
    lib.yf_mbqm.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int]
    blob = (ctypes.c_uint8 * 11304).in_dll(lib, "yf_weights_blob")
    out, ix = ctypes.c_void_p(), Index()
    rc = lib.yf_prepare_tables(blob, 11304, ctypes.byref(out), ctypes.byref(ix))
    assert rc == 0
    tab = bytes((ctypes.c_uint8 * ix.total_bytes).from_address(out.value))
    return dict(lib=lib, blob=bytes(blob), ix=ix, tab=tab)


@pytest.fixture(scope="module")
def pack():
    from oracle.np_restatement import load_yfm
    return load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))


def test_table_layout_matches_the_plan_compiled_into_the_kernels(prep):
    """The kernels address the table blob at compile-time offsets (yf_kernels.hip.h, TablePlan); the host preparation
    lays the blob out at run time.  Both must agree (the engine refuses to start otherwise)."""
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc")], stdout=subprocess.DEVNULL)
    net = ctypes.CDLL(os.path.join(PKG, "lib", "libyf_network.so"))
    n = 43 + 3 * N_CS
    plan = (ctypes.c_int32 * n)()
    assert net.yf_network_table_plan(plan, n) == n
    ix = prep["ix"]
    want = [d.w_off for d in ix.dense] + [d.c_off for d in ix.dense] + [d.g_off for d in ix.dw] + [ix.lut_off, ix.total_bytes] + \
        list(ix.cs_v_off) + list(ix.cs_v_bytes) + list(ix.cs_s_off)
    assert list(plan) == want
    assert net.yf_network_table_plan(None, 0) == n


def test_constant_blocks_of_the_fused_kernel_regroup_the_same_numbers(prep):
    """Round 3: every const-stage owns one contiguous block (weights | {mult2, zr} per pass | residual-add tables) that a single
    LDS-DMA brings into an LDS ring slot, plus a compact array of the scalar side ({c64, rshift} per pass).  They must hold
    exactly the bytes of the per-stage records the requantisation tests below check."""
    ix, tab = prep["ix"], prep["tab"]
    assert len(CONST_STAGES) == N_CS
    for cs, (kind, i, add) in enumerate(CONST_STAGES):
        v, vb, so = ix.cs_v_off[cs], ix.cs_v_bytes[cs], ix.cs_s_off[cs]
        assert v % 16 == 0 and vb % 16 == 0 and so % 16 == 0 and vb <= 2816       # one ring slot of the kernel
        if kind == "d":
            d = ix.dense[i]
            wb, npass = d.cout_pad4 * d.krow, d.cout_pad4 // 4
            assert tab[v:v + wb] == tab[d.w_off:d.w_off + wb]
            for p in range(npass):
                rec = tab[d.c_off + 80 * p: d.c_off + 80 * (p + 1)]
                assert tab[v + wb + 32 * p: v + wb + 32 * (p + 1)] == rec[:32]
                assert tab[so + 48 * p: so + 48 * (p + 1)] == rec[32:]
            if add is not None:
                a0 = ix.lut_off + N_LUT * 256 + 2048 * add
                assert tab[v + wb + 32 * npass: v + wb + 32 * npass + 2048] == tab[a0:a0 + 2048]
        else:
            d = ix.dw[i]
            for g in range(d.ngroups):
                rec = tab[d.g_off + 224 * g: d.g_off + 224 * (g + 1)]
                assert tab[v + 176 * g: v + 176 * (g + 1)] == rec[:144 + 32]
                assert tab[so + 48 * g: so + 48 * (g + 1)] == rec[144 + 32:]
    scal = sorted(ix.cs_s_off)
    assert scal[-1] + 48 * 10 - scal[0] <= 8192, "the scalar arrays stay compact (scalar-cache resident)"


def test_quantize_multiplier_agrees_with_oracle(prep, oracle):
    rng = np.random.default_rng(0)
    vals = np.concatenate([10.0 ** rng.uniform(-12, 3, 4000), [0.0, 0.5, 1.0, 0.25, 1 - 2.0**-40, 2.0**-33, 3e-12]])
    for v in vals:
        a, b, c, d = ctypes.c_int32(), ctypes.c_int(), ctypes.c_int32(), ctypes.c_int()
        prep["lib"].yf_quantize_multiplier(float(v), ctypes.byref(a), ctypes.byref(b))
        oracle.lib.yfo_quantize_multiplier(float(v), ctypes.byref(c), ctypes.byref(d))
        assert (a.value, b.value) == (c.value, d.value), v


def test_mbqm_agrees_with_oracle(prep, oracle):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.integers(-2**27, 2**27, 3000), [0, 1, -1, 2**30, -2**30]])
    for x in xs:
        m = int(rng.integers(1 << 30, (1 << 31) - 1))
        sh = int(rng.integers(-20, 1))
        assert prep["lib"].yf_mbqm(int(x), m, sh) == oracle.lib.yfo_mbqm(int(x), m, sh)


def test_index_is_embedded_and_blocks_are_aligned(prep):
    ix, tab = prep["ix"], prep["tab"]
    assert tab[:ctypes.sizeof(Index)] == bytes(ix)
    assert ctypes.sizeof(Index) <= 1024
    for d in ix.dense:
        assert d.w_off % 16 == 0 and d.c_off % 16 == 0 and d.krow % 16 == 0 and d.cout_pad4 % 4 == 0
    for d in ix.dw:
        assert d.g_off % 16 == 0
    assert ix.lut_off % 16 == 0 and ix.total_bytes >= ix.lut_off + N_LUT * 256 + 3 * 2 * 1024 + 16
    assert ix.in_zp == -128
    assert list(ix.halo_zp) == [-108, -99, -96, -94, -107, -102, -114]    # zero points of tflite tensors 52,57,64,73,80,86,95


def test_leaky_and_requant_luts_equal_oracle(prep, oracle, pack):
    ix, tab = prep["ix"], prep["tab"]
    lut = np.frombuffer(tab, np.int8, N_LUT * 256, ix.lut_off).reshape(N_LUT, 256)
    for op, lid in LEAKY_LUT_IDS.items():
        assert np.array_equal(lut[lid], oracle.leaky_lut(op)), f"LEAKY_RELU #{op}"
    T = pack["tensors"]

    def requant(t_in, t_out):
        m, sh = ctypes.c_int32(), ctypes.c_int()
        oracle.lib.yfo_quantize_multiplier(float(np.float32(T[t_in]["scale"][0])) / float(np.float32(T[t_out]["scale"][0])),
                                           ctypes.byref(m), ctypes.byref(sh))
        return np.array([np.clip(oracle.lib.yfo_mbqm(q - T[t_in]["zp"], m.value, sh.value) + T[t_out]["zp"], -128, 127)
                         for q in range(-128, 128)], np.int8)
    assert np.array_equal(np.roll(lut[3], 128), requant(58, 103))   # QUANTIZE #21, raw-indexed (index = int8 bit pattern)
    assert np.array_equal(np.roll(lut[9], 128), requant(74, 101))   # QUANTIZE #45, raw-indexed
    q44 = requant(92, 102)
    l43 = oracle.leaky_lut(43)
    assert np.array_equal(lut[15], q44[l43.astype(int) + 128])  # QUANTIZE #44 o LEAKY_RELU #43


O = 0x40000000          # YF_ACC_OFFSET: the MFMA accumulator starts at the inline constant 2.0 (bit pattern 2^30)


def _chan(tab, off, ch):
    """Channel ch of a yf_pass array (yf_tables.h): {mult2[4], zr[4], c64[4][2], rshift[4]} per 4 channels, 80 bytes.
    Returns (M, rshift, ZR, C64) of the device form."""
    base = off + 80 * (ch // 4)
    j = ch % 4
    mult2, = struct.unpack_from("<I", tab, base + 4 * j)
    zr, = struct.unpack_from("<I", tab, base + 16 + 4 * j)
    lo, hi = struct.unpack_from("<II", tab, base + 32 + 8 * j)
    rs, = struct.unpack_from("<i", tab, base + 64 + 4 * j)
    assert mult2 % 2 == 0
    return mult2 >> 1, rs, zr, lo | (hi << 32)


def _device_requant(dot, mult, rs, zr, c64):
    """The kernel's arithmetic on 32/64-bit unsigned words (yf_kernels.hip.h, rq4): the MFMA leaves acc_p = O + dot;
    {carry, N} = acc_p * 2M + C64 (v_mad_u64_u32); t = hi32(N) + ZR + carry (v_addc_co_u32); y = t >> rs (arithmetic)."""
    acc_p = O + dot
    assert 0 < acc_p < (1 << 32)
    n = acc_p * (2 * mult) + c64
    carry, n = n >> 64, n & ((1 << 64) - 1)
    assert carry in (0, 1)
    t = ((n >> 32) + zr + carry) & 0xFFFFFFFF
    t = t - (1 << 32) if t >= (1 << 31) else t
    return t >> rs


def _check_requant_identity(oracle, bias2, mult, rs, zr, c64, zp_out, abs_w, rng):
    """device form == MBQM(acc, mult, -rs) + zp_out + 128 on random accumulators and on accumulators whose first
    rounding lands exactly on a tie of the second shift (both signs: the carry stands in for TFLite's sign term)."""
    lim = 255 * abs_w
    dots = np.concatenate([rng.integers(-lim, lim + 1, 300), [0, 1, -1, lim, -lim, -bias2, -bias2 - 1, -bias2 + 1]])
    half = 1 << (rs - 1)
    for k in (-40, -3, -2, -1, 0, 1, 2, 40):
        target = k * (1 << rs) + half              # s1 value that is a tie of RoundingDivideByPOT
        a = int(round(target * 2.0**31 / mult))    # accumulator whose SRDHM is (about) that value
        dots = np.concatenate([dots, [a - bias2 + d for d in (-2, -1, 0, 1, 2)], [-a - bias2 + d for d in (-2, -1, 0, 1, 2)]])
    assert zr == (zp_out + 128) << rs
    for dot in dots:
        dot = int(dot)
        ref = oracle.lib.yfo_mbqm(dot + bias2, mult, -rs) + zp_out + 128
        assert _device_requant(dot, mult, rs, zr, c64) == ref, (dot, bias2, mult, rs)


def test_dense_tables(prep, oracle, pack):
    ix, tab, blob = prep["ix"], prep["tab"], prep["blob"]
    T, ops = pack["tensors"], pack["ops"]
    rng = np.random.default_rng(2)
    for s, op in enumerate(DENSE_OPS):
        o, d = ops[op], ix.dense[s]
        t_in = o["ins"][0] if op not in (1,) else 0
        wt, bt, to = T[o["ins"][1]], T[o["ins"][2]], T[o["out"]]
        w = wt["data"].reshape(wt["shape"]).astype(np.int64)       # OHWI
        cout = wt["shape"][0]
        assert d.cout == cout and d.cout_pad4 == (cout + 3) // 4 * 4
        zp_in = T[t_in]["zp"]
        for ch in range(cout):
            row = np.frombuffer(tab, np.int8, d.krow, d.w_off + ch * d.krow).astype(np.int64)
            wf = w[ch].reshape(-1)
            if op == 1:      # RGBX slots, three 16-byte k-steps (yf_tables.h)
                exp = np.zeros(48, np.int64)
                for ky in range(3):
                    for kx in range(3):
                        base = (ky * 3 + kx) * 4
                        exp[base:base + 3] = w[ch, ky, kx, :]
                assert np.array_equal(row, exp)
            elif op == 23:   # T14 channel order: pool [0,18) | conv [20,38)
                exp = np.zeros(48, np.int64)
                exp[:18] = wf[:18]
                exp[20:38] = wf[18:]
                assert np.array_equal(row, exp)
            else:
                assert np.array_equal(row[:wf.size], wf) and not row[wf.size:].any()
            mult, rs, zr, c64 = _chan(tab, d.c_off, ch)
            bias2 = int(bt["data"][ch]) - zp_in * int(wf.sum())
            m, sh = ctypes.c_int32(), ctypes.c_int()
            eff = float(np.float32(T[t_in]["scale"][0])) * float(np.float32(wt["scale"][ch])) / float(np.float32(to["scale"][0]))
            oracle.lib.yfo_quantize_multiplier(eff, ctypes.byref(m), ctypes.byref(sh))
            assert (mult, -rs) == (m.value, sh.value) and 1 <= rs <= 20 and mult > (1 << 30)
            assert c64 == ((bias2 - O) * 2 * mult + (1 << 31) + (((1 << (rs - 1)) - 1) << 32)) % (1 << 64)
            _check_requant_identity(oracle, bias2, mult, rs, zr, c64, to["zp"], int(np.abs(wf).sum()), rng)


def test_depthwise_tables(prep, oracle, pack):
    ix, tab = prep["ix"], prep["tab"]
    T, ops = pack["tensors"], pack["ops"]
    rng = np.random.default_rng(3)
    for s, op in enumerate(DW_OPS):
        o, d = ops[op], ix.dw[s]
        t_in = o["ins"][0]
        if ops[op - 1]["op"] == 34:            # explicit PAD in front: quantisation comes from the PAD input
            t_in = ops[op - 1]["ins"][0]
        wt, bt, to = T[o["ins"][1]], T[o["ins"][2]], T[o["out"]]
        w = wt["data"].reshape(wt["shape"]).astype(np.int64)       # 1HWC
        c = wt["shape"][3]
        assert d.c == c and d.ngroups == (c + 3) // 4
        for g in range(d.ngroups):
            base = d.g_off + g * (36 * 4 + 80)
            wd = np.frombuffer(tab, "<u4", 36, base).reshape(9, 4)
            for j in range(4):
                ch = g * 4 + j
                if ch >= c:
                    assert not wd[:, j].any()
                    continue
                taps = w[0].reshape(9, c)[:, ch]
                assert np.array_equal(wd[:, j], (taps & 255).astype(np.uint32) << (8 * j))
                mult, rs, zr, c64 = _chan(tab, base + 144, j)
                bias2 = int(bt["data"][ch]) - T[t_in]["zp"] * int(taps.sum())
                m, sh = ctypes.c_int32(), ctypes.c_int()
                eff = float(np.float32(T[t_in]["scale"][0])) * float(np.float32(wt["scale"][ch])) / float(np.float32(to["scale"][0]))
                oracle.lib.yfo_quantize_multiplier(eff, ctypes.byref(m), ctypes.byref(sh))
                assert (mult, -rs) == (m.value, sh.value) and 1 <= rs <= 20
                assert c64 == ((bias2 - O) * 2 * mult + (1 << 31) + (((1 << (rs - 1)) - 1) << 32)) % (1 << 64)
                _check_requant_identity(oracle, bias2, mult, rs, zr, c64, to["zp"], int(np.abs(taps).sum()), rng)


def test_add_tables(prep, oracle, pack):
    T, ops = pack["tensors"], pack["ops"]
    for s, op in enumerate((18, 35, 41)):
        a, o = prep["ix"].add[s], ops[op]
        t1, t2, to = T[o["ins"][0]], T[o["ins"][1]], T[o["out"]]
        assert (a.zp1, a.zp2, a.zpo) == (t1["zp"], t2["zp"], to["zp"])
        s1, s2, so = np.float32(t1["scale"][0]), np.float32(t2["scale"][0]), np.float32(to["scale"][0])
        twice = float(np.float32(2) * max(s1, s2))
        for real, (mm, ss) in ((float(s1) / twice, (a.m1, a.s1)), (float(s2) / twice, (a.m2, a.s2)),
                               (twice / float(np.float32(1 << 20) * so), (a.mo, a.so))):
            m, sh = ctypes.c_int32(), ctypes.c_int()
            oracle.lib.yfo_quantize_multiplier(real, ctypes.byref(m), ctypes.byref(sh))
            assert (mm, ss) == (m.value, sh.value) and ss <= 0
        assert a.rso == -a.so >= 1 and a.kco == (1 << (a.rso - 1)) + (a.zpo << a.rso)
        # device tables: A[q1], B[q2] and the fused final requantisation reproduce TFLite's ADD for every (q1, q2) sampled
        ix, tab = prep["ix"], prep["tab"]
        al = np.frombuffer(tab, "<i4", 3 * 512, ix.lut_off + N_LUT * 256).reshape(3, 2, 256)
        rng = np.random.default_rng(40 + s)
        for q1, q2 in np.concatenate([rng.integers(-128, 128, (400, 2)), [[-128, -128], [127, 127], [-128, 127], [t1["zp"], t2["zp"]]]]):
            q1, q2 = int(q1), int(q2)
            sa = oracle.lib.yfo_mbqm((q1 - a.zp1) * (1 << 20), a.m1, a.s1)
            sb = oracle.lib.yfo_mbqm((q2 - a.zp2) * (1 << 20), a.m2, a.s2)
            assert (al[s, 0, q1 + 128], al[s, 1, q2 + 128]) == (sa, sb + O)       # the accumulator offset rides in table B
            sm = oracle.lib.yfo_srdhm(sa + sb, a.mo)
            fused = (sm + a.kco + (sm >> 31)) >> a.rso
            ref = oracle.lib.yfo_mbqm(sa + sb, a.mo, a.so) + a.zpo
            assert fused == ref
            assert a.mo2 == 2 * a.mo and a.zro == (a.zpo + 128) << a.rso
            assert _device_requant(sa + sb, a.mo, a.rso, a.zro, a.c64o[0] | (a.c64o[1] << 32)) == ref + 128


# ---- selectable rounding (round 6): library rounding -> (oracle variant, oracle mbqm mode of dense convs, of everything else)
ROUNDINGS = {1: ("ties_up", 1, 1, 0), 2: ("ties_up_all", 2, 1, 1), 3: ("single", 4, 2, 0)}


GENERIC = 0x100          # YF_ROUND_GENERIC_KERNELS: constants for the four-instruction kernels instead of the sign-free dense form


def _prep_rounding(prep, rounding):
    lib = prep["lib"]
    lib.yf_prepare_tables_rounding.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(Index)]
    blob = (ctypes.c_uint8 * 11304).in_dll(lib, "yf_weights_blob")
    out, ix = ctypes.c_void_p(), Index()
    assert lib.yf_prepare_tables_rounding(blob, 11304, rounding, ctypes.byref(out), ctypes.byref(ix)) == 0
    return ix, bytes((ctypes.c_uint8 * ix.total_bytes).from_address(out.value))


def _device_requant3(dot, mult, rs, c64):
    """The sign-free epilogue of the kernels in namespaces yfu / yf160u (yf_kernels.hip.h, rq4 SIGNLESS): t = hi32(acc_p * 2M + C64) as a signed
    32-bit word (v_mad_u64_u32; the carry-out is not read, ZR is inside C64), y = t >> rs (v_ashrrev)."""
    t = (((O + dot) * (2 * mult) + c64) >> 32) & 0xFFFFFFFF
    t = t - (1 << 32) if t >= (1 << 31) else t
    return t >> rs


def _check_requant_identity_mode(oracle, bias2, mult, rs, zr, c64, zp_out, abs_w, rng, mode, folded=False):
    """the kernel's four instructions with a non-reference rounding's constants == the oracle's statement of that rounding + zp_out + 128, on
    random accumulators and on both signs of every kind of tie; the multiply-add must never carry out (its carry is TFLite's sign term in the
    reference form and has to stay 0 in the others).  folded: the three-instruction form (ZR inside C64, ZR field 0)."""
    lim = 255 * abs_w
    dots = np.concatenate([rng.integers(-lim, lim + 1, 200), [0, 1, -1, lim, -lim, -bias2, -bias2 - 1, -bias2 + 1]])
    half = 1 << (rs - 1)
    for k in (-40, -3, -2, -1, 0, 1, 2, 40):
        target = k * (1 << rs) + half
        a = int(round(target * 2.0**31 / mult))
        dots = np.concatenate([dots, [a - bias2 + d for d in (-2, -1, 0, 1, 2)], [-a - bias2 + d for d in (-2, -1, 0, 1, 2)]])
    assert zr == (0 if folded else (((zp_out + 128) << rs) - (1 << 31)) % (1 << 32))
    for dot in dots:
        dot = int(dot)
        acc_p = O + dot
        n = acc_p * (2 * mult) + c64
        want = oracle.lib.yfo_mbqm_mode(dot + bias2, mult, -rs, mode) + zp_out + 128
        if folded:
            assert 0 < acc_p < (1 << 32) and _device_requant3(dot, mult, rs, c64) == want, (dot, bias2, mult, rs, mode)
            continue
        assert 0 < acc_p < (1 << 32) and (1 << 61) <= n < (1 << 64), "no carry-out"
        assert _device_requant(dot, mult, rs, zr, c64) == want, (dot, bias2, mult, rs, mode)


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("rounding", sorted(ROUNDINGS))
def test_rounding_modes_are_other_constants_in_the_same_layout(prep, oracle, pack, rounding, generic):
    """yf_network_set_requant_rounding: every rounding is a table blob of the SAME layout (the kernels address it at compiled-in offsets) whose
    weights are the reference blob's, and whose {C64, ZR} per channel, byte LUTs and add tables state the rounding the oracle's variant states:
    dense convs by the variant's dense form, depthwise convs / LEAKY_RELU / QUANTIZE / ADD by its form for everything else.  The dense stages'
    constants come in two forms: FOLDED (default: ZR inside C64, for the kernels whose dense epilogue is three instructions -- the engine launches
    those, yf_rounding_signless_dense) and, with YF_ROUND_GENERIC_KERNELS, the form the four-instruction kernels take for any rounding."""
    name, variant, m_dense, m_other = ROUNDINGS[rounding]
    ix, tab = _prep_rounding(prep, rounding | (GENERIC if generic else 0))
    lib = prep["lib"]
    assert lib.yf_rounding_signless_dense(rounding) == 1 and lib.yf_rounding_signless_dense(rounding | GENERIC) == 0 and lib.yf_rounding_signless_dense(0) == 0
    ix0, tab0 = prep["ix"], prep["tab"]
    skip = Index.add.offset, Index.add.offset + Index.add.size                     # the add records carry rounding-dependent constants (checked below)
    assert bytes(ix)[:skip[0]] == bytes(ix0)[:skip[0]] and bytes(ix)[skip[1]:] == bytes(ix0)[skip[1]:] and len(tab) == len(tab0)
    T, ops = pack["tensors"], pack["ops"]
    rng = np.random.default_rng(60 + rounding)
    n_changed = 0
    for s, op in enumerate(DENSE_OPS):
        o, d = ops[op], ix.dense[s]
        t_in = o["ins"][0] if op != 1 else 0
        wt, bt, to = T[o["ins"][1]], T[o["ins"][2]], T[o["out"]]
        w = wt["data"].reshape(wt["shape"]).astype(np.int64)
        assert tab[d.w_off:d.w_off + d.cout_pad4 * d.krow] == tab0[d.w_off:d.w_off + d.cout_pad4 * d.krow]
        for ch in range(wt["shape"][0]):
            mult, rs, zr, c64 = _chan(tab, d.c_off, ch)
            assert (mult, rs) == _chan(tab0, d.c_off, ch)[:2]
            wf = w[ch].reshape(-1)
            bias2 = int(bt["data"][ch]) - T[t_in]["zp"] * int(wf.sum())
            tail = (1 << 63) if generic else (((to["zp"] + 128) << rs) << 32)
            if m_dense == 1:
                assert c64 == ((bias2 - O) * 2 * mult + (1 << 31) + ((1 << (rs - 1)) << 32) + tail) % (1 << 64)
            else:
                assert c64 == ((bias2 - O) * 2 * mult + (1 << (31 + rs)) + tail) % (1 << 64)
            _check_requant_identity_mode(oracle, bias2, mult, rs, zr, c64, to["zp"], int(np.abs(wf).sum()), rng, m_dense, folded=not generic)
            n_changed += 1
    for s, op in enumerate(DW_OPS):
        o, d = ops[op], ix.dw[s]
        t_in = ops[op - 1]["ins"][0] if ops[op - 1]["op"] == 34 else o["ins"][0]
        wt, bt, to = T[o["ins"][1]], T[o["ins"][2]], T[o["out"]]
        w = wt["data"].reshape(wt["shape"]).astype(np.int64)
        c = wt["shape"][3]
        for g in range(d.ngroups):
            base = d.g_off + g * (36 * 4 + 80)
            assert tab[base:base + 144] == tab0[base:base + 144]
            for j in range(4):
                ch = g * 4 + j
                if ch >= c:
                    continue
                mult, rs, zr, c64 = _chan(tab, base + 144, j)
                taps = w[0].reshape(9, c)[:, ch]
                bias2 = int(bt["data"][ch]) - T[t_in]["zp"] * int(taps.sum())
                if m_other == 0:
                    assert (mult, rs, zr, c64) == _chan(tab0, base + 144, j)            # untouched by a dense-only rounding
                else:
                    _check_requant_identity_mode(oracle, bias2, mult, rs, zr, c64, to["zp"], int(np.abs(taps).sum()), rng, m_other)
    assert n_changed == 338                                                        # dense output channels (206 depthwise: 544 in all)
    # byte LUTs: the oracle's tables for the variant; a dense-only rounding leaves all 19 and the add tables as they were
    lut = np.frombuffer(tab, np.int8, N_LUT * 256, ix.lut_off).reshape(N_LUT, 256)
    for op, lid in LEAKY_LUT_IDS.items():
        assert np.array_equal(lut[lid], oracle.leaky_lut(op, variant)), f"LEAKY_RELU #{op}"
    a0 = ix.lut_off + N_LUT * 256
    if m_other == 0:
        assert tab[ix.lut_off:a0 + 3 * 2048 + 256] == tab0[ix.lut_off:a0 + 3 * 2048 + 256]
        assert bytes(ix.add) == bytes(ix0.add)
        return
    assert not np.array_equal(lut, np.frombuffer(tab0, np.int8, N_LUT * 256, ix.lut_off).reshape(N_LUT, 256))

    def requant(t_in, t_out):
        m, sh = ctypes.c_int32(), ctypes.c_int()
        oracle.lib.yfo_quantize_multiplier(float(np.float32(T[t_in]["scale"][0])) / float(np.float32(T[t_out]["scale"][0])), ctypes.byref(m), ctypes.byref(sh))
        return np.array([np.clip(oracle.lib.yfo_mbqm_mode(q - T[t_in]["zp"], m.value, sh.value, m_other) + T[t_out]["zp"], -128, 127) for q in range(-128, 128)], np.int8)
    assert np.array_equal(np.roll(lut[3], 128), requant(58, 103)) and np.array_equal(np.roll(lut[9], 128), requant(74, 101))
    assert np.array_equal(lut[15], requant(92, 102)[oracle.leaky_lut(43, variant).astype(int) + 128])
    al = np.frombuffer(tab, "<i4", 3 * 512, a0).reshape(3, 2, 256)
    for s in range(3):
        a = ix.add[s]
        for q1, q2 in np.concatenate([np.random.default_rng(70 + s).integers(-128, 128, (300, 2)), [[-128, -128], [127, 127], [-128, 127]]]):
            q1, q2 = int(q1), int(q2)
            sa = oracle.lib.yfo_mbqm_mode((q1 - a.zp1) * (1 << 20), a.m1, a.s1, m_other)
            sb = oracle.lib.yfo_mbqm_mode((q2 - a.zp2) * (1 << 20), a.m2, a.s2, m_other)
            assert (al[s, 0, q1 + 128], al[s, 1, q2 + 128]) == (sa, sb + O)
            ref = oracle.lib.yfo_mbqm_mode(sa + sb, a.mo, a.so, m_other) + a.zpo
            assert _device_requant(sa + sb, a.mo, a.rso, a.zro, a.c64o[0] | (a.c64o[1] << 32)) == ref + 128


def test_prepare_rejects_bad_arguments(prep):
    lib = prep["lib"]
    out, ix = ctypes.c_void_p(), Index()
    small = (ctypes.c_uint8 * 100)()
    assert lib.yf_prepare_tables(small, 100, ctypes.byref(out), ctypes.byref(ix)) == 1
    assert lib.yf_prepare_tables(None, 11304, ctypes.byref(out), ctypes.byref(ix)) == 1
    lib.yf_prepare_tables_rounding.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(Index)]
    blob = (ctypes.c_uint8 * 11304).in_dll(lib, "yf_weights_blob")
    for bad in (-1, 4, 99, 0x104, 0x200):
        assert lib.yf_prepare_tables_rounding(blob, 11304, bad, ctypes.byref(out), ctypes.byref(ix)) == 1
    assert lib.yf_prepare_tables_rounding(blob, 11304, 0x100, ctypes.byref(out), ctypes.byref(ix)) == 0      # reference rounding, generic kernels: the default blob
    assert bytes((ctypes.c_uint8 * ix.total_bytes).from_address(out.value)) == prep["tab"]


def test_fp16_prefetch_wait_count_matches_the_isa(tmp_path):
    """yf_fp16.hip: the barrier behind a stage that issued the next frame's input prefetch waits with s_waitcnt vmcnt(N), N = the number
    of prefetch load instructions per thread -- it must cover the stage's weight DMA (older) and may leave the N prefetch loads (younger) in
    flight.  Fewer than N vector loads between the DMA and that wait would leave the DMA unwaited.  The ISA is checked, not assumed."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "stm32h7-yolo_amd", "csrc")
    flags = [ln.split("=", 1)[1].split() for ln in open(os.path.join(csrc, "flags.mk")) if ln.startswith("FP16FLAGS_BASE")][0]
    asm = tmp_path / "f16.s"
    subprocess.check_call([hipcc, *flags, "-S", "--cuda-device-only", os.path.join(csrc, "yf_fp16.hip"), "-o", str(asm)], stderr=subprocess.DEVNULL)
    lines = [ln.strip() for ln in open(asm) if ln.strip() and not ln.strip().startswith(";")]
    waits = [i for i, ln in enumerate(lines) if re.match(r"s_waitcnt vmcnt\([1-9]\d*\)", ln) and "lgkmcnt" not in ln]
    checked = 0
    for i in waits:
        n = int(re.search(r"vmcnt\((\d+)\)", lines[i]).group(1))
        nxt = next(k for k in range(i, len(lines)) if lines[k].startswith("s_barrier") or lines[k].startswith("v_") or lines[k].startswith("ds_"))
        if not lines[nxt].startswith("s_barrier"):
            continue                                        # a compiler-placed counted wait in front of a use, not the stage barrier
        dma = max(k for k in range(i) if lines[k].startswith("global_load_lds_dwordx4"))
        loads = [ln for ln in lines[dma + 1:i] if re.match(r"(global|flat|buffer)_load_dword", ln)]
        stores = [ln for ln in lines[dma + 1:i] if re.match(r"(global|flat|buffer|scratch)_store", ln)]
        assert len(loads) >= n and not stores, (n, loads, stores)
        checked += 1
    assert checked >= 1


def test_profile_summary_separates_overlapped_launches_from_back_to_back_ones(tmp_path):
    """tools/summarize_pmc.py (what profiles/*_pmc_summary.json come from): a launch counts as OVERLAPPED only against a neighbour on ANOTHER stream (or by more than a
    microsecond) -- back-to-back launches of one stream show the next start a few hundred nanoseconds before the previous end stamp, and round 6's first summary filed 728
    of them under "overlapped", which moved the "kernel alone" average.  Synthetic trace: ten launches back to back on stream 4 (each starting 150 ns early), then six
    alternating between streams 5 and 6 with real overlap, plus a small-grid launch that must not count."""
    import json
    import subprocess
    import sys
    d = tmp_path / "prof"
    (d / "trace").mkdir(parents=True)
    rows, t = [], 1_000_000
    name = "void yf::yoloface56_fused<2, 8, false, false>(yf::NetParams)"
    for k in range(10):                                   # one stream, back to back: start 150 ns before the previous end
        rows.append((t, t + 140_000, 4, 262144)); t += 140_000 - 150
    t += 1_000_000
    for k in range(6):                                    # two streams: each launch starts 130 us before the previous one ends
        rows.append((t, t + 260_000, 5 + k % 2, 262144)); t += 130_000
    rows.append((t + 2_000_000, t + 2_024_000, 4, 512))   # a one-frame launch of the same kernel: not a full batch
    with open(d / "trace" / "t_kernel_trace.csv", "w") as f:
        f.write("Kind,Agent_Id,Queue_Id,Stream_Id,Kernel_Name,Start_Timestamp,End_Timestamp,Grid_Size_X\n")
        for b, e, s, g in rows:
            f.write(f'KERNEL_DISPATCH,1,1,{s},"{name}",{b},{e},{g}\n')
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), str(d)], stdout=subprocess.DEVNULL)
    res = json.load(open(d / "summary.json"))
    assert res["trace"]["calls"] == 10 and res["trace"]["avg_ns"] == 140_000
    assert res["trace_overlapped"]["calls"] == 6 and res["trace_overlapped"]["runs"] == 1 and res["trace_overlapped"]["avg_duration_ns"] == 260_000
    assert abs(res["trace_overlapped"]["wall_ns_per_launch"] - (5 * 130_000 + 260_000) / 6) < 1
