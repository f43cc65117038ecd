"""CPU tests of the oracle (test infrastructure) -- PARITY UNPINNED against the real TFLite interpreter; these are
the partial pins SURVEY.md 8(c) lists plus self-consistency checks."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, has_reference

LEAKY_OPS = [2, 4, 7, 11, 14, 16, 20, 24, 28, 31, 33, 37, 39, 43, 48, 50, 52]


def test_fixed_point_primitives_known_answers(oracle):
    lib = oracle.lib
    # SURVEY.md Appendix A.3 worked example (conv2d_1's LeakyReLU alpha branch).  TFLite evaluates
    # s_in*alpha/s_out in float32 and widens it (activations.cc LeakyReluPrepare) -> (1460035712, -2); the survey
    # quotes the double-evaluated (1460035715, -2).  Both give the same table; the worked value is unchanged.
    s_in, s_out, alpha = np.float32(0.11715607345104218), np.float32(0.06892728805541992), np.float32(0.1)
    m, sh = ctypes.c_int32(), ctypes.c_int()
    lib.yfo_quantize_multiplier(float(np.float32(s_in * alpha / s_out)), ctypes.byref(m), ctypes.byref(sh))
    assert (m.value, sh.value) == (1460035712, -2)
    lib.yfo_quantize_multiplier(float(s_in) * float(alpha) / float(s_out), ctypes.byref(m), ctypes.byref(sh))
    assert (m.value, sh.value) == (1460035715, -2)
    for mult in (1460035712, 1460035715):
        assert lib.yfo_srdhm(-114, mult) == -78
        assert lib.yfo_mbqm(-114, mult, -2) == -20
    assert lib.yfo_rdivpot(-78, 2) == -20
    # gemmlowp corner cases
    assert lib.yfo_srdhm(-2**31, -2**31) == 2**31 - 1
    assert lib.yfo_srdhm(1, 2**30) == 1            # +0.5 rounds up
    assert lib.yfo_srdhm(-1, 2**30) == 0           # -0.5 rounds up (toward +inf)
    assert lib.yfo_srdhm(-3, 2**30) == -1          # -1.5 -> -1
    assert lib.yfo_rdivpot(-2, 2) == -1            # -0.5 rounds away from zero
    assert lib.yfo_rdivpot(2, 2) == 1
    assert lib.yfo_rdivpot(-1, 2) == 0
    assert lib.yfo_rdivpot(5, 0) == 5
    lib.yfo_quantize_multiplier(0.0, ctypes.byref(m), ctypes.byref(sh))
    assert (m.value, sh.value) == (0, 0)
    lib.yfo_quantize_multiplier(0.5, ctypes.byref(m), ctypes.byref(sh))
    assert (m.value, sh.value) == (1 << 30, 0)
    lib.yfo_quantize_multiplier(1.0 - 2.0**-40, ctypes.byref(m), ctypes.byref(sh))   # rounds up to 2^31 -> renormalised
    assert (m.value, sh.value) == (1 << 30, 1)


def test_model_pack_shapes(oracle):
    assert oracle.num_ops == 54
    assert oracle.out_shape(56, 56) == (7, 7, 18)
    assert oracle.dump_bytes(56, 56) == 196199          # SURVEY.md Appendix A total
    assert oracle.out_shape(160, 160) == (20, 20, 18)   # fully convolutional (config 5)


def test_c_oracle_equals_numpy_restatement(oracle):
    from oracle.np_restatement import NpModel
    npm = NpModel(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    rng = np.random.default_rng(7)
    x = rng.integers(-128, 128, (3, 56, 56, 3), dtype=np.int8)
    x[1] = -128
    heads, dump = oracle.run(x, dump=True)
    for f in range(3):
        h2, outs = npm.run(x[f], dump=True)
        assert np.array_equal(h2, heads[f])
        assert np.array_equal(np.concatenate([o.reshape(-1) for o in outs]), dump[f])


def test_c_oracle_equals_numpy_restatement_on_extreme_frames(oracle):
    """The two independent restatements (C, numpy) on frames that push accumulators and requantisation to their edges: all +127, the corner colour
    (+127, -128, +127), a period-1 checkerboard, a period-2 row stripe, a cold pixel in the top-left corner of a hot frame, random +127 / -128 pixels.
    Every one of the 54 op outputs must agree, not only the head."""
    from oracle.np_restatement import NpModel
    npm = NpModel(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    yy, xx = np.mgrid[0:56, 0:56]
    frames = [np.full((56, 56, 3), 127, np.int8), np.broadcast_to(np.array([127, -128, 127], np.int8), (56, 56, 3)).copy(),
              np.broadcast_to(np.where(((xx + yy) % 2)[..., None] == 1, 127, -128).astype(np.int8), (56, 56, 3)).copy(),
              np.broadcast_to(np.where(((yy // 2) % 2)[..., None] == 1, 127, -128).astype(np.int8), (56, 56, 3)).copy()]
    f = np.full((56, 56, 3), 127, np.int8)
    f[0, 0] = -128
    frames += [f, np.where(np.random.default_rng(11).integers(0, 2, (56, 56, 3)) == 1, 127, -128).astype(np.int8)]
    x = np.stack(frames)
    heads, dump = oracle.run(x, dump=True)
    for k in range(x.shape[0]):
        h2, outs = npm.run(x[k], dump=True)
        assert np.array_equal(h2, heads[k]), k
        assert np.array_equal(np.concatenate([o.reshape(-1) for o in outs]), dump[k]), k


def test_c_oracle_equals_numpy_restatement_other_size(oracle):
    """Ragged / non-56 input (80x72): exercises SAME/VALID shape inference and the pool borders."""
    from oracle.np_restatement import NpModel
    npm = NpModel(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    x = np.random.default_rng(3).integers(-128, 128, (1, 80, 72, 3), dtype=np.int8)
    head = oracle.run(x)
    assert head.shape == (1, 10, 9, 18)
    assert np.array_equal(npm.run(x[0]), head[0])


def test_st_luts_are_float_rounded_leaky_and_differ_from_tflite(oracle):
    """Known-answer data from the reference (network.c:2218..2902): ST's 17 LUTs equal round(float leaky) exactly
    (pins the quant params and the float formula) and differ from TFLite's fixed-point LEAKY_RELU in 11-22
    entries per layer, all by 1 LSB on the negative branch."""
    from oracle.np_restatement import load_yfm
    st = np.fromfile(os.path.join(GOLDEN, "st_leaky_luts.bin"), np.int8).reshape(17, 256)
    m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    q = np.arange(-128, 128)
    for k, op in enumerate(LEAKY_OPS):
        o = m["ops"][op]
        ti, to = m["tensors"][o["ins"][0]], m["tensors"][o["out"]]
        real = (q - ti["zp"]).astype(np.float64) * float(ti["scale"][0])
        real = np.where(real >= 0, real, real * float(np.float32(o["alpha"])))
        flt = np.clip(np.round(real / float(to["scale"][0])) + to["zp"], -128, 127).astype(np.int8)
        assert np.array_equal(flt, st[k]), f"ST LUT of op {op} is not the float-rounded leaky"
        tfl = oracle.leaky_lut(op)
        diff = tfl.astype(int) - st[k].astype(int)
        assert 11 <= np.count_nonzero(diff) <= 22
        assert np.abs(diff).max() == 1
        assert np.all((q - ti["zp"])[diff != 0] < 0)
    # the documented example: conv2d_1, q=-125 -> TFLite -128, ST -127
    assert oracle.leaky_lut(2)[-125 + 128] == -128 and st[0][-125 + 128] == -127


@pytest.mark.skipif(not has_reference(), reason="container only: needs /root/reference")
def test_fixtures_match_reference_data_files():
    """Weights blob / LUT fixture regenerate identically from the reference's data (tools/gen_model.py asserts the
    blob against network_data.c).  The generator writes into a temporary directory: the tree (and the mtimes the
    library's Makefile looks at) stays untouched."""
    import subprocess
    import sys
    import tempfile
    files = ("oracle/model/yoloface_int8.yfm", "tests/golden/st_leaky_luts.bin", "tests/golden/decode_tables_f32.bin",
             "stm32h7-yolo_amd/csrc/gen/yf_model_gen.h", "stm32h7-yolo_amd/csrc/gen/yf_weights_blob_gen.c",
             "stm32h7-yolo_amd/csrc/gen/yf_decode_tables_gen.h")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_model.py"), "--out-root", tmp], stdout=subprocess.DEVNULL)
        for f in files:
            assert open(os.path.join(tmp, f), "rb").read() == open(os.path.join(ROOT, f), "rb").read(), f


def test_golden_vectors(oracle, golden):
    heads, dump = oracle.run(golden["inputs"], dump=True)
    assert np.array_equal(heads, golden["heads"])
    assert np.array_equal(dump[0], golden["dump0"])
    from oracle.np_restatement import load_yfm
    m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    sizes = [int(np.prod(m["tensors"][o["out"]]["shape"][1:])) for o in m["ops"]]
    for f, fr in enumerate(golden["meta"]["frames"]):
        off = 0
        for n, sha in zip(sizes, fr["op_sha256_16"]):
            assert hashlib.sha256(dump[f, off:off + n].tobytes()).hexdigest()[:16] == sha
            off += n
        py = oracle.decode_py(heads[f], f)
        assert [(d[1], d[2], d[3], d[4], d[6], d[7], d[8], d[9]) for d in py] == \
               [(g["anchor"], g["row"], g["col"], g["q_conf"], g["x1"], g["y1"], g["x2"], g["y2"]) for g in fr["detections_py"]]
        fw = oracle.decode_c(heads[f], f)
        assert [(d[1], d[2], d[3], d[6], d[7], d[8], d[9]) for d in fw] == \
               [(g["anchor"], g["row"], g["col"], g["x1"], g["y1"], g["x2"], g["y2"]) for g in fr["detections_fw"]]


def test_int8_graph_tracks_float_graph(oracle, golden):
    """Structural sanity: the int8 evaluation stays within a few LSB of a float64
    evaluation of the dequantised graph -- a padding-side / concat-order / channel-order slip shows as a jump."""
    from oracle.np_restatement import NpModel
    npm = NpModel(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    x = golden["inputs"][5]                       # the real image
    f = npm.run_float(x)
    q = (oracle.run(x[None])[0].astype(np.float64) + 15) * 0.14218327403068542
    err = np.abs(f - q) / 0.14218327403068542
    assert err.mean() < 2.5 and err.max() < 25


def test_int8_oracle_tracks_the_references_fp32_onnx_on_its_sample_images(oracle):
    """Cross-artifact pin.  The reference ships TWO exports of the trained network: yoloface_int8.tflite (the int8 graph
    the oracle restates) and yoloface-50k.onnx (fp32, reference yoloface/pytorch/).  On the reference's own 27 sample
    images the dequantised int8 head must follow an independent fp32 numpy evaluation of the ONNX weights: Pearson
    r > 0.95 per image and the most confident (cell, anchor) identical on most images.  A wrong padding side, concat
    order, channel order or requantisation constant in the restatement destroys this agreement (PTQ noise alone gives
    r ~ 0.97).  It does not prove bit-exactness with the TFLite interpreter (parity stays 'unpinned')."""
    from oracle.np_fp32 import load_yfw, run_fp32
    frames = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    assert frames.shape[0] == 27
    convs = load_yfw(os.path.join(ROOT, "stm32h7-yolo_amd", "model", "yoloface_fp32.yfw"))
    heads = oracle.run(frames)
    same_top, rs = 0, []
    for x, h in zip(frames, heads):
        h8 = (h.astype(np.float64) + 15) * 0.14218327403068542
        hf = run_fp32(convs, (x.astype(np.int16) + 128).astype(np.float32) / 255.0)
        rs.append(np.corrcoef(h8.ravel(), hf.ravel())[0, 1])
        c8, cf = h8.reshape(7, 7, 3, 6)[..., 4], hf.reshape(7, 7, 3, 6)[..., 4]
        same_top += int(c8.argmax() == cf.argmax())
    assert min(rs) > 0.95, rs
    assert same_top >= 14, same_top                       # 18 of 27 when this test was written


# ---- rounding variants (round 6) --------------------------------------------------------------------------------------
def test_rounding_variants_agree_between_the_two_restatements(oracle):
    """Every variant of the requantisation rounding (oracle/yf_oracle.h YFO_RV_*) is stated twice -- C and numpy -- and the two must agree on
    all 54 op outputs, on random frames and on two of the reference's sample images; the primitives on known answers."""
    from oracle.np_restatement import NpModel
    from oracle.oracle import VARIANTS
    lib = oracle.lib
    half = 1 << 30                                                               # multiplier 0.5
    for x, sh, want in ((-6, -2, (-1, -1, -1)),      # -0.75: -1 in every form
                        (-4, -2, (-1, 0, 0)),        # -0.5 exactly: away from zero | upward | single rounding also upward
                        (4, -2, (1, 1, 1)),          # +0.5: up in every form
                        (-3, -1, (-1, 0, -1)),       # -0.75 as -1.5 -> -1 (first rounding, upward) -> -0.5: double rounding lands on a tie, single does not
                        (-5, -1, (-1, -1, -1))):     # -1.25
        assert tuple(lib.yfo_mbqm_mode(x, half, sh, mode) for mode in (0, 1, 2)) == want, (x, sh)
    npm = NpModel(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    real = np.fromfile(os.path.join(GOLDEN, "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    x = np.concatenate([np.random.default_rng(7).integers(-128, 128, (2, 56, 56, 3), dtype=np.int8), real[:2]])
    ref = oracle.run(x)
    for name, v in VARIANTS.items():
        heads, dump = oracle.run(x, dump=True, variant=v)
        assert (name == "R") == np.array_equal(heads, ref), name             # every other variant really is another function
        for f in range(x.shape[0]):
            h2, outs = npm.run(x[f], dump=True, variant=v)
            assert np.array_equal(np.concatenate([o.reshape(-1) for o in outs]), dump[f]), (name, f)


# What changes when the ONE unverifiable choice under every parity claim is made differently.  Pinned: a change of the oracle, of the
# fixtures or of a variant's definition shows here.  Per input set and variant: head bytes that differ from (R), max |delta|, frames with any
# differing head byte, frames whose Python-decode box LIST differs, of those the ones that differ in coordinates only (same cells fire).
EXPOSURE = {
    "real27": dict(frames=27, boxes=40,
                   U=(9987, 9, 27, 10, 9), U_all=(12683, 22, 27, 17, 13), X=(10665, 9, 27, 11, 10), S=(10366, 10, 27, 13, 12)),
    "golden6": dict(frames=6, boxes=2,
                    U=(1838, 8, 6, 0, 0), U_all=(2821, 18, 6, 1, 1), X=(2199, 8, 6, 1, 1), S=(2059, 8, 6, 0, 0)),
    "seeded4096": dict(frames=4096, boxes=608,
                       U=(1436229, 11, 4096, 384, 183), U_all=(1843024, 20, 4096, 492, 114), X=(1548997, 11, 4096, 417, 209), S=(1481192, 10, 4096, 395, 194)),
}


def rounding_exposure(oracle, frames, threads=8):
    """(boxes under R, {variant: (head bytes differing, max |delta|, frames with a differing byte, frames whose box list differs, ... in coordinates only)})"""
    from oracle.oracle import VARIANTS

    def boxes(h, f):
        return [(d[1], d[2], d[3], d[6], d[7], d[8], d[9]) for d in oracle.decode_py(h, f)]
    ref = oracle.run(frames, threads=threads)
    ref_boxes = [boxes(ref[f], f) for f in range(len(frames))]
    out = {}
    for name, v in VARIANTS.items():
        if name == "R":
            continue
        h = oracle.run(frames, threads=threads, variant=v)
        d = h.astype(np.int32) - ref.astype(np.int32)
        lists = [boxes(h[f], f) for f in range(len(frames))]
        changed = [f for f in range(len(frames)) if lists[f] != ref_boxes[f]]
        coords = [f for f in changed if [b[:3] for b in lists[f]] == [b[:3] for b in ref_boxes[f]]]
        out[name.replace("-", "_")] = (int(np.count_nonzero(d)), int(np.abs(d).max()), int(np.count_nonzero(np.any(d.reshape(len(frames), -1) != 0, axis=1))),
                                       len(changed), len(coords))
    return sum(len(b) for b in ref_boxes), out


def test_rounding_variant_exposure(oracle, golden):
    """the oracle restates TFLite's builtin REFERENCE kernels (ties away from zero); the reference's script
    (yoloface/tflite/tflite_prediction.py:23) builds its interpreter with default arguments = the default resolver, whose per-channel int8
    CONV_2D goes through ruy (right shift ties UPWARD, variant U; single rounding on its portable path, S) -- or through XNNPACK (fp32
    requantisation, X) where that delegate is on.  None can be run here.  This test MEASURES the distance on the reference's 27 sample images,
    the six golden frames and 4096 seeded frames (BASELINE configs[1]'s input): about 40 % of the head bytes change (by up to 9-11 LSB),
    and the Python box list changes on 10 of the 27 real frames (coordinates only on 9, a box appears or disappears on 1).  So "bit-exact
    vs tflite" is exact against the variant one names -- which is why the library's rounding is selectable (yf_network_set_requant_rounding)
    and DESIGN.md section 2 carries this table."""
    sets = {"real27": np.fromfile(os.path.join(GOLDEN, "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3),
            "golden6": golden["inputs"],
            "seeded4096": np.random.default_rng(1).integers(-128, 128, (4096, 56, 56, 3), dtype=np.int8)}
    for name, frames in sets.items():
        want = EXPOSURE[name]
        assert frames.shape[0] == want["frames"]
        n_boxes, got = rounding_exposure(oracle, frames)
        assert n_boxes == want["boxes"], name
        for v in ("U", "U_all", "X", "S"):
            assert got[v] == want[v], (name, v, got[v])


def test_golden_vectors_of_the_rounding_variants(oracle, golden):
    """tests/golden/golden_heads_variants.npz + golden_variants_meta.json (make_golden_variants.py): every variant's heads on the golden frames bit for bit, the
    sha256 of every head of the 27 real frames, the Python-decode boxes of the golden frames.  Interpreter-unverified like every golden here; what they buy: a
    refactor of a variant is pinned, and an interpreter run elsewhere has something to be compared with."""
    import json
    from oracle.oracle import VARIANTS
    want = np.load(os.path.join(GOLDEN, "golden_heads_variants.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "golden_variants_meta.json")))["variants"]
    real = np.fromfile(os.path.join(GOLDEN, "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    for name, v in VARIANTS.items():
        if name == "R":
            continue
        key = name.replace("-", "_")
        h = oracle.run(golden["inputs"], variant=v)
        assert np.array_equal(h, want[key]), name
        assert [hashlib.sha256(a.tobytes()).hexdigest() for a in oracle.run(real, variant=v)] == meta[key]["real_frames_head_sha256"], name
        boxes = [[[int(d[1]), int(d[2]), int(d[3]), int(d[6]), int(d[7]), int(d[8]), int(d[9])] for d in oracle.decode_py(h[f], f)] for f in range(h.shape[0])]
        assert boxes == meta[key]["golden_detections_py"], name


def test_which_tflite_names_the_variant(tmp_path):
    """tools/which_tflite.py (what an integrator with TensorFlow runs to settle the rounding question): the heads of each variant are recognised, foreign heads are not."""
    import subprocess
    import sys
    v = np.load(os.path.join(GOLDEN, "golden_heads_variants.npz"))
    tool = os.path.join(ROOT, "tools", "which_tflite.py")
    for name, arr in (("R", np.fromfile(os.path.join(GOLDEN, "golden_heads.bin"), np.int8)), ("U", v["U"]), ("S", v["S"])):
        p = tmp_path / f"{name}.bin"
        arr.tofile(p)
        r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
        assert r.returncode == 0 and f"computes variant {name}:" in r.stdout, r.stdout
    bad = v["U"].copy()
    bad[0, 0, 0, 0] ^= 1
    bad.tofile(tmp_path / "bad.bin")
    r = subprocess.run([sys.executable, tool, str(tmp_path / "bad.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "NO variant matches" in r.stdout


def test_decode_threshold_identity(oracle):
    """conf > 0.7 (py) and conf >= 0.7 (firmware) are both equivalent to q_conf >= -9."""
    sig = oracle.sig
    assert sig[-9 + 128] > 0.7 and sig[-10 + 128] < 0.7
    head = np.full((7, 7, 18), -128, np.int8)
    head[3, 4, 6 + 4] = -9
    head[0, 0, 4] = -10
    py = oracle.decode_py(head)
    assert len(py) == 1 and py[0][1:4] == (1, 3, 4)
    fw = oracle.decode_c(head)
    assert len(fw) == 1 and fw[0][1:4] == (1, 3, 4)


def test_decode_py_matches_numpy_mirror(oracle, golden, yf):
    import importlib
    ip = importlib.import_module("stm32h7-yolo_amd.interpreter")
    rng = np.random.default_rng(11)
    for k in range(20):
        head = rng.integers(-40, 30, (7, 7, 18), dtype=np.int8)
        ref = ip.decode_boxes(head, 0.7, 410 / 56.0, 362 / 56.0)
        got = oracle.decode_py(head, 0, 410 / 56.0, 362 / 56.0)
        assert [tuple(int(v) for v in b) for b in ref] == [(d[6], d[7], d[8], d[9]) for d in got]


def test_prepare_rgb565(oracle):
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, 112 * 112 * 2, dtype=np.uint8)
    out = oracle.prepare_rgb565(raw)
    px = (raw[0::2].astype(np.uint16) << 8 | raw[1::2]).reshape(112, 112)
    r, g, b = (px >> 11) & 31, (px >> 5) & 63, px & 31
    box = lambda c: (c[0::2, 0::2].astype(int) + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2]) >> 2  # noqa: E731
    exp = np.stack([(box(r) << 3) - 128, (box(g) << 2) - 128, (box(b) << 3) - 128], axis=-1).astype(np.int8)
    assert np.array_equal(out, exp)


def test_threads_do_not_change_results(oracle):
    x = np.random.default_rng(9).integers(-128, 128, (37, 56, 56, 3), dtype=np.int8)
    assert np.array_equal(oracle.run(x, threads=1), oracle.run(x, threads=5))


def test_empty_batch(oracle):
    assert oracle.run(np.zeros((0, 56, 56, 3), np.int8)).shape == (0, 7, 7, 18)
