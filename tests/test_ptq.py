"""SURVEY.md 8(f)4 -- post-training quantisation (reference yoloface/tflite/tflite_quantize.py:29-96).
The reference ships BOTH the float model (yoloface/tflite/yoloface.tflite) and its int8 quantisation
(yoloface_int8.tflite); stm32h7-yolo_amd/ptq.py restates the converter's rules and is checked against that pair:
weights and biases exactly, activation ranges within a tolerance (the calibration images are resized with PIL here,
with OpenCV in the reference)."""
import importlib
import os

import numpy as np
import pytest

from conftest import ROOT

CONV, DWCONV, MAXPOOL, PAD, QUANTIZE = 3, 4, 17, 34, 114


@pytest.fixture(scope="module")
def ptq():
    return importlib.import_module("stm32h7-yolo_amd.ptq")


@pytest.fixture(scope="module")
def models():
    from oracle.np_restatement import NpModel
    npm = NpModel(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    z = np.load(os.path.join(ROOT, "tests", "golden", "ptq_float_convs.npz"))
    convs = [(z[f"w{k}"], z[f"b{k}"], bool(z[f"dw{k}"])) for k in range(24)]
    return npm, convs


def test_int8_weights_and_biases_are_the_ptq_of_the_float_model(ptq, models):
    """All 9126 int8 weights, 544 float32 scales and 544 int32 biases of yoloface_int8.tflite are reproduced
    bit for bit from the float weights of yoloface.tflite."""
    npm, convs = models
    T = npm.m["tensors"]
    ops = [o for o in npm.m["ops"] if o["op"] in (CONV, DWCONV)]
    assert len(ops) == 24
    n_w = n_b = 0
    for o, (w, b, dw) in zip(ops, convs):
        assert dw == (o["op"] == DWCONV)
        wt, bt, it = T[o["ins"][1]], T[o["ins"][2]], T[o["ins"][0]]
        q, scale = ptq.quantize_conv_weights(w, 3 if dw else 0)
        assert np.array_equal(scale, wt["scale"])
        assert np.array_equal(q.ravel(), wt["data"])
        assert np.array_equal(ptq.quantize_bias(b, it["scale"][0], scale), bt["data"])
        n_w += q.size
        n_b += b.size
    assert n_w == 9126 and n_b == 544


def test_activation_ranges_from_the_calibration_set(ptq, models):
    """Calibrating on the reference's representative dataset (small_dataset/*.jpg -> tests/golden/real_frames_56.bin)
    with a float evaluation of the graph gives the input parameters exactly (1/255, -128) and the int8 model's activation
    scales to a few percent from conv2d_17 on (median 2.6 %); the first layers come out 10-27 % narrower because the
    fixture was resized with PIL (antialiased) and the reference resizes with OpenCV (not antialiased)."""
    npm, convs = models
    T, ops = npm.m["tensors"], npm.m["ops"]
    frames = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    cal = ptq.Calibrator()
    for x in frames:
        npm.run_float(x, float_convs=[(w, b) for w, b, _ in convs], observe=cal.observe)
    s_in, zp_in = cal.qparams(npm.m["input"])
    t_in = T[npm.m["input"]]
    assert abs(s_in - t_in["scale"][0]) < 1e-9 and zp_in == t_in["zp"] == -128
    rel, zpd = [], []
    for o in ops:
        if o["op"] in (MAXPOOL, PAD, QUANTIZE) or T[o["out"]]["ns"] != 1:
            continue                        # pool and pad keep their input's parameters; QUANTIZE outputs take the concat's
        s, zp = cal.qparams(o["out"])
        rel.append(abs(s - T[o["out"]]["scale"][0]) / T[o["out"]]["scale"][0])
        zpd.append(abs(zp - T[o["out"]]["zp"]))
    rel = np.array(rel)
    assert len(rel) >= 40
    assert np.median(rel) < 0.05 and rel.max() < 0.30, (np.median(rel), rel.max())
    assert np.median(rel[len(rel) // 2:]) < 0.03              # the deeper half of the network
    assert np.median(zpd) <= 3
