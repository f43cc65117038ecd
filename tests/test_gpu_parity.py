"""Parity tests proper (run on a real MI355X with -m gpu): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs -- bit-exact (integer/byte work).  Full BASELINE sizes are additionally checked
through size-independent properties (determinism, permutation equivariance, golden frames embedded in the batch)."""
import ctypes
import importlib
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

STAGES = [("T1", 2), ("T2", 4), ("T3", 5), ("T4", 7), ("Q21", 21), ("T6", 11), ("T7", 12), ("T8", 14), ("T9", 16),
          ("T11", 18), ("T14", 22), ("T15", 24), ("Q45", 45), ("T17", 28), ("T18", 29), ("T19", 31), ("T20", 33),
          ("T22", 35), ("T23", 37), ("T24", 39), ("T26", 41), ("T30", 46), ("T31", 48), ("T32", 50), ("T33", 52),
          # tensors the fused stages never materialise, dumped for the per-node observer: raw max-pools, convolutions in front of the adds
          ("P8", 8), ("C17", 17), ("P25", 25), ("C34", 34), ("C40", 40), ("L43", 43)]
VARIANTS = [(1, 8), (2, 8)]     # the product's two shapes (the lab library's other shapes: test_lab_library_shapes_agree)
LAB_LIB = os.path.join(ROOT, "stm32h7-yolo_amd", "lib_lab", "libyf_network.so")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def rnd(seed, n):
    return np.random.default_rng(seed).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)


def test_native_library_is_loaded(network):
    maps = open("/proc/self/maps").read()
    assert "libyf_network.so" in maps
    assert "yoloface56_fused" in network.kernel_name


@pytest.mark.gpu
def test_reference_sample_images(network, oracle):
    """The reference's 27 sample images (tests/golden/real_frames_56.bin): real activation statistics instead of
    uniform noise -- head bit-exact against the oracle and against the committed sha256 of every head."""
    import hashlib, json
    frames = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "real_frames_56.json")))
    got = network.run(frames)
    assert np.array_equal(got, oracle.run(frames))
    assert [hashlib.sha256(h.tobytes()).hexdigest() for h in got] == meta["head_sha256"]


@pytest.mark.parametrize("n", [1, 2, 3, 5, 64, 257])
def test_ai_network_run_host_path_equals_oracle(network, oracle, n):
    """reference call: ai_network_run(network, &ai_input, &ai_output) with n_batches = n (yoloface.c:226-231)"""
    network.configure(2, 8)
    x = rnd(100 + n, n)
    assert np.array_equal(network.run(x), oracle.run(x, threads=8))


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 300, 512, 513])
def test_small_batches_run_one_frame_per_workgroup(network, oracle, torch_cuda, n):
    """Automatic shape choice (yf_network_configure(-1, ..)): up to 512 frames one frame per workgroup, more the throughput shape; up to 8
    frames ai_network_run lets the kernel read the caller's frames from pinned host memory.  Host path and device path against the oracle."""
    torch = torch_cuda
    network.configure(-1, -1)
    assert "F=1,NW=8" in network.kernel_name_for(512) and "F=2,NW=8" in network.kernel_name_for(513)
    x = rnd(300 + n, n)
    ref = oracle.run(x, threads=8)
    assert np.array_equal(network.run(x), ref)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), ref)
    network.configure(2, 8)
    assert "F=2,NW=8" in network.kernel_name_for(1)


@pytest.mark.parametrize("n", [2047, 2048, 3073, 4096, 7169])
def test_pipelined_host_path(network, oracle, torch_cuda, n):
    """ai_network_run on large host batches: chunks on two streams, heads downloaded by the worker thread (yf_engine_run_host).  Every head
    against the device path, the first, a middle and the last 128 frames against the oracle; two calls in a row reuse the staging."""
    torch = torch_cuda
    network.configure(-1, -1)
    x = rnd(400 + n, n)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dev = d_out.cpu().numpy()
    out = np.full((n, 7, 7, 18), 77, np.int8)
    for _ in range(2):
        out[:] = 77
        assert np.array_equal(network.run(x, out=out), dev)
    for lo in (0, n // 2, n - 128):
        assert np.array_equal(out[lo:lo + 128], oracle.run(x[lo:lo + 128], threads=8))
    assert network.lib.ai_network_forward(network.handle, ctypes.byref(b_make(x))) == n       # no output array: nothing is downloaded


def b_make(x):
    b = importlib.import_module("stm32h7-yolo_amd.binding")
    return b.make_buffer(b.AI_BUFFER_FORMAT_S8, 56, 56, 3, x.shape[0], x.ctypes.data)


@pytest.mark.parametrize("order", ["lib_first", "lib_init_first", "torch_first"])
def test_library_and_pytorch_in_either_order(order):
    """One process, one HIP runtime: the binding pre-loads PyTorch's copy of libamdhip64 when PyTorch is installed, so the library may be
    loaded (and even initialised) before `import torch` -- __graft_entry__.build() followed by smoke() in one process is that order."""
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "order_probe.py"), order], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "init ok" in out.stdout and "failed" not in out.stdout, out.stdout + out.stderr
    if order != "torch_first":
        assert "cuda available True" in out.stdout


def test_golden_fixtures(network, golden):
    heads = network.run(golden["inputs"])
    assert np.array_equal(heads, golden["heads"])


def test_edge_frames(network, oracle):
    x = np.stack([np.full((56, 56, 3), v, np.int8) for v in (-128, -1, 0, 1, 127)])
    x = np.concatenate([x, rnd(5, 3) // 64, (rnd(6, 2) | 0x7F).astype(np.int8)])   # low-contrast and saturated frames
    assert np.array_equal(network.run(x), oracle.run(x))


def test_structured_extreme_frames(network, oracle, torch_cuda):
    """Frames built to drive accumulators and requantisation to their edges rather than to look like images: the eight corner colours, stripes and
    checkerboards of +127 / -128 at periods 1, 2, 4 and 7 (stride-2 layers see them in and out of phase), single hot and cold pixels at the borders and
    corners (the halo / padding paths), frames matched to the SIGN of conv2d_1's weights for each of its eight output channels (the largest accumulators
    that layer can produce, both signs), and per-pixel random extremes.  Every head byte must equal the oracle's, through the device path on a ragged batch."""
    torch = torch_cuda
    from oracle.np_restatement import load_yfm
    m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    frames = []
    for r in (-128, 127):
        for g in (-128, 127):
            for b in (-128, 127):
                frames.append(np.broadcast_to(np.array([r, g, b], np.int8), (56, 56, 3)).copy())
    yy, xx = np.mgrid[0:56, 0:56]
    for period in (1, 2, 4, 7):
        for pat in ((xx // period) % 2, (yy // period) % 2, ((xx // period) + (yy // period)) % 2):
            f = np.where(pat[..., None] == 1, 127, -128).astype(np.int8)
            frames += [np.broadcast_to(f, (56, 56, 3)).copy(), (-1 - np.broadcast_to(f, (56, 56, 3))).astype(np.int8)]
    for (y, x) in ((0, 0), (0, 55), (55, 0), (55, 55), (0, 27), (27, 0), (55, 28), (28, 55), (27, 27)):
        for base, hot in ((-128, 127), (127, -128), (0, 127)):
            f = np.full((56, 56, 3), base, np.int8)
            f[y, x] = hot
            frames.append(f)
    conv1 = next(o for o in m["ops"] if o["op"] == 3)                            # CONV_2D #1: conv2d_1, 3x3 stride 2 behind the explicit top/left PAD
    w = np.asarray(m["tensors"][conv1["ins"][1]]["data"]).reshape(8, 3, 3, 3)   # OHWI int8
    ky = np.where((np.arange(56) + 1) % 2 == 1, 1, 0)                             # input row y is padded row y + 1: odd -> the window's middle tap, even -> its first
    for o in range(8):
        for sign in (1, -1):
            t = w[o][ky][:, ky].astype(np.int32) * sign                           # [56, 56, 3]: the weight each input value meets in (one of) its windows
            frames.append(np.where(t >= 0, 127, -128).astype(np.int8))
    rng = np.random.default_rng(99)
    frames += list(np.where(rng.integers(0, 2, (12, 56, 56, 3)) == 1, 127, -128).astype(np.int8))
    x = np.stack(frames)
    ref = oracle.run(x, threads=16)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.full((x.shape[0] + 1, 7, 7, 18), 55, dtype=torch.int8, device="cuda")
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), x.shape[0])
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:-1], ref) and (got[-1] == 55).all()
    assert len({ref[i].tobytes() for i in range(ref.shape[0])}) > ref.shape[0] // 2      # the frames do reach different heads (not all saturated to one answer)
    assert np.array_equal(network.run(x), ref)                                            # and through ai_network_run on host arrays


def test_every_fused_stage_equals_the_matching_tflite_op(network, oracle, torch_cuda):
    """Observer-style dump (yf_network_run_device_dump) vs the oracle's per-op outputs: the 25 fused stage tensors and the six
    tensors only the per-node observer needs (raw max-pools, the convolutions in front of the residual adds, LEAKY_RELU #43 alone)."""
    torch = torch_cuda
    from oracle.np_restatement import load_yfm
    m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    sizes = [int(np.prod(m["tensors"][o["out"]]["shape"][1:])) for o in m["ops"]]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n = 5
    x = rnd(42, n)
    head_ref, dump_ref = oracle.run(x, dump=True)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    d_dump = torch.zeros((n, network.dump_bytes()), dtype=torch.int8, device="cuda")
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), n, None, d_dump.data_ptr())
    torch.cuda.synchronize()
    dump = d_dump.cpu().numpy()
    off = 0
    for name, op in STAGES:
        got = dump[:, off:off + sizes[op]]
        ref = dump_ref[:, offs[op]:offs[op] + sizes[op]]
        assert np.array_equal(got, ref), f"stage {name} (tflite op {op})"
        off += sizes[op]
    assert off == network.dump_bytes()
    assert np.array_equal(d_out.cpu().numpy(), head_ref)


@pytest.mark.parametrize("fw", VARIANTS)
def test_kernel_variants_and_ragged_tails(yf, network, oracle, torch_cuda, fw):
    torch = torch_cuda
    network.configure(*fw)
    for n in (1, 7, 130):
        x = rnd(7 * n + fw[0], n)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.full((n + 1, 7, 7, 18), 77, dtype=torch.int8, device="cuda")     # canary frame after the batch
        network.run_device(d_in.data_ptr(), d_out.data_ptr(), n)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        ref = oracle.run(x, threads=8)
        assert np.array_equal(got[:n], ref)
        assert (got[n] == 77).all(), "wrote past the last frame"
        # the same shape with the box decode fused into the launch (the decoding waves depend on F and NW)
        cap = 4
        d_d = torch.zeros((n + 1, cap, 28), dtype=torch.uint8, device="cuda")
        d_c = torch.full((n + 1,), -7, dtype=torch.int32, device="cuda")
        d_out.fill_(77)
        network.run_decode_device(d_in.data_ptr(), d_out.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap, 1)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy()[:n], ref)
        counts = d_c.cpu().numpy()
        assert counts[n] == -7 and not d_d[n].any(), "decode wrote past the last frame"
        buf = d_d.cpu().numpy().view(yf.DET_DTYPE).reshape(n + 1, cap)
        for f in range(n):
            want = oracle.decode_c(ref[f], f)
            assert counts[f] == len(want)
            assert [(int(d["anchor"]), int(d["row"]), int(d["col"]), int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"]))
                    for d in buf[f, :min(cap, counts[f])]] == [(d[1], d[2], d[3], d[6], d[7], d[8], d[9]) for d in want][:cap]
    network.configure(2, 8)


def test_baseline_config2_batch_4096(network, oracle, golden, torch_cuda):
    """BASELINE.json configs[1]: batch 4096 on one GPU; golden frames are embedded at the front of the batch
    (SURVEY.md 8(d)); whole batch compared with the oracle, plus determinism and permutation equivariance."""
    torch = torch_cuda
    n = 4096
    x = rnd(1, n)
    x[:6] = golden["inputs"]
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), n)
    torch.cuda.synchronize()
    a = d_out.cpu().numpy()
    assert np.array_equal(a[:6], golden["heads"])
    assert np.array_equal(a, oracle.run(x, threads=16))
    d_out.zero_()
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), n)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), a)                       # deterministic
    perm = np.random.default_rng(2).permutation(n)
    d_in2 = torch.from_numpy(x[perm]).cuda()
    network.run_device(d_in2.data_ptr(), d_out.data_ptr(), n)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), a[perm])                # frames are independent


# library rounding -> the oracle variant that states it (oracle/yf_oracle.h YFO_RV_*)
ROUNDING_TO_VARIANT = {0: 0, 1: 1, 2: 2, 3: 4}


@pytest.mark.parametrize("rounding", [1, 2, 3, 0x101, 0x103])
def test_selectable_requant_rounding_equals_the_oracle_variant(yf, network, oracle, golden, torch_cuda, rounding):
    """yf_network_set_requant_rounding (round 6): another published rounding of TFLite's requantisation -- ties upward on the dense convs (ruy, what
    tflite_prediction.py:23's default resolver most likely runs), ties upward everywhere, single rounding on the dense convs -- bit-exact against the
    oracle's statement of that variant: the six golden frames, the reference's 27 sample images, 4096 seeded frames (BASELINE configs[1]'s input) with
    the fused decode, every fused stage through the dump build, a 160x160 block, ai_network_run on host arrays (small batches: the one-frame-per-
    workgroup shape).  By default these roundings run the second kernel set, whose dense convolutions requantise in three instructions (no sign term:
    no carry, ZR folded into C64); 0x100 | rounding (YF_ROUND_GENERIC_KERNELS) keeps the reference rounding's four-instruction kernels with that
    rounding's constants -- same results.  Switching back restores the reference rounding, and its kernels, bit for bit."""
    torch = torch_cuda
    variant = ROUNDING_TO_VARIANT[rounding & 0xFF]
    real = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    x = rnd(1, 4096)
    x[:6] = golden["inputs"]
    x[6:33] = real
    ref0 = oracle.run(x, threads=16)
    want, want_dump = oracle.run(x, threads=16, variant=variant), None
    assert not np.array_equal(want, ref0)
    assert network.requant_rounding == 0
    try:
        network.set_requant_rounding(rounding)
        assert network.requant_rounding == rounding
        assert ("sign-free dense epilogue" in network.kernel_name) == (rounding < 0x100) and ("sign-free" in network.kernel_name_for(5)) == (rounding < 0x100)
        d_in = torch.from_numpy(x).cuda()
        d_out = torch.zeros((4096, 7, 7, 18), dtype=torch.int8, device="cuda")
        cap = 4
        d_d = torch.zeros((4096, cap, 28), dtype=torch.uint8, device="cuda")
        d_c = torch.zeros((4096,), dtype=torch.int32, device="cuda")
        network.run_decode_device(d_in.data_ptr(), d_out.data_ptr(), 4096, d_d.data_ptr(), d_c.data_ptr(), cap, 0)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert np.array_equal(got, want)
        assert np.array_equal(got[:6], np.load(os.path.join(ROOT, "tests", "golden", "golden_heads_variants.npz"))[{1: "U", 2: "U_all", 4: "S"}[variant]])     # the committed fixture
        counts, buf = d_c.cpu().numpy(), d_d.cpu().numpy().view(yf.DET_DTYPE).reshape(4096, cap)
        for f in list(range(40)) + list(np.nonzero(counts)[0][:200]):
            py = oracle.decode_py(want[f], f)
            assert counts[f] == len(py)
            assert [(int(d["anchor"]), int(d["row"]), int(d["col"]), int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"])) for d in buf[f, :min(cap, counts[f])]] == \
                   [(d[1], d[2], d[3], d[6], d[7], d[8], d[9]) for d in py][:cap]
        # small batches (one frame per workgroup), the host path, and every fused stage through the dump build
        assert np.array_equal(network.run(x[:33]), want[:33])
        from oracle.np_restatement import load_yfm
        m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
        sizes = [int(np.prod(m["tensors"][o["out"]]["shape"][1:])) for o in m["ops"]]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        _, dump_ref = oracle.run(x[4:9], dump=True, variant=variant)
        d_dump = torch.zeros((5, network.dump_bytes()), dtype=torch.int8, device="cuda")
        network.run_device(d_in[4:9].data_ptr(), d_out.data_ptr(), 5, None, d_dump.data_ptr())
        torch.cuda.synchronize()
        dump, off = d_dump.cpu().numpy(), 0
        for name, op in STAGES:
            assert np.array_equal(dump[:, off:off + sizes[op]], dump_ref[:, offs[op]:offs[op] + sizes[op]]), f"stage {name} (tflite op {op})"
            off += sizes[op]
        # 160x160: the banded kernels read the same tables
        block = np.random.default_rng(4).integers(-128, 128, (3, 160, 160, 3), dtype=np.int8)
        d_b = torch.from_numpy(block).cuda()
        d_o = torch.zeros((3, 20, 20, 18), dtype=torch.int8, device="cuda")
        network.run_device_hw(160, 160, d_b.data_ptr(), d_o.data_ptr(), 3)
        torch.cuda.synchronize()
        assert np.array_equal(d_o.cpu().numpy(), oracle.run(block, threads=3, variant=variant))
        # camera frames -> heads + firmware-mode records in one launch (the camera-input build of the kernel set in force); and ragged batches on both shapes
        raw = np.random.default_rng(34).integers(0, 256, (131, 112 * 112 * 2), dtype=np.uint8)
        cam_ref = oracle.run(np.stack([oracle.prepare_rgb565(r) for r in raw]), threads=8, variant=variant)
        d_raw = torch.from_numpy(raw).cuda()
        d_ch = torch.zeros((131, 7, 7, 18), dtype=torch.int8, device="cuda")
        d_cd = torch.zeros((131, cap, 28), dtype=torch.uint8, device="cuda")
        d_cc = torch.zeros((131,), dtype=torch.int32, device="cuda")
        network.run_camera_device(d_raw.data_ptr(), d_ch.data_ptr(), 131, d_cd.data_ptr(), d_cc.data_ptr(), cap, yf.YF_DECODE_FW)
        torch.cuda.synchronize()
        assert np.array_equal(d_ch.cpu().numpy(), cam_ref)
        cc, cbuf = d_cc.cpu().numpy(), d_cd.cpu().numpy().view(yf.DET_DTYPE).reshape(131, cap)
        for f in range(131):
            fw = oracle.decode_c(cam_ref[f], f)
            assert cc[f] == len(fw) and [(int(d["anchor"]), int(d["row"]), int(d["col"]), int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"])) for d in cbuf[f, :min(cap, cc[f])]] == \
                   [(d[1], d[2], d[3], d[6], d[7], d[8], d[9]) for d in fw][:cap]
        for shape in ((2, 8), (1, 8)):
            network.configure(*shape)
            for nn in (1, 7, 513, 1027):
                d_r = torch.full((nn + 1, 7, 7, 18), 77, dtype=torch.int8, device="cuda")
                network.run_device(d_in.data_ptr(), d_r.data_ptr(), nn)
                torch.cuda.synchronize()
                r_got = d_r.cpu().numpy()
                assert np.array_equal(r_got[:nn], want[:nn]) and (r_got[nn] == 77).all(), (shape, nn)
        network.configure(-1, -1)
        with pytest.raises(Exception) as ei:
            network.set_requant_rounding(7)
        assert ei.value.type == 0x14 and network.requant_rounding == rounding               # AI_ERROR_INVALID_PARAM latched, nothing changed
    finally:
        network.set_requant_rounding(0)
    assert "sign-free" not in network.kernel_name
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), 4096)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), ref0)


def test_requant_rounding_from_the_environment_steers_an_unmodified_caller(oracle, golden):
    """YF_REQUANT_ROUNDING is read by ai_network_create: a caller that cannot be modified (the reference's aiInit: create + init back to back,
    yoloface.c:188-211) runs with another rounding; an unknown word fails ai_network_init loudly.  Fresh processes (the variable is read at create)."""
    import sys
    code = ("import sys, importlib, numpy as np\nsys.path.insert(0, %r)\nyf = importlib.import_module('stm32h7-yolo_amd')\n"
            "x = np.fromfile(%r, np.int8).reshape(-1, 56, 56, 3)\n"
            "try:\n    net = yf.Network(device=0).init()\nexcept Exception as e:\n    print('INIT FAILED', e); sys.exit(3)\n"
            "print(net.requant_rounding); sys.stdout.flush(); sys.stdout.buffer.write(net.run(x).tobytes())\n") % (ROOT, os.path.join(ROOT, "tests", "golden", "golden_inputs.bin"))
    for word, variant, value in (("ties_up", 1, 1), ("ties_up+generic", 1, 0x101), ("ref", 0, 0)):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300, env=dict(os.environ, YF_REQUANT_ROUNDING=word))
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        head, _, raw = r.stdout.partition(b"\n")
        assert int(head) == value
        assert np.array_equal(np.frombuffer(raw, np.int8).reshape(-1, 7, 7, 18), oracle.run(golden["inputs"], variant=variant))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300, env=dict(os.environ, YF_REQUANT_ROUNDING="nearest"))
    assert r.returncode == 3 and b"YF_REQUANT_ROUNDING" in r.stdout


@pytest.mark.parametrize("shape", [(2, 8), (1, 8)])
def test_tail_pairing_patterns(network, oracle, torch_cuda, shape):
    """Tail batching runs the 7x7 stages once per PAIR of a workgroup's frame groups.  Batch sizes around the multiples of the
    resident grid give workgroups with 1, 2, 3, 4 ... groups (paired, unpaired last, half-filled last group); frames are drawn
    from a 256-frame block whose heads come from the oracle, so every head of every batch is compared."""
    torch = torch_cuda
    block = rnd(31, 256)
    ref = oracle.run(block, threads=16)
    rng = np.random.default_rng(32)
    network.configure(*shape)
    try:
        for n in (511, 1023, 1024, 1025, 2047, 2049, 3071, 3073, 5001):
            pick = rng.integers(0, 256, n)
            d_in = torch.from_numpy(block[pick]).cuda()
            d_out = torch.full((n + 1, 7, 7, 18), 55, dtype=torch.int8, device="cuda")
            network.run_device(d_in.data_ptr(), d_out.data_ptr(), n)
            torch.cuda.synchronize()
            got = d_out.cpu().numpy()
            assert np.array_equal(got[:n], ref[pick]), (shape, n)
            assert (got[n] == 55).all()
    finally:
        network.configure(2, 8)


def test_overlapping_launches_on_two_streams(network, oracle, torch_cuda):
    """The fused kernel parks one group's T15 per workgroup in an HBM scratch (tail batching).  Launches of one network
    instance that overlap on different streams must not share those slots: two streams, twelve launches each of different
    batches and different (odd) sizes, issued alternately without synchronisation; every head is compared with the oracle."""
    torch = torch_cuda
    sizes = [4096, 2050]
    xs = [rnd(21 + k, sizes[k]) for k in range(2)]
    refs = [oracle.run(x, threads=16) for x in xs]
    streams = [torch.cuda.Stream() for _ in range(2)]
    d_ins = [torch.from_numpy(x).cuda() for x in xs]
    d_outs = [[torch.zeros((sizes[k], 7, 7, 18), dtype=torch.int8, device="cuda") for _ in range(12)] for k in range(2)]
    torch.cuda.synchronize()
    st0 = network.scratch_stats()
    for it in range(12):
        for k in range(2):
            network.run_device(d_ins[k].data_ptr(), d_outs[k][it].data_ptr(), sizes[k], streams[k].cuda_stream)
    torch.cuda.synchronize()
    for k in range(2):
        for it in range(12):
            assert np.array_equal(d_outs[k][it].cpu().numpy(), refs[k]), (k, it)
    # what the scratch map did, as a host reads it (yf_network_scratch_stats, round 6): 24 launches were marked -- in company they record their events (the very first
    # may still have been alone) --, two regions at least exist, and nobody waited for the device
    st1 = network.scratch_stats()
    assert (st1["events_recorded"] - st0["events_recorded"]) + (st1["events_skipped"] - st0["events_skipped"]) == 24
    assert st1["events_recorded"] - st0["events_recorded"] >= 22 and st1["regions"] >= 2 and st1["device_syncs"] == st0["device_syncs"]
    # (whether a LONE stream skips its events depends on what earlier callers left behind -- a region launched on without an event stays busy until its stream
    # launches again or is released -- so that half of the policy is asserted where the state is known: tests/csrc/scratch_map_test.cpp, cases 2 and 9)
    one = torch.cuda.Stream()
    for it in range(6):
        network.run_device(d_ins[0].data_ptr(), d_outs[0][it].data_ptr(), sizes[0], one.cuda_stream)
    torch.cuda.synchronize()
    st2 = network.scratch_stats()
    assert (st2["events_recorded"] - st1["events_recorded"]) + (st2["events_skipped"] - st1["events_skipped"]) == 6 and st2["device_syncs"] == st1["device_syncs"]


def test_short_lived_streams_keep_the_scratch_bounded(network, oracle, torch_cuda):
    """A host that creates a stream per request (yf_stream_scratch.h): 64 streams, each used for one int8, one fp16 and one 160x160
    launch and then left WITHOUT a release call (the handles stay alive until their launches are through: a stream is identified by its handle
    value, so destroying one with a launch still in flight needs yf_network_release_stream first -- INTEGRATION.md).  Regions of completed launches change hands, at most eight regions per kind
    exist, so the footprint stays under 8 x (park slots + fp16 park slots + one 160x160 arena chunk) whatever the number of
    streams (round 3 kept one region per stream handle ever seen: 64 x 43 MB here, 64 x 320 MB with 1024-frame 160x160 batches).  Every int8 head is compared with the oracle;
    an explicit release of a stream returns its bytes at once."""
    torch = torch_cuda
    block = rnd(41, 512)
    ref = oracle.run(block, threads=16)
    blk160 = np.random.default_rng(42).integers(-128, 128, (4, 160, 160, 3), dtype=np.int8)
    ref160 = oracle.run(blk160, threads=16)
    network.fp16_init()
    d_in = torch.from_numpy(block).cuda()
    d_f16 = torch.from_numpy((np.random.default_rng(43).integers(0, 256, (64, 56, 56, 3)) / 255.0).astype(np.float16)).cuda()
    d_160 = torch.from_numpy(np.tile(blk160, (16, 1, 1, 1))).cuda()
    torch.cuda.synchronize()
    outs = []
    peak = 0
    for k in range(64):
        st = torch.cuda.Stream()
        o = torch.empty((512, 7, 7, 18), dtype=torch.int8, device="cuda")      # (no fill kernel on the default stream racing the launches on `st`)
        of = torch.empty((64, 7, 7, 18), dtype=torch.float32, device="cuda")
        o160 = torch.empty((64, 20, 20, 18), dtype=torch.int8, device="cuda")
        network.configure(2, 8)                         # the batched shape: it parks T15 tensors in the stream's region
        network.run_device(d_in.data_ptr(), o.data_ptr(), 512, st.cuda_stream)
        network.fp16_run_device(d_f16.data_ptr(), of.data_ptr(), 64, st.cuda_stream)
        network.run_device_hw(160, 160, d_160.data_ptr(), o160.data_ptr(), 64, st.cuda_stream)
        outs.append((o, o160, st))
        peak = max(peak, network.scratch_bytes())
    torch.cuda.synchronize()
    network.configure(-1, 0)
    for o, o160, _ in outs:
        assert np.array_equal(o.cpu().numpy(), ref)
        assert np.array_equal(o160.cpu().numpy(), np.tile(ref160, (16, 1, 1, 1)))
    # eight regions per kind: a 64-frame 160x160 arena chunk (18.5 MB at 289 KB per frame), the int8 park slots (5.5 MB), the fp16 park slots (19.3 MB)
    bound = 8 * 48 * 2 ** 20
    assert 0 < peak <= bound, (peak, bound)
    before = network.scratch_bytes()
    network.release_stream(outs[-1][2].cuda_stream)
    assert network.scratch_bytes() < before


def test_asymmetric_launches_on_several_streams(network, oracle, torch_cuda):
    """Scratch ownership by stream (yf_stream_scratch.h).  ONE long launch on stream A (65 535 frames, ~2.7 ms) is followed,
    without synchronisation, by five 4096-frame launches on stream B and three on stream C: with scratch regions handed out
    round-robin per launch (the round-2 scheme) B's fourth launch would have parked its T15 tensors in the region A is still
    using.  Then two overlapping 160x160 launches on two streams (one arena per stream) and two overlapping fp16 launches.
    Every head is compared with the oracle (int8) / with the single-stream result (fp16)."""
    torch = torch_cuda
    block = rnd(31, 4096)
    ref = oracle.run(block, threads=16)
    rng = np.random.default_rng(32)
    pick_a = rng.integers(0, 4096, 65535)
    d_a = torch.from_numpy(block[pick_a]).cuda()
    o_a = torch.zeros((65535, 7, 7, 18), dtype=torch.int8, device="cuda")
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    picks_b = [rng.integers(0, 4096, 4096) for _ in range(5)]
    picks_c = [rng.integers(0, 4096, 2050) for _ in range(3)]
    d_b = [torch.from_numpy(block[p]).cuda() for p in picks_b]
    d_c = [torch.from_numpy(block[p]).cuda() for p in picks_c]
    o_b = [torch.zeros((4096, 7, 7, 18), dtype=torch.int8, device="cuda") for _ in range(5)]
    o_c = [torch.zeros((2050, 7, 7, 18), dtype=torch.int8, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    network.run_device(d_a.data_ptr(), o_a.data_ptr(), 65535, sa.cuda_stream)
    for k in range(5):
        network.run_device(d_b[k].data_ptr(), o_b[k].data_ptr(), 4096, sb.cuda_stream)
        if k < 3:
            network.run_device(d_c[k].data_ptr(), o_c[k].data_ptr(), 2050, sc.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(o_a.cpu().numpy(), ref[pick_a])
    for k in range(5):
        assert np.array_equal(o_b[k].cpu().numpy(), ref[picks_b[k]]), k
    for k in range(3):
        assert np.array_equal(o_c[k].cpu().numpy(), ref[picks_c[k]]), k
    del d_a, o_a, d_b, o_b, d_c, o_c
    # 160x160: one arena per stream
    blk = np.random.default_rng(33).integers(-128, 128, (8, 160, 160, 3), dtype=np.int8)
    ref160 = oracle.run(blk, threads=16)
    idx = [np.random.default_rng(34 + k).integers(0, 8, 700 + 37 * k) for k in range(2)]
    d_in = [torch.from_numpy(blk[i]).cuda() for i in idx]
    d_out = [[torch.zeros((len(i), 20, 20, 18), dtype=torch.int8, device="cuda") for _ in range(2)] for i in idx]
    torch.cuda.synchronize()
    for it in range(2):
        for k, st in enumerate((sa, sb)):
            network.run_device_hw(160, 160, d_in[k].data_ptr(), d_out[k][it].data_ptr(), len(idx[k]), st.cuda_stream)
    torch.cuda.synchronize()
    for k in range(2):
        for it in range(2):
            assert np.array_equal(d_out[k][it].cpu().numpy(), ref160[idx[k]]), (k, it)
    del d_in, d_out
    # fp16: a long launch on one stream, short ones on another; results must equal the single-stream ones bit for bit
    network.fp16_init()
    x16 = torch.from_numpy((np.random.default_rng(35).integers(0, 256, (4096, 56, 56, 3)).astype(np.float32) / 255).astype(np.float16)).cuda()
    big = x16.repeat(8, 1, 1, 1).contiguous()
    want = torch.zeros((4096, 7, 7, 18), dtype=torch.float32, device="cuda")
    network.fp16_run_device(x16.data_ptr(), want.data_ptr(), 4096)
    torch.cuda.synchronize()
    o_big = torch.zeros((32768, 7, 7, 18), dtype=torch.float32, device="cuda")
    o_small = [torch.zeros((4096, 7, 7, 18), dtype=torch.float32, device="cuda") for _ in range(5)]
    network.fp16_run_device(big.data_ptr(), o_big.data_ptr(), 32768, sa.cuda_stream)
    for k in range(5):
        network.fp16_run_device(x16.data_ptr(), o_small[k].data_ptr(), 4096, sb.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(o_big.view(8, 4096, 7, 7, 18), want.expand(8, 4096, 7, 7, 18))
    for k in range(5):
        assert torch.equal(o_small[k], want), k


def test_full_size_32768_properties(network, oracle, torch_cuda):
    """BASELINE.json configs[2] size (32768 frames, here on one GPU): the batch is 8 shuffled copies of a 4096-frame
    block, so every copy must reproduce the block's heads (checked against the oracle on the block)."""
    torch = torch_cuda
    block = rnd(2, 4096)
    ref = oracle.run(block, threads=16)
    rng = np.random.default_rng(3)
    perms = [rng.permutation(4096) for _ in range(8)]
    x = np.concatenate([block[p] for p in perms])
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((32768, 7, 7, 18), dtype=torch.int8, device="cuda")
    network.run_device(d_in.data_ptr(), d_out.data_ptr(), 32768)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    for k, p in enumerate(perms):
        assert np.array_equal(got[k * 4096:(k + 1) * 4096], ref[p])


def test_baseline_config5_160x160(network, oracle, torch_cuda):
    """BASELINE.json configs[4]: 160x160 input, batch 1024 per GPU.  Same weights and quantisation (fully
    convolutional); the oracle is the same C restatement at H=W=160 (head 20x20x18).  A 12-frame block (random,
    constant and saturated frames) is compared bit for bit; the 1024-frame batch is shuffled copies of the block."""
    torch = torch_cuda
    rng = np.random.default_rng(4)
    block = rng.integers(-128, 128, (12, 160, 160, 3), dtype=np.int8)
    block[1] = -128
    block[2] = 127
    block[3] = (block[3] // 32).astype(np.int8)
    ref = oracle.run(block, threads=16)
    assert ref.shape == (12, 20, 20, 18)
    idx = rng.integers(0, 12, 1024)
    idx[:12] = np.arange(12)
    d_in = torch.from_numpy(block[idx]).cuda()
    d_out = torch.full((1025, 20, 20, 18), 77, dtype=torch.int8, device="cuda")
    network.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), 1024)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got[:1024], ref[idx])
    assert (got[1024] == 77).all()
    # ragged launches: fewer band jobs than resident workgroups, a second chunk of the 1024-frame arena with one frame
    for n in (1, 3, 1025):
        pick = rng.integers(0, 12, n)
        d_in = torch.from_numpy(block[pick]).cuda()
        d_out = torch.full((n + 1, 20, 20, 18), 77, dtype=torch.int8, device="cuda")
        network.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), n)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert np.array_equal(got[:n], ref[pick]), n
        assert (got[n] == 77).all()
    # the generic entry point also accepts 56x56 (fused kernel) and refuses other sizes with a latched error
    x = rnd(77, 9)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.zeros((9, 7, 7, 18), dtype=torch.int8, device="cuda")
    network.run_device_hw(56, 56, d_in.data_ptr(), d_out.data_ptr(), 9)
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), oracle.run(x))
    with pytest.raises(Exception) as ei:
        network.run_device_hw(64, 64, d_in.data_ptr(), d_out.data_ptr(), 1)
    assert (ei.value.type, ei.value.code) == (0x12, 0x18)


def test_160x160_band_edges(network, oracle, torch_cuda):
    """The banded 160x160 kernels cut a frame into row bands (band_k1: 16 rows of the 80x80 grid = 32 input rows; band_k23: 8 rows of 40x40 = 32 input rows;
    band_k4 takes T15 in two halves) and pass halo rows between them through HBM.  Frames that put their only structure ON those cuts -- single hot rows,
    hot pixels and short vertical bars at input rows 30..33, 62..65, 94..97, 126..129 and at the image's first / last rows and columns, stripes whose period
    is the band height or half of it, a frame per band that is noise inside ONE band and constant elsewhere -- must give the oracle's heads bit for bit."""
    torch = torch_cuda
    frames = []
    for cut in (32, 64, 96, 128):
        for dy in (-2, -1, 0, 1):
            f = np.full((160, 160, 3), -128, np.int8)
            f[cut + dy, :, :] = 127                                           # one hot row next to / on a band cut
            frames.append(f)
        f = np.full((160, 160, 3), 127, np.int8)
        f[cut - 3:cut + 3, 40:43, :] = -128                                   # a short cold bar across the cut
        f[cut - 1, 0, :] = -128; f[cut, 159, :] = -128                        # ... and cold pixels on the cut at both borders
        frames.append(f)
    for y, x in ((0, 0), (0, 159), (159, 0), (159, 159), (0, 80), (159, 79), (80, 0), (79, 159)):
        f = np.zeros((160, 160, 3), np.int8)
        f[y, x] = (127, -128, 127)
        frames.append(f)
    yy = np.arange(160)[:, None, None]
    for period in (8, 16, 32):
        frames.append(np.broadcast_to(np.where((yy // period) % 2 == 1, 127, -128), (160, 160, 3)).astype(np.int8).copy())
        frames.append(np.broadcast_to(np.where(((yy + period // 2) // period) % 2 == 1, 127, -128), (160, 160, 3)).astype(np.int8).copy())
    rng = np.random.default_rng(160)
    for band in range(5):
        f = np.full((160, 160, 3), 3, np.int8)
        f[32 * band:32 * band + 32] = rng.integers(-128, 128, (32, 160, 3), dtype=np.int8)      # noise inside one band only
        frames.append(f)
    x = np.stack(frames)
    ref = oracle.run(x, threads=16)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.full((x.shape[0] + 1, 20, 20, 18), 77, dtype=torch.int8, device="cuda")
    network.run_device_hw(160, 160, d_in.data_ptr(), d_out.data_ptr(), x.shape[0])
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    bad = [i for i in range(x.shape[0]) if not np.array_equal(got[i], ref[i])]
    assert not bad and (got[-1] == 77).all(), f"frames {bad} differ from the oracle"
    assert len({ref[i].tobytes() for i in range(ref.shape[0])}) > ref.shape[0] // 2


@pytest.mark.skipif(not os.path.exists(LAB_LIB), reason="the lab library is not built (make -C stm32h7-yolo_amd/csrc lab)")
def test_160x160_layer_by_layer_form_agrees(torch_cuda):
    """The layer-by-layer form of the 160x160 path (one kernel per stage over an HBM arena; the LAB library with YF_160_LAYERWISE=1, kept as
    the plain statement the banded kernels are debugged against) gives the same heads: tests/dev/parity_160.py compares it with the oracle
    in a fresh process, because library and form are chosen when the engine is created."""
    import subprocess, sys
    env = dict(os.environ, YF_160_LAYERWISE="1", YF_LIB_PATH=LAB_LIB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "parity_160.py"), "4"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "160x160 head ok" in r.stdout


@pytest.mark.skipif(not os.path.exists(LAB_LIB), reason="the lab library is not built (make -C stm32h7-yolo_amd/csrc lab)")
def test_lab_library_shapes_agree(torch_cuda):
    """The other fused shapes (<1,4>, <2,4>, <4,8>) live in the lab library only (make lab): bit-exact against the oracle on ragged batches
    (tests/dev/lab_shapes.py, fresh process with YF_LIB_PATH)."""
    import subprocess, sys
    env = dict(os.environ, YF_LIB_PATH=LAB_LIB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "lab_shapes.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "lab shapes ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(LAB_LIB), reason="the lab library is not built (make -C stm32h7-yolo_amd/csrc lab)")
def test_every_fused_stage_in_the_production_stage_order(torch_cuda):
    """test_every_fused_stage_equals_the_matching_tflite_op runs the DEBUG build (staged pool order, no tail batching, 59 barriers);
    the kernel that ships runs the pools beside the branch on 3 + 5 waves, keeps conv2d_10's output on concat_22's bytes and runs the thirteen 7x7 stages once
    per PAIR of groups on four frames through the HBM park.  The laboratory library holds a dump build of exactly that order (namespace yfpd); on a grid of one
    workgroup the groups of 5, 3 and 8 frames pair up (5: a pair and an unpaired last group that holds one frame), then the full grid; the 25 fused-stage
    tensors of every frame equal the oracle's ops, and nothing is written behind the batch.  Fresh process (the lab library instead of the product)."""
    import sys
    for extra in ([], ["--ties-up"]):            # the reference rounding's kernel, and the second kernel set (sign-free dense epilogue) against the oracle's variant (U)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "stage_parity.py"), "--prod-order"] + extra + ["5", "3", "8", "130"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "stage parity (production order) ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        assert r.stdout.count("25 stage tensors + head ok") == 8
        assert ("sign-free dense epilogue" in r.stdout) == bool(extra)


@pytest.mark.skipif(not os.path.exists(LAB_LIB), reason="the lab library is not built (make -C stm32h7-yolo_amd/csrc lab)")
def test_a_failed_launch_does_not_keep_its_scratch_region(torch_cuda):
    """nine launches with an invalid grid on nine streams (lab library, YF_LAB_FAIL_LAUNCHES), then sixteen good launches on
    nine other streams -- every one gets a region and the oracle's heads (tests/dev/failed_launch.py; the map's policy itself is tested on the CPU
    against a fake runtime: tests/test_sanitizers.py::test_stream_scratch_map_policy_under_asan)."""
    import subprocess, sys
    env = dict(os.environ, YF_LIB_PATH=LAB_LIB, YF_LAB_FAIL_LAUNCHES="9")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "failed_launch.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failed-launch rehearsal ok" in r.stdout, r.stdout + r.stderr


def test_batches_whose_byte_offsets_pass_32_bits(torch_cuda):
    """481 280 int8 frames (4.5 GB) in one fused launch with the decode, 240 640 fp16 frames, 60 160 frames of 160x160: block-periodic inputs, so every copy must
    give the block's heads (the oracle's for int8) and every detection record its own frame index -- a 32-bit byte offset anywhere would show at the far end
    (tests/dev/huge_batch.py, fresh process)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "huge_batch.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "huge batches ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_a_hundred_lifecycles_leak_nothing(torch_cuda):
    """create -> init (-> init again) -> run on every path (device on a side stream, host zero-copy / one-shot / pipelined, fp16, 160x160) -> destroy, a
    hundred times in one process (tests/dev/lifecycle_soak.py): device memory comes back, host memory does not grow, every head stays equal to the oracle's."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "lifecycle_soak.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "lifecycle soak ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_hip_stream_per_thread_is_keyed_by_thread(network, oracle, torch_cuda):
    """hipStreamPerThread is ONE handle value (2) for a different stream per host thread: the scratch map keys it by (handle, thread), so two host threads
    that launch on it -- taking turns, the entry points are serialised by the caller like the reference's -- never share a region.  Two threads, five
    launches of a 1024-frame batch each on hipStreamPerThread, every head equal to the oracle's, and the library holds (at least) two tail-scratch regions."""
    import ctypes
    import threading
    torch = torch_cuda
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    per_thread = 2                                                 # hipStreamPerThread
    block = rnd(91, 1024)
    ref = oracle.run(block, threads=16)
    d_in = torch.from_numpy(block).cuda()
    torch.cuda.synchronize()
    network.configure(2, 8)
    before = network.scratch_bytes()
    turn, results, errors = threading.Lock(), {}, []

    def worker(tid):
        try:
            outs = []
            for _ in range(5):
                o = torch.empty((1024, 7, 7, 18), dtype=torch.int8, device="cuda")
                with turn:
                    network.run_device(d_in.data_ptr(), o.data_ptr(), 1024, per_thread)
                outs.append(o)
            assert hip.hipStreamSynchronize(per_thread) == 0      # this thread's own stream
            results[tid] = [np.array_equal(o.cpu().numpy(), ref) for o in outs]
        except Exception as e:                                     # noqa: BLE001
            errors.append(repr(e))
    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    network.configure(-1, 0)
    assert not errors, errors
    assert all(results[0]) and all(results[1]) and len(results[0]) == 5
    assert network.scratch_bytes() >= before and network.scratch_bytes() >= 2 * 5 * 2 ** 20      # two regions of 5.5 MB for the two threads' streams


def test_streams_that_are_destroyed_without_a_release(torch_cuda):
    """A stream per request, synchronised and destroyed with real hipStreamCreate / hipStreamDestroy and no yf_network_release_stream (INTEGRATION.md allows
    it): the scratch map holds regions whose stream handle is dead and must never hand one to the runtime -- this runtime segfaults on a destroyed stream
    (a first form of round 5's map did exactly that).  200 requests, every int8 head equal to the oracle's, footprint within eight regions per kind
    (tests/dev/destroyed_streams.py, fresh process)."""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "destroyed_streams.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "destroyed-streams rehearsal ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_baseline_config4_fp16_tolerance(yf, network, golden, torch_cuda):
    """BASELINE.json configs[3]: fp16 weights from the reference's ONNX export, 56x56, MFMA-f16 dense convs.
    Checked against an fp32 numpy evaluation of the same graph (oracle/np_fp32.py; itself cross-checked against torch
    CPU convs): head logits within atol 2e-2 / rtol 2e-2, and the same detection set wherever the fp32
    confidence logit is not within the tolerance of the 0.7 threshold."""
    torch = torch_cuda
    from oracle.np_fp32 import load_yfw, run_fp32
    convs = load_yfw(os.path.join(ROOT, "stm32h7-yolo_amd", "model", "yoloface_fp32.yfw"))
    rng = np.random.default_rng(3)
    u8 = rng.integers(0, 256, (8, 56, 56, 3), dtype=np.uint8)
    u8[:6] = (golden["inputs"].astype(np.int16) + 128).astype(np.uint8)           # includes the real image
    x32 = u8.astype(np.float32) / 255
    ref = np.stack([run_fp32(convs, f) for f in x32])
    network.fp16_init()
    n = 4096                                                                         # the config's batch: tiled copies
    idx = np.arange(n) % 8
    d_in = torch.from_numpy(x32.astype(np.float16)[idx]).cuda()
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.float32, device="cuda")
    network.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.array_equal(got.reshape(n // 8, 8, 7, 7, 18), np.broadcast_to(got[:8], (n // 8, 8, 7, 7, 18)))   # every copy, bit for bit
    # ragged batches: the tail runs on PAIRS of a workgroup's frames (tail batching); an unpaired last frame runs it alone
    for m in (1, 3, 513, 1027):
        d_o = torch.full((m + 1, 7, 7, 18), 7.0, dtype=torch.float32, device="cuda")
        network.fp16_run_device(d_in.data_ptr(), d_o.data_ptr(), m)
        torch.cuda.synchronize()
        g = d_o.cpu().numpy()
        assert np.array_equal(g[:m], got[:m]), m
        assert (g[m] == 7.0).all()
    err = np.abs(got[:8] - ref)
    assert np.all(err <= 2e-2 + 2e-2 * np.abs(ref)), f"max abs err {err.max():.4f}"
    # ... and bit for bit what this kernel gave when the fixture was written (tests/golden/make_fp16_golden.py): pins refactors of the kernel -- arena plan,
    # DMA placement -- that the tolerance above would let through.  Not a reference value: regenerated when the arithmetic order changes on purpose.
    pinned = np.load(os.path.join(ROOT, "tests", "golden", "fp16_logits_8.npy"))
    assert np.array_equal(got[:8], pinned), f"fp16 logits differ from the committed fixture (max abs diff {np.abs(got[:8] - pinned).max():.3g})"
    thr = np.log(0.7 / 0.3)                                                          # sigmoid(t) > 0.7  <=>  t > ln(7/3)
    cref, cgot = ref.reshape(8, 49, 3, 6)[..., 4], got[:8].reshape(8, 49, 3, 6)[..., 4]
    clear = np.abs(cref - thr) > 2e-2 + 2e-2 * np.abs(cref)
    assert np.array_equal((cref > thr)[clear], (cgot > thr)[clear])
    assert (cref[5] > thr).any()                                                     # the real image fires in fp32 too


def test_fp16_tolerance_on_structured_extreme_frames(network, torch_cuda):
    """The fp16 configuration on frames that are all contrast: black / white, the eight corner colours, stripes and checkerboards of period 1, 2, 4, 7, random
    black-and-white pixels.  Activations are as large as this network makes them (logits up to 19) and fp16 rounding accumulates most; the tolerance of
    SURVEY.md 8(d) -- atol 2e-2 + rtol 2e-2 against the fp32 numpy evaluation of the ONNX graph -- still holds on every logit (the worst one uses 99 % of it)."""
    torch = torch_cuda
    from oracle.np_fp32 import load_yfw, run_fp32
    convs = load_yfw(os.path.join(ROOT, "stm32h7-yolo_amd", "model", "yoloface_fp32.yfw"))
    frames = [np.full((56, 56, 3), v, np.uint8) for v in (0, 255)]
    for r in (0, 255):
        for g in (0, 255):
            for b in (0, 255):
                frames.append(np.broadcast_to(np.array([r, g, b], np.uint8), (56, 56, 3)).copy())
    yy, xx = np.mgrid[0:56, 0:56]
    for period in (1, 2, 4, 7):
        for pat in ((xx // period) % 2, (yy // period) % 2, ((xx // period) + (yy // period)) % 2):
            frames.append(np.broadcast_to(np.where(pat[..., None] == 1, 255, 0).astype(np.uint8), (56, 56, 3)).copy())
    frames += list(np.where(np.random.default_rng(7).integers(0, 2, (6, 56, 56, 3)) == 1, 255, 0).astype(np.uint8))
    x32 = np.stack(frames).astype(np.float32) / 255
    ref = np.stack([run_fp32(convs, f) for f in x32])
    n = x32.shape[0]
    network.fp16_init()
    d_in = torch.from_numpy(x32.astype(np.float16)).cuda()
    d_out = torch.zeros((n, 7, 7, 18), dtype=torch.float32, device="cuda")
    network.fp16_run_device(d_in.data_ptr(), d_out.data_ptr(), n)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    err = np.abs(got - ref)
    assert np.isfinite(got).all() and np.abs(ref).max() > 10
    assert np.all(err <= 2e-2 + 2e-2 * np.abs(ref)), f"max abs err {err.max():.4f}, worst use of the tolerance {(err / (2e-2 + 2e-2 * np.abs(ref))).max():.3f}"


def test_max_n_batches_65535_through_the_abi(network, oracle):
    """ai_buffer.n_batches is 16 bit (ai_platform.h:519): the largest single ai_network_run call."""
    lib = network.lib
    b = importlib.import_module("stm32h7-yolo_amd.binding")
    n = 65535
    base = rnd(4, 256)
    x = np.tile(base, (n // 256 + 1, 1, 1, 1))[:n]
    out = np.empty((n, 7, 7, 18), np.int8)
    bi = b.make_buffer(b.AI_BUFFER_FORMAT_S8, 56, 56, 3, n, x.ctypes.data)
    bo = b.make_buffer(b.AI_BUFFER_FORMAT_S8, 7, 7, 18, n, out.ctypes.data)
    assert lib.ai_network_run(network.handle, ctypes.byref(bi), ctypes.byref(bo)) == n
    ref = oracle.run(base, threads=8)
    assert np.array_equal(out[:256], ref)
    assert np.array_equal(out[-255:], ref[:255])           # 65535 = 255 * 256 + 255
    assert np.array_equal(out[256 * 100:256 * 101], ref)


def test_host_buffers_at_odd_addresses(network, oracle):
    """ai_network_run takes the caller's arrays as they are (the firmware aligns in_data / out_data to 32 bytes, yoloface.c:10-16; a host application
    may not): frames and heads at addresses that are 1, 3 and 7 bytes off any alignment, on the zero-copy (n <= 8), one-shot and pipelined paths."""
    b = importlib.import_module("stm32h7-yolo_amd.binding")
    lib = network.lib
    for n in (3, 300, 2500):
        x = rnd(100 + n, n)
        ref = oracle.run(x, threads=16)
        for off in (1, 3, 7):
            raw_in = np.zeros(n * 9408 + 64, np.int8)
            raw_out = np.full(n * 882 + 64, 91, np.int8)
            raw_in[off:off + n * 9408] = x.reshape(-1)
            bi = b.make_buffer(b.AI_BUFFER_FORMAT_S8, 56, 56, 3, n, raw_in.ctypes.data + off)
            bo = b.make_buffer(b.AI_BUFFER_FORMAT_S8, 7, 7, 18, n, raw_out.ctypes.data + off)
            assert lib.ai_network_run(network.handle, ctypes.byref(bi), ctypes.byref(bo)) == n
            assert np.array_equal(raw_out[off:off + n * 882].reshape(n, 7, 7, 18), ref), (n, off)
            assert (raw_out[:off] == 91).all() and (raw_out[off + n * 882:] == 91).all()      # nothing written outside the caller's heads


def test_run_argument_errors_are_latched(network):
    """Error behaviour of the reference boundary: <= 0 return, first error readable once (network.h:120-132)."""
    lib = network.lib
    b = importlib.import_module("stm32h7-yolo_amd.binding")
    x = np.zeros((2, 56, 56, 3), np.int8)
    out = np.zeros((2, 7, 7, 18), np.int8)
    good_in = b.make_buffer(b.AI_BUFFER_FORMAT_S8, 56, 56, 3, 2, x.ctypes.data)
    good_out = b.make_buffer(b.AI_BUFFER_FORMAT_S8, 7, 7, 18, 2, out.ctypes.data)
    cases = [
        (b.make_buffer(b.AI_BUFFER_FORMAT_U8, 56, 56, 3, 2, x.ctypes.data), good_out, (0x12, 0x19)),
        (b.make_buffer(b.AI_BUFFER_FORMAT_S8, 56, 56, 1, 2, x.ctypes.data), good_out, (0x12, 0x18)),
        (b.make_buffer(b.AI_BUFFER_FORMAT_S8, 56, 56, 3, 0, x.ctypes.data), good_out, (0x12, 0x21)),
        (b.make_buffer(b.AI_BUFFER_FORMAT_S8, 56, 56, 3, 2, None), good_out, (0x12, 0x17)),
        (good_in, b.make_buffer(b.AI_BUFFER_FORMAT_S8, 7, 7, 18, 1, out.ctypes.data), (0x13, 0x21)),
        (good_in, b.make_buffer(b.AI_BUFFER_FORMAT_S8, 7, 7, 17, 2, out.ctypes.data), (0x13, 0x18)),
    ]
    for bi, bo, want in cases:
        assert lib.ai_network_run(network.handle, ctypes.byref(bi), ctypes.byref(bo)) == 0
        e = lib.ai_network_get_error(network.handle)
        assert (e.type, e.code) == want
        assert lib.ai_network_get_error(network.handle).type == 0
    assert lib.ai_network_run(network.handle, ctypes.byref(good_in), None) == 0
    assert lib.ai_network_get_error(network.handle).type == 0x13
    assert lib.ai_network_forward(network.handle, ctypes.byref(good_in)) == 2          # run without output
    assert lib.ai_network_run(network.handle, ctypes.byref(good_in), ctypes.byref(good_out)) == 2
    # the device-pointer extensions use the same latch: null pointers, bad decode mode, zero capacity
    import torch
    d = torch.zeros((2 * 9408,), dtype=torch.int8, device="cuda")
    for args in [(None, d.data_ptr(), 2, 0, 1.0, 1.0, d.data_ptr(), d.data_ptr(), 4, None),
                 (d.data_ptr(), d.data_ptr(), 2, 7, 1.0, 1.0, d.data_ptr(), d.data_ptr(), 4, None),
                 (d.data_ptr(), d.data_ptr(), 2, 0, 1.0, 1.0, d.data_ptr(), d.data_ptr(), 0, None),
                 (d.data_ptr(), d.data_ptr(), 2, 0, 1.0, 1.0, None, d.data_ptr(), 4, None)]:
        assert lib.yf_network_run_decode_device(network.handle, *args) <= 0
        assert lib.ai_network_get_error(network.handle).type != 0
        assert lib.ai_network_get_error(network.handle).type == 0
    assert lib.yf_network_run_decode_device(network.handle, d.data_ptr(), d.data_ptr(), 0, 0, 1.0, 1.0, d.data_ptr(), d.data_ptr(), 4, None) == 0   # n = 0: nothing to do
    assert lib.ai_network_get_error(network.handle).type == 0
    # device pointers the kernels cannot take -- frames not 4-byte aligned (dword staging), heads not 2-byte aligned, fp16 frames not 4-byte aligned, 160x160
    # frames not 4-byte aligned -- are refused before anything is launched, with a latched error
    h = torch.zeros((2 * 882 + 8,), dtype=torch.int8, device="cuda")
    for call in (lambda: lib.yf_network_run_device(network.handle, d.data_ptr() + 1, h.data_ptr(), 2, None),
                 lambda: lib.yf_network_run_device(network.handle, d.data_ptr() + 2, h.data_ptr(), 2, None),
                 lambda: lib.yf_network_run_device(network.handle, d.data_ptr(), h.data_ptr() + 1, 2, None),
                 lambda: lib.yf_network_run_device_hw(network.handle, 160, 160, d.data_ptr() + 2, h.data_ptr(), 0 + 1, None)):
        assert call() <= 0
        assert lib.ai_network_get_error(network.handle).type != 0
        assert lib.ai_network_get_error(network.handle).type == 0
    torch.cuda.synchronize()


def test_weights_come_from_the_callers_blob(yf, network, oracle):
    """ai_network_init reads the blob it is handed (network.c:3108-3267 binds the caller's blob): perturbing one
    weight byte of a caller-owned copy changes the output; the pristine copy reproduces the oracle."""
    lib = network.lib
    m = ctypes.cast(lib.ai_network_data_weights_get(), ctypes.POINTER(ctypes.c_void_p))
    blob = np.frombuffer((ctypes.c_uint8 * 11304).from_address(m[1]), np.uint8).copy()
    x = rnd(8, 4)
    ref = oracle.run(x)
    network.init(weights=blob)
    assert np.array_equal(network.run(x), ref)
    bad = blob.copy()
    bad[10656 + 5] ^= 0x40            # one weight of the head conv (ST blob offset 10656, network.c:3259)
    network.init(weights=bad)
    assert not np.array_equal(network.run(x), ref)
    network.init()
    assert np.array_equal(network.run(x), ref)


def _dets(buf, counts, cap):
    out = []
    for f in range(counts.shape[0]):
        out.append([(int(d["anchor"]), int(d["row"]), int(d["col"]), int(d["q_conf"]), float(d["conf"]),
                     int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"])) for d in buf[f, :min(int(counts[f]), cap)]])
    return out


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_box_decode_on_gpu_equals_oracle(yf, network, oracle, golden, torch_cuda, mode):
    torch = torch_cuda
    rng = np.random.default_rng(21)
    heads = rng.integers(-60, 40, (300, 7, 7, 18), dtype=np.int8)
    heads[:6] = golden["heads"]
    heads[6] = 127                                    # every candidate fires: 147 detections, every box edge beyond int32
    heads[7] = -128
    n, cap = heads.shape[0], 147
    d_h = torch.from_numpy(heads).cuda()
    d_d = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_c = torch.zeros((n,), dtype=torch.int32, device="cuda")
    ws, hs = (410 / 56.0, 362 / 56.0) if mode == 0 else (1.0, 1.0)
    network.decode_device(d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap, mode, ws, hs)
    torch.cuda.synchronize()
    buf = d_d.cpu().numpy().view(yf.DET_DTYPE).reshape(n, cap)
    counts = d_c.cpu().numpy()
    got = _dets(buf, counts, cap)
    for f in range(n):
        ref = oracle.decode_py(heads[f], f, ws, hs) if mode == 0 else oracle.decode_c(heads[f], f, host_x86=(mode == 2))
        assert counts[f] == len(ref)
        assert got[f] == [(d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9]) for d in ref], f"frame {f}"
        assert all(int(d["frame"]) == f for d in buf[f, :counts[f]])
    assert counts[6] == 147 and counts[7] == 0
    # fixed capacity smaller than the true count: count is still the true count, records are the first `cap`
    d_d2 = torch.zeros((n, 5, 28), dtype=torch.uint8, device="cuda")
    network.decode_device(d_h.data_ptr(), n, d_d2.data_ptr(), d_c.data_ptr(), 5, mode, ws, hs)
    torch.cuda.synchronize()
    assert np.array_equal(d_c.cpu().numpy(), counts)
    buf2 = d_d2.cpu().numpy().view(yf.DET_DTYPE).reshape(n, 5)
    assert _dets(buf2, counts, 5)[6] == got[6][:5]


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_network_and_decode_in_one_launch(yf, network, oracle, torch_cuda, mode):
    """yf_network_run_decode_device: heads AND detection records from one launch (the heads are decoded while still
    in LDS) equal the oracle's network followed by the oracle's decode; ragged batch (odd n) and a small capacity."""
    torch = torch_cuda
    real = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    x = np.concatenate([real, np.random.default_rng(33).integers(-128, 128, (174, 56, 56, 3), dtype=np.int8)])
    n, cap = x.shape[0], 8
    assert n % 2 == 1
    ws, hs = (410 / 56.0, 362 / 56.0) if mode == 0 else (1.0, 1.0)
    d_x = torch.from_numpy(x).cuda()
    d_h = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    d_d = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_c = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    network.run_decode_device(d_x.data_ptr(), d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap, mode, ws, hs)
    torch.cuda.synchronize()
    heads = d_h.cpu().numpy()
    assert np.array_equal(heads, oracle.run(x))
    buf = d_d.cpu().numpy().view(yf.DET_DTYPE).reshape(n, cap)
    counts = d_c.cpu().numpy()
    got = _dets(buf, counts, cap)
    n_det = 0
    for f in range(n):
        ref = oracle.decode_py(heads[f], f, ws, hs) if mode == 0 else oracle.decode_c(heads[f], f, host_x86=(mode == 2))
        assert counts[f] == len(ref)
        assert got[f] == [(d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9]) for d in ref][:cap], f"frame {f}"
        n_det += len(ref)
    assert n_det > 0


def test_frame_preparation_on_gpu_equals_oracle(network, oracle, torch_cuda):
    """yoloface.c:26-93 (RGB565 112x112 -> int8 56x56x3), then the whole camera-format -> head pipeline."""
    torch = torch_cuda
    rng = np.random.default_rng(31)
    raw = rng.integers(0, 256, (9, 112 * 112 * 2), dtype=np.uint8)
    raw[0] = 0
    raw[1] = 255
    d_raw = torch.from_numpy(raw).cuda()
    d_x = torch.zeros((9, 56, 56, 3), dtype=torch.int8, device="cuda")
    d_out = torch.zeros((9, 7, 7, 18), dtype=torch.int8, device="cuda")
    network.prepare_rgb565_device(d_raw.data_ptr(), d_x.data_ptr(), 9)
    network.run_device(d_x.data_ptr(), d_out.data_ptr(), 9)
    torch.cuda.synchronize()
    ref = np.stack([oracle.prepare_rgb565(r) for r in raw])
    assert np.array_equal(d_x.cpu().numpy(), ref)
    assert np.array_equal(d_out.cpu().numpy(), oracle.run(ref))


def test_camera_frames_to_detections_in_one_launch(yf, network, oracle, torch_cuda):
    """the firmware's frame preparation fused into conv2d_1's load.  yf_network_run_camera_device takes
    112x112 RGB565 camera frames and returns heads and firmware-mode detection records from ONE launch; both must equal the
    oracle's prepare -> network -> decode and the library's own two-step path (separate preparation kernel)."""
    torch = torch_cuda
    rng = np.random.default_rng(33)
    n = 2051                                                     # odd: half-filled last group, unpaired tail
    base = rng.integers(0, 256, (40, 112 * 112 * 2), dtype=np.uint8)
    base[0] = 0
    base[1] = 255
    base[2, 0::2] = 0xF8; base[2, 1::2] = 0x00                   # pure red
    base[3, 0::2] = 0x07; base[3, 1::2] = 0xE0                   # pure green
    base[4, 0::2] = 0x00; base[4, 1::2] = 0x1F                   # pure blue
    pick = rng.integers(0, 40, n); pick[:40] = np.arange(40)
    raw = base[pick]
    x_ref = np.stack([oracle.prepare_rgb565(r) for r in base])
    h_ref = oracle.run(x_ref, threads=16)
    cap = 8
    d_raw = torch.from_numpy(raw).cuda()
    d_h = torch.full((n + 1, 7, 7, 18), 9, dtype=torch.int8, device="cuda")
    d_d = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_c = torch.zeros((n,), dtype=torch.int32, device="cuda")
    network.run_camera_device(d_raw.data_ptr(), d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap, yf.YF_DECODE_FW)
    torch.cuda.synchronize()
    heads = d_h.cpu().numpy()
    assert np.array_equal(heads[:n], h_ref[pick])
    assert (heads[n] == 9).all()
    # the two-step path of this library on the same frames
    d_x = torch.zeros((n, 56, 56, 3), dtype=torch.int8, device="cuda")
    d_h2 = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    d_d2 = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_c2 = torch.zeros((n,), dtype=torch.int32, device="cuda")
    network.prepare_rgb565_device(d_raw.data_ptr(), d_x.data_ptr(), n)
    network.run_decode_device(d_x.data_ptr(), d_h2.data_ptr(), n, d_d2.data_ptr(), d_c2.data_ptr(), cap, yf.YF_DECODE_FW)
    torch.cuda.synchronize()
    assert torch.equal(d_h[:n], d_h2) and torch.equal(d_c, d_c2) and torch.equal(d_d, d_d2)
    # records against the oracle's firmware decode of the oracle's heads
    dets, counts = d_d.cpu().numpy().view(yf.DET_DTYPE).reshape(n, cap), d_c.cpu().numpy()
    for f in range(64):
        want = oracle.decode_c(h_ref[pick[f]], f, 147)
        assert counts[f] == len(want)
        got = [(int(d["anchor"]), int(d["row"]), int(d["col"]), int(d["x1"]), int(d["y1"]), int(d["x2"]), int(d["y2"])) for d in dets[f, :min(cap, counts[f])]]
        assert got == [(w[1], w[2], w[3], w[6], w[7], w[8], w[9]) for w in want][:cap]
    # heads only
    d_h.fill_(9)
    network.run_camera_device(d_raw.data_ptr(), d_h.data_ptr(), 5)
    torch.cuda.synchronize()
    assert np.array_equal(d_h.cpu().numpy()[:5], h_ref[pick[:5]])
    # the staging loads 16-byte vectors: a misaligned frame pointer is refused with a latched error, nothing is launched
    with pytest.raises(Exception) as ei:
        network.run_camera_device(d_raw.data_ptr() + 2, d_h.data_ptr(), 1)
    assert ei.value.type != 0


def test_interpreter_mirror(oracle, golden, network):
    """tflite_prediction.py:23-41 call sequence on the mirror class (shares the library's single network)."""
    ip = importlib.import_module("stm32h7-yolo_amd.interpreter")
    it = ip.Interpreter(model_path="yoloface_int8.tflite")
    it.allocate_tensors()
    i_d, o_d = it.get_input_details(), it.get_output_details()
    assert tuple(i_d[0]["shape"]) == (1, 56, 56, 3) and i_d[0]["dtype"] == np.int8
    it.set_tensor(i_d[0]["index"], golden["inputs"][5:6])
    it.invoke()
    out = it.get_tensor(o_d[0]["index"])
    assert np.array_equal(out, golden["heads"][5:6])
    boxes = ip.decode_boxes(out[0], 0.7, 1.0, 1.0)
    want = golden["meta"]["frames"][5]["detections_py"]
    assert [tuple(int(v) for v in b) for b in boxes] == [(d["x1"], d["y1"], d["x2"], d["y2"]) for d in want]
    with pytest.raises(Exception, match="superseded"):       # the mirror owns the library's single instance now
        network.run_device(0, 0, 1)
    # tf.lite.Interpreter's experimental_op_resolver_type selects the kernel set, here the rounding: BUILTIN_REF = the oracle's definition,
    # BUILTIN (the optimized kernels: ruy on the dense convs) = ties upward, stated by the oracle's variant 1; AUTO = the library's default (reference)
    for resolver, variant in (("BUILTIN_REF", 0), ("BUILTIN", 1), ("AUTO", 0)):
        it = ip.Interpreter(model_path="yoloface_int8.tflite", experimental_op_resolver_type=resolver)
        it.allocate_tensors()
        it.resize_tensor_input(0, [6, 56, 56, 3])
        it.set_tensor(0, golden["inputs"])
        it.invoke()
        assert np.array_equal(it.get_tensor(100), oracle.run(golden["inputs"], variant=variant)), resolver
    with pytest.raises(ValueError):
        ip.Interpreter(experimental_op_resolver_type="XNNPACK")
    network.reclaim().init()  # hand the singleton back to the session fixture for the later tests
    assert network.requant_rounding == 0


@pytest.mark.parametrize("binary", ["abi_ref_caller", "abi_ref_runtime_caller"])
def test_reference_header_caller_binary(golden, tmp_path, binary):
    """oracle/_ref/abi_ref_caller = reference network_data.c + a caller on the reference's headers;
    oracle/_ref/abi_ref_runtime_caller additionally compiles the reference's GENERATED network.c unchanged, so only
    the ST runtime library is replaced.  Both linked against libyf_network.so in the build container
    (oracle/Makefile.ref); both initialisation forms."""
    exe = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(exe):
        pytest.skip(f"oracle/_ref/{binary} was not built (needs /root/reference at build time)")
    fin = os.path.join(ROOT, "tests", "golden", "golden_inputs.bin")
    for extra in ([], ["map"]):
        fout = str(tmp_path / "heads.bin")
        r = subprocess.run([exe, fin, fout, "6"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "OK 6" in r.stdout and "macc 1344320" in r.stdout
        assert np.array_equal(np.fromfile(fout, np.int8).reshape(6, 7, 7, 18), golden["heads"])


def test_reports_of_a_running_network_equal_the_reference_network_c(tmp_path):
    """Row a4 on the GPU box: after init + run, ai_network_get_report / ai_network_get_info of this library and of the reference's own generated
    network.c (over platform_abi.c) print the same fields -- now with the weights / activations buffers bound --, compile_datetime excepted."""
    outs = []
    for binary in ("abi_ref_caller", "abi_ref_runtime_caller"):
        exe = os.path.join(ROOT, "oracle", "_ref", binary)
        if not os.path.exists(exe):
            pytest.skip(f"oracle/_ref/{binary} was not built (needs /root/reference at build time)")
        per_form = []
        for extra in ([], ["map"]):
            r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "golden_inputs.bin"), str(tmp_path / "h.bin"), "6"] + extra, capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, r.stdout + r.stderr
            per_form.append([ln for ln in r.stdout.splitlines() if ln.startswith(("report[", "info[")) and ".compile_datetime " not in ln])
        outs.append(per_form)
    assert outs[0] == outs[1] and len(outs[0][0]) == 41
    d = dict(ln.split(" ", 1) for ln in outs[0][0])
    assert d["report[ready].model_signature"] == "6e73621b74e22de6d49ea182d7122906"
    assert "data=set" in d["report[ready].map_weights.buffer[]"] and "data=set" in d["report[ready].map_activations.buffer[]"]
    assert "data=set" in d["info[ready].params"] and "channels=29784 data=set" in d["info[ready].activations"]


def test_per_node_observer_through_the_reference_headers(oracle, tmp_path):
    """oracle/_ref/abi_observer_probe = the reference's generated network.c (unchanged) + a client of ai_platform_observer_* written against
    the reference's headers (ai_platform_interface.h:684-731, 981-1024).  An observed ai_network_run calls the client before and after each
    of the 31 c-nodes of every frame; in the POST call the node's output tensor -- read through the caller's own tensor objects -- must hold
    what the matching TFLite op produces (node ids are the op numbers: network_generate_report.txt:290-480)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "abi_observer_probe")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/abi_observer_probe was not built (needs /root/reference at build time)")
    from oracle.np_restatement import load_yfm
    m = load_yfm(os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"))
    sizes = [int(np.prod(m["tensors"][o["out"]]["shape"][1:])) for o in m["ops"]]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    n = 3
    x = np.fromfile(os.path.join(ROOT, "tests", "golden", "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)[:n]
    fin, fout = str(tmp_path / "frames.bin"), str(tmp_path / "nodes.bin")
    x.tofile(fin)
    r = subprocess.run([exe, fin, str(n), fout], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    ids = [2, 4, 5, 7, 11, 12, 14, 16, 17, 18, 20, 8, 22, 24, 28, 29, 31, 33, 34, 35, 37, 39, 40, 41, 43, 25, 46, 48, 50, 52, 53]
    lines = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("node ")]
    assert [int(t[3]) for t in lines] == ids and [int(t[1]) for t in lines] == list(range(31))
    assert int(lines[0][9], 16) & 0x100 and int(lines[30][9], 16) & 0x200            # FIRST / LAST event flags
    assert "info c_idx 12: id 22" in r.stdout and f"PRE {31 * n} POST {31 * n}" in r.stdout and "heads equal" in r.stdout and f"OK {n}" in r.stdout
    head_ref, dump_ref = oracle.run(x, dump=True)
    got = np.fromfile(fout, np.int8)
    want = np.concatenate([dump_ref[f, offs[k]:offs[k] + sizes[k]] for f in range(n) for k in ids])
    assert got.size == want.size
    pos = 0
    for f in range(n):
        for k in ids:
            assert np.array_equal(got[pos:pos + sizes[k]], dump_ref[f, offs[k]:offs[k] + sizes[k]]), f"frame {f}, node id {k}"
            pos += sizes[k]


def _rgb565_frames(golden):
    """112x112 big-endian RGB565 camera frames: random colours, and the golden 56x56 frames blown up 2x (each 2x2 block
    one colour, so the firmware's box average gives the frame back up to the 5/6/5 truncation)."""
    rng = np.random.default_rng(77)
    raw = [rng.integers(0, 256, (112, 112, 2), dtype=np.uint8) for _ in range(3)]
    for x in golden["inputs"][3:6]:
        u = (x.astype(np.int16) + 128).astype(np.uint16)
        px = ((u[..., 0] >> 3) << 11) | ((u[..., 1] >> 2) << 5) | (u[..., 2] >> 3)
        px = np.repeat(np.repeat(px, 2, axis=0), 2, axis=1)
        raw.append(np.stack([(px >> 8).astype(np.uint8), (px & 255).astype(np.uint8)], axis=-1))
    return np.stack(raw).reshape(len(raw), 112 * 112 * 2)


def test_reference_yoloface_c_links_unchanged_and_runs_on_the_gpu(yf, network, oracle, golden, torch_cuda, tmp_path):
    """SURVEY.md 8(a) row a15.  oracle/_ref/abi_yoloface_caller = the reference's UNMODIFIED stm32/X-CUBE-AI/App/yoloface.c
    (aiInit, aiRun, the 112->56 resize, prepare_yolo_data, post_process) + its network_data.c, compiled where they lie
    against the reference's AI headers, linked to libyf_network.so, driven like stm32/User/main.c:42-54.  Its three
    artefacts must equal this library's own: in_data = the GPU frame preparation, out_data = the head of the fused
    kernel, stdout = the UART text formatted from the GPU's firmware-mode records (host float -> int convention: the
    binary is an x86-64 build of yoloface.c).  A boundary test, not an oracle pin."""
    torch = torch_cuda
    exe = os.path.join(ROOT, "oracle", "_ref", "abi_yoloface_caller")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/abi_yoloface_caller was not built (needs /root/reference at build time)")
    binding = importlib.import_module("stm32h7-yolo_amd.binding")
    raw = _rgb565_frames(golden)
    n = raw.shape[0]
    f_raw, f_in, f_out = str(tmp_path / "cam.bin"), str(tmp_path / "in_data.bin"), str(tmp_path / "out_data.bin")
    raw.tofile(f_raw)
    r = subprocess.run([exe, f_raw, str(n), f_in, f_out], capture_output=True, timeout=180)
    assert r.returncode == 0, r.stdout.decode(errors="replace") + r.stderr.decode(errors="replace")
    in_data = np.fromfile(f_in, np.int8).reshape(n, 56, 56, 3)
    out_data = np.fromfile(f_out, np.int8).reshape(n, 7, 7, 18)
    # this library's pipeline on the same camera frames, all on the GPU
    d_raw = torch.from_numpy(raw).cuda()
    d_x = torch.zeros((n, 56, 56, 3), dtype=torch.int8, device="cuda")
    d_h = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
    cap = 147
    d_d = torch.zeros((n, cap, 28), dtype=torch.uint8, device="cuda")
    d_c = torch.zeros((n,), dtype=torch.int32, device="cuda")
    network.prepare_rgb565_device(d_raw.data_ptr(), d_x.data_ptr(), n)
    network.run_decode_device(d_x.data_ptr(), d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap, yf.YF_DECODE_FW_HOST)
    torch.cuda.synchronize()
    assert np.array_equal(in_data, d_x.cpu().numpy()), "prepare_yolo_data (reference C) vs the GPU frame preparation"
    assert np.array_equal(in_data, np.stack([oracle.prepare_rgb565(f) for f in raw]))
    assert np.array_equal(out_data, d_h.cpu().numpy()), "out_data behind the reference's aiRun vs the fused kernel"
    assert np.array_equal(out_data, oracle.run(in_data))
    buf = d_d.cpu().numpy().view(yf.DET_DTYPE).reshape(n, cap)
    counts = d_c.cpu().numpy()
    text = b"".join(binding.format_uart(k + 1, buf[k, :min(int(counts[k]), cap)], int(counts[k])) for k in range(n))
    assert r.stdout == text, "UART text of the reference application vs yf_network_format_uart over the GPU records"
    assert int(counts.sum()) > 0, "no frame produced a detection: the text comparison would be vacuous"


def test_bench_two_ranks_exchange_detections(tmp_path):
    """The N > 1 path of bench.py end to end on ONE GPU, entered the way a plain `python bench.py --gpus N` enters it (the
    parent starts the ranks itself): two fresh processes, backend gloo -- the collective goes through host copies,
    everything else (sharding, double-buffered record exchange of sharding.DetectionExchange, per-rank parity check,
    rank-major record order) is the code the 8-GPU run executes with RCCL.  The exchange is detections only:
    fixed-capacity records + counts, <= 0.6 MB per rank per step at 4096 frames."""
    import json
    import sys
    # plain `python bench.py --gpus 2`: bench.py starts its own ranks (child torch.distributed.run) and relays rank 0's line
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["all_gather_ok"] is True and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 8192 and line["parity"].startswith("every rank")
    assert 0 < line["config"]["exchange_bytes_per_rank_per_step"] <= 600_000
    assert line["config"]["launch_streams"] == 2 and line["pipelining"]["launch_streams"] == 2        # the default: steps alternate between two streams


def test_bench_two_ranks_gather_heads_and_a_failing_rank():
    """Two more rehearsals of the N > 1 path on one GPU (gloo), before the first real 8-rank run: (1) --gather-heads (the exchange record grows by
    the 3.6 MB of int8 heads per rank); (2) rank 1's parity check forced to fail (YF_BENCH_TEST_FAIL_RANK, read only here): rank 0 still prints
    its line, the line says FAILED, and the command exits non-zero."""
    import json
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd + ["--gather-heads"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["all_gather_ok"] is True and line["config"]["exchange_bytes_per_rank_per_step"] > 4096 * 882
    # (1b) the two exchange knobs of round 5: ONE collective per two steps (three steps: the run ends inside a buffer, drain() sends it) carrying
    # 12-byte wire records; every rank decodes the sparse heads the gathered records stand for on the GPU and must find its own records in them.
    # One launch stream here, two in the runs above and below (the default).
    r = subprocess.run(cmd[:-4] + ["--steps", "3", "--warmup", "1", "--gather-every", "2", "--compact-records", "--streams", "1"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["all_gather_ok"] is True and line["config"]["gather_every"] == 2 and line["config"]["launch_streams"] == 1
    assert line["config"]["exchange_bytes_per_rank_per_step"] == 4096 * (4 * 12 + 4) and "12-byte wire form" in line["config"]["workload"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(env, YF_BENCH_TEST_FAIL_RANK="1"))
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(lines[0])["parity"].startswith("FAILED")


def test_bench_four_ranks_on_one_gpu_preflight_for_the_eight_gpu_run():
    """Pre-flight of the driver's N = 8 run.  Nothing had ever run above world size 2; the world-size-EIGHT execution of the exchange and of bench.py's
    rank-major check is the CPU gloo test (tests/test_sharding.py::test_eight_rank_gloo_detection_exchange).  On the GPU box the pool allows at most SIX
    processes with the card open at once ("process guard": a six-rank form of this test was killed at 8 -- six ranks, the launcher's agent and this test
    runner, which holds the session's network), so bench.py itself -- self_launch with its children, per-rank inputs default_rng([1, rank, k]), rank >= 2
    in the gathered-buffer indexing, the rank-major check sampling every rank's shard -- is rehearsed with FOUR ranks sharing the one GPU (gloo;
    --input-batches 2 so that the ranks do not generate 1.2 GB of input).  And a failing LAST rank makes the command exit non-zero after rank 0 has
    printed its line."""
    import json
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--input-batches", "2"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["config"]["global_batch"] == 4 * 4096 and line["all_gather_ok"] is True and line["scaling"] == "weak"
    assert line["rank_major_check"]["ranks_sampled"] == [0, 1, 2, 3]                        # the order check reached into every rank's shard
    assert line["config"]["input_batches_rotated"] == 2 and line["parity"].startswith("every rank")
    assert line["config"]["check_steps"] == 4                                                # two launch streams x four exchange buffers: every pair checked
    r = subprocess.run(cmd + ["--gather-every", "2", "--compact-records", "--gather-heads"], cwd=ROOT, capture_output=True, text=True, timeout=1200,
                       env=dict(env, YF_BENCH_TEST_FAIL_RANK="3"))
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["parity"].startswith("FAILED") and line["all_gather_ok"] is True and line["rank_major_check"]["ranks_sampled"] == [0, 1, 2, 3]


def test_the_bench_line_carries_its_own_context():
    """The N = 1 line at the driver's flags: the one-stream rate beside the two-stream `value` (launch policy separated from
    kernel progress without profiles/), the CPU baseline on ALL usable cores beside the 16-thread figure, and the check steps -- one per launch stream
    and buffer, each compared with the oracle."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-secondary"], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    p, cb = line["pipelining"], line["cpu_baseline"]
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 4096 and line["parity"].startswith("bit-exact")
    assert p["launch_streams"] == 2 and p["one_stream_ms_per_step"] > 0 and p["one_stream_images_per_s"] > 0
    assert abs(p["one_stream_images_per_s"] - 4096 / p["one_stream_ms_per_step"] * 1e3) < 1e-3 * p["one_stream_images_per_s"]
    assert cb["cores"] == min(cb["affinity_cores"], 16) and cb["all_cores_threads"] == cb["affinity_cores"] and cb["all_cores_images_per_s"] > 0
    assert line["config"]["check_steps"] == 2
    assert line["roofline"]["kernel_ms"] > 0 and line["roofline"]["frac"] > 0


def test_the_bench_step_from_a_pure_c_host():
    """tools/c_host/yf_bench.c: the reference's aiInit sequence and the bench step (one yf_network_run_decode_device launch per 4096-frame batch, two
    alternating HIP streams) from a C program that links libyf_network.so and the HIP runtime only -- no Python, no PyTorch.  north_star: "host code in C
    calling HIP through a thin FFI".  The golden frames inside its first batch must give the golden heads.  Its rate is REPORTED, not asserted (it depends on
    the box and its clock regime: profiles/ has the figures); YF_TEST_ASSERT_RATES=1 turns the round-5 floors back on."""
    import json
    exe = os.path.join(ROOT, "stm32h7-yolo_amd", "lib", "yf_c_bench")
    if not os.path.exists(exe):
        pytest.skip("stm32h7-yolo_amd/lib/yf_c_bench was not built (make -C stm32h7-yolo_amd/csrc chost)")
    r = subprocess.run([exe, ROOT, "200", "50", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["golden_heads_equal"] is True and line["detections_on_the_real_frame"] > 0 and line["launch_streams"] == 2
    assert line["value"] > 0 and line["kernel_ms_alone"] > 0
    print(f"C host: {line['value'] / 1e6:.2f} M images/s, kernel alone {line['kernel_ms_alone'] * 1e3:.1f} us")
    if os.environ.get("YF_TEST_ASSERT_RATES") == "1":
        assert line["value"] > 20e6 and 0.10 < line["kernel_ms_alone"] < 0.20


def test_the_n_rank_flow_from_a_pure_c_host():
    """tools/c_host/yf_ranks.c: one PROCESS per GPU forked before any HIP call, rank 0's ncclUniqueId handed round by pipe, ncclCommInitRank, the bench step +
    yf_network_all_gather_device with four alternating record buffers on two streams, golden heads on rank 0, every rank's block at its own place and the
    gathered counts in rank order on every rank; one JSON line, non-zero exit if any rank fails -- no Python, no torch.distributed (north_star: "host code in C";
    INTEGRATION.md "Multi-GPU" has the N = 8 command).  Here: (1) N = 1 through RCCL (a one-rank communicator: RCCL refuses two ranks on one device);
    (2) three ranks sharing the GPU without the collective (--no-exchange): fork / pipes / barrier / MAX over ranks with rank >= 1."""
    import json
    exe = os.path.join(ROOT, "stm32h7-yolo_amd", "lib", "yf_c_ranks")
    if not os.path.exists(exe):
        pytest.skip("stm32h7-yolo_amd/lib/yf_c_ranks was not built (make -C stm32h7-yolo_amd/csrc chost)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, ROOT, "1", "40", "10", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["global_batch"] == 4096 and line["status"] == 0 and line["exchange"].startswith("RCCL all-gather")
    assert line["golden_heads_equal"] is True and line["own_block_at_own_place"] is True and line["gathered_counts_in_rank_order_on_every_rank"] is True
    assert line["detections_on_the_real_frame"] > 0 and line["value"] > 0 and line["exchange_bytes_per_rank_per_step"] == 4096 * (4 * 28 + 4)
    print(f"C host, one rank through RCCL: {line['value'] / 1e6:.2f} M images/s ({line['ms_per_step'] * 1e3:.1f} us per step, kernel alone {line['kernel_ms_alone_max_over_ranks'] * 1e3:.1f} us)")
    r = subprocess.run([exe, ROOT, "3", "10", "2", "2", "--no-exchange"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 3 and line["global_batch"] == 3 * 4096 and line["status"] == 0 and line["golden_heads_equal"] is True
    assert line["gathered_counts_in_rank_order_on_every_rank"] is None and line["value"] > 0


def test_a_stream_destroyed_with_a_launch_in_flight_does_not_share_its_scratch():
    """"the one place where a host mistake corrupts results silently".  The runtime hands a destroyed stream's handle VALUE to the next
    stream at once; rounds 3-5 keyed the launch scratch by that value.  tools/c_host/yf_stream_reuse.c (a C host on /opt/rocm's runtime, which exports
    hipStreamGetId; PyTorch 2.10's bundled runtime does not) launches on a stream, destroys it with the launch in flight, launches on its successor -- twenty times:
    the successor gets a region of its OWN (two alive at once whenever the handle value came back) and every launch equals its synchronous reference run."""
    import json
    exe = os.path.join(ROOT, "stm32h7-yolo_amd", "lib", "yf_c_stream_reuse")
    if not os.path.exists(exe):
        pytest.skip("stm32h7-yolo_amd/lib/yf_c_stream_reuse was not built (make -C stm32h7-yolo_amd/csrc chost)")
    r = subprocess.run([exe, ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    print(line)
    assert line["launches_that_differ_from_their_reference"] == 0
    if line["runtime_has_hipStreamGetId"] and line["successor_got_the_same_handle_value"]:
        assert line["regions_alive_at_once_max"] >= 2


def test_compact_wire_records_on_the_gpu(yf, network, oracle, torch_cuda):
    """yf_network_pack_detections_device / _unpack_ (the 12-byte wire form of the multi-GPU exchange, one launch each): the packed bytes equal the tensor-op
    statement of the format (sharding.pack_compact) also where the record buffer holds stale bytes beyond a frame's count; the sparse heads equal
    sharding.unpack_compact's; and decoding the sparse heads on the GPU reproduces the sender's records -- min(count, cap) per frame, same order, every
    field -- for both a cap that holds every candidate and one that does not."""
    torch = torch_cuda
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    frames = np.fromfile(os.path.join(ROOT, "tests", "golden", "real_frames_56.bin"), np.int8).reshape(-1, 56, 56, 3)
    x = np.concatenate([frames, rnd(77, 229)])                     # real images (they fire) and noise, 256 frames
    n = x.shape[0]
    d_in = torch.from_numpy(x).cuda()
    for cap in (2, 6):
        d_h = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
        d_d = torch.full((n * cap * 28,), 0xAB, dtype=torch.uint8, device="cuda")      # stale bytes everywhere the kernel does not write
        d_c = torch.zeros((n,), dtype=torch.int32, device="cuda")
        network.run_decode_device(d_in.data_ptr(), d_h.data_ptr(), n, d_d.data_ptr(), d_c.data_ptr(), cap)
        wire = torch.full((n * cap * 12,), 0xCD, dtype=torch.uint8, device="cuda")
        network.pack_detections_device(d_d.data_ptr(), d_c.data_ptr(), d_h.data_ptr(), wire.data_ptr(), n, cap)
        torch.cuda.synchronize()
        assert int((d_c > cap).sum()) > 0 or cap == 6
        want = sh.pack_compact(d_d, d_c, d_h.view(torch.uint8).view(-1), n, cap)
        assert torch.equal(wire, want)
        sparse = torch.zeros((n, 7, 7, 18), dtype=torch.int8, device="cuda")
        network.unpack_detections_device(wire.data_ptr(), d_c.data_ptr(), sparse.data_ptr(), n, cap)
        torch.cuda.synchronize()
        assert torch.equal(sparse, sh.unpack_compact(wire, d_c, n, cap))
        r_d = torch.zeros((n * cap * 28,), dtype=torch.uint8, device="cuda")
        r_c = torch.zeros((n,), dtype=torch.int32, device="cuda")
        network.decode_device(sparse.data_ptr(), n, r_d.data_ptr(), r_c.data_ptr(), cap)
        torch.cuda.synchronize()
        kept = torch.arange(cap, device="cuda")[None, :] < d_c.clamp(max=cap)[:, None]
        assert torch.equal(r_c, d_c.clamp(max=cap)) and int(r_c.sum()) > 0
        assert torch.equal(r_d.view(n, cap, 28)[kept], d_d.view(n, cap, 28)[kept])


def test_bench_exchange_through_rccl_in_a_one_rank_group():
    """The torch.distributed / RCCL calls of bench.py's N > 1 path have never run with more than one rank (no multi-GPU box), and RCCL refuses two
    ranks on one device -- so they are executed here in a ONE-rank group (`--rccl-one-rank`): process group with backend nccl bound to the device,
    asynchronous all_gather_into_tensor of the packed uint8 records on RCCL's stream behind kernels on TWO launch streams, `wait()` from the launch
    stream before a buffer is reused, barrier, all-reduce of the time and of the verdict, gathered-record checks.  Once per step with 28-byte records,
    once per two steps with the 12-byte wire records (the receiver-side decode of the sparse heads runs on the GPU)."""
    import json
    import sys
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rccl-one-rank", "--steps", "5", "--warmup", "2", "--clock-settle-ms", "5"]
    for extra, rec in ([], 4096 * (4 * 28 + 4)), (["--gather-every", "2", "--compact-records"], 4096 * (4 * 12 + 4)):
        r = subprocess.run(cmd + extra, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert line["n_gpus"] == 1 and line["all_gather_ok"] is True and line["parity"].startswith("every rank")
        assert line["config"]["exchange_bytes_per_rank_per_step"] == rec and line["config"]["launch_streams"] == 2 and "rehearsal" in line["config"]
        assert line["config"]["collectives_issued"] >= 1


def test_c_level_all_gather_through_rccl(network, torch_cuda):
    """yf_network_all_gather_device: the exchange step a C host application calls (one process per GPU, its own ncclComm_t).
    With one GPU on the box the communicator has one rank; the call still goes through librccl's ncclAllGather on the
    caller's stream.  Bad arguments are latched like every other entry point."""
    import ctypes
    torch = torch_cuda
    try:
        rccl = ctypes.CDLL("librccl.so")
    except OSError:
        pytest.skip("librccl.so not loadable")

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid, comm = UniqueId(), ctypes.c_void_p()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    n = 4096 * (4 * 28 + 4)
    send = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    recv = torch.zeros_like(send)
    lib = network.lib
    assert lib.yf_network_all_gather_device(network.handle, comm, send.data_ptr(), recv.data_ptr(), n, None) == n
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    assert lib.yf_network_all_gather_device(network.handle, None, send.data_ptr(), recv.data_ptr(), n, None) == 0
    assert network.get_error()[0] == 0x14                                   # AI_ERROR_INVALID_PARAM, latched once
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    rccl.ncclCommDestroy(comm)
