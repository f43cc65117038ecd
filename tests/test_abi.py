"""The drop-in boundary: exported symbols, struct layouts, error behaviour without a GPU, and (container only) the
reference-header caller.  No compute calls need a GPU here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_reference

HEADER = os.path.join(ROOT, "include", "yf_network.h")
PROBE = os.path.join(ROOT, "tests", "abi", "layout_probe.c")


def _declared_functions():
    src = open(HEADER).read()
    return re.findall(r"YF_API\s+[\w\s\*]+?\b(\w+)\s*\(", src)


def test_library_exports_every_declared_symbol(yf):
    lib = yf.load()
    names = _declared_functions()
    assert len(names) >= 21 and "ai_network_run" in names and "ai_platform_bind_network_params" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/yf_network.h but not exported"
    binding = __import__("importlib").import_module("stm32h7-yolo_amd.binding")
    assert sorted(binding.EXPORTS) == sorted(names)


def _run_probe(tmp_path, own):
    exe = str(tmp_path / ("probe_own" if own else "probe_ref"))
    inc = ["-I" + os.path.join(ROOT, "include"), "-DYF_OWN_HEADER"] if own else ["-I/root/reference/stm32/Middlewares/ST/AI/Inc"]
    subprocess.check_call(["gcc", "-std=gnu11", PROBE, "-o", exe] + inc)
    return subprocess.check_output([exe]).decode()


def test_struct_layout_matches_ctypes_binding(tmp_path, yf):
    b = __import__("importlib").import_module("stm32h7-yolo_amd.binding")
    out = _run_probe(tmp_path, True)
    lines = dict(l.split(" ", 1) for l in out.strip().splitlines())
    assert int(lines["ai_buffer"].split(":")[0]) == ctypes.sizeof(b.AiBuffer) == 32
    assert [int(v) for v in lines["ai_buffer"].split(":")[1].split()] == [
        b.AiBuffer.format.offset, b.AiBuffer.n_batches.offset, b.AiBuffer.height.offset, b.AiBuffer.width.offset,
        b.AiBuffer.channels.offset, b.AiBuffer.data.offset, b.AiBuffer.meta_info.offset]
    assert int(lines["ai_network_params"].split(":")[0]) == ctypes.sizeof(b.AiNetworkParams) == 64
    assert int(lines["ai_network_report"].split(":")[0]) == ctypes.sizeof(b.AiNetworkReport) == 168
    assert int(lines["ai_buffer_array"].split(":")[0]) == ctypes.sizeof(b.AiBufferArray) == 16
    assert "S8=0x%08x U8=0x%08x" % (b.AI_BUFFER_FORMAT_S8, b.AI_BUFFER_FORMAT_U8) in out


@pytest.mark.skipif(not has_reference(), reason="container only: needs /root/reference")
def test_own_header_is_abi_identical_to_reference_header(tmp_path):
    """sizeof / offsetof / enum values of include/yf_network.h vs the reference's ai_platform.h
    (Middlewares/ST/AI/Inc/ai_platform.h:392-413,467-470,517-536,546-586,606-655)."""
    assert _run_probe(tmp_path, True) == _run_probe(tmp_path, False)


@pytest.mark.skipif(not has_reference(), reason="container only: needs /root/reference")
def test_reference_network_data_links_against_library():
    """The reference's unmodified network_data.c + a caller on the reference's headers link against the library
    (ai_platform_bind_network_params is the only runtime symbol that file needs, network_data.c:431)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "oracle", "_ref", "abi_ref_caller")
    undefined = subprocess.check_output(["nm", "-u", exe]).decode()
    needed = sorted(set(re.findall(r"U (ai_\w+)", undefined)))
    assert needed == ["ai_network_create", "ai_network_destroy", "ai_network_get_error", "ai_network_get_info", "ai_network_get_report",
                      "ai_network_init", "ai_network_run", "ai_platform_bind_network_params"]


def _report_lines(binary, *args):
    exe = os.path.join(ROOT, "oracle", "_ref", binary)
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return [ln for ln in r.stdout.splitlines() if ln.startswith(("report[", "info["))]


@pytest.mark.skipif(not has_reference(), reason="container only: needs /root/reference")
def test_reports_equal_the_reference_network_c_field_for_field():
    """ai_network_get_report / ai_network_get_info (row a4): abi_ref_caller asks THIS library's functions, abi_ref_runtime_caller the reference's own
    generated network.c:3271-3361 (which pre-fills a report and hands it to ai_platform_api_get_network_report, here platform_abi.c).  Every string
    and integer of both reports must be equal -- model signature, date, tool revision, the three versions, the arm of the params union each call
    fills -- except compile_datetime (the compile time of network.c / of this library)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    mine, ref = _report_lines("abi_ref_caller", "report"), _report_lines("abi_ref_runtime_caller", "report")
    assert len(mine) == len(ref) == 23 + 20
    keep = lambda lines: [ln for ln in lines if ".compile_datetime " not in ln]      # noqa: E731
    assert keep(mine) == keep(ref)
    d = dict(ln.split(" ", 1) for ln in mine)
    # ... and against the #defines of the reference's generated files themselves (network.c:38-49, network_config.h:25-46)
    netc = open("/root/reference/stm32/X-CUBE-AI/App/network.c", encoding="latin-1").read()
    assert d["report[created].model_signature"] == re.search(r'#define AI_NETWORK_MODEL_SIGNATURE\s+"(\w+)"', netc).group(1)
    assert d["report[created].model_datetime"] == re.search(r'#define AI_TOOLS_DATE_TIME\s+"([^"]+)"', netc).group(1)
    assert d["report[created].tool_revision"] == "[]" and d["report[created].tool_version"] == "7.0.0.0" and d["report[created].tool_api_version"] == "1.4.0.0"
    assert d["report[created].map_signature"] == "0xa1facade" and "info[created].map_signature" not in d
    assert d["info[created].params"].startswith("format=0x40040440 n_batches=1 height=1 width=1 channels=11304")


@pytest.mark.skipif(not has_reference(), reason="container only: needs /root/reference")
def test_reference_generated_network_c_links_against_library(yf):
    """Runtime-level drop-in: the reference's GENERATED network.c, unchanged, resolves every
    ST-runtime symbol it references (ai_platform_*, forward_*, nl_func_array_integer, ai_sum_*) from this library."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "oracle", "_ref", "abi_ref_runtime_caller")
    undefined = subprocess.check_output(["nm", "-u", exe]).decode()
    needed = sorted(set(re.findall(r"U ((?:ai_|forward_|nl_func)\w+)", undefined)))
    obj = subprocess.check_output(["bash", "-c", "gcc -std=gnu11 -c /root/reference/stm32/X-CUBE-AI/App/network.c "
                                   "-I/root/reference/stm32/X-CUBE-AI/App -I/root/reference/stm32/Middlewares/ST/AI/Inc "
                                   "-o /tmp/_yf_network_ref.o && nm -u /tmp/_yf_network_ref.o"]).decode()
    wanted = sorted(set(re.findall(r"U ((?:ai_|forward_|nl_func)\w+)", obj)) | {"ai_platform_bind_network_params"})
    assert needed == wanted and len(wanted) == 22
    lib = yf.load()
    for n in wanted:
        assert hasattr(lib, n)
    # without a GPU the binary fails the way the firmware would report it: init error type 0x30
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "golden_inputs.bin"), "/tmp/_yf_heads.bin", "6"],
                           capture_output=True, text=True)
        assert r.returncode == 4 and "type=48" in r.stdout


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks the behaviour WITHOUT a GPU")
def test_product_fails_loudly_without_gpu(yf):
    """No CPU fallback: init reports AI_ERROR_INIT_FAILED and run refuses (reference error convention,
    network.h:120-132: first error latched, reading resets it)."""
    net = yf.Network()
    with pytest.raises(yf.NetworkError) as ei:
        net.init()
    assert ei.value.type == 0x30 and "HIP" in ei.value.text
    assert net.get_error() == (0, 0)                       # latch was reset by the read
    with pytest.raises(yf.NetworkError) as ei:
        net.run(np.zeros((1, 56, 56, 3), np.int8))
    assert (ei.value.type, ei.value.code) == (0x11, 0x30)  # INVALID_STATE / MISSED_INIT
    net.destroy()


@pytest.mark.skipif(not _no_gpu(), reason="checks the behaviour WITHOUT a GPU")
def test_repeated_init_without_gpu_does_not_grow(yf):
    """ai_network_init may be called again and again (network.c:3385-3399): every failure path of the engine's creation gives back
    what it had acquired (yf_engine_create -> yf_engine_destroy), so a hundred refused initialisations leave the process where it was."""
    import resource
    net = yf.Network()
    def attempt():
        with pytest.raises(yf.NetworkError):
            net.init()
    for _ in range(5):
        attempt()
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for _ in range(100):
        attempt()
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert after - before < 2048, (before, after)            # KiB
    net.destroy()


def test_load_without_a_build_tool_checks_the_build_id():
    """binding.load() decides without a child process whether the in-tree library is current (stamped ids == the ids of the sources as they stand,
    library newer than every source) and then loads it without running make -- under rocprofv3 every child of the process would be instrumented;
    YF_NO_BUILD=1 forces that path.  When it does have to build and the box has no make / hipcc, an existing library is loaded with a warning.
    In every case the ids baked into the library are compared with the ids computed from the sources, and another build is refused."""
    import subprocess
    import sys
    code = (
        "import importlib, os, sys, warnings, subprocess\n"
        "os.environ['PATH'] = '/nonexistent'\n"
        "b = importlib.import_module('stm32h7-yolo_amd.binding')\n"
        "mode = sys.argv[1]\n"
        "if 'bad' in mode: b.expected_build_id = lambda *a: '0' * 16\n"
        "if 'stale' in mode: b.library_is_current = lambda: False\n"
        "calls = []\n"
        "real = subprocess.check_call\n"
        "def spy(*a, **k): calls.append(a); return real(*a, **k)\n"
        "subprocess.check_call = spy\n"
        "with warnings.catch_warnings(record=True) as w:\n"
        "    warnings.simplefilter('always')\n"
        "    try:\n"
        "        lib = b.load()\n"
        "        print('LOADED', len(calls), sum('could not run the build' in str(x.message) for x in w), lib.yf_network_build_id().decode() == b.expected_build_id())\n"
        "    except RuntimeError as e:\n"
        "        print('REFUSED', len(calls), 'build id' in str(e))\n")
    env = dict(os.environ)
    env.pop("YF_LIB_PATH", None)
    env.pop("YF_NO_BUILD", None)

    def run(mode, **extra):
        r = subprocess.run([sys.executable, "-c", code, mode], cwd=ROOT, env=dict(env, **extra), capture_output=True, text=True, timeout=300)
        return r.stdout.strip(), r.stdout + r.stderr
    out, log = run("good")
    assert out == "LOADED 0 0 True", log                    # current library: no make, no child process at all
    out, log = run("stale")
    assert out == "LOADED 1 1 True", log                    # has to build, cannot (no make on PATH): warns, checks the ids, loads
    out, log = run("stale-bad")
    assert out == "REFUSED 1 True", log
    out, log = run("stale", YF_NO_BUILD="1")
    assert out == "LOADED 0 0 True", log                    # YF_NO_BUILD=1: never starts make ...
    out, log = run("stale-bad", YF_NO_BUILD="1")
    assert out == "REFUSED 0 True", log                     # ... and refuses a library built from other sources


def test_requant_rounding_is_a_property_of_the_created_network(yf):
    """yf_network_set_requant_rounding without a GPU in sight (host logic only): the choice is stored on a created network (and used by the next
    ai_network_init), an unknown value latches AI_ERROR_INVALID_PARAM and changes nothing, ai_network_create resets it to $YF_REQUANT_ROUNDING (unset:
    the reference rounding), YF_ROUND_GENERIC_KERNELS rides along, a foreign handle is refused, and the scratch statistics of a network that never
    launched are zero."""
    b = __import__("importlib").import_module("stm32h7-yolo_amd.binding")
    lib = yf.load()
    assert lib.yf_network_get_requant_rounding(ctypes.c_void_p(0x1234)) == -1 and lib.yf_network_set_requant_rounding(ctypes.c_void_p(0x1234), 1) == -1
    net = yf.Network()
    assert net.requant_rounding == b.YF_ROUND_TFLITE_REF
    for r in (b.YF_ROUND_TIES_UP, b.YF_ROUND_TIES_UP_ALL, b.YF_ROUND_SINGLE, b.YF_ROUND_TIES_UP | b.YF_ROUND_GENERIC_KERNELS, b.YF_ROUND_TFLITE_REF):
        assert net.set_requant_rounding(r).requant_rounding == r
    net.set_requant_rounding(b.YF_ROUND_TIES_UP)
    for bad in (4, -1, 0x104, 0x200):
        with pytest.raises(yf.NetworkError) as ei:
            net.set_requant_rounding(bad)
        assert ei.value.type == 0x14 and net.requant_rounding == b.YF_ROUND_TIES_UP
    assert net.scratch_stats() == dict(events_recorded=0, events_skipped=0, event_waits=0, device_syncs=0, acquire_waits=0, regions=0)
    net2 = yf.Network()                                  # ai_network_create again: back to the environment's choice
    assert net2.requant_rounding == b.YF_ROUND_TFLITE_REF
    net2.destroy()


@pytest.mark.skipif(not _no_gpu(), reason="checks the behaviour WITHOUT a GPU")
def test_c_rank_host_fails_cleanly_without_a_gpu():
    """tools/c_host/yf_ranks.c on a box without a GPU: every rank fails at its first HIP call, the parent neither hangs on its pipes nor dies of SIGPIPE
    (a first form read a dead rank's report as the ncclUniqueId), prints its one line with status 2 and exits 2."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "stm32h7-yolo_amd", "lib", "yf_c_ranks")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "stm32h7-yolo_amd", "csrc"), "chost"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for args in (["2", "1", "0", "1"], ["3", "1", "0", "1", "--no-exchange"]):
        r = subprocess.run([exe, ROOT] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 2, (r.returncode, r.stdout, r.stderr)
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["status"] == 2 and line["n_gpus"] == int(args[0]) and line["value"] == 0.0
    assert subprocess.run([exe], capture_output=True).returncode == 2


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang-offload-bundler"), reason="needs the ROCm LLVM tools")
def test_frozen_kernels_are_instruction_identical(yf):
    """The int8 kernels (rounds 4-5), the fp16 kernel and the 160x160 band kernels are FROZEN.  Round 6 edits device SOURCE files for host-side
    reasons only -- the engine's table upload, scratch statistics, the laboratory's production-order dump build inside yf_fused56.hip.h, the selectable rounding
    (other CONSTANTS for the same kernels) -- so the build id moves, but every kernel of the product library must disassemble to the instruction stream of the
    round-5 library (profiles/isa_hashes_frozen.txt; tools/isa_hashes.py).  A kernel change is a decision: regenerate the list and say so."""
    import subprocess
    import sys
    yf.load()                                     # builds the library if it is stale
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hashes.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "12 of 12 frozen kernels: instruction streams identical" in r.stdout, r.stdout + r.stderr


def test_handle_and_param_validation(yf):
    lib = yf.load()
    b = __import__("importlib").import_module("stm32h7-yolo_amd.binding")
    # create rejects a non-NULL config and a NULL handle pointer
    h = ctypes.c_void_p()
    cfg = b.make_buffer(b.AI_BUFFER_FORMAT_U8, 1, 1, 4)
    assert lib.ai_network_create(ctypes.byref(h), ctypes.byref(cfg)).type == 0x33
    assert lib.ai_network_create(None, None).type == 0x33
    assert lib.ai_network_create(ctypes.byref(h), None).type == 0
    # a foreign handle is refused everywhere
    bogus = ctypes.c_void_p(0x1234)
    assert lib.ai_network_get_error(bogus).type == 0x10
    assert not lib.ai_network_init(bogus, None)
    assert lib.ai_network_run(bogus, None, None) == 0
    # init: NULL params, short weights, short activations
    assert not lib.ai_network_init(h, None)
    e = lib.ai_network_get_error(h)
    assert (e.type, e.code) == (0x30, 0x11)
    p = b.AiNetworkParams()
    blob = (ctypes.c_uint8 * 100)()
    p.params = b.make_buffer(b.AI_BUFFER_FORMAT_U8 | b.AI_BUFFER_FMT_FLAG_CONST, 1, 1, 100, 1, ctypes.addressof(blob))
    assert not lib.ai_network_init(h, ctypes.byref(p))
    e = lib.ai_network_get_error(h)
    assert (e.type, e.code) == (0x30, 0x18)
    p.params = b.make_buffer(b.AI_BUFFER_FORMAT_U8 | b.AI_BUFFER_FMT_FLAG_CONST, 1, 1, 11304, 1, lib.ai_network_data_weights_get())
    act = (ctypes.c_uint8 * 64)()
    p.activations = b.make_buffer(b.AI_BUFFER_FORMAT_U8, 1, 1, 64, 1, ctypes.addressof(act))
    assert not lib.ai_network_init(h, ctypes.byref(p))
    e = lib.ai_network_get_error(h)
    assert (e.type, e.code) == (0x30, 0x13)
    # first error is latched: two failures, the first one is reported, then the latch is clear
    assert not lib.ai_network_init(h, None)
    assert lib.ai_network_run(h, None, None) == 0
    e = lib.ai_network_get_error(h)
    assert (e.type, e.code) == (0x30, 0x11)
    assert lib.ai_network_get_error(h).type == 0
    # weights getters: MAGIC-framed pointer map and the ai_buffer_array form
    m = ctypes.cast(lib.ai_network_data_weights_get(), ctypes.POINTER(ctypes.c_void_p))
    assert m[0] == 0xA1FACADE and m[2] == 0xA1FACADE and m[1]
    mp = b.AiNetworkParams()
    assert lib.ai_network_data_params_get(h, ctypes.byref(mp))
    assert mp.map_signature == 0xA1FACADE and mp.map_weights.size == 1 and mp.map_weights.buffer[0].channels == 11304
    assert mp.map_weights.buffer[0].data == m[1]
    # report works on a created (not yet initialised) network
    r = b.AiNetworkReport()
    assert lib.ai_network_get_report(h, ctypes.byref(r))
    assert r.model_name == b"network" and r.n_macc == 1344320 and r.n_nodes == 31 and r.n_inputs == 1
    assert r.model_signature == b"6e73621b74e22de6d49ea182d7122906" and r.model_datetime == b"Thu Nov  6 21:23:50 2025" and r.tool_revision == b""
    assert (r.tool_version.major, r.tool_version.minor, r.tool_version.micro) == (7, 0, 0) and (r.tool_api_version.major, r.tool_api_version.minor) == (1, 4)
    assert (r.api_version.major, r.api_version.minor, r.interface_api_version.major, r.interface_api_version.minor) == (1, 1, 1, 3)
    assert (r.inputs[0].height, r.inputs[0].width, r.inputs[0].channels) == (56, 56, 3)
    assert (r.outputs[0].height, r.outputs[0].width, r.outputs[0].channels) == (7, 7, 18)
    assert lib.ai_network_destroy(h) is None
    assert lib.ai_network_destroy(bogus) == 0x1234        # not destroyed: handle comes back


def test_library_weight_blob_is_the_generated_one(yf):
    lib = yf.load()
    m = ctypes.cast(lib.ai_network_data_weights_get(), ctypes.POINTER(ctypes.c_void_p))
    blob = bytes((ctypes.c_uint8 * 11304).from_address(m[1]))
    src = open(os.path.join(ROOT, "stm32h7-yolo_amd", "csrc", "gen", "yf_weights_blob_gen.c")).read()
    body = src[src.index("{") + 1: src.rindex("}")]
    assert blob == bytes(int(v) for v in re.findall(r"\d+", body))


@pytest.mark.skipif(not has_reference(), reason="container only: needs /root/reference")
def test_graph_views_match_reference_layout(tmp_path):
    """csrc/st_graph_view.h restates the layout of the objects in the reference's generated network.c (ai_network, layer
    base, conv2d / pool layers, tensor chain, storage klass, array).  tests/abi/graph_probe.c prints every size and offset
    the library reads, once from ST's headers and once from the library's own views."""
    outs = []
    for flags in (["-I/root/reference/stm32/Middlewares/ST/AI/Inc"], ["-DYF_OWN_VIEWS", "-I" + os.path.join(ROOT, "stm32h7-yolo_amd", "csrc")]):
        exe = str(tmp_path / ("probe" + str(len(outs))))
        subprocess.check_call(["gcc", "-std=gnu11"] + flags + [os.path.join(ROOT, "tests", "abi", "graph_probe.c"), "-o", exe])
        outs.append(subprocess.check_output([exe]).decode())
    assert outs[0] == outs[1] and "ai_layer_conv2d 104" in outs[0]


def test_tampered_reference_graph_is_refused():
    """Runtime-level drop-in, SURVEY.md 8(f)3: ai_platform_network_init walks the CALLER's node list (the reference's
    generated network.c) and refuses any graph that is not the one the fused engine implements.  oracle/_ref/abi_graph_tamper
    includes the reference's unmodified network.c and changes one field of one layer in memory before ai_network_init."""
    exe = os.path.join(ROOT, "oracle", "_ref", "abi_graph_tamper")
    if has_reference():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/abi_graph_tamper was not built (needs /root/reference at build time)")
    expect = {"stride": "node 14 (id 28", "groups": "groups 1, expected 36", "pad": "padding {1,1,1,0}, expected {1,1,0,0}",
              "nl": "fused non-linearity present", "pool": "window 8x4, expected 8x8", "shape": "output shape 7x7x12, expected 7x7x18",
              "order": "layer id 7", "weights": "weight tensor has 1000 elements, expected 1920",
              # quantisation tables (network.c:663-1341): one float32 ulp of one scale, one step of one zero point
              "scale": "node 13 (id 24, expected conv2d id 24): output tensor scale", "zp": "node 15 (id 29, expected conv2d id 29): output tensor zero point -3, expected -4",
              "wscale": "node 3 (id 7, expected conv2d id 7): weight tensor channel 17 scale", "prescale": "node 27 (id 48, expected conv2d id 48): pre-activation tensor scale"}
    for what, text in expect.items():
        r = subprocess.run([exe, what], capture_output=True, text=True, timeout=120)
        assert r.returncode == 4, (what, r.stdout, r.stderr)
        assert "type=0x30 code=0x10" in r.stdout and "not the yoloface graph" in r.stdout and text in r.stdout, (what, r.stdout)
    r = subprocess.run([exe, "none"], capture_output=True, text=True, timeout=120)      # the unedited graph passes the check
    assert "not the yoloface graph" not in r.stdout
    if _no_gpu():
        assert r.returncode == 4 and "no HIP device" in r.stdout          # ... and then fails loudly for want of a GPU
    else:
        assert r.returncode == 0 and "init ok" in r.stdout
