"""ASan + UBSan over the CPU-side C code (oracle restatement and the product's host table preparation).  GPU
sanitizers are not available on the pool, so this is where memory errors in the C sources would surface."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def fnv(b):
    h = 1469598103934665603
    for v in b:
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_sources_are_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_main")
    csrc = os.path.join(ROOT, "stm32h7-yolo_amd", "csrc")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", "-Wall", "-Wextra", "-o", exe,
                           os.path.join(ROOT, "tests", "csrc", "sanitize_main.c"), os.path.join(ROOT, "oracle", "yf_oracle.c"),
                           os.path.join(csrc, "yf_host_prep.c"), os.path.join(csrc, "gen", "yf_weights_blob_gen.c"),
                           "-lm", "-lpthread"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(ROOT, "oracle", "model", "yoloface_int8.yfm"), os.path.join(GOLDEN, "golden_inputs.bin"),
                        os.path.join(GOLDEN, "decode_tables_f32.bin")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    heads = np.fromfile(os.path.join(GOLDEN, "golden_heads.bin"), np.uint8)
    assert r.stdout.split()[1] == f"{fnv(heads.tobytes()):016x}"          # and the sanitized build reproduces the goldens


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_stream_scratch_map_policy_under_asan(tmp_path):
    """csrc/yf_stream_scratch.h against a fake HIP runtime (tests/csrc/fake_hip: streams that count what was enqueued, events that complete when
    the test drains a stream): a launch that fails between get() and mark() leaves nothing acquired (max_regions failures, then a new stream
    still gets a region), a stream that is alone records no event and goes back to none once the others' launches have completed, the handle of
    a DESTROYED stream is never handed to the runtime (the fake aborts if it is: the real runtime segfaults) and its region comes back through a
    device synchronise, a reused handle value finds its old region, 64 dropped streams stay within 8 regions, a failed allocation leaves the map
    usable."""
    exe = str(tmp_path / "scratch_map_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "tests", "csrc", "fake_hip"), "-I" + os.path.join(ROOT, "stm32h7-yolo_amd", "csrc"),
                           os.path.join(ROOT, "tests", "csrc", "scratch_map_test.cpp"), "-o", exe, "-pthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1"))
    assert r.returncode == 0 and "scratch map: ok" in r.stdout, r.stdout + r.stderr
