#!/usr/bin/env python3
"""A host that creates a HIP stream per request, synchronises it, DESTROYS it and never calls yf_network_release_stream (allowed: INTEGRATION.md, "Dropping a
stream") -- with real hipStreamCreate / hipStreamDestroy through the runtime's C API, not PyTorch's pooled streams, whose handles are never destroyed.  The
scratch map then holds regions whose stream handle is dead.  The map never hands a STORED stream handle to the runtime (this runtime does not validate
handles: hipStreamQuery / hipEventRecord on a destroyed stream is a segmentation fault, tools/probe/dead_stream_probe.py -- a first form of round 5 did
exactly that and died here): a region left "dirty" (launched on without an event) stays dirty until its stream launches again, is released, or -- at the
cap, when only dirty regions are left -- the DEVICE is synchronised (counted: scratch_stats()["device_syncs"]), and then changes hands.  200 requests (int8 batches of 512 frames on the batched shape, every fourth also fp16 and 160x160), at most three streams alive at a time; every int8
head is compared with the oracle, the footprint must stay within eight regions per kind.  Test helper: tests/test_gpu_parity.py runs it in a fresh process."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
yf = importlib.import_module("stm32h7-yolo_amd")
net = yf.Network(device=0).init()
net.configure(2, 8)
net.fp16_init()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))      # the runtime this process already uses (binding._one_hip_runtime)
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
block = np.random.default_rng(61).integers(-128, 128, (512, 56, 56, 3), dtype=np.int8)
ref = Oracle().run(block, threads=16)
d_in = torch.from_numpy(block).cuda()
d_f16 = torch.from_numpy((np.random.default_rng(62).integers(0, 256, (64, 56, 56, 3)) / 255.0).astype(np.float16)).cuda()
d_160 = torch.from_numpy(np.random.default_rng(63).integers(-128, 128, (16, 160, 160, 3), dtype=np.int8)).cuda()
torch.cuda.synchronize()
alive, bad, peak, handles = [], 0, 0, set()
rng = np.random.default_rng(64)
for req in range(200):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0                 # hipStreamNonBlocking
    handles.add(s.value)
    out = torch.empty((512, 7, 7, 18), dtype=torch.int8, device="cuda")
    net.run_device(d_in.data_ptr(), out.data_ptr(), 512, s.value)
    if req % 4 == 0:
        of = torch.empty((64, 7, 7, 18), dtype=torch.float32, device="cuda")
        o160 = torch.empty((16, 20, 20, 18), dtype=torch.int8, device="cuda")
        net.fp16_run_device(d_f16.data_ptr(), of.data_ptr(), 64, s.value)
        net.run_device_hw(160, 160, d_160.data_ptr(), o160.data_ptr(), 16, s.value)
    alive.append((s, out))
    peak = max(peak, net.scratch_bytes())
    while len(alive) > int(rng.integers(0, 3)):                                   # finish some requests: synchronise, check, DESTROY the stream, no release call
        st, o = alive.pop(0)
        assert hip.hipStreamSynchronize(st) == 0
        bad += not np.array_equal(o.cpu().numpy(), ref)
        assert hip.hipStreamDestroy(st) == 0
for st, o in alive:
    assert hip.hipStreamSynchronize(st) == 0
    bad += not np.array_equal(o.cpu().numpy(), ref)
    assert hip.hipStreamDestroy(st) == 0
torch.cuda.synchronize()
st = net.scratch_stats()                    # what the maps did (yf_network_scratch_stats): a host can SEE that dropping streams without a release costs device synchronisations
ok = bad == 0 and 0 < peak <= 8 * 48 * 2 ** 20 and 0 < st["regions"] <= 8 + 4 + 8 and st["events_recorded"] + st["events_skipped"] >= 200
print(f"200 requests on {len(handles)} distinct stream handle values: {bad} mismatches, peak scratch {peak / 2**20:.1f} MiB")
print("scratch stats:", st)
print("destroyed-streams rehearsal ok" if ok else "destroyed-streams rehearsal FAILED")
sys.exit(0 if ok else 1)
