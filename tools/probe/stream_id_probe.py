#!/usr/bin/env python3
"""Does hipStreamGetId name a STREAM or a HANDLE VALUE?  DEV PROBE (GPU box; /opt/rocm's runtime: PyTorch 2.10's bundled libamdhip64 does not export the call).

The scratch map keys a region by the stream's handle value (csrc/yf_stream_scratch.h): a stream destroyed with a launch still in flight and a new stream that is
handed the same handle value would share a region.  If the runtime's stream ids are unique over the life of the process, the id is the better key wherever the
call exists.  Prints (handle, id) of streams created, destroyed and created again."""
import ctypes
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipStreamGetId.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
seen = {}
reused = 0
for k in range(40):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    i = ctypes.c_ulonglong()
    rc = hip.hipStreamGetId(s, ctypes.byref(i))
    if s.value in seen:
        reused += 1
        print(f"handle {s.value:#x} again: id {i.value} (before: {seen[s.value]}) rc {rc}")
    seen[s.value] = i.value
    assert hip.hipStreamDestroy(s) == 0
i = ctypes.c_ulonglong()
print("null stream:", hip.hipStreamGetId(None, ctypes.byref(i)), i.value)
print(f"{len(seen)} distinct handle values over 40 create/destroy cycles, {reused} reuses; ids seen: {sorted(set(seen.values()))[:12]} ...")
