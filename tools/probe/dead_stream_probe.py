"""Does this HIP runtime validate stream handles?  Creates a stream, records an event on it, synchronises and destroys it, then calls hipStreamQuery and
hipEventRecord on the dead handle.  On ROCm 7.0 / 7.2 the first of them is a SEGMENTATION FAULT (the last line printed is "destroy 0"): a library must
never hand a stream handle it has stored to the runtime (csrc/yf_stream_scratch.h).  DEV PROBE: it is expected to crash."""
import ctypes, os, sys, torch, faulthandler
faulthandler.enable()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
torch.cuda.init(); torch.zeros(1, device="cuda")
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamQuery.argtypes = [ctypes.c_void_p]
s = ctypes.c_void_p(); e = ctypes.c_void_p()
print("create", hip.hipStreamCreateWithFlags(ctypes.byref(s), 1), hex(s.value)); sys.stdout.flush()
print("event", hip.hipEventCreateWithFlags(ctypes.byref(e), 2)); sys.stdout.flush()
print("record live", hip.hipEventRecord(e, s)); sys.stdout.flush()
print("sync", hip.hipStreamSynchronize(s)); print("destroy", hip.hipStreamDestroy(s)); sys.stdout.flush()
print("query dead", hip.hipStreamQuery(s)); sys.stdout.flush()
print("record dead", hip.hipEventRecord(e, s)); sys.stdout.flush()
print("survived")
