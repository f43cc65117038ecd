#!/usr/bin/env python3
"""Host<->device copy rates of the box: pageable vs pinned, chunked, host memcpy.  DEV TOOL (sizes of one 4096-frame batch)."""
import time, ctypes, numpy as np, torch
n = 4096 * 9408
src = np.random.default_rng(0).integers(-128, 128, n, dtype=np.int8)
t_src = torch.from_numpy(src)
pin = torch.empty(n, dtype=torch.int8).pin_memory()
dev = torch.empty(n, dtype=torch.int8, device="cuda")
def timeit(f, k=10):
    f(); torch.cuda.synchronize()
    t = []
    for _ in range(k):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return min(t)
t = timeit(lambda: dev.copy_(t_src, non_blocking=True)); print(f"pageable H2D {n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
t = timeit(lambda: dev.copy_(pin, non_blocking=True)); print(f"pinned   H2D {n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
t = timeit(lambda: pin.copy_(t_src)); print(f"host memcpy pageable->pinned (torch, may be multi-threaded) {n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
dst = np.empty_like(src)
t0 = time.perf_counter(); ctypes.memmove(dst.ctypes.data, src.ctypes.data, n); t = time.perf_counter() - t0
t0 = time.perf_counter(); ctypes.memmove(dst.ctypes.data, src.ctypes.data, n); t = time.perf_counter() - t0
print(f"single-thread memmove {n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
out_n = 4096 * 882
d_out = torch.empty(out_n, dtype=torch.int8, device="cuda"); h_out = torch.empty(out_n, dtype=torch.int8); p_out = torch.empty(out_n, dtype=torch.int8).pin_memory()
t = timeit(lambda: h_out.copy_(d_out)); print(f"pageable D2H 3.6 MB {out_n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
t = timeit(lambda: p_out.copy_(d_out, non_blocking=True)); print(f"pinned   D2H 3.6 MB {out_n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
# chunked pinned copies on two streams
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def chunked(c):
    per = n // c
    for k in range(c):
        with torch.cuda.stream(s[k % 2]): dev[k * per:(k + 1) * per].copy_(pin[k * per:(k + 1) * per], non_blocking=True)
for c in (2, 4, 8): t = timeit(lambda: chunked(c)); print(f"pinned H2D in {c} chunks on two streams {n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
# hipHostRegister cost
hip = ctypes.CDLL("libamdhip64.so")
buf = np.empty(n, dtype=np.int8); buf[:] = 1
t0 = time.perf_counter(); rc = hip.hipHostRegister(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n), 0); t = time.perf_counter() - t0
print(f"hipHostRegister 38.5 MB rc={rc}: {t*1e3:.3f} ms")
t_reg = torch.from_numpy(buf)
t = timeit(lambda: dev.copy_(t_reg, non_blocking=True)); print(f"registered H2D (torch path) {n/t/1e9:6.1f} GB/s ({t*1e3:.3f} ms)")
t0 = time.perf_counter(); hip.hipHostUnregister(ctypes.c_void_p(buf.ctypes.data)); print(f"hipHostUnregister {1e3*(time.perf_counter()-t0):.3f} ms")
