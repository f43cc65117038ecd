// Device scratch keyed by the launch stream (host side, HIP).
//
// The fused kernels park intermediate tensors in HBM (tail batching: a workgroup's slot is indexed by blockIdx.x; the 160x160
// banded kernels: a per-frame arena).  Two launches may use the same bytes only if they cannot run at the same time.  Launches
// issued to ONE stream serialise, launches on different streams may overlap in any order and for any length -- so the scratch
// is owned by the stream: a small stream -> region map, a region allocated on the first launch from a stream and grown when a
// later launch needs more.  The reference has one context and no streams at all (network.c:2929-2939); an entry point that
// accepts a stream has to be safe on any.  The map is mutex-protected (launches may come from several host threads).
#ifndef YF_STREAM_SCRATCH_H
#define YF_STREAM_SCRATCH_H
#include <hip/hip_runtime.h>
#include <mutex>
#include <vector>

struct yf_stream_scratch {
  struct Region { hipStream_t stream; char* ptr; size_t bytes; };
  std::mutex mu;
  std::vector<Region> regions;

  // Region of at least `bytes` bytes for launches on `s`.  Growing frees the old block first; hipFree waits for the device,
  // so no launch still reads the block that goes away.
  hipError_t get(hipStream_t s, size_t bytes, char** out) {
    std::lock_guard<std::mutex> lock(mu);
    for (Region& r : regions) {
      if (r.stream != s) continue;
      if (r.bytes < bytes) {
        if (r.ptr) (void)hipFree(r.ptr);
        r.ptr = nullptr; r.bytes = 0;
        const hipError_t rc = hipMalloc((void**)&r.ptr, bytes);
        if (rc != hipSuccess) return rc;
        r.bytes = bytes;
      }
      *out = r.ptr;
      return hipSuccess;
    }
    Region r = {s, nullptr, 0};
    const hipError_t rc = hipMalloc((void**)&r.ptr, bytes);
    if (rc != hipSuccess) return rc;
    r.bytes = bytes;
    regions.push_back(r);
    *out = r.ptr;
    return hipSuccess;
  }
  size_t count() {
    std::lock_guard<std::mutex> lock(mu);
    return regions.size();
  }
  void release() {
    std::lock_guard<std::mutex> lock(mu);
    for (Region& r : regions) if (r.ptr) (void)hipFree(r.ptr);
    regions.clear();
  }
};
#endif
