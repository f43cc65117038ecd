// Device scratch keyed by the launch stream (host side, HIP), BOUNDED.
//
// The fused kernels park intermediate tensors in HBM (int8 tail batching: a workgroup's slot is indexed by blockIdx.x; the fp16 kernel's
// park slots; the 160x160 banded kernels: a per-frame arena).  Two launches may use the same bytes only if they cannot run at the same
// time.  Launches issued to ONE stream serialise, launches on different streams may overlap in any order and for any length -- so a
// region is owned by a stream for as long as a launch on it may still run.  The reference has one context and no streams at all
// (network.c:2929-2939); an entry point that accepts a stream has to be safe on any, and a host that creates a stream per request must
// not grow the footprint without bound (round 3 kept one region per stream handle ever seen):
//   - a launch records an event on its region (mark()); a region whose event has completed is IDLE and is handed to the next
//     stream that asks (no allocation).  While ONE stream is all the object has ever seen, no event is recorded (an event between two
//     kernels of a stream costs ~4 us of back-to-back overlap per launch: 2.7 % of the 145 us headline kernel): the region is DIRTY
//     instead -- never idle -- so the first launch on a second stream gets a region of its own, and from then on every launch records
//     its event (the first one on the dirty region's stream covers its earlier launches: stream order);
//   - at most max_regions regions exist; when all are busy on other streams the caller waits for the one marked longest ago;
//   - release_stream() gives a stream's region back at once (yf_network_release_stream);
//   - hipStreamPerThread is one handle value for a different stream per host thread: the key is (handle, thread).
// The first launch on a new stream may allocate (a blocking hipMalloc): INTEGRATION.md says so.  The map is mutex-protected.
#ifndef YF_STREAM_SCRATCH_H
#define YF_STREAM_SCRATCH_H
#include <hip/hip_runtime.h>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct yf_stream_scratch {
  size_t max_regions = 8;                          // the owner may lower it (the 160x160 arena: 4)
  struct Region { hipStream_t stream; size_t thread; char* ptr; size_t bytes; hipEvent_t done; bool marked; unsigned long long stamp; bool dirty; bool acquired; };   // dirty: launched on without an event; acquired: handed out by get(), its launch not yet marked
  std::mutex mu;
  std::vector<Region> regions;
  unsigned long long clock = 0;

  static size_t thread_key(hipStream_t s) { return s == hipStreamPerThread ? std::hash<std::thread::id>()(std::this_thread::get_id()) : 0; }
  Region* find(hipStream_t s, size_t tk) { for (Region& r : regions) if (r.stream == s && r.thread == tk) return &r; return nullptr; }
  static bool idle(const Region& r) { return !r.acquired && !r.dirty && (!r.marked || hipEventQuery(r.done) == hipSuccess); }

  // Region of at least `bytes` bytes for a launch on `s`; the caller launches and then calls mark(s).
  hipError_t get(hipStream_t s, size_t bytes, char** out) {
    std::lock_guard<std::mutex> lock(mu);
    const size_t tk = thread_key(s);
    Region* r = find(s, tk);
    if (!r) {
      for (Region& c : regions) if (idle(c)) { r = &c; break; }                       // an idle region changes hands
      if (!r && regions.size() < max_regions) {
        Region n = {s, tk, nullptr, 0, nullptr, false, 0, false, false};
        const hipError_t rc = hipEventCreateWithFlags(&n.done, hipEventDisableTiming);
        if (rc != hipSuccess) return rc;
        regions.push_back(n);
        r = &regions.back();
      }
      if (!r) {                                                                       // all busy on other streams: wait for the one marked longest ago
        for (Region& c : regions) if (!c.dirty && !c.acquired && (!r || c.stamp < r->stamp)) r = &c;
        hipError_t rc;
        if (r) rc = hipEventSynchronize(r->done);
        else {                                                                        // only dirty regions left (nothing names their launches): wait for the device
          for (Region& c : regions) if (!c.acquired) { r = &c; break; }
          if (!r) return hipErrorNotReady;                                            // every region is between get() and mark() on another thread
          rc = hipDeviceSynchronize();
        }
        if (rc != hipSuccess) return rc;
      }
      r->stream = s; r->thread = tk; r->marked = false; r->dirty = false;
    }
    if (r->bytes < bytes) {                                                           // grow: hipFree waits for the device, nothing still reads the old block
      if (r->ptr) (void)hipFree(r->ptr);
      r->ptr = nullptr; r->bytes = 0;
      const hipError_t rc = hipMalloc((void**)&r->ptr, bytes);
      if (rc != hipSuccess) return rc;
      r->bytes = bytes;
    }
    r->stamp = ++clock;
    r->acquired = true;                                                               // not idle between get() and mark(), whoever asks
    *out = r->ptr;
    return hipSuccess;
  }
  // after the launch(es) that use the region obtained for `s`
  hipError_t mark(hipStream_t s) {
    std::lock_guard<std::mutex> lock(mu);
    Region* r = find(s, thread_key(s));
    if (!r) return hipSuccess;
    r->acquired = false;
    if (regions.size() == 1) { r->dirty = true; return hipSuccess; }                  // one stream so far: no event between its kernels (see above)
    r->marked = true; r->dirty = false;
    return hipEventRecord(r->done, s);
  }
  // the caller is done with `s` (about to destroy it): its region is freed once its last launch has completed
  hipError_t release_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(mu);
    const size_t tk = thread_key(s);
    for (size_t i = 0; i < regions.size(); ++i) {
      Region& r = regions[i];
      if (r.stream != s || r.thread != tk) continue;
      if (r.marked && !r.dirty) (void)hipEventSynchronize(r.done);
      if (r.ptr) (void)hipFree(r.ptr);                                                // (hipFree waits for the device: a dirty region's launches are through)
      (void)hipEventDestroy(r.done);
      regions.erase(regions.begin() + (long)i);
      return hipSuccess;
    }
    return hipSuccess;
  }
  size_t count() { std::lock_guard<std::mutex> lock(mu); return regions.size(); }
  size_t bytes_held() { std::lock_guard<std::mutex> lock(mu); size_t b = 0; for (const Region& r : regions) b += r.bytes; return b; }
  void release() {
    std::lock_guard<std::mutex> lock(mu);
    for (Region& r : regions) { if (r.ptr) (void)hipFree(r.ptr); if (r.done) (void)hipEventDestroy(r.done); }
    regions.clear();
  }
};
#endif
