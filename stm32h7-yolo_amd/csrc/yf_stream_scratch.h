// Device scratch keyed by the launch stream (host side, HIP), BOUNDED.
//
// The fused kernels park intermediate tensors in HBM (int8 tail batching: a workgroup's slot is indexed by blockIdx.x; the fp16 kernel's
// park slots; the 160x160 banded kernels: a per-frame arena).  Two launches may use the same bytes only if they cannot run at the same
// time.  Launches issued to ONE stream serialise, launches on different streams may overlap in any order and for any length -- so a
// region is owned by a stream for as long as a launch on it may still run.  The reference has one context and no streams at all
// (network.c:2929-2939); an entry point that accepts a stream has to be safe on any, and a host that creates a stream per request must
// not grow the footprint without bound (round 3 kept one region per stream handle ever seen):
//   - get() returns a LEASE; the caller launches and calls lease.mark().  A lease that goes out of scope unmarked -- any early return
//     between get() and mark(): a failed launch, a failed later launch of a multi-launch call -- is marked by its destructor, so a region
//     never stays "acquired" (round 4 leaked one region per failed launch);
//   - mark() names the region's launches by an event, and a region whose event has completed is IDLE: the next stream that asks takes it
//     over (no allocation).  The event is SKIPPED while no other region is in use (every other region idle): an event between two kernels
//     of a stream costs ~4 us of back-to-back overlap per launch (2.7 % of the headline kernel).  The region is then DIRTY -- launched on,
//     not named, never idle.  The decision is per mark(), not "one region exists": a stream that has the object to itself again (the
//     others' launches completed) stops paying for events (round 4: once a second stream had been seen every launch recorded one for good);
//   - the map NEVER hands a stored stream handle to the runtime: a host may have destroyed that stream (below), and this runtime does not validate
//     handles -- hipStreamQuery / hipEventRecord on a destroyed stream is a segmentation fault (tools/probe/dead_stream_probe.py).  Every HIP call here
//     takes the CALLER's stream, or an event.  A dirty region whose stream never launches again therefore stays dirty (nothing can name its launches)
//     until release_stream(), or until every other region is busy and the device is synchronised (the all-busy path);
//   - at most max_regions regions exist; when all are busy on other streams the caller waits for the one marked longest ago -- or, when
//     only dirty regions are left, for the DEVICE (counted: stats().device_syncs, so a host can see when it pays that), or, when every region
//     is between another thread's get() and mark(), for one of those marks (a condition variable; round 5 failed that launch with
//     hipErrorNotReady, reachable with more host threads than regions);
//   - release_stream() gives a stream's region back at once (yf_network_release_stream);
//   - hipStreamPerThread is one handle value for a different stream per host thread: the key is (handle, thread), and such a region always
//     records its event.
// STREAM IDENTITY (round 6).  This runtime hands the SAME handle value to the next stream as soon as one is destroyed (tools/probe/stream_id_probe.py: forty
// create / destroy cycles, one handle value), so the handle alone cannot tell a stream from its successor -- a launch of a destroyed stream still in flight
// and a new stream would share a region (rounds 3-5: a documented contract, "drop a stream only once its launches have completed").  Where the runtime
// exports hipStreamGetId (ROCm 7.1's libamdhip64; resolved with dlsym: PyTorch 2.10's bundled runtime does not have it) a region is keyed by (handle, thread,
// stream ID): ids are unique over the life of the process, a successor under the same handle value is a NEW stream and gets a region of its own, and the
// predecessor's region changes hands like any other -- when its event has completed, or through the all-busy path.  Without the call the key is the handle
// value and the contract stands.
// The first launch on a new stream may allocate (a blocking hipMalloc): INTEGRATION.md says so.  The map is mutex-protected.
#ifndef YF_STREAM_SCRATCH_H
#define YF_STREAM_SCRATCH_H
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct yf_stream_scratch {
  size_t max_regions = 8;                          // the owner may lower it (the 160x160 arena: 4)
  // dirty: launched on without an event; acquired: handed out by get(), its launch not yet marked
  struct Region { hipStream_t stream; size_t thread; char* ptr; size_t bytes; hipEvent_t done; bool marked; unsigned long long stamp; bool dirty; bool acquired;
                  unsigned long long uid; bool has_uid; };        // uid: hipStreamGetId of the owner (has_uid: the runtime has the call)
  std::mutex mu;
  std::condition_variable cv_marked;               // a mark() or release: somebody waiting in get() for a region that is not acquired may look again
  std::vector<Region> regions;
  unsigned long long clock = 0;
  // diagnostics (yf_network_scratch_stats; tests): events recorded / skipped by mark(); on the all-busy path of get(): waits for the oldest
  // region's event, device synchronisations (only dirty regions left), waits for another thread's mark() (every region acquired)
  unsigned long long events_recorded = 0, events_skipped = 0, event_waits = 0, device_syncs = 0, acquire_waits = 0;
  struct Stats { unsigned long long events_recorded, events_skipped, event_waits, device_syncs, acquire_waits, regions; };
  Stats stats() { std::lock_guard<std::mutex> lock(mu); return Stats{events_recorded, events_skipped, event_waits, device_syncs, acquire_waits, regions.size()}; }

  // What get() hands out: the region's bytes, and the duty to mark it.  Movable, not copyable.
  struct Lease {
    yf_stream_scratch* owner = nullptr; hipStream_t stream = nullptr; char* ptr = nullptr;
    Lease() = default;
    Lease(const Lease&) = delete; Lease& operator=(const Lease&) = delete;
    Lease(Lease&& o) noexcept : owner(o.owner), stream(o.stream), ptr(o.ptr) { o.owner = nullptr; }
    Lease& operator=(Lease&& o) noexcept { if (this != &o) { settle(); owner = o.owner; stream = o.stream; ptr = o.ptr; o.owner = nullptr; } return *this; }
    ~Lease() { settle(); }
    // after the launch(es) that use the region
    hipError_t mark() { yf_stream_scratch* o = owner; owner = nullptr; return o ? o->mark(stream) : hipSuccess; }
   private:
    void settle() { if (owner) { (void)owner->mark(stream); owner = nullptr; } }     // an abandoned lease: whatever WAS launched is named, nothing stays acquired
  };

  static size_t thread_key(hipStream_t s) { return s == hipStreamPerThread ? std::hash<std::thread::id>()(std::this_thread::get_id()) : 0; }
  // the runtime's id of a LIVE stream (the caller's own: never a stored handle), false where the runtime has no hipStreamGetId
  static bool stream_uid(hipStream_t s, unsigned long long* id) {
#ifdef YF_FAKE_HIP_STREAM_ID                      /* the CPU test's fake runtime (tests/csrc/fake_hip) */
    return fake_hip::has_stream_id() && hipStreamGetId(s, id) == hipSuccess;
#else
    typedef hipError_t (*fn_t)(hipStream_t, unsigned long long*);
    static const fn_t fn = (fn_t)dlsym(RTLD_DEFAULT, "hipStreamGetId");
    return fn != nullptr && fn(s, id) == hipSuccess;
#endif
  }
  Region* find(hipStream_t s, size_t tk) {
    unsigned long long uid = 0;
    const bool has = stream_uid(s, &uid);
    for (Region& r : regions) if (r.stream == s && r.thread == tk && (!has || !r.has_uid || r.uid == uid)) return &r;
    return nullptr;
  }
  static bool idle(const Region& r) { return !r.acquired && !r.dirty && (!r.marked || hipEventQuery(r.done) == hipSuccess); }
  // Region of at least `bytes` bytes for a launch on `s`; the caller launches and then calls lease.mark().
  hipError_t get(hipStream_t s, size_t bytes, Lease* lease) {
    if (!lease || lease->owner) return hipErrorInvalidValue;                          // a lease that is still live is marked by its holder, not overwritten here (its settle() takes mu)
    std::unique_lock<std::mutex> lock(mu);
    const size_t tk = thread_key(s);
    Region* r = find(s, tk);
    while (r && r->acquired) {                                                        // another thread is between get() and mark() on this very stream: its turn first
      ++acquire_waits; cv_marked.wait(lock); r = find(s, tk);
    }
    while (!r) {
      for (Region& c : regions) if (idle(c)) { r = &c; break; }                       // an idle region changes hands
      if (!r && regions.size() < max_regions) {
        Region n = {s, tk, nullptr, 0, nullptr, false, 0, false, false, 0, false};
        const hipError_t rc = hipEventCreateWithFlags(&n.done, hipEventDisableTiming);
        if (rc != hipSuccess) return rc;
        regions.push_back(n);
        r = &regions.back();
      }
      if (!r) {                                                                       // all busy on other streams: wait for the one marked longest ago
        for (Region& c : regions) if (!c.dirty && !c.acquired && (!r || c.stamp < r->stamp)) r = &c;
        hipError_t rc;
        if (r) { ++event_waits; rc = hipEventSynchronize(r->done); }
        else {                                                                        // only dirty regions left (nothing names their launches): wait for the device
          for (Region& c : regions) if (!c.acquired) { r = &c; break; }
          if (!r) { ++acquire_waits; cv_marked.wait(lock); continue; }                // every region is between get() and mark() on another thread: wait for a mark, look again
          ++device_syncs; rc = hipDeviceSynchronize();
        }
        if (rc != hipSuccess) return rc;
      }
      r->stream = s; r->thread = tk; r->marked = false; r->dirty = false;
      r->has_uid = stream_uid(s, &r->uid);
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
    lease->owner = this; lease->stream = s; lease->ptr = r->ptr;
    return hipSuccess;
  }
  // after the launch(es) that use the region obtained for `s` (through the lease)
  hipError_t mark(hipStream_t s) {
    std::lock_guard<std::mutex> lock(mu);
    const size_t tk = thread_key(s);
    Region* r = find(s, tk);
    if (!r) return hipSuccess;
    r->acquired = false;
    cv_marked.notify_all();
    bool alone = tk == 0;                                                             // nobody else in sight: no event between this stream's kernels
    for (const Region& c : regions) if (&c != r && !idle(c)) { alone = false; break; }
    if (alone) { r->marked = false; r->dirty = true; ++events_skipped; return hipSuccess; }      // dirty outranks whatever an older event says: never idle until named
    r->marked = true; r->dirty = false; ++events_recorded;
    return hipEventRecord(r->done, s);
  }
  // the caller is done with `s` (about to destroy it): its region is freed once its last launch has completed
  hipError_t release_stream(hipStream_t s) {
    std::lock_guard<std::mutex> lock(mu);
    Region* mine = find(s, thread_key(s));
    for (size_t i = 0; i < regions.size(); ++i) {
      Region& r = regions[i];
      if (&r != mine) continue;
      if (r.marked && !r.dirty) (void)hipEventSynchronize(r.done);
      if (r.ptr) (void)hipFree(r.ptr);                                                // (hipFree waits for the device: a dirty region's launches are through)
      (void)hipEventDestroy(r.done);
      regions.erase(regions.begin() + (long)i);
      cv_marked.notify_all();
      return hipSuccess;
    }
    return hipSuccess;
  }
  size_t count() { std::lock_guard<std::mutex> lock(mu); return regions.size(); }
  size_t acquired_count() { std::lock_guard<std::mutex> lock(mu); size_t n = 0; for (const Region& r : regions) n += r.acquired; return n; }
  size_t bytes_held() { std::lock_guard<std::mutex> lock(mu); size_t b = 0; for (const Region& r : regions) b += r.bytes; return b; }
  void release() {
    std::lock_guard<std::mutex> lock(mu);
    for (Region& r : regions) { if (r.ptr) (void)hipFree(r.ptr); if (r.done) (void)hipEventDestroy(r.done); }
    regions.clear();
  }
};
#endif
