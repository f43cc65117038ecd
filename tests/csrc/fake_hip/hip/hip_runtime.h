/* A FAKE of the few HIP entry points csrc/yf_stream_scratch.h uses, for the CPU test of the stream-scratch map (tests/csrc/scratch_map_test.cpp):
 * own-authored, host-only, no GPU.  A "stream" is a small object that counts what was enqueued on it; a launch is `stream->enqueue()`; an
 * event recorded on a stream completes when the test calls `stream->drain()` (or `fake_hip::drain_all()`).  Destroyed streams are remembered:
 * handing one to the runtime ABORTS the test -- the real runtime does not validate stream handles, a call on a destroyed stream is a segmentation fault
 * (tools/probe/dead_stream_probe.py) -- so the map under test must never do it. */
#ifndef FAKE_HIP_RUNTIME_H
#define FAKE_HIP_RUNTIME_H
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorInvalidHandle = 400, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
enum { hipEventDisableTiming = 2 };
#define YF_FAKE_HIP_STREAM_ID 1      /* yf_stream_scratch.h calls hipStreamGetId below directly instead of resolving it with dlsym */
struct fake_stream { unsigned long long id = 0; std::atomic<long> enqueued{0}, completed{0}; void enqueue() { ++enqueued; } void drain() { completed = enqueued.load(); } };   /* atomic: the threaded cases drain their own streams while the map drains all */
struct fake_event { fake_stream* on = nullptr; long at = 0; };
typedef fake_stream* hipStream_t;
typedef fake_event* hipEvent_t;
namespace fake_hip {
inline std::set<fake_stream*>& live() { static std::set<fake_stream*> s; return s; }
inline std::vector<fake_stream*>& all() { static std::vector<fake_stream*> s; return s; }
inline long& mallocs() { static long n = 0; return n; }
inline long& frees() { static long n = 0; return n; }
inline long& device_syncs() { static long n = 0; return n; }
inline long& fail_mallocs() { static long n = 0; return n; }
inline bool& has_stream_id() { static bool b = true; return b; }      /* false: a runtime without hipStreamGetId (PyTorch 2.10's bundled libamdhip64) */
inline unsigned long long& next_id() { static unsigned long long n = 100; return n; }
inline hipStream_t create() { fake_stream* s = new fake_stream(); s->id = next_id()++; live().insert(s); all().push_back(s); return s; }
/* "the runtime hands the handle value of a destroyed stream to a NEW stream": same address, new id, nothing enqueued on the new one */
inline void reincarnate(hipStream_t s) { s->id = next_id()++; live().insert(s); }
inline void destroy(hipStream_t s) { live().erase(s); }            /* the object stays allocated: a stale handle value is still a valid address */
inline void free_all() { for (fake_stream* s : all()) delete s; all().clear(); live().clear(); }
inline void drain_all() { for (fake_stream* s : all()) s->drain(); }
}
#define hipStreamPerThread ((hipStream_t)2)
inline hipError_t hipStreamGetId(hipStream_t s, unsigned long long* id) {
  if (s != hipStreamPerThread && !fake_hip::live().count(s)) { std::fprintf(stderr, "fake HIP: hipStreamGetId on a destroyed stream\n"); std::abort(); }
  *id = s == hipStreamPerThread ? 1 : s->id; return hipSuccess;
}
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, int) { *e = new fake_event(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  if (!fake_hip::live().count(s)) { std::fprintf(stderr, "fake HIP: hipEventRecord on a destroyed stream (a crash in the real runtime)\n"); std::abort(); }
  e->on = s; e->at = s->enqueued; return hipSuccess;
}
inline hipError_t hipEventQuery(hipEvent_t e) { return (!e->on || e->on->completed >= e->at) ? hipSuccess : hipErrorNotReady; }
inline hipError_t hipEventSynchronize(hipEvent_t e) { if (e->on && e->on->completed < e->at) e->on->completed = (long)e->at; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { ++fake_hip::device_syncs(); fake_hip::drain_all(); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { if (fake_hip::fail_mallocs() > 0) { --fake_hip::fail_mallocs(); return hipErrorOutOfMemory; } *p = std::malloc(n); ++fake_hip::mallocs(); return hipSuccess; }
inline hipError_t hipFree(void* p) { fake_hip::drain_all(); std::free(p); ++fake_hip::frees(); return hipSuccess; }
#endif
