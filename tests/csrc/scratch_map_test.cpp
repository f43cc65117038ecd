// CPU test of csrc/yf_stream_scratch.h against a fake HIP runtime (tests/csrc/fake_hip): the policy of the stream-keyed scratch map, in particular
// the paths a GPU test cannot reach deterministically -- a launch that fails between get() and mark(), events skipped while a stream is alone
// and recorded again when another stream turns up, destroyed streams whose handles must never reach the runtime.  Built and run by tests/test_sanitizers.py.
#include "yf_stream_scratch.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::fflush(stdout); std::exit(1); } } while (0)

static hipError_t good_launch(yf_stream_scratch& m, hipStream_t s, size_t bytes = 1024, char** ptr = nullptr) {
  yf_stream_scratch::Lease l;
  const hipError_t rc = m.get(s, bytes, &l);
  if (rc != hipSuccess) return rc;
  if (ptr) *ptr = l.ptr;
  s->enqueue();
  return l.mark();
}
static hipError_t failing_launch(yf_stream_scratch& m, hipStream_t s) {      // the shape of launch(): HIPCHK returns between get() and mark()
  yf_stream_scratch::Lease l;
  const hipError_t rc = m.get(s, 1024, &l);
  if (rc != hipSuccess) return rc;
  return hipErrorInvalidHandle;                                                // "hipGetLastError() failed": nothing was enqueued, the lease is dropped
}

int main() {
  {   // 1. a failed launch does not keep its region acquired: max_regions failures on distinct streams, then a new stream still gets one
    yf_stream_scratch m; m.max_regions = 4;
    hipStream_t s[6];
    for (auto& x : s) x = fake_hip::create();
    for (int i = 0; i < 4; ++i) { CHECK(failing_launch(m, s[i]) == hipErrorInvalidHandle); CHECK(m.acquired_count() == 0); }
    CHECK(m.count() <= 4);
    CHECK(good_launch(m, s[4]) == hipSuccess);                                 // round 4: hipErrorNotReady here, for good
    CHECK(good_launch(m, s[5]) == hipSuccess);
    CHECK(m.count() <= 4 && m.acquired_count() == 0);
    fake_hip::drain_all();
    for (int i = 1; i < 6; ++i) CHECK(good_launch(m, s[i]) == hipSuccess);
    CHECK(m.count() <= 4);
    m.release();
  }
  {   // 2. a stream that is alone records no event; a second stream makes both record; alone again -> no events again
    yf_stream_scratch m;
    hipStream_t a = fake_hip::create(), b = fake_hip::create();
    for (int i = 0; i < 10; ++i) CHECK(good_launch(m, a) == hipSuccess);
    CHECK(m.events_recorded == 0 && m.events_skipped == 10 && m.count() == 1);
    char *pa = nullptr, *pb = nullptr;
    CHECK(good_launch(m, b, 1024, &pb) == hipSuccess);                         // a's launches may still run: b must not get a's bytes
    CHECK(good_launch(m, a, 1024, &pa) == hipSuccess);
    CHECK(m.count() == 2 && pa != pb);
    const unsigned long long rec = m.events_recorded;
    CHECK(rec >= 2);                                                           // b's launch (a's region is dirty, i.e. not idle) and a's
    fake_hip::drain_all();                                                     // everything completed: whoever launches next is alone
    CHECK(good_launch(m, a) == hipSuccess && good_launch(m, a) == hipSuccess);
    CHECK(m.events_recorded == rec);                                           // round 4: every launch recorded an event for the life of the object
    m.release();
  }
  {   // 3. a stream that was synchronised and DESTROYED while its region was dirty (launched on without an event): the map must never hand that handle to
      //    the runtime (the fake aborts if it does).  While there is room the next stream gets a new region; at the cap a named region is waited for; and
      //    when only the dead stream's dirty region is left, the device is what can be waited for
    yf_stream_scratch m; m.max_regions = 2;
    hipStream_t a = fake_hip::create(), b = fake_hip::create(), c = fake_hip::create();
    CHECK(good_launch(m, a) == hipSuccess);                                    // alone: dirty, no event
    a->drain();
    fake_hip::destroy(a);
    CHECK(good_launch(m, b) == hipSuccess && m.count() == 2);                  // a's region is not idle (nothing names its launches): b gets its own, and records
    const long syncs = fake_hip::device_syncs();
    CHECK(good_launch(m, c) == hipSuccess && m.count() == 2);                  // the cap: b's region, after waiting for b's event
    CHECK(fake_hip::device_syncs() == syncs && m.acquired_count() == 0);
    m.release();
    yf_stream_scratch one; one.max_regions = 1;
    hipStream_t d = fake_hip::create(), e = fake_hip::create();
    CHECK(good_launch(one, d) == hipSuccess);
    d->drain();
    fake_hip::destroy(d);
    CHECK(good_launch(one, e) == hipSuccess && one.count() == 1 && fake_hip::device_syncs() == syncs + 1);
    one.release();
  }
  {   // 4. the same handle VALUE comes back for a new stream (this runtime reuses the address at once: tools/probe/stream_id_probe.py).
      //    (a) a runtime WITHOUT hipStreamGetId: the key is the handle value, the new stream finds the old region -- safe only because a stream may be
      //        dropped only once its launches have completed;
    fake_hip::has_stream_id() = false;
    {
      yf_stream_scratch m;
      hipStream_t a = fake_hip::create();
      char *p1 = nullptr, *p2 = nullptr;
      CHECK(good_launch(m, a, 1024, &p1) == hipSuccess);
      a->drain();
      fake_hip::destroy(a);
      fake_hip::reincarnate(a);                                                // "a new stream at the same address"
      CHECK(good_launch(m, a, 1024, &p2) == hipSuccess && p1 == p2 && m.count() == 1);
      m.release();
    }
    fake_hip::has_stream_id() = true;
    {   // (b) (round 6) WITH hipStreamGetId the key carries the stream's id: the successor is a NEW stream.  The predecessor was destroyed with its launch STILL IN
        //     FLIGHT (nothing drained, no release: the host's mistake) -- rounds 3-5 would have handed its region to the successor, both launches on the same bytes
      yf_stream_scratch m;
      hipStream_t b = fake_hip::create(), a = fake_hip::create();
      char *pb = nullptr, *p1 = nullptr, *p2 = nullptr;
      CHECK(good_launch(m, b, 1024, &pb) == hipSuccess);                       // company, so that a's launch is named by an event
      CHECK(good_launch(m, a, 1024, &p1) == hipSuccess);                       // in flight ...
      fake_hip::destroy(a);                                                    // ... and its stream destroyed
      fake_hip::reincarnate(a);
      CHECK(good_launch(m, a, 1024, &p2) == hipSuccess);
      CHECK(p2 != p1 && m.count() == 3);                                       // a region of its own: the bytes of the launch in flight are not shared
      // once the predecessor's launch has completed, its region is idle and changes hands like any other
      fake_hip::drain_all();
      hipStream_t c = fake_hip::create();
      char* pc = nullptr;
      CHECK(good_launch(m, c, 1024, &pc) == hipSuccess && m.count() == 3 && (pc == p1 || pc == pb || pc == p2));
      // and release_stream() of the successor frees ITS region, not the predecessor's
      const size_t before = m.count();
      CHECK(m.release_stream(a) == hipSuccess && m.count() == before - 1);
      m.release();
    }
  }
  {   // 5. 64 short-lived streams, three launches each, dropped after their work completed, never released: bounded footprint
    yf_stream_scratch m;
    for (int i = 0; i < 64; ++i) {
      hipStream_t s = fake_hip::create();
      for (int k = 0; k < 3; ++k) CHECK(good_launch(m, s) == hipSuccess);
      s->drain();
      fake_hip::destroy(s);
    }
    CHECK(m.count() <= 8 && m.acquired_count() == 0);
    m.release();
  }
  {   // 6. a failed allocation leaves nothing acquired and the next get() works
    yf_stream_scratch m;
    hipStream_t a = fake_hip::create();
    fake_hip::fail_mallocs() = 1;
    yf_stream_scratch::Lease l;
    CHECK(m.get(a, 4096, &l) == hipErrorOutOfMemory && m.acquired_count() == 0);
    CHECK(good_launch(m, a, 4096) == hipSuccess);
    m.release();
  }
  {   // 7. (round 6, ADVICE) get() into a lease that is still LIVE is refused before the lock is taken: overwriting it would settle() it under
      //    the map's own mutex -- a self-deadlock on a non-recursive std::mutex.  The live lease stays valid and marks normally.
    yf_stream_scratch m;
    hipStream_t a = fake_hip::create(), b = fake_hip::create();
    yf_stream_scratch::Lease l;
    CHECK(m.get(a, 1024, &l) == hipSuccess && m.acquired_count() == 1);
    CHECK(m.get(b, 1024, &l) == hipErrorInvalidValue);                         // round 5: never returned
    CHECK(m.get(a, 1024, nullptr) == hipErrorInvalidValue);
    CHECK(m.acquired_count() == 1 && l.ptr != nullptr);
    a->enqueue();
    CHECK(l.mark() == hipSuccess && m.acquired_count() == 0);
    CHECK(m.get(b, 1024, &l) == hipSuccess);                                   // a marked lease may be filled again
    CHECK(l.mark() == hipSuccess);
    m.release();
  }
  {   // 8. (round 6, ADVICE) more host threads than regions, every region between another thread's get() and mark(): the extra thread WAITS for a
      //    mark and then launches (round 5: hipErrorNotReady, a failed launch).  Reachable on the 160x160 path: arena160 has four regions and holds
      //    its lease across the whole chunk loop.
    yf_stream_scratch m; m.max_regions = 2;
    hipStream_t s[3];
    for (auto& x : s) x = fake_hip::create();
    yf_stream_scratch::Lease l0, l1;
    CHECK(m.get(s[0], 1024, &l0) == hipSuccess && m.get(s[1], 1024, &l1) == hipSuccess && m.acquired_count() == 2);
    std::atomic<int> state{0};
    hipError_t rc_third = hipErrorUnknown;
    std::thread third([&] { state = 1; rc_third = good_launch(m, s[2]); state = 2; });
    while (state.load() == 0) std::this_thread::yield();
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
    CHECK(state.load() == 1);                                                  // still waiting: nothing has been marked
    s[0]->enqueue();
    CHECK(l0.mark() == hipSuccess);                                            // wakes the waiter; s[0]'s region is dirty (it was alone), so the waiter...
    s[1]->enqueue();
    CHECK(l1.mark() == hipSuccess);                                            // ...gets a region through an event wait or the device
    third.join();                                                              // (nothing is drained before: the waiter has to wait for an event, the fake completes it)
    s[0]->drain(); s[1]->drain();
    CHECK(rc_third == hipSuccess && m.count() == 2 && m.acquired_count() == 0);
    const yf_stream_scratch::Stats st = m.stats();
    CHECK(st.acquire_waits >= 1 && st.regions == 2);
    CHECK(st.event_waits + st.device_syncs >= 1);                              // what the third stream paid is visible to the host
    // six threads, four launches each, on two regions: everybody gets through
    std::vector<std::thread> pool;
    std::atomic<int> failures{0};
    hipStream_t t[6];
    for (auto& x : t) x = fake_hip::create();
    for (int i = 0; i < 6; ++i) pool.emplace_back([&, i] { for (int k = 0; k < 4; ++k) { if (good_launch(m, t[i]) != hipSuccess) ++failures; t[i]->drain(); } });
    for (auto& th : pool) th.join();
    CHECK(failures.load() == 0 && m.count() == 2 && m.acquired_count() == 0);
    m.release();
  }
  {   // 9. the counters a host reads through yf_network_scratch_stats: a lone stream skips its events, company makes both record
    yf_stream_scratch m;
    hipStream_t a = fake_hip::create(), b = fake_hip::create();
    for (int i = 0; i < 5; ++i) CHECK(good_launch(m, a) == hipSuccess);
    yf_stream_scratch::Stats st = m.stats();
    CHECK(st.events_skipped == 5 && st.events_recorded == 0 && st.device_syncs == 0 && st.event_waits == 0 && st.acquire_waits == 0 && st.regions == 1);
    CHECK(good_launch(m, b) == hipSuccess);
    st = m.stats();
    CHECK(st.events_recorded == 1 && st.regions == 2);
    m.release();
  }
  fake_hip::free_all();
  std::printf("scratch map: ok\n");
  return 0;
}
