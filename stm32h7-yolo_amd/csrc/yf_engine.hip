// HIP engine behind the C-ABI: device context, table upload, launches of the fused kernel, box decode and
// frame-preparation kernels.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include "yf_engine.h"
#include "yf_stream_scratch.h"
#include "yf_decode.hip.h"
#define YF_NS yf
#include "yf_kernels.hip.h"
#undef YF_NS
#undef YF_STAGE_FN
#undef YF_H0
// 160x160 (BASELINE configs[4]): the same stage code on band-local buffers (three banded kernels; YF_LAB: also layer by layer over an HBM arena)
#define YF_NS yf160
#define YF_H0 160
#define YF_GENERIC 1
#include "yf_kernels.hip.h"
#undef YF_NS
#undef YF_H0
#undef YF_GENERIC
// The same kernels once more with the SIGN-FREE three-instruction epilogue on their dense convolutions (yf_kernels.hip.h, rq4 SIGNLESS): what runs when the network's
// requantisation rounding has no sign term (yf_network_set_requant_rounding: ties upward / single rounding; the host then folds ZR into C64 for the dense stages).
// Namespaces yfu (56x56) and yf160u (160x160).  The kernels of yf / yf160 above -- the reference rounding's -- are untouched by this.
#undef YF_STAGE_FN
#define YF_NS yfu
#define YF_RQ3_DENSE 1
#include "yf_kernels.hip.h"
#undef YF_NS
#undef YF_STAGE_FN
#undef YF_H0
#define YF_NS yf160u
#define YF_H0 160
#define YF_GENERIC 1
#include "yf_kernels.hip.h"
#undef YF_NS
#undef YF_H0
#undef YF_GENERIC
#undef YF_RQ3_DENSE
#ifdef YF_LAB
// laboratory: the 56x56 kernel once more, as a dump build that keeps the PRODUCTION stage order (yf_fused56.hip.h, YF_PDUMP) -- per-stage parity of what ships
#undef YF_STAGE_FN
#define YF_NS yfpd
#define YF_DUMP_PROD_ORDER 1
#include "yf_kernels.hip.h"
#undef YF_NS
#undef YF_H0
#undef YF_STAGE_FN
#define YF_NS yfpdu                 /* ... and of the kernel set with the sign-free dense epilogue */
#define YF_RQ3_DENSE 1
#include "yf_kernels.hip.h"
#undef YF_NS
#undef YF_RQ3_DENSE
#undef YF_DUMP_PROD_ORDER
#undef YF_H0
#endif
#include "gen/yf_decode_tables_gen.h"

namespace {

using yfdec::d_sig_bits;
using yfdec::d_exp_bits;

// ---------------------------------------------------------------------------------------------- box decode
// Stand-alone form (heads already in HBM): one wave per frame, see yf_decode.hip.h.
__global__ void __launch_bounds__(256) decode_kernel(const int8_t* __restrict__ heads, long n, int mode, float w_scale,
                                                     float h_scale, yf_det* __restrict__ dets, int* __restrict__ counts, int cap) {
  const int lane = threadIdx.x & 63;
  const long frame = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (frame >= n) return;
  yfdec::decode_frame(heads + frame * 882, frame, lane, mode, w_scale, h_scale, dets, counts, cap);
}

// ---------------------------------------------------------------------------------------------- compact wire records (multi-GPU exchange)
// A yf_det (28 B) carries its frame index (= its position), a float confidence and four int32 box edges -- all functions of the firing cell's six int8
// head values.  The WIRE form of a record is 12 bytes: {u8 anchor, row, col, 0, i8 q[6], u16 0}; lossless whatever the edges are.  One thread per record;
// slots beyond min(count, cap) are written as zeros (the record buffer may hold stale bytes there).
__global__ void __launch_bounds__(256) pack_dets_kernel(const yf_det* __restrict__ dets, const int* __restrict__ counts, const int8_t* __restrict__ heads,
                                                        uint32_t* __restrict__ wire, long n, int cap) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * cap) return;
  const long f = i / cap;
  const int k = (int)(i - f * cap);
  uint32_t w0 = 0, w1 = 0, w2 = 0;
  if (k < min(counts[f], cap)) {
    const yf_det d = dets[i];
    const int8_t* q = heads + f * 882 + ((int)d.row * 7 + (int)d.col) * 18 + 6 * (int)d.anchor;
    w0 = (uint32_t)d.anchor | ((uint32_t)d.row << 8) | ((uint32_t)d.col << 16);
    w1 = (uint32_t)(uint8_t)q[0] | ((uint32_t)(uint8_t)q[1] << 8) | ((uint32_t)(uint8_t)q[2] << 16) | ((uint32_t)(uint8_t)q[3] << 24);
    w2 = (uint32_t)(uint8_t)q[4] | ((uint32_t)(uint8_t)q[5] << 8);
  }
  wire[3 * i] = w0; wire[3 * i + 1] = w1; wire[3 * i + 2] = w2;
}
// The receiving side: the sparse int8 head a frame's wire records stand for -- -128 everywhere (a confidence logit that never fires), the six values of
// every transmitted (anchor, row, col) in their place.  Decoding it (decode_kernel) gives the sender's records in the sender's order.  One wave per frame.
__global__ void __launch_bounds__(256) unpack_dets_kernel(const uint32_t* __restrict__ wire, const int* __restrict__ counts, int8_t* __restrict__ heads, long n, int cap) {
  const int lane = threadIdx.x & 63;
  const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= n) return;
  int8_t* h = heads + f * 882;
  for (int b = lane; b < 882; b += 64) h[b] = (int8_t)-128;
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const int m = min(counts[f], cap);
  for (int k = lane; k < m; k += 64) {
    const uint32_t w0 = wire[3 * (f * cap + k)], w1 = wire[3 * (f * cap + k) + 1], w2 = wire[3 * (f * cap + k) + 2];
    const int a = (int)(w0 & 255u), row = (int)((w0 >> 8) & 255u), col = (int)((w0 >> 16) & 255u);
    if (a > 2 || row > 6 || col > 6) continue;                  // not a record this library packed
    int8_t* q = h + (row * 7 + col) * 18 + 6 * a;
    q[0] = (int8_t)(w1 & 255u); q[1] = (int8_t)((w1 >> 8) & 255u); q[2] = (int8_t)((w1 >> 16) & 255u); q[3] = (int8_t)(w1 >> 24);
    q[4] = (int8_t)(w2 & 255u); q[5] = (int8_t)((w2 >> 8) & 255u);
  }
}

// ---------------------------------------------------------------------------------------------- frame preparation
// stm32/X-CUBE-AI/App/yoloface.c:26-93: 112x112 big-endian RGB565 -> 2x2 box average per 5/6/5 field -> re-pack ->
// shift-expand -> value-128.  One thread per output pixel; 4 x 2-byte loads, 3 byte stores.
__global__ void __launch_bounds__(256) prepare_rgb565_kernel(const uint8_t* __restrict__ src, int8_t* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 3136) return;
  const long f = i / 3136; const int r = (int)(i - f * 3136);
  const int y = r / 56, x = r - y * 56;
  const uint8_t* s = src + f * (112 * 112 * 2);
  unsigned sr = 0, sg = 0, sb = 0;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int o = ((2 * y + dy) * 112 + (2 * x + dx)) * 2;
      const unsigned px = ((unsigned)s[o] << 8) | s[o + 1];
      sr += (px >> 11) & 0x1F; sg += (px >> 5) & 0x3F; sb += px & 0x1F;
    }
  const unsigned color = (((sr >> 2) & 0x1F) << 11) | (((sg >> 2) & 0x3F) << 5) | ((sb >> 2) & 0x1F);
  int8_t* o8 = dst + f * 9408 + r * 3;
  o8[0] = (int8_t)(((color & 0xF800) >> 8) - 128);
  o8[1] = (int8_t)(((color & 0x07E0) >> 3) - 128);
  o8[2] = (int8_t)(((color & 0x001F) << 3) - 128);
}

typedef void (*fused_fn)(const yf::NetParams);
struct Variant { int f, nw; bool dump; bool cam; fused_fn fn; size_t lds; size_t park; const char* name; bool prod_order; bool signless; };
// park: scratch bytes per frame slot of a workgroup; prod_order: the laboratory's dump build in the production stage order; signless: namespace yfu (sign-free dense epilogue)

#define YF_VARIANT(F, NW, DUMP) { F, NW, DUMP, false, (fused_fn)yf::yoloface56_fused<F, NW, DUMP>, yf::lds_bytes<F, NW, DUMP>(), yf::scratch_bytes_per_frame_slot<DUMP>(), \
                                  "yoloface56_fused<F=" #F ",NW=" #NW ">" }
#define YF_VARIANT_CAM(F, NW) { F, NW, false, true, (fused_fn)yf::yoloface56_fused<F, NW, false, true>, yf::lds_bytes<F, NW, false>(), \
                                yf::scratch_bytes_per_frame_slot<false>(), "yoloface56_fused<F=" #F ",NW=" #NW ",RGB565 input>" }
// The product: the batched shape <2,8>, the one-frame-per-workgroup shape <1,8> for small batches, the camera-input form of <2,8> and ONE debug
// (per-stage dump / stop_stage) build for the per-node observer.  A -DYF_LAB build (make lab) adds the other shapes for tools and tests.
#define YF_VARIANT_U(F, NW, DUMP) { F, NW, DUMP, false, (fused_fn)yfu::yoloface56_fused<F, NW, DUMP>, yfu::lds_bytes<F, NW, DUMP>(), yfu::scratch_bytes_per_frame_slot<DUMP>(), \
                                    "yoloface56_fused<F=" #F ",NW=" #NW ",sign-free dense epilogue>", false, true }
#define YF_VARIANT_U_CAM(F, NW) { F, NW, false, true, (fused_fn)yfu::yoloface56_fused<F, NW, false, true>, yfu::lds_bytes<F, NW, false>(), \
                                  yfu::scratch_bytes_per_frame_slot<false>(), "yoloface56_fused<F=" #F ",NW=" #NW ",RGB565 input,sign-free dense epilogue>", false, true }
const Variant k_variants[] = {
  YF_VARIANT(2, 8, false), YF_VARIANT(1, 8, false), YF_VARIANT(2, 8, true), YF_VARIANT_CAM(2, 8),
  // ... and the same four with the three-instruction epilogue on the dense convolutions, for the roundings without a sign term
  YF_VARIANT_U(2, 8, false), YF_VARIANT_U(1, 8, false), YF_VARIANT_U(2, 8, true), YF_VARIANT_U_CAM(2, 8),
#ifdef YF_LAB
  YF_VARIANT(1, 4, false), YF_VARIANT(2, 4, false), YF_VARIANT(4, 8, false), YF_VARIANT(2, 4, true),
  // the dump build in the production stage order (same NetParams layout; selected by YF_LAB_DUMP_PROD_ORDER=1 as the engine's dump variant)
  { 2, 8, true, false, (fused_fn)yfpd::yoloface56_fused<2, 8, true>, yfpd::lds_bytes<2, 8, true>(), yfpd::scratch_bytes_per_frame_slot<true>(),
    "yoloface56_fused<F=2,NW=8,dump in production order>", true },
  { 2, 8, true, false, (fused_fn)yfpdu::yoloface56_fused<2, 8, true>, yfpdu::lds_bytes<2, 8, true>(), yfpdu::scratch_bytes_per_frame_slot<true>(),
    "yoloface56_fused<F=2,NW=8,dump in production order,sign-free dense epilogue>", true, true },
#endif
};

}  // namespace

struct yf_engine {
  int device = 0;
  int cus = 0;
  size_t lds_per_cu = 0;                         // hipDeviceProp_t::maxSharedMemoryPerMultiProcessor
  int band_wgs_per_cu[2][3] = {{1, 1, 1}, {1, 1, 1}};   // resident workgroups per CU of the three banded 160x160 kernels on this device (occupancy query at creation), per kernel set
  bool signless = false;                         // the tables are built for -- and the launches take -- the kernels with the sign-free dense epilogue (namespaces yfu / yf160u)
#ifdef YF_LAB
  int grid_div = 1;                              // laboratory (YF_LAB_GRID_DIV): a launch takes 1 / grid_div of the resident grid (launch-policy what-ifs: several launches side by side)
  int fail_next_launch = 0;                      // laboratory: the next k fused launches get an invalid grid (tests the scratch lease on the failure path)
  bool dump_prod_order = false;                  // laboratory (YF_LAB_DUMP_PROD_ORDER=1): the dump entry points run the dump build that keeps the production stage order
#endif
  uint8_t* d_tab = nullptr;
  yf_table_index ix;
  const Variant* var = nullptr;
  const Variant* var_dump = nullptr;
  void* d_in = nullptr; void* d_out = nullptr; long stage_cap = 0;
  bool layerwise160 = false;
  long chunk160 = 1024;                          // frames per 160x160 arena chunk (289 KB per frame: 296 MB for BASELINE's 1024-frame batch; the arena is sized by the
                                                 // actual batch).  YF_160_CHUNK overrides: 512 costs 3.5 %, 256 costs 22 % of the 1024-frame rate (three launches per chunk)
  yf_stream_scratch arena160;                    // 160x160 per-frame HBM arena, one per launch stream (yf_stream_scratch.h)
  yf_stream_scratch park; size_t park_region = 0;   // tail batching scratch of the fused kernel, one region per launch stream
  hipStream_t own_stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;
  // ---- host-buffer path (ai_network_run on the caller's arrays, yf_engine_run_host)
  const Variant* var_small = nullptr;            // batches of at most SMALL_N frames: one frame per workgroup (lowest latency); null once a shape is configured
  hipStream_t pipe_stream[2] = {nullptr, nullptr};   // large batches: chunks alternate between two streams (upload of chunk k+1 behind the kernel of chunk k)
  hipEvent_t pipe_ev[2] = {nullptr, nullptr};
  void* h_small_in = nullptr; void* h_small_out = nullptr;   // pinned, device-mapped: tiny batches are read / written by the kernel in place (no copy submissions)
  struct Downloader* dl = nullptr;               // worker thread: results of finished chunks go to the caller's array while later chunks upload
};
constexpr long SMALL_N = 512;                    // frames: below this every frame gets a workgroup of its own
constexpr long ZERO_COPY_N = 8;                  // frames: the kernel reads the caller's frames from pinned host memory
constexpr long PIPE_MIN_N = 2048, PIPE_CHUNK_DEFAULT = 3072, PIPE_LAST_DEFAULT = 1024;

// Downloads device -> pageable host memory on a thread of their own.  A copy into pageable memory blocks the calling thread until
// it is done; issued from the thread that also uploads, every download would hold up the next chunk's upload.
struct Downloader {
  struct Job { hipEvent_t ready; const void* src; void* dst; size_t bytes; };
  int device = 0;
  hipStream_t stream = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_job, cv_idle;
  std::deque<Job> q;
  int pending = 0;
  bool stop = false;
  hipError_t status = hipSuccess;
  void loop() {
    (void)hipSetDevice(device);
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_job.wait(lk, [&] { return stop || !q.empty(); });
        if (q.empty()) return;
        j = q.front(); q.pop_front();
      }
      hipError_t rc = hipStreamWaitEvent(stream, j.ready, 0);
      if (rc == hipSuccess) rc = hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, stream);
      if (rc == hipSuccess) rc = hipStreamSynchronize(stream);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (rc != hipSuccess && status == hipSuccess) status = rc;
        --pending; cv_idle.notify_all();
      }
    }
  }
  hipError_t start(int dev) {
    device = dev;
    const hipError_t rc = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (rc != hipSuccess) return rc;
    th = std::thread([this] { loop(); });
    return hipSuccess;
  }
  void submit(const Job& j) { { std::lock_guard<std::mutex> lk(mu); q.push_back(j); ++pending; } cv_job.notify_one(); }
  hipError_t drain() { std::unique_lock<std::mutex> lk(mu); cv_idle.wait(lk, [&] { return pending == 0; }); const hipError_t rc = status; status = hipSuccess; return rc; }
  // wait until fewer than `below` downloads are outstanding (the submitter re-uses a chunk's event two chunks later)
  hipError_t drain_to(int below) { std::unique_lock<std::mutex> lk(mu); cv_idle.wait(lk, [&] { return pending < below; }); return status; }
  ~Downloader() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_job.notify_all();
    if (th.joinable()) th.join();
    if (stream) (void)hipStreamDestroy(stream);
  }
};

#define HIPCHK(e_, call) do { hipError_t rc_ = (call); if (rc_ != hipSuccess) { \
    (e_)->err = std::string(#call) + ": " + hipGetErrorString(rc_); return YF_ENG_ERR_HIP; } } while (0)

static const Variant* shape_for(const yf_engine* e, long n);
static const Variant* find_variant(int f, int nw, bool dump, bool cam = false, bool prod_order = false, bool signless = false) {
  for (const Variant& v : k_variants) if (v.f == f && v.nw == nw && v.dump == dump && v.cam == cam && v.prod_order == prod_order && v.signless == signless) return &v;
  return nullptr;
}

// the debug (dump) build of a shape: the staged-order build the per-node observer runs; a laboratory engine started with YF_LAB_DUMP_PROD_ORDER=1 takes the
// dump build that keeps the production stage order instead (shape <2,8> only)
static const Variant* dump_variant_for(const yf_engine* e, int f, int nw) {
#ifdef YF_LAB
  if (e->dump_prod_order) { const Variant* v = find_variant(f, nw, true, false, true, e->signless); if (v) return v; }
#endif
  return find_variant(f, nw, true, false, false, e->signless);
}
// the engine's three kernel shapes for a configuration (f, nw; 0 = the automatic choice), in the kernel set its tables are built for
static void select_variants(yf_engine* e, int f, int nw, bool automatic) {
  e->var = find_variant(f, nw, false, false, false, e->signless);
  e->var_dump = dump_variant_for(e, f, nw);
  e->var_small = automatic ? find_variant(1, 8, false, false, false, e->signless) : nullptr;
}

#ifdef YF_LAB
// 160x160 variant: launches the 27 per-stage kernels in order
template <int ST>
static int launch160_from(yf_engine* e, const yf160::GenParams& prm, unsigned grid, hipStream_t s) {
  if constexpr (ST < yf160::GEN_STAGES) {
    hipLaunchKernelGGL((yf160::generic_stage_kernel<ST, 8>), dim3(grid), dim3(512), yf160::LUT_BYTES, s, prm);
    { hipError_t rc_ = hipGetLastError(); if (rc_ != hipSuccess) { e->err = std::string("generic stage launch: ") + hipGetErrorString(rc_); return YF_ENG_ERR_HIP; } }
    return launch160_from<ST + 1>(e, prm, grid, s);
  } else {
    return YF_ENG_OK;
  }
}

#endif

// 160x160, banded form: three kernels, each fusing a group of stages over row bands staged through LDS
struct BandKernel { const void* fn; const char* name; unsigned threads; size_t lds; int jobs_per_frame; };   // (workgroups per CU: per ENGINE, yf_engine::band_wgs_per_cu)
#ifndef YF_K1_NW
#define YF_K1_NW 8
#endif
#define YF_K1_NW_ YF_K1_NW
static const BandKernel k_band_fused[2][3] = {   // round 3: K2 and K3 fused (three tensors cross HBM instead of five); [1]: the set with the sign-free dense epilogue
  {{(const void*)yf160::band::band_k1<YF_K1_NW_>, "band_k1", YF_K1_NW_ * 64, (size_t)yf160::band::K1_LDS, yf160::band::K1_BANDS},
   {(const void*)yf160::band::band_k23<8>, "band_k23", 512, (size_t)yf160::band::K23_LDS, yf160::band::K23_BANDS},
   {(const void*)yf160::band::band_k4<8>,  "band_k4", 512,  (size_t)yf160::band::K4_LDS, 1}},
  {{(const void*)yf160u::band::band_k1<YF_K1_NW_>, "band_k1 (sign-free dense epilogue)", YF_K1_NW_ * 64, (size_t)yf160u::band::K1_LDS, yf160u::band::K1_BANDS},
   {(const void*)yf160u::band::band_k23<8>, "band_k23 (sign-free dense epilogue)", 512, (size_t)yf160u::band::K23_LDS, yf160u::band::K23_BANDS},
   {(const void*)yf160u::band::band_k4<8>,  "band_k4 (sign-free dense epilogue)", 512,  (size_t)yf160u::band::K4_LDS, 1}},
};
static int launch160_banded(yf_engine* e, const yf160::band::Params& prm, hipStream_t s) {
  for (int i = 0; i < 3; ++i) {
    const BandKernel& k = k_band_fused[e->signless][i];
    const long jobs = prm.n * k.jobs_per_frame;
    const long full = (long)e->cus * e->band_wgs_per_cu[e->signless][i];  // persistent grid: every workgroup resident, jobs grid-strided
    const unsigned grid = (unsigned)(jobs < full ? jobs : full);
    void* args[] = {(void*)&prm};
    const hipError_t rc = hipLaunchKernel(k.fn, dim3(grid), dim3(k.threads), args, k.lds, s);
    if (rc != hipSuccess) { e->err = std::string(k.name) + " launch: " + hipGetErrorString(rc); return YF_ENG_ERR_HIP; }
  }
  return YF_ENG_OK;
}

// the kernels address the tables at compiled-in offsets (yf_kernels.hip.h, TablePlan): the blob must be laid out that way
static bool layout_is_the_compiled_plan(const yf_table_index* ix) {
  bool same = (int)ix->lut_off == yf::PLAN.lut_off && (int)ix->total_bytes == yf::PLAN.total;
  for (int i = 0; i < YF_N_DENSE; ++i) same = same && (int)ix->dense[i].w_off == yf::PLAN.w_off[i] && (int)ix->dense[i].c_off == yf::PLAN.c_off[i];
  for (int i = 0; i < YF_N_DW; ++i) same = same && (int)ix->dw[i].g_off == yf::PLAN.g_off[i];
  for (int i = 0; i < YF_N_CS; ++i)
    same = same && (int)ix->cs_v_off[i] == yf::PLAN.vb_off[i] && (int)ix->cs_v_bytes[i] == yf::PLAN.vb_bytes[i] && (int)ix->cs_s_off[i] == yf::PLAN.sb_off[i];
  return same;
}

extern "C" {

int yf_engine_create(int device, const uint8_t* table_blob, const yf_table_index* ix, int signless_dense, yf_engine** out, char* err, size_t errlen) {
  auto fail = [&](const std::string& m, int code) { if (err && errlen) snprintf(err, errlen, "%s", m.c_str()); return code; };
  if (!table_blob || !ix || !out) return fail("bad arguments", YF_ENG_ERR_ARG);
  int ndev = 0;
  hipError_t rc = hipGetDeviceCount(&ndev);
  if (rc != hipSuccess || ndev <= 0)
    return fail(std::string("no HIP device: ") + (rc != hipSuccess ? hipGetErrorString(rc) : "device count is 0"), YF_ENG_ERR_NO_DEVICE);
  if (device < 0 || device >= ndev) return fail("device index out of range", YF_ENG_ERR_ARG);
  yf_engine* e = new yf_engine();
  e->device = device; e->ix = *ix;
  e->arena160.max_regions = 4;                   // at most four streams' 160x160 arenas at a time (others wait for the one used longest ago)
  // every failure path releases what has been acquired so far (ai_network_init may be called again and again: network.c:3385-3399)
  auto bail = [&](hipError_t r, const char* what) { std::string m = std::string(what) + ": " + hipGetErrorString(r); yf_engine_destroy(e); return fail(m, YF_ENG_ERR_HIP); };
  auto quit = [&](const std::string& m, int code) { yf_engine_destroy(e); return fail(m, code); };
  if ((rc = hipSetDevice(device)) != hipSuccess) return bail(rc, "hipSetDevice");
  hipDeviceProp_t prop;
  if ((rc = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(rc, "hipGetDeviceProperties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return quit(std::string("unsupported GPU ") + prop.gcnArchName + " (this library is gfx950 only)", YF_ENG_ERR_NO_DEVICE);
  e->cus = prop.multiProcessorCount;
  e->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;      // 160 KB on gfx950: how many workgroups of a shape a CU holds (grid size, scratch slots)
  for (const Variant& v : k_variants)
    if (e->lds_per_cu < v.lds) return quit("the device reports " + std::to_string(e->lds_per_cu) + " bytes of LDS per CU, " + v.name + " needs " + std::to_string(v.lds), YF_ENG_ERR_NO_DEVICE);
  if (!layout_is_the_compiled_plan(ix)) return quit("table blob layout differs from the layout compiled into the kernels", YF_ENG_ERR_ARG);
  if ((rc = hipMalloc((void**)&e->d_tab, ix->total_bytes)) != hipSuccess) return bail(rc, "hipMalloc(tables)");
  if ((rc = hipMemcpy(e->d_tab, table_blob, ix->total_bytes, hipMemcpyHostToDevice)) != hipSuccess) return bail(rc, "hipMemcpy(tables)");
  for (int i = 1; i < 256; ++i) {           // the fused decode compares quantised confidences (yf_decode_q_threshold): the table must not decrease
    float a, b; memcpy(&a, &yf_sigmoid_bits[i - 1], 4); memcpy(&b, &yf_sigmoid_bits[i], 4);
    if (b < a) return bail(hipErrorInvalidValue, "sigmoid table is not monotonic");
  }
  if ((rc = hipMemcpyToSymbol(HIP_SYMBOL(d_sig_bits), yf_sigmoid_bits, sizeof yf_sigmoid_bits)) != hipSuccess) return bail(rc, "hipMemcpyToSymbol(sigmoid)");
  if ((rc = hipMemcpyToSymbol(HIP_SYMBOL(d_exp_bits), yf_exp_bits, sizeof yf_exp_bits)) != hipSuccess) return bail(rc, "hipMemcpyToSymbol(exp)");
  for (const Variant& v : k_variants) {
    // The byte LUTs are addressed absolutely (LDS offset LUT_ID*256): the dynamic segment must start at LDS address 0,
    // i.e. the kernel must not have picked up any static LDS.
    hipFuncAttributes at;
    if ((rc = hipFuncGetAttributes(&at, (const void*)v.fn)) != hipSuccess) return bail(rc, "hipFuncGetAttributes");
    if (at.sharedSizeBytes != 0) return quit(std::string(v.name) + ": kernel has static LDS, absolute LUT addressing is invalid", YF_ENG_ERR_HIP);
    if ((rc = hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)v.lds)) != hipSuccess)
      return bail(rc, "hipFuncSetAttribute(max dynamic LDS)");
  }
#ifdef YF_LAB
  {
    hipFuncAttributes at;
    if ((rc = hipFuncGetAttributes(&at, (const void*)yf160::generic_stage_kernel<1, 8>)) != hipSuccess) return bail(rc, "hipFuncGetAttributes");
    if (at.sharedSizeBytes != 0) return quit("generic stage kernel has static LDS", YF_ENG_ERR_HIP);
  }
  { const char* lw = getenv("YF_160_LAYERWISE"); e->layerwise160 = lw && lw[0] == '1'; }     // the lab library's layer-by-layer form (debugging)
#endif
  auto prepare_band = [&](const BandKernel& k, int* wgs_per_cu) -> int {
    hipFuncAttributes at;
    if ((rc = hipFuncGetAttributes(&at, k.fn)) != hipSuccess) return bail(rc, "hipFuncGetAttributes");
    if (at.sharedSizeBytes != 0) return quit(std::string(k.name) + ": kernel has static LDS, absolute LUT addressing is invalid", YF_ENG_ERR_HIP);
    if ((rc = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k.lds)) != hipSuccess)
      return bail(rc, "hipFuncSetAttribute(max dynamic LDS)");
    int occ = 0;
    if ((rc = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k.fn, (int)k.threads, k.lds)) != hipSuccess) return bail(rc, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
    *wgs_per_cu = occ > 0 ? occ : 1;                          // a property of THIS engine's device (round 5 wrote it into the process-wide kernel table)
    return YF_ENG_OK;
  };
  for (int u = 0; u < 2; ++u)
    for (int i = 0; i < 3; ++i) { const int r = prepare_band(k_band_fused[u][i], &e->band_wgs_per_cu[u][i]); if (r != YF_ENG_OK) return r; }
  { const char* ck = getenv("YF_160_CHUNK"); if (ck && atol(ck) > 0) e->chunk160 = atol(ck); }
#ifdef YF_LAB
  { const char* fl = getenv("YF_LAB_FAIL_LAUNCHES"); if (fl) e->fail_next_launch = atoi(fl); }
  { const char* gd = getenv("YF_LAB_GRID_DIV"); if (gd && atoi(gd) > 1) e->grid_div = atoi(gd); }
  { const char* pd = getenv("YF_LAB_DUMP_PROD_ORDER"); e->dump_prod_order = pd && pd[0] == '1'; }
#endif
  {   // every shape keeps at most 4 frames in flight per CU (grid x frames per group)
    size_t park = 0;
    for (const Variant& v : k_variants) park = v.park > park ? v.park : park;
    size_t slots = 4;
    for (const Variant& v : k_variants) { const size_t need = (e->lds_per_cu / v.lds) * (size_t)v.f; slots = need > slots ? need : slots; }   // (a lab what-if with overlapping arenas holds three workgroups per CU)
    e->park_region = (size_t)e->cus * slots * park;      // allocated per stream on its first launch (launch())
  }
  if ((rc = hipStreamCreate(&e->own_stream)) != hipSuccess) return bail(rc, "hipStreamCreate");
  if ((rc = hipEventCreate(&e->ev0)) != hipSuccess || (rc = hipEventCreate(&e->ev1)) != hipSuccess) return bail(rc, "hipEventCreate");
  for (int k = 0; k < 2; ++k) {
    if ((rc = hipStreamCreateWithFlags(&e->pipe_stream[k], hipStreamNonBlocking)) != hipSuccess) return bail(rc, "hipStreamCreateWithFlags");
    if ((rc = hipEventCreateWithFlags(&e->pipe_ev[k], hipEventDisableTiming)) != hipSuccess) return bail(rc, "hipEventCreateWithFlags");
  }
  if ((rc = hipHostMalloc(&e->h_small_in, (size_t)ZERO_COPY_N * 9408, hipHostMallocMapped)) != hipSuccess ||
      (rc = hipHostMalloc(&e->h_small_out, (size_t)ZERO_COPY_N * 882, hipHostMallocMapped)) != hipSuccess) return bail(rc, "hipHostMalloc(pinned staging)");
  e->signless = signless_dense != 0;
  select_variants(e, 2, 8, true);
  *out = e;
  return YF_ENG_OK;
}

int yf_engine_set_tables(yf_engine* e, const uint8_t* table_blob, const yf_table_index* ix, int signless_dense) {
  if (!e || !table_blob || !ix) return YF_ENG_ERR_ARG;
  if (!find_variant(e->var->f, e->var->nw, false, false, false, signless_dense != 0)) { e->err = "the configured kernel shape has no build for this rounding"; return YF_ENG_ERR_VARIANT; }
  if (!layout_is_the_compiled_plan(ix)) { e->err = "table blob layout differs from the layout compiled into the kernels"; return YF_ENG_ERR_ARG; }
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, hipDeviceSynchronize());             // launches in flight read the old constants to their end
  HIPCHK(e, hipMemcpy(e->d_tab, table_blob, ix->total_bytes, hipMemcpyHostToDevice));
  e->ix = *ix;
  const bool automatic = e->var_small != nullptr;
  e->signless = signless_dense != 0;             // the tables and the kernels that read them change together (the device is idle here)
  select_variants(e, e->var->f, e->var->nw, automatic);
  return YF_ENG_OK;
}

void yf_engine_destroy(yf_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();                  // a launch still in flight finishes before its tables and scratch are freed (no reliance on hipFree's implicit wait)
  if (e->d_tab) (void)hipFree(e->d_tab);
  if (e->d_in) (void)hipFree(e->d_in);
  if (e->d_out) (void)hipFree(e->d_out);
  e->arena160.release();
  e->park.release();
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e->dl;                                  // joins the download thread
  for (int k = 0; k < 2; ++k) {
    if (e->pipe_ev[k]) (void)hipEventDestroy(e->pipe_ev[k]);
    if (e->pipe_stream[k]) (void)hipStreamDestroy(e->pipe_stream[k]);
  }
  if (e->h_small_in) (void)hipHostFree(e->h_small_in);
  if (e->h_small_out) (void)hipHostFree(e->h_small_out);
  delete e;
}

int yf_engine_table_plan(int32_t* out, int cap) {
  const int need = 2 * YF_N_DENSE + YF_N_DW + 2 + 3 * YF_N_CS;
  if (!out || cap < need) return need;
  int k = 0;
  for (int i = 0; i < YF_N_DENSE; ++i) out[k++] = yf::PLAN.w_off[i];
  for (int i = 0; i < YF_N_DENSE; ++i) out[k++] = yf::PLAN.c_off[i];
  for (int i = 0; i < YF_N_DW; ++i) out[k++] = yf::PLAN.g_off[i];
  out[k++] = yf::PLAN.lut_off; out[k++] = yf::PLAN.total;
  for (int i = 0; i < YF_N_CS; ++i) out[k++] = yf::PLAN.vb_off[i];
  for (int i = 0; i < YF_N_CS; ++i) out[k++] = yf::PLAN.vb_bytes[i];
  for (int i = 0; i < YF_N_CS; ++i) out[k++] = yf::PLAN.sb_off[i];
  return need;
}

int yf_engine_variant_exists(int frames_per_wg, int waves_per_wg) { return find_variant(frames_per_wg, waves_per_wg, false) != nullptr; }

int yf_engine_configure(yf_engine* e, int frames_per_wg, int waves_per_wg) {
  if (!e) return YF_ENG_ERR_ARG;
  if (frames_per_wg < 0) {                       /* back to the automatic choice: throughput shape, small batches one frame per workgroup */
    select_variants(e, 2, 8, true);
    return YF_ENG_OK;
  }
  const int f = frames_per_wg > 0 ? frames_per_wg : e->var->f, nw = waves_per_wg > 0 ? waves_per_wg : e->var->nw;
  if (!find_variant(f, nw, false, false, false, e->signless)) { e->err = "no such kernel variant"; return YF_ENG_ERR_VARIANT; }
  select_variants(e, f, nw, false);              /* an explicitly configured shape runs every batch size; debug build of the SAME shape, or none: the dump / stage-timing
                                                    entry points refuse instead of running another shape */
  return YF_ENG_OK;
}

const char* yf_engine_error(const yf_engine* e) { return e ? e->err.c_str() : "null engine"; }
const char* yf_engine_kernel_name(const yf_engine* e) { return e && e->var ? e->var->name : ""; }
const char* yf_engine_kernel_name_for(const yf_engine* e, long n) { return e && e->var ? shape_for(e, n)->name : ""; }
#ifndef YF_BUILD_ID
#define YF_BUILD_ID "unstamped"
#endif
const char* yf_engine_build_id(void) { return YF_BUILD_ID; }
long yf_engine_dump_bytes(void) { return yf::DumpOffsets::TOTAL; }

// The fused decode tests the QUANTISED confidence: the first entry of the (monotonic) sigmoid table that passes the mode's comparison --
// conf > 0.7f for the Python decode, (double)conf >= 0.7 for the firmware's -- as an int8 value; 128 = none passes.
static int yf_decode_q_threshold(int mode) {
  for (int i = 0; i < 256; ++i) {
    float v; memcpy(&v, &yf_sigmoid_bits[i], 4);
    if (mode == YF_DECODE_PY ? (v > 0.7f) : ((double)v >= 0.7)) return i - 128;
  }
  return 128;
}
struct DecodeArgs { void* dets; void* counts; int cap, mode; float w_scale, h_scale; };

// Kernel shape for a batch of n frames: up to SMALL_N frames run one frame per workgroup (every frame on a CU of its own: a 1-frame
// batch takes 24 us instead of 34), larger batches the throughput shape.
static const Variant* shape_for(const yf_engine* e, long n) { return (e->var_small && n <= SMALL_N) ? e->var_small : e->var; }

static int launch(yf_engine* e, const Variant* v, const void* d_in, void* d_out, void* d_dump, long n, hipStream_t s, int stop_stage = -1,
                  const DecodeArgs* dec = nullptr) {
  if (n <= 0) return YF_ENG_OK;
  if (((uintptr_t)d_in & 3) != 0) { e->err = "input must be 4-byte aligned"; return YF_ENG_ERR_ARG; }
  if (((uintptr_t)d_out & 1) != 0) { e->err = "output must be 2-byte aligned"; return YF_ENG_ERR_ARG; }
  yf::NetParams prm;
  prm.in = (const int8_t*)d_in; prm.out = (int8_t*)d_out; prm.n = n; prm.tab = e->d_tab; prm.dump = (int8_t*)d_dump; prm.stop_stage = stop_stage;
  prm.dets = nullptr; prm.counts = nullptr; prm.cap = 0; prm.mode = 0; prm.w_scale = prm.h_scale = 1.f;
  prm.q_thr = 128;
  if (dec) { prm.dets = (yf_det*)dec->dets; prm.counts = (int*)dec->counts; prm.cap = dec->cap; prm.mode = dec->mode; prm.w_scale = dec->w_scale; prm.h_scale = dec->h_scale;
             prm.q_thr = yf_decode_q_threshold(dec->mode); }
  const long groups = (n + v->f - 1) / v->f;
  const int per_cu = (int)(e->lds_per_cu / v->lds) > 0 ? (int)(e->lds_per_cu / v->lds) : 1;
  long grid = (long)e->cus * per_cu;
#ifdef YF_LAB
  if (e->grid_div > 1) grid = grid / e->grid_div > 0 ? grid / e->grid_div : 1;
#endif
  if (grid > groups) grid = groups;
  prm.scratch = nullptr;
  yf_stream_scratch::Lease lease;    // marks its region on EVERY way out of this function (a failed launch must not keep a region acquired)
  if (v->park) {   // tail batching: a workgroup parks one group's T15 (f frames) in HBM, slot = blockIdx.x.  The region belongs to
                   // the launch STREAM: launches on one stream serialise, launches on different streams never share bytes.
    const size_t need = (size_t)grid * v->f * v->park;
    if (need > e->park_region) { e->err = "tail scratch region too small for this kernel shape"; return YF_ENG_ERR_VARIANT; }
    HIPCHK(e, e->park.get(s, e->park_region, &lease));
    prm.scratch = lease.ptr;
  }
#ifdef YF_LAB
  if (e->fail_next_launch > 0) { --e->fail_next_launch; grid = 0; }    // laboratory only (YF_LAB_FAIL_LAUNCHES=k): an invalid grid for the first k launches, to rehearse the failure path below
#endif
  hipLaunchKernelGGL(v->fn, dim3((unsigned)grid), dim3(v->nw * 64), v->lds, s, prm);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, lease.mark());                                   // the region is busy until this launch has completed (yf_stream_scratch.h)
  return YF_ENG_OK;
}

int yf_engine_run_device(yf_engine* e, const void* d_in, void* d_out, void* d_dump, long n, void* stream) {
  if (!e || !d_in || !d_out || n < 0) return YF_ENG_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
#ifdef YF_BARPROF
  return launch(e, e->var, d_in, d_out, d_dump, n, (hipStream_t)stream);      // profile build: the production variant fills d_dump with barrier waits
#else
  if (d_dump && !e->var_dump) { e->err = "no debug (dump) build of the configured kernel shape"; return YF_ENG_ERR_VARIANT; }
  return launch(e, d_dump ? e->var_dump : shape_for(e, n), d_in, d_out, d_dump, n, (hipStream_t)stream);
#endif
}

int yf_engine_run_decode_device(yf_engine* e, const void* d_in, void* d_out, long n, int mode, float w_scale, float h_scale,
                                void* d_dets, void* d_counts, int cap, void* stream) {
  if (!e || !d_in || !d_out || !d_dets || !d_counts || n < 0 || cap <= 0 || (mode != YF_DECODE_PY && mode != YF_DECODE_FW && mode != YF_DECODE_FW_HOST)) return YF_ENG_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  const DecodeArgs dec = {d_dets, d_counts, cap, mode, w_scale, h_scale};
  return launch(e, shape_for(e, n), d_in, d_out, nullptr, n, (hipStream_t)stream, -1, &dec);
}

// camera frames (112x112 big-endian RGB565) -> heads (+ boxes when d_dets is given): the frame preparation runs inside the
// fused kernel's input staging.  Only the shipped shape has this build.
int yf_engine_run_camera_device(yf_engine* e, const void* d_rgb565, void* d_out, long n, int mode, float w_scale, float h_scale,
                                void* d_dets, void* d_counts, int cap, void* stream) {
  if (!e || !d_rgb565 || !d_out || n < 0) return YF_ENG_ERR_ARG;
  if (d_dets && (!d_counts || cap <= 0 || (mode != YF_DECODE_PY && mode != YF_DECODE_FW && mode != YF_DECODE_FW_HOST))) return YF_ENG_ERR_ARG;
  if (((uintptr_t)d_rgb565 & 15) != 0) { e->err = "camera frames must be 16-byte aligned"; return YF_ENG_ERR_ARG; }
  HIPCHK(e, hipSetDevice(e->device));
  const Variant* v = find_variant(e->var->f, e->var->nw, false, true, false, e->signless);
  if (!v) { e->err = "no camera-input build of the configured kernel shape"; return YF_ENG_ERR_VARIANT; }
  const DecodeArgs dec = {d_dets, d_counts, cap, mode, w_scale, h_scale};
  return launch(e, v, d_rgb565, d_out, nullptr, n, (hipStream_t)stream, -1, d_dets ? &dec : nullptr);
}

// ai_network_run on the caller's host arrays (the reference's own call path: yoloface.c:216-240 hands in_data / out_data).
//   n <= ZERO_COPY_N : the frames are copied into pinned, device-mapped memory and the kernel reads them -- and writes the heads --
//                      over PCIe itself: no copy submissions, one launch, one synchronisation.
//   n >= PIPE_MIN_N  : chunks of PIPE_CHUNK frames alternate between two streams, so the upload of chunk k+1 runs behind the
//                      kernel of chunk k, and a worker thread downloads the heads of finished chunks meanwhile (the last chunk is
//                      cut short: what cannot overlap anything is its kernel and its download).
//   otherwise        : upload, one launch, download.
// Device staging grows on the first call that needs more (never shrinks): no allocation on the call path after that.
int yf_engine_run_host(yf_engine* e, const void* h_in, void* h_out, long n) {
  if (!e || !h_in || n < 0) return YF_ENG_ERR_ARG;        /* h_out may be NULL (ai_network_forward) */
  if (n == 0) return YF_ENG_OK;
  HIPCHK(e, hipSetDevice(e->device));
  if (n <= ZERO_COPY_N) {
    memcpy(e->h_small_in, h_in, (size_t)n * 9408);
    void *d_i = nullptr, *d_o = nullptr;
    HIPCHK(e, hipHostGetDevicePointer(&d_i, e->h_small_in, 0));
    HIPCHK(e, hipHostGetDevicePointer(&d_o, e->h_small_out, 0));
    const int rc = launch(e, shape_for(e, n), d_i, d_o, nullptr, n, e->own_stream);
    if (rc) return rc;
    HIPCHK(e, hipStreamSynchronize(e->own_stream));
    if (h_out) memcpy(h_out, e->h_small_out, (size_t)n * 882);
    return YF_ENG_OK;
  }
  if (n > e->stage_cap) {
    if (e->d_in) (void)hipFree(e->d_in);
    if (e->d_out) (void)hipFree(e->d_out);
    e->d_in = e->d_out = nullptr; e->stage_cap = 0;
    HIPCHK(e, hipMalloc(&e->d_in, (size_t)n * 9408));
    HIPCHK(e, hipMalloc(&e->d_out, (size_t)n * 882));
    e->stage_cap = n;
  }
  if (n < PIPE_MIN_N) {
    HIPCHK(e, hipMemcpyAsync(e->d_in, h_in, (size_t)n * 9408, hipMemcpyHostToDevice, e->own_stream));
    const int rc = launch(e, shape_for(e, n), e->d_in, e->d_out, nullptr, n, e->own_stream);
    if (rc) return rc;
    if (h_out) HIPCHK(e, hipMemcpyAsync(h_out, e->d_out, (size_t)n * 882, hipMemcpyDeviceToHost, e->own_stream));
    HIPCHK(e, hipStreamSynchronize(e->own_stream));
    return YF_ENG_OK;
  }
  if (h_out && !e->dl) {
    e->dl = new Downloader();
    const hipError_t rc = e->dl->start(e->device);
    if (rc != hipSuccess) { delete e->dl; e->dl = nullptr; e->err = std::string("download stream: ") + hipGetErrorString(rc); return YF_ENG_ERR_HIP; }
  }
  static const long PIPE_CHUNK = [] { const char* v = getenv("YF_PIPE_CHUNK"); const long c = v ? atol(v) : 0; return c >= 256 ? c : PIPE_CHUNK_DEFAULT; }();   // tuning knobs
  static const long PIPE_LAST = [] { const char* v = getenv("YF_PIPE_LAST"); const long c = v ? atol(v) : 0; return c >= 64 ? c : PIPE_LAST_DEFAULT; }();
  int rc = YF_ENG_OK;
  hipError_t hrc = hipSuccess;
  long done = 0;
  for (int k = 0; done < n && rc == YF_ENG_OK && hrc == hipSuccess; ++k) {
    long c = n - done < PIPE_CHUNK ? n - done : PIPE_CHUNK;
    if (c == n - done && c > 2 * PIPE_LAST) c -= PIPE_LAST;              // the batch ends with a short chunk
    hipStream_t st = e->pipe_stream[k & 1];
    const char* src = (const char*)h_in + done * 9408;
    char* d_i = (char*)e->d_in + done * 9408;
    char* d_o = (char*)e->d_out + done * 882;
    // the previous use of this stream's event (chunk k-2) has been consumed by the download stream's wait before its copy ran; with
    // two events in flight and in-order downloads, re-recording here cannot overtake a pending wait only if that download is done:
    if (h_out && k >= 2) { const hipError_t r = e->dl->drain_to(2); if (r != hipSuccess) { hrc = r; break; } }
    hrc = hipMemcpyAsync(d_i, src, (size_t)c * 9408, hipMemcpyHostToDevice, st);
    if (hrc != hipSuccess) break;
    rc = launch(e, e->var, d_i, d_o, nullptr, c, st);
    if (rc != YF_ENG_OK) break;
    if (h_out) {
      hrc = hipEventRecord(e->pipe_ev[k & 1], st);
      if (hrc != hipSuccess) break;
      e->dl->submit(Downloader::Job{e->pipe_ev[k & 1], d_o, (char*)h_out + done * 882, (size_t)c * 882});
    }
    done += c;
  }
  // everything issued is waited for, whatever happened above: the caller's arrays must not be touched after this returns
  const hipError_t d_rc = e->dl && h_out ? e->dl->drain() : hipSuccess;
  const hipError_t s0 = hipStreamSynchronize(e->pipe_stream[0]), s1 = hipStreamSynchronize(e->pipe_stream[1]);
  if (rc != YF_ENG_OK) return rc;
  for (hipError_t r : {hrc, d_rc, s0, s1}) if (r != hipSuccess) { e->err = std::string("pipelined host run: ") + hipGetErrorString(r); return YF_ENG_ERR_HIP; }
  return YF_ENG_OK;
}

// Debug form of yf_engine_run_host for the per-node observer (platform_abi.c): the dump build of the configured shape on n host
// frames; heads and the per-stage dump records (yf_engine_dump_bytes() per frame) come back to host memory.  Allocates per call.
int yf_engine_run_host_dump(yf_engine* e, const void* h_in, void* h_out, void* h_dump, long n) {
  if (!e || !h_in || !h_out || !h_dump || n <= 0) return YF_ENG_ERR_ARG;
  if (!e->var_dump) { e->err = "no debug (dump) build of the configured kernel shape"; return YF_ENG_ERR_VARIANT; }
  HIPCHK(e, hipSetDevice(e->device));
  const size_t ds = (size_t)yf::DumpOffsets::TOTAL;
  char *d_i = nullptr, *d_o = nullptr, *d_d = nullptr;
  int rc = YF_ENG_OK;
  hipError_t h = hipMalloc((void**)&d_i, (size_t)n * 9408);
  if (h == hipSuccess) h = hipMalloc((void**)&d_o, (size_t)n * 882);
  if (h == hipSuccess) h = hipMalloc((void**)&d_d, (size_t)n * ds);
  if (h == hipSuccess) h = hipMemcpyAsync(d_i, h_in, (size_t)n * 9408, hipMemcpyHostToDevice, e->own_stream);
  if (h == hipSuccess) rc = launch(e, e->var_dump, d_i, d_o, d_d, n, e->own_stream);
  if (h == hipSuccess && rc == YF_ENG_OK) h = hipMemcpyAsync(h_out, d_o, (size_t)n * 882, hipMemcpyDeviceToHost, e->own_stream);
  if (h == hipSuccess && rc == YF_ENG_OK) h = hipMemcpyAsync(h_dump, d_d, (size_t)n * ds, hipMemcpyDeviceToHost, e->own_stream);
  const hipError_t hs = hipStreamSynchronize(e->own_stream);
  if (d_i) (void)hipFree(d_i);
  if (d_o) (void)hipFree(d_o);
  if (d_d) (void)hipFree(d_d);
  if (rc != YF_ENG_OK) return rc;
  if (h != hipSuccess || hs != hipSuccess) { e->err = std::string("observed run: ") + hipGetErrorString(h != hipSuccess ? h : hs); return YF_ENG_ERR_HIP; }
  return YF_ENG_OK;
}

// byte offset of a dumped tensor inside a frame's dump record, by tflite op number (the op that produces it); -1 = not dumped
long yf_engine_dump_offset(int tflite_op) {
  typedef yf::DumpOffsets D;
  switch (tflite_op) {
    case 2: return D::T1; case 4: return D::T2; case 5: return D::T3; case 7: return D::T4; case 21: return D::Q21; case 11: return D::T6;
    case 12: return D::T7; case 14: return D::T8; case 16: return D::T9; case 18: return D::T11; case 22: return D::T14; case 24: return D::T15;
    case 45: return D::Q45; case 28: return D::T17; case 29: return D::T18; case 31: return D::T19; case 33: return D::T20; case 35: return D::T22;
    case 37: return D::T23; case 39: return D::T24; case 41: return D::T26; case 46: return D::T30; case 48: return D::T31; case 50: return D::T32;
    case 52: return D::T33; case 8: return D::P8; case 17: return D::C17; case 25: return D::P25; case 34: return D::C34; case 40: return D::C40;
    case 43: return D::L43;
    default: return -1;
  }
}

int yf_engine_time_device(yf_engine* e, const void* d_in, void* d_out, long n, int iters, void* stream, float* ms_per_launch) {
  if (!e || !d_in || !d_out || n <= 0 || iters <= 0 || !ms_per_launch) return YF_ENG_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(e, hipEventRecord(e->ev0, s));
  for (int i = 0; i < iters; ++i) { const int rc = launch(e, e->var, d_in, d_out, nullptr, n, s); if (rc) return rc; }
  HIPCHK(e, hipEventRecord(e->ev1, s));
  HIPCHK(e, hipEventSynchronize(e->ev1));
  float ms = 0.f;
  HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *ms_per_launch = ms / iters;
  return YF_ENG_OK;
}

// ---------------------------------------------------------------------------------------------- 160x160 variant
int yf_engine_run_device_160(yf_engine* e, const void* d_in, void* d_out, long n, void* stream) {
  if (!e || !d_in || !d_out || n < 0) return YF_ENG_ERR_ARG;
  if (n == 0) return YF_ENG_OK;
  if (((uintptr_t)d_in & 3) != 0 || ((uintptr_t)d_out & 1) != 0) { e->err = "input must be 4-byte, output 2-byte aligned"; return YF_ENG_ERR_ARG; }
  HIPCHK(e, hipSetDevice(e->device));
  const long cap = n < e->chunk160 ? n : e->chunk160;         // frames per chunk of the HBM arena (289 KB per frame)
#ifdef YF_LAB
  const size_t per_frame = e->layerwise160 ? (size_t)yf160::FRAME_BYTES : (size_t)yf160::band::ARENA_BYTES;
#else
  const size_t per_frame = (size_t)yf160::band::ARENA_BYTES;
#endif
  yf_stream_scratch::Lease lease;                            // owned by the launch stream: overlapping launches never share it; marked on every way out
  HIPCHK(e, e->arena160.get((hipStream_t)stream, (size_t)cap * per_frame, &lease));
  char* arena = lease.ptr;
  for (long done = 0; done < n; done += cap) {
    const long m = (n - done) < cap ? (n - done) : cap;
    int rc;
#ifdef YF_LAB
    if (e->layerwise160) {
      yf160::GenParams prm;
      prm.in = (const int8_t*)d_in + done * yf160::IN_FRAME_BYTES;
      prm.out = (int8_t*)d_out + done * yf160::OUT_FRAME_BYTES;
      prm.n = m; prm.tab = e->d_tab; prm.arena = arena;
      rc = launch160_from<0>(e, prm, (unsigned)m, (hipStream_t)stream);
    } else
#endif
    {
      yf160::band::Params prm;
      prm.in = (const int8_t*)d_in + done * yf160::IN_FRAME_BYTES;
      prm.out = (int8_t*)d_out + done * yf160::OUT_FRAME_BYTES;
      prm.n = m; prm.tab = e->d_tab; prm.arena = arena;
      rc = launch160_banded(e, prm, (hipStream_t)stream);
    }
    if (rc) return rc;
  }
  HIPCHK(e, lease.mark());
  return YF_ENG_OK;
}

int yf_engine_release_stream(yf_engine* e, void* stream) {
  if (!e) return YF_ENG_ERR_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  HIPCHK(e, e->park.release_stream((hipStream_t)stream));
  HIPCHK(e, e->arena160.release_stream((hipStream_t)stream));
  return YF_ENG_OK;
}

size_t yf_engine_scratch_bytes(yf_engine* e) { return e ? e->park.bytes_held() + e->arena160.bytes_held() : 0; }
// counters of the two scratch maps, summed: out[0..5] = events recorded, events skipped, event waits, device synchronisations, waits for another
// thread's mark (the last three on the all-busy path of yf_stream_scratch::get), regions in existence
void yf_engine_scratch_stats(yf_engine* e, unsigned long long out[6]) {
  for (int i = 0; i < 6; ++i) out[i] = 0;
  if (!e) return;
  for (yf_stream_scratch* m : {&e->park, &e->arena160}) {
    const yf_stream_scratch::Stats s = m->stats();
    out[0] += s.events_recorded; out[1] += s.events_skipped; out[2] += s.event_waits; out[3] += s.device_syncs; out[4] += s.acquire_waits; out[5] += s.regions;
  }
}

int yf_engine_time_stages(yf_engine* e, const void* d_in, void* d_out, long n, int iters, int stop_stage, void* stream, float* ms_per_launch) {
  if (!e || !d_in || !d_out || n <= 0 || iters <= 0 || !ms_per_launch) return YF_ENG_ERR_ARG;
  if (!e->var_dump) { e->err = "no debug (stage-timing) build of the configured kernel shape"; return YF_ENG_ERR_VARIANT; }
  HIPCHK(e, hipSetDevice(e->device));
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(e, hipEventRecord(e->ev0, s));
  for (int i = 0; i < iters; ++i) { const int rc = launch(e, e->var_dump, d_in, d_out, nullptr, n, s, stop_stage); if (rc) return rc; }
  HIPCHK(e, hipEventRecord(e->ev1, s));
  HIPCHK(e, hipEventSynchronize(e->ev1));
  float ms = 0.f;
  HIPCHK(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
  *ms_per_launch = ms / iters;
  return YF_ENG_OK;
}

int yf_engine_decode_device(yf_engine* e, const void* d_heads, long n, int mode, float w_scale, float h_scale,
                            void* d_dets, void* d_counts, int cap, void* stream) {
  if (!e || !d_heads || !d_dets || !d_counts || n < 0 || cap <= 0 || (mode != YF_DECODE_PY && mode != YF_DECODE_FW && mode != YF_DECODE_FW_HOST)) return YF_ENG_ERR_ARG;
  if (n == 0) return YF_ENG_OK;
  HIPCHK(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const int8_t*)d_heads, n, mode, w_scale, h_scale, (yf_det*)d_dets, (int*)d_counts, cap);
  HIPCHK(e, hipGetLastError());
  return YF_ENG_OK;
}

int yf_engine_pack_detections_device(yf_engine* e, const void* d_dets, const void* d_counts, const void* d_heads, void* d_wire, long n, int cap, void* stream) {
  if (!e || !d_dets || !d_counts || !d_heads || !d_wire || n < 0 || cap <= 0 || ((uintptr_t)d_wire & 3) != 0 || ((uintptr_t)d_dets & 3) != 0) return YF_ENG_ERR_ARG;
  if (n == 0) return YF_ENG_OK;
  HIPCHK(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(pack_dets_kernel, dim3((unsigned)((n * cap + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const yf_det*)d_dets, (const int*)d_counts, (const int8_t*)d_heads, (uint32_t*)d_wire, n, cap);
  HIPCHK(e, hipGetLastError());
  return YF_ENG_OK;
}

int yf_engine_unpack_detections_device(yf_engine* e, const void* d_wire, const void* d_counts, void* d_heads, long n, int cap, void* stream) {
  if (!e || !d_wire || !d_counts || !d_heads || n < 0 || cap <= 0 || ((uintptr_t)d_wire & 3) != 0) return YF_ENG_ERR_ARG;
  if (n == 0) return YF_ENG_OK;
  HIPCHK(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(unpack_dets_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)d_wire, (const int*)d_counts, (int8_t*)d_heads, n, cap);
  HIPCHK(e, hipGetLastError());
  return YF_ENG_OK;
}

int yf_engine_prepare_rgb565_device(yf_engine* e, const void* d_rgb565, void* d_out, long n, void* stream) {
  if (!e || !d_rgb565 || !d_out || n < 0) return YF_ENG_ERR_ARG;
  if (n == 0) return YF_ENG_OK;
  HIPCHK(e, hipSetDevice(e->device));
  hipLaunchKernelGGL(prepare_rgb565_kernel, dim3((unsigned)((n * 3136 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)d_rgb565, (int8_t*)d_out, n);
  HIPCHK(e, hipGetLastError());
  return YF_ENG_OK;
}

}  // extern "C"
