// The 160x160 path: three banded kernels (band_k1, band_k23, band_k4), each a group of fused stages over row bands staged through LDS.
// Part of yf_kernels.hip.h (included from inside namespace YF_NS; not a stand-alone header).


// ------------------------------------------------------------------------------------------------ banded form
// Second form for sizes that do not fit in LDS (160x160): FOUR kernels, each fusing a group of stages over a BAND of rows of
// one frame.  A workgroup copies the band of its input tensor (with the halo rows the group needs) from the per-frame HBM
// arena into LDS with coalesced loads, runs the SAME stage functions as the 56x56 kernel on band-local buffers, and writes
// the band of its output tensor back.  Only five tensors cross HBM (T4, the pooled half of concat_22, T7, T8, T15):
// ~0.78 MB per frame instead of 1.7 MB, all of it in full-row transfers.
//   K1  input rows -> conv2d_1 -> conv2d_3 (dw) -> conv2d_5 -> conv2d_6 -> T4            band = 8 rows of the 80x80 grid
//   K2  T4 -> pool_8 (+QUANTIZE) -> P8 ; conv2d_10 (dw) -> conv2d_12 -> T7 -> conv2d_13 -> T8     band = 4 rows of 40x40
//   K3  T8 -> conv2d_15 (dw) -> conv2d_17 + add(T7) -> conv2d_19 | P8 -> conv2d_23 -> T15         band = 8 rows of 40x40
//   K4  T15 -> pool_25, conv2d_27 (dw) ... conv2d_53 -> head                                       whole 20x20 grid
// Halo rules: a band's input copy spans whole halo'd rows of the global tensor, so image borders bring their zero-point
// halo with them and interior band edges bring real neighbour rows; the producer fills halo columns (and the first / last
// band the top / bottom halo row) before the copy-out.
namespace band {
#define YF_BAND_PRIO(P) __builtin_amdgcn_s_setprio(P)      // priority ladder over a band job's stages (see the 56x56 kernel)
constexpr int LB = LUT_BYTES;                                   // LUTs at LDS offset 0 (absolute addressing)
// per-frame HBM arena of the banded form (bytes); rows are padded to multiples of 16 bytes so bands move as 16-byte vectors
constexpr int T4_RS = G1 + 4, T8_RS = G2 + 4, T15_RS = G2 + 2;   // pixels per halo'd row
constexpr int T4_ROW = T4_RS * 20, T8_ROW = T8_RS * 36, T15_ROW = T15_RS * 24;
static_assert(T4_ROW % 16 == 0 && T8_ROW % 16 == 0 && T15_ROW % 16 == 0 && (G2 * 20) % 16 == 0 && (G2 * 8) % 16 == 0, "16-byte rows");
constexpr int A_T4 = 0;                                          // [G1 + 1 halo'd rows][T4_RS][20]    top/left halo
constexpr int A_P8 = (A_T4 + (G1 + 1) * T4_ROW + 63) & ~63;      // [G2][G2][20]                       pool_8 + QUANTIZE#21
constexpr int A_T7 = (A_P8 + G2 * G2 * 20 + 63) & ~63;           // [G2][G2][8]
constexpr int A_T8 = (A_T7 + G2 * G2 * 8 + 63) & ~63;            // [G2 + 2][T8_RS][36]                halo ring
constexpr int A_T15 = (A_T8 + (G2 + 2) * T8_ROW + 63) & ~63;     // [G2 + 1][T15_RS][24]               top/left halo
constexpr int ARENA_BYTES = (A_T15 + (G2 + 1) * T15_ROW + 63) & ~63;

struct Params { const int8_t* in; int8_t* out; long n; const uint8_t* tab; char* arena; };

// Workgroup barrier that orders LDS only.  __syncthreads() also waits for every outstanding global access (vmcnt(0)):
// that would drain the next band's prefetch loads and this band's copy-out stores at every stage boundary.  Nothing a
// band kernel writes to HBM is read back by the same kernel, so LDS ordering is all the stages need.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Virtual job index v (= blockIdx.x + k * gridDim.x) -> (frame, band).  Workgroup b runs on XCD b mod 8 and every XCD has an L2 of its own.  The bands of ONE frame
// read overlapping rows (band_k1: 5 of a band's 37 input rows, band_k23: 6 of its 22 T4 rows), so they go to workgroups of ONE XCD -- v mod 8 picks the frame's
// class, v / 8 walks (frame, band) inside it -- and the overlap is read from HBM once instead of once per band.  Needs a grid and a batch that are multiples of 8
// (uniform over the launch); other launches keep the plain order.
#ifndef YF_BAND_XCD
#define YF_BAND_XCD 1
#endif
template <int BANDS>
__device__ __forceinline__ void split_job(long v, bool xcd, long& fr, int& band) {
  if (xcd) { const long l = v >> 3; const long q = l / BANDS; fr = (v & 7) + 8 * q; band = (int)(l - q * BANDS); }
  else { fr = v / BANDS; band = (int)(v - fr * BANDS); }
}

// N bytes LDS -> HBM as 16-byte vectors (both 16-byte aligned, N a multiple of 16)
template <int NT>
__device__ __forceinline__ void store_rows(char* dst, const char* src, int bytes, int tid) {
  for (int i = tid; i < bytes / 16; i += NT) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
}
template <int NT>
__device__ __forceinline__ void fill_dwords(char* dst, uint32_t v, int bytes, int tid) {
  for (int i = tid; i < bytes / 4; i += NT) reinterpret_cast<uint32_t*>(dst)[i] = v;
}
// column `col` (pixel units) of `rows` rows of a buffer with ROW bytes per row and S bytes per pixel <- v
template <int NT, int ROW, int S>
__device__ __forceinline__ void fill_column(char* base, int col, int rows, uint32_t v, int tid) {
  constexpr int DW = S / 4;
  for (int i = tid; i < rows * DW; i += NT) {
    const int r = i / DW, d = i - r * DW;
    *reinterpret_cast<uint32_t*>(base + r * ROW + col * S + 4 * d) = v;
  }
}
__device__ __forceinline__ uint32_t splat(int zp) { return (uint32_t)(zp & 255) * 0x01010101u; }

template <int NT>
__device__ __forceinline__ void load_luts(uint8_t* luts, const uint8_t* __restrict__ tab, int tid) {
  for (int i = tid; i < LUT_BYTES / 16; i += NT)
    reinterpret_cast<uint4*>(luts)[i] = reinterpret_cast<const uint4*>(tab + PLAN.lut_off)[i];
}

// A band's input, prefetched: CNT 16-byte vectors of a contiguous HBM range, vector i owned by thread i % NT.  fetch() issues
// the loads for the NEXT job right after the current job's data has been committed to LDS; nothing waits for them until
// the commit at the top of the next iteration, so the HBM latency hides behind the whole band's compute.
template <int NT, int CNT>
struct Prefetch {
  static constexpr int PER = (CNT + NT - 1) / NT;
  v4u v[PER];                                   // native vectors: HIP's uint4 class keeps the array in scratch
};
template <int NT, int CNT>
__device__ __forceinline__ void pf_fetch(Prefetch<NT, CNT>& p, const char* src, int n16, int tid) {      // n16 <= CNT vectors
#pragma unroll
  for (int k = 0; k < Prefetch<NT, CNT>::PER; ++k) p.v[k] = reinterpret_cast<const v4u*>(src)[min(tid + k * NT, n16 - 1)];
}
template <int NT, int CNT>
__device__ __forceinline__ void pf_commit(const Prefetch<NT, CNT>& p, char* dst, int n16, int tid) {
#pragma unroll
  for (int k = 0; k < Prefetch<NT, CNT>::PER; ++k) { const int i = tid + k * NT; if (i < n16) reinterpret_cast<v4u*>(dst)[i] = p.v[k]; }
}
// the same into rows of PITCH bytes in LDS (ROWVEC 16-byte vectors per row in HBM): a pitch that is not a multiple of 16 bytes (a row skew
// against bank conflicts) takes dword stores
template <int ROWVEC, int PITCH, int NT, int CNT>
__device__ __forceinline__ void pf_commit_rows(const Prefetch<NT, CNT>& p, char* dst, int n16, int tid) {
#pragma unroll
  for (int k = 0; k < Prefetch<NT, CNT>::PER; ++k) {
    const int i = tid + k * NT;
    if (i < n16) {
      const int r = (int)((uint32_t)i / (uint32_t)ROWVEC), c = i - r * ROWVEC;
      uint32_t* d = reinterpret_cast<uint32_t*>(dst + r * PITCH + 16 * c);
      if constexpr (PITCH % 16 == 0) *reinterpret_cast<v4u*>(d) = p.v[k];
      else { d[0] = p.v[k][0]; d[1] = p.v[k][1]; d[2] = p.v[k][2]; d[3] = p.v[k][3]; }
    }
  }
}

// ---- lean stage forms in the band kernels (round 3) -----------------------------------------------------------------------
// band_k1 and band_k23 use the 56x56 kernel's lean stage forms (namespace v2) with their constants RESIDENT: a workgroup runs many band
// jobs with the same few stages, so the vector-side blocks of those stages (1.4 KB for band_k1, 8 KB for band_k23) are loaded once per
// workgroup -- into bytes of the LUT / residual-add-table area [0, LB) that the kernel's own stages never index -- instead of fetched from
// global memory behind every stage boundary of every job (1.5-2.5 k cycles each; the band jobs ran at half the 56x56 kernel's per-pixel rate).
template <int CS0, int CS1, int BASE_, int ZERO_, int JT_, int JT_BYTES_>
struct BandLay {
  static constexpr int ZERO = ZERO_, JT = JT_, JT_BYTES = JT_BYTES_, FIRST = CS0, LAST = CS1, BASE = BASE_;
  static constexpr int slot(int cs) { int off = BASE_; for (int i = CS0; i < cs; ++i) off += PLAN.vb_bytes[i]; return off; }
  static constexpr int END = slot(CS1 + 1);
};
// after load_luts: zeros, then the blocks of const-stages FIRST .. LAST at their slots (16-byte vectors, every thread)
template <class LAY, int NT>
__device__ __forceinline__ void load_resident(char* smem, const uint8_t* __restrict__ tab, int tid) {
  for (int i = tid; i < v2::ZERO_B / 16; i += NT) reinterpret_cast<uint4*>(smem + LAY::ZERO)[i] = uint4{0, 0, 0, 0};
#pragma unroll
  for (int cs = LAY::FIRST; cs <= LAY::LAST; ++cs)
    for (int i = tid; i < PLAN.vb_bytes[cs] / 16; i += NT)
      reinterpret_cast<uint4*>(smem + LAY::slot(cs))[i] = reinterpret_cast<const uint4*>(tab + PLAN.vb_off[cs])[i];
}
// band_k1 indexes LUTs 0-2 only ([0, 768)): blocks of conv2d_1 / 3 / 5 / 6 behind them, job table and zeros at the end of the area
typedef BandLay<0, 3, 768, LB - v2::ZERO_B, LB - v2::ZERO_B - 768, 768> LayK1;
// band_k23 indexes LUTs 3-8 ([768, 2304)) and no add table of the area (conv2d_17's block brings its own): zeros and job tables on
// LUTs 0-2, the seven blocks of conv2d_10 .. conv2d_23 from 2304 on
typedef BandLay<4, 10, 2304, 0, v2::ZERO_B, 768 - v2::ZERO_B> LayK23;
static_assert(LayK1::END <= LayK1::JT && LayK1::ZERO + v2::ZERO_B <= LB && LayK23::END <= LB && v2::ZERO_B + LayK23::JT_BYTES <= 768, "resident constants fit the unused LUT / add-table bytes");

#ifndef YF_BAND_TPJ
#define YF_BAND_TPJ 1          /* 1: five passes per job for conv2d_6 in band_k1 (10 jobs per band instead of 20: -2.5 % of that kernel; the same in band_k23 spills or loses) */
#endif
// ---- K1 ----------------------------------------------------------------------------------------------------------------
#ifndef YF_K1_BH
#define YF_K1_BH 16
#endif
#ifndef YF_K1_OCC
#define YF_K1_OCC 4
#endif
constexpr int K1_BH = YF_K1_BH, K1_BANDS = G1 / K1_BH, K1_NIN = 2 * K1_BH + 5, K1_NT1 = K1_BH + 2;
static_assert(G1 % K1_BH == 0, "band height must divide the grid");
constexpr int cmax(int a, int b) { return a > b ? a : b; }
#ifndef YF_BAND_SKEW
#define YF_BAND_SKEW 4              /* bytes of row skew in the band kernels' depthwise inputs with 8- / 40-byte pixels (see Buf::SK) */
#endif
constexpr int K1_T1_ROW = (G1 + 2) * 8 + YF_BAND_SKEW;
constexpr int K1_IN_BYTES = K1_NIN * (G0 + 4) * 4, K1_T1_BYTES = (K1_NT1 * K1_T1_ROW + 15) & ~15;
constexpr int K1_R0 = cmax(K1_IN_BYTES + K1_T1_BYTES, K1_BH * T4_ROW);                       // IN + T1, later T4
typedef Buf<LB,                                  G0, K1_NIN - 1, 4, G0 + 4, 1, 4> L1_IN;    // RGBX rows: local row l = global halo'd row 2(a-1)+l
typedef Buf<L1_IN::OFF + K1_IN_BYTES,            G1, K1_NT1,     8, G1 + 2, 0, 1, BUF_FS, YF_BAND_SKEW> L1_T1;    // local row t = T1 row a-1+t, halo columns 0 and G1+1
typedef Buf<LB + K1_R0,                          G1, K1_BH,      8, G1,     0, 0> L1_T2;
typedef Buf<L1_T2::OFF + K1_BH * G1 * 8,         G1, K1_BH,      4, G1,     0, 0> L1_T3;
typedef Buf<LB,                                  G1, K1_BH,     20, T4_RS,  0, 1> L1_T4;    // left halo column; aliases IN and T1 (dead after conv2d_3)
constexpr int K1_LDS = L1_T3::OFF + K1_BH * G1 * 4;

template <int NW>
__global__ void __launch_bounds__(NW * 64, YF_K1_OCC) band_k1(const Params prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64, F = 1;
  constexpr int RSW = G0 + 4, WQ = G0 / 4, ITEMS = K1_NIN * WQ, PER = (ITEMS + NT - 1) / NT;     // item = 4 pixels = 12 input bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t* __restrict__ tab = prm.tab;
  int vz = 0;
  asm volatile("" : "+v"(vz));
  load_luts<NT>(reinterpret_cast<uint8_t*>(smem), tab, tid);
  __syncthreads();                                                // the LUT area is written; its unused bytes now take the resident pieces
  load_resident<LayK1, NT>(smem, tab, tid);
  v2::fill_jobtab<1, 1, L1_T1, L1_T2, 0, LayK1>(smem, tid);
  const AddK no_add = {};
  const uint32_t z_in = splat((int)uniform_u32(tab + offsetof(yf_table_index, in_zp)));
  const uint32_t z_t1 = splat(load_halo_zp(tab, YF_W_DW3)), z_t4 = splat(load_halo_zp(tab, YF_W_DW10));
  char* frames = smem;                                            // band-local buffers live at their LDS offsets
  const long jobs = prm.n * K1_BANDS;
  const bool xcd = YF_BAND_XCD && gridDim.x % 8 == 0 && prm.n % 8 == 0;
  uint32_t pre[PER][3];
  // input rows of a band: local row l <-> input row 2(a-1)+l-1, out of range = zero point
  auto fetch = [&](long job) {
    long fr; int bnd; split_job<K1_BANDS>(job, xcd, fr, bnd);
    const int a = bnd * K1_BH;
    const int8_t* in = prm.in + fr * (long)IN_FRAME_BYTES;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + k * NT;
      const int l = i / WQ, xq = i - l * WQ;
      const int r = 2 * (a - 1) + l - 1;
      pre[k][0] = pre[k][1] = pre[k][2] = z_in;
      if (i < ITEMS && r >= 0 && r < G0) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(in + r * (G0 * 3) + xq * 12);
        pre[k][0] = src[0]; pre[k][1] = src[1]; pre[k][2] = src[2];
      }
    }
  };
  long job = blockIdx.x;
  if (job < jobs) fetch(job);
  for (; job < jobs; job += gridDim.x) {
    long fr; int bnd; split_job<K1_BANDS>(job, xcd, fr, bnd);
    const int a = bnd * K1_BH;                                     // first T4 row of the band
    char* arena = prm.arena + fr * (long)ARENA_BYTES;
    YF_BAND_PRIO(3);
    lds_barrier();                                                // previous band's buffers are dead
#pragma unroll
    for (int k = 0; k < PER; ++k) {                               // RGB -> RGBX dwords behind the halo column
      const int i = tid + k * NT;
      if (i < ITEMS) {
        const int l = i / WQ, xq = i - l * WQ;
        uint4 px;
        px.x = pre[k][0]; px.y = funnel(pre[k][1], pre[k][0], 24); px.z = funnel(pre[k][2], pre[k][1], 16); px.w = pre[k][2] >> 8;
        *reinterpret_cast<uint4*>(frames + L1_IN::OFF + l * RSW * 4 + 16 + 16 * xq) = px;
      }
    }
    if (tid < K1_NIN) *reinterpret_cast<uint32_t*>(frames + L1_IN::OFF + tid * RSW * 4 + 12) = z_in;        // halo column (dword 3)
    lds_barrier();
    if (job + gridDim.x < jobs) fetch(job + gridDim.x);
    v2::conv1_2_stage<F, NW, 0, L1_IN, L1_T1, LayK1>(frames, tab, wave, lane);
    fill_column<NT, K1_T1_ROW, 8>(frames + L1_T1::OFF, 0, K1_NT1, z_t1, tid);
    fill_column<NT, K1_T1_ROW, 8>(frames + L1_T1::OFF, G1 + 1, K1_NT1, z_t1, tid);
    lds_barrier();
    if (a == 0) fill_dwords<NT>(frames + L1_T1::OFF, z_t1, (G1 + 2) * 8, tid);                               // T1 row -1 = halo
    if (a + K1_BH == G1) fill_dwords<NT>(frames + L1_T1::OFF + (K1_NT1 - 1) * K1_T1_ROW, z_t1, (G1 + 2) * 8, tid);   // T1 row G1
    if (a == 0 || a + K1_BH == G1) lds_barrier();
    YF_BAND_PRIO(2);
    v2::dw2_stage<F, NW, 1, L1_T1, L1_T2, 8, YF_L_LEAKY4, 1, 0, LayK1>(frames, tab, wave, lane);
    lds_barrier();
    YF_BAND_PRIO(1);
    v2::dense2_stage<F, NW, 1, 1, 8, L1_T2, L1_T3, 0, 4, EPI_RAW, 0, L1_T3, 2, -1, 0, -1, LayK1>(frames, nullptr, tab, no_add, wave, lane);
    lds_barrier();
    v2::dense2_stage<F, NW, YF_BAND_TPJ ? 5 : 3, 1, 4, L1_T3, L1_T4, 0, 18, EPI_LUT, YF_L_LEAKY7, L1_T4, 3, -1, 0, -1, LayK1>(frames, nullptr, tab, no_add, wave, lane);
    fill_column<NT, T4_ROW, 20>(frames + L1_T4::OFF, 0, K1_BH, z_t4, tid);
    lds_barrier();
    YF_BAND_PRIO(0);
#if !(defined(YF_LAB) && defined(YF_WHATIF_NO_T4_HBM))   // what-if (WRONG results): T4 never written ...
    store_rows<NT>(arena + A_T4 + (a + 1) * T4_ROW, frames + L1_T4::OFF, K1_BH * T4_ROW, tid);              // halo'd rows a+1 ..
#endif
    if (a == 0) fill_dwords<NT>(arena + A_T4, z_t4, T4_ROW, tid);                                           // top halo row
  }
}

// ---- K23: K2 and K3 fused (round 3) ------------------------------------------------------------------------------------
// One band job = 8 rows of the 40x40 grid through pool_8 .. conv2d_23: T4 rows in, T15 rows out; the pooled half of concat_22, T7 and T8
// never leave the chip (three tensors cross HBM instead of five: 0.73 -> 0.49 MB per frame).  conv2d_15's 3x3 window needs T8 rows
// p0-1 .. p0+8, so conv2d_10 / 12 / 13 run on TEN rows per band (the T4 rows they need, 2p0-3 .. 2p0+17, are inside the 22 rows the
// pool already loads); rows outside the image are computed from whatever the LDS holds and then overwritten with the halo.
// LDS: [LUTs][T4 22 rows -> T8 10 rows | T9][HB 22 rows -> T6 10 rows | T7 10 rows -> T11 | T15][T14 8 rows] = 79 KB, two per CU.
constexpr int K23_BP = 8, K23_BANDS = G2 / K23_BP, K23_NR = 2 * K23_BP + 6, K23_NM = K23_BP + 2;      // pooled rows, T4 rows, T6/T7/T8 rows
static_assert(G2 % K23_BP == 0 && K23_BP % 4 == 0, "band height must divide the grid; the vertical pool pass sweeps 4 rows");
constexpr int K23_RA = LB, K23_RA_BYTES = K23_NR * T4_ROW;                                  // region A: T4, later T8 | T9
constexpr int K23_RH = K23_RA + K23_RA_BYTES, K23_RH_BYTES = K23_NR * G2 * 20;               // region H: HB, later T6 | T7, later T11 | T15
constexpr int K23_R14 = K23_RH + K23_RH_BYTES;                                              // concat_22 rows of the band
typedef Buf<K23_RA,                              G1, K23_NR, 20, T4_RS,  0, 1> L23_T4;       // local row l = T4 row 2p0-3+l
typedef Buf<K23_RA,                              G1, K23_NR, 20, T4_RS,  0, 1> L23_T4_DW;    // conv2d_10's view: output row t (T6 row p0-1+t) reads local rows 2t .. 2t+2
typedef Buf<K23_RH,                              G2, K23_NR, 20, G2,     0, 0> L23_HB;
// pool_8's passes run BESIDE the conv branch (round 4, as in the fused 56x56 kernel): {horizontal pass || conv2d_10}, {conv2d_12}, {vertical pass || conv2d_13}.
// HB then lives until the vertical pass, so conv2d_10's output T6 sits on concat_22's rows (unwritten until that pass) and T7 behind T9 at the end of region A
// (T4 is dead there once conv2d_10 is through).  YF_K23_POOL_MERGE=0: the staged order of round 3 (T6 | T7 on HB's bytes).
#ifndef YF_K23_POOL_MERGE
#define YF_K23_POOL_MERGE 1
#endif
#ifndef YF_K23_PH
#define YF_K23_PH 3            /* waves of the horizontal pass (the other NW - PH run conv2d_10) */
#endif
#ifndef YF_K23_PV
#define YF_K23_PV 3            /* waves of the vertical pass (the others run conv2d_13) */
#endif
constexpr int K23_T8T9_BYTES = K23_NM * (T8_ROW + 16) + K23_BP * G2 * 48;                   // T8 (16 bytes of row skew) | T9 in region A
#if YF_K23_POOL_MERGE
typedef Buf<K23_R14,                             G2, K23_NM, 32, G2,     0, 0> L23_T6;       // on concat_22's rows (written by the vertical pass, after conv2d_12)
typedef Buf<K23_RA + K23_T8T9_BYTES,             G2, K23_NM,  8, G2,     0, 0> L23_T7;       // rows p0-1 .. p0+8, behind T9 in region A
#else
typedef Buf<K23_RH,                              G2, K23_NM, 32, G2,     0, 0> L23_T6;       // aliases HB (dead after the vertical pool pass)
typedef Buf<K23_RH + K23_NM * G2 * 32,           G2, K23_NM,  8, G2,     0, 0> L23_T7;       // rows p0-1 .. p0+8
#endif
typedef Buf<L23_T7::OFF + G2 * 8,                G2, K23_BP,  8, G2,     0, 0> L23_T7C;      // rows p0 .. p0+7: the residual input of eltwise_18
// T8's 1584-byte rows put the second row of a 32-lane tap read 12 banks behind the first (4 of 16 lanes collide); 16 bytes of skew make it 16
#ifndef YF_K23_T8_SKEW
#define YF_K23_T8_SKEW 16
#endif
constexpr int K23_T8_ROW = T8_ROW + YF_K23_T8_SKEW;
typedef Buf<K23_RA,                              G2, K23_NM, 36, T8_RS,  0, 1, BUF_FS, YF_K23_T8_SKEW> L23_T8;       // aliases T4 (dead after conv2d_10); halo'd rows p0 .. p0+9
typedef Buf<K23_RA + K23_NM * K23_T8_ROW,        G2, K23_BP, 48, G2,     0, 0> L23_T9;
typedef Buf<K23_RH,                              G2, K23_BP,  8, G2,     0, 0> L23_T11;      // aliases T6 (dead after conv2d_12)
typedef Buf<K23_R14,                             G2, K23_BP, 48, G2,     0, 0> L23_T14;
typedef Buf<K23_RH + K23_BP * G2 * 8,            G2, K23_BP, 24, T15_RS, 0, 1> L23_T15;      // behind T11, on T6's old bytes
constexpr int K23_LDS = K23_R14 + K23_BP * G2 * 48;
#if YF_K23_POOL_MERGE
static_assert(YF_K23_T8_SKEW == 16 && K23_T8T9_BYTES == K23_NM * K23_T8_ROW + K23_BP * G2 * 48 && K23_T8T9_BYTES + K23_NM * G2 * 8 <= K23_RA_BYTES && K23_T8T9_BYTES % 16 == 0, "T8 | T9 | T7 fit region A");
static_assert(K23_NM * G2 * 32 <= K23_BP * G2 * 48 && K23_BP * G2 * 8 + K23_BP * T15_ROW <= K23_RH_BYTES, "T6 fits concat_22's rows; T11 | T15 fit HB's bytes");
#else
static_assert(K23_NM * G2 * 32 + K23_NM * G2 * 8 <= K23_RH_BYTES && K23_NM * K23_T8_ROW + K23_BP * G2 * 48 <= K23_RA_BYTES, "aliases fit");
static_assert(K23_BP * G2 * 8 + K23_BP * T15_ROW <= K23_NM * G2 * 32, "T11 | T15 fit T6's bytes (T7 behind them stays alive until conv2d_17)");
#endif
static_assert(K23_LDS <= 81920 && K23_RA % 16 == 0 && K23_RH % 16 == 0 && K23_R14 % 16 == 0 && L23_T9::OFF % 16 == 0 && L23_T15::OFF % 16 == 0, "two workgroups per CU, aligned buffers");

template <int NW>
__global__ void __launch_bounds__(NW * 64, 4) band_k23(const Params prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64, F = 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t* __restrict__ tab = prm.tab;
  int vz = 0;
  asm volatile("" : "+v"(vz));
  load_luts<NT>(reinterpret_cast<uint8_t*>(smem), tab, tid);
  const AddK no_add = {};
  auto addctx = [&](int k) {
    const uint8_t* a = tab + offsetof(yf_table_index, add) + k * sizeof(yf_add);
    return AddK{uniform_u32(a + offsetof(yf_add, mo2)), uniform_u32(a + offsetof(yf_add, zro)),
                (unsigned long)uniform_u32(a + offsetof(yf_add, c64o)) | ((unsigned long)uniform_u32(a + offsetof(yf_add, c64o) + 4) << 32),
                (int)uniform_u32(a + offsetof(yf_add, rso))};
  };
  const uint32_t z_t8 = splat(load_halo_zp(tab, YF_W_DW15)), z_t15 = splat(load_halo_zp(tab, YF_W_DW27));
  constexpr int JT_DW10 = 0, JT_DW15 = 8 * v2::DwGeo<1, 2, L23_T4_DW, L23_T6>::JPG;
  __syncthreads();
  load_resident<LayK23, NT>(smem, tab, tid);
  v2::fill_jobtab<1, 2, L23_T4_DW, L23_T6, JT_DW10, LayK23>(smem, tid);
  v2::fill_jobtab<1, 1, L23_T8, L23_T9, JT_DW15, LayK23>(smem, tid);
  char* frames = smem;
  const long jobs = prm.n * K23_BANDS;
  const bool xcd = YF_BAND_XCD && gridDim.x % 8 == 0 && prm.n % 8 == 0;
  Prefetch<NT, K23_NR * T4_ROW / 16> pre;
  // T4 halo'd rows [2p0-2, 2p0-2+NR) that exist (0 .. G1): contiguous in the arena
  auto range = [&](long job, const char*& src, int& lo_local, int& n16) {
    long fr; int bnd; split_job<K23_BANDS>(job, xcd, fr, bnd);
    const int p0 = bnd * K23_BP;
    const int h0 = 2 * p0 - 2, lo = max(h0, 0), hi = min(h0 + K23_NR, G1 + 1);
    src = prm.arena + fr * (long)ARENA_BYTES + A_T4 + lo * T4_ROW;
    lo_local = lo - h0; n16 = (hi - lo) * (T4_ROW / 16);
  };
  long job = blockIdx.x;
#if defined(YF_LAB) && defined(YF_WHATIF_NO_T4_HBM)       // ... and never read: band_k23 commits whatever its prefetch registers hold.  Together the bound for
#define YF_K23_FETCH(...) do { } while (0)                // any way of making T4's crossing of HBM cheaper (unpadded T4, crossing at T3): profiles/r05_160/whatif.txt
  for (auto& v : pre.v) v = v4u{0u, 0u, 0u, 0u};
#else
#define YF_K23_FETCH(...) pf_fetch(__VA_ARGS__)
#endif
  if (job < jobs) { const char* src; int ll, n16; range(job, src, ll, n16); YF_K23_FETCH(pre, src, n16, tid); }
  for (; job < jobs; job += gridDim.x) {
    long fr; int bnd; split_job<K23_BANDS>(job, xcd, fr, bnd);
    const int p0 = bnd * K23_BP;                                   // first 40x40 row of the band
    char* arena = prm.arena + fr * (long)ARENA_BYTES;
    lds_barrier();
    YF_BAND_PRIO(3);
    { const char* src; int ll, n16; range(job, src, ll, n16); pf_commit(pre, frames + L23_T4::OFF + ll * T4_ROW, n16, tid); }
    lds_barrier();
    if (job + gridDim.x < jobs) { const char* src; int ll, n16; range(job + gridDim.x, src, ll, n16); YF_K23_FETCH(pre, src, n16, tid); }
    // pool_8 horizontal pass over every band row (rows outside the image are never read back), on threads [0, nt)
    auto pool_h = [&](int t0, int nt) {
      constexpr int NO = 5, NCH = G2 / NO;
      static_assert(G2 % NO == 0, "sweeps of 5 outputs");
      for (int i = t0; i < K23_NR * NCH * 5; i += nt) {
        const int cg = i % 5; int t = i / 5;
        const int k = t % NCH; const int l = t / NCH;
        const char* row = frames + L23_T4::OFF + l * T4_ROW + 20 + 4 * cg;                // pixel 0 sits behind the halo column
        char* dst = frames + L23_HB::OFF + l * (G2 * 20) + 4 * cg;
        pool8_sweep<NO, G1 - 1>(k * NO, [&](int x) { return lds_u32(row + x * 20); },
                                [&](int ox, const SplitB& v) { *reinterpret_cast<uint32_t*>(dst + ox * 20) = v.merge(); });
      }
    };
    // vertical pass + QUANTIZE#21 straight into the pooled half of the band's concat_22 rows
    auto pool_v = [&](int t0, int nt) {
      constexpr int NO = 4, NSW = K23_BP / NO;
      for (int i = t0; i < NSW * G2 * 5; i += nt) {
        const int cg = i % 5; int t = i / 5;
        const int ox = t % G2; const int sw = t / G2;
        const char* col = frames + L23_HB::OFF + ox * 20 + 4 * cg;
        char* dst = frames + L23_T14::OFF + ox * 48 + 4 * cg;
        pool8_sweep<NO, G1 - 1>(p0 + sw * NO, [&](int r) { return lds_u32(col + (r - (2 * p0 - 3)) * (G2 * 20)); },
                                [&](int oy, const SplitB& v) { *reinterpret_cast<uint32_t*>(dst + (oy - p0) * (G2 * 48)) = lut4_raw<YF_L_Q21>(v); });
      }
    };
#if YF_K23_POOL_MERGE
    constexpr int PH = YF_K23_PH, PV = YF_K23_PV;
    YF_BAND_PRIO(2);
    if (wave < PH) pool_h(tid, PH * 64);                                                                                      // T4 -> HB ...
    else v2::dw2_stage<F, NW - PH, 2, L23_T4_DW, L23_T6, 18, YF_L_LEAKY11, 4, JT_DW10, LayK23>(frames, tab, wave - PH, lane);  // ... beside conv2d_10 (ten rows): T4 -> T6
    lds_barrier();
    v2::dense2_stage<F, NW, 1, 2, 16, L23_T6, L23_T7, 0, 6, EPI_RAW, 0, L23_T7, 5, -1, 0, -1, LayK23>(frames, nullptr, tab, no_add, wave, lane);
    lds_barrier();
    if (wave < PV) pool_v(tid, PV * 64);                                                                                      // HB -> concat_22 ...
    else v2::dense2_stage<F, NW - PV, 3, 1, 8, L23_T7, L23_T8, 0, 36, EPI_LUT, YF_L_LEAKY14, L23_T8, 6, -1, 0, -1, LayK23>(frames, nullptr, tab, no_add, wave - PV, lane);   // ... beside conv2d_13: T7 -> T8
    fill_column<NT, K23_T8_ROW, 36>(frames + L23_T8::OFF, 0, K23_NM, z_t8, tid);
    fill_column<NT, K23_T8_ROW, 36>(frames + L23_T8::OFF, G2 + 1, K23_NM, z_t8, tid);
    lds_barrier();
#else
    pool_h(tid, NT);
    lds_barrier();
    pool_v(tid, NT);
    lds_barrier();                                                 // T6 (written next) aliases HB
    YF_BAND_PRIO(2);
    v2::dw2_stage<F, NW, 2, L23_T4_DW, L23_T6, 18, YF_L_LEAKY11, 4, JT_DW10, LayK23>(frames, tab, wave, lane);               // ten rows
    lds_barrier();
    v2::dense2_stage<F, NW, 1, 2, 16, L23_T6, L23_T7, 0, 6, EPI_RAW, 0, L23_T7, 5, -1, 0, -1, LayK23>(frames, nullptr, tab, no_add, wave, lane);
    lds_barrier();
    v2::dense2_stage<F, NW, 3, 1, 8, L23_T7, L23_T8, 0, 36, EPI_LUT, YF_L_LEAKY14, L23_T8, 6, -1, 0, -1, LayK23>(frames, nullptr, tab, no_add, wave, lane);
    fill_column<NT, K23_T8_ROW, 36>(frames + L23_T8::OFF, 0, K23_NM, z_t8, tid);
    fill_column<NT, K23_T8_ROW, 36>(frames + L23_T8::OFF, G2 + 1, K23_NM, z_t8, tid);
    lds_barrier();
#endif
    if (p0 == 0) fill_dwords<NT>(frames + L23_T8::OFF, z_t8, T8_ROW, tid);                                       // T8 row -1 = halo
    if (p0 + K23_BP == G2) fill_dwords<NT>(frames + L23_T8::OFF + (K23_NM - 1) * K23_T8_ROW, z_t8, T8_ROW, tid);    // T8 row G2
    if (p0 == 0 || p0 + K23_BP == G2) lds_barrier();
    YF_BAND_PRIO(1);
    v2::dw2_stage<F, NW, 1, L23_T8, L23_T9, 36, YF_L_LEAKY16, 7, JT_DW15, LayK23>(frames, tab, wave, lane);
    lds_barrier();
    v2::dense2_stage<F, NW, 1, 3, 16, L23_T9, L23_T11, 0, 6, EPI_ADD, YF_A_ADD18, L23_T7C, 8, -1, 0, -1, LayK23>(frames, nullptr, tab, addctx(YF_A_ADD18), wave, lane);
    lds_barrier();
    v2::dense2_stage<F, NW, 2, 1, 8, L23_T11, L23_T14, YF_T14_CONV_BASE, 18, EPI_LUT, YF_L_LEAKY20, L23_T14, 9, -1, 0, -1, LayK23>(frames, nullptr, tab, no_add, wave, lane);
    lds_barrier();
    YF_BAND_PRIO(0);
    v2::dense2_stage<F, NW, 2, 3, 16, L23_T14, L23_T15, 0, 24, EPI_LUT, YF_L_LEAKY24, L23_T15, 10, -1, 0, -1, LayK23>(frames, nullptr, tab, no_add, wave, lane);
    fill_column<NT, T15_ROW, 24>(frames + L23_T15::OFF, 0, K23_BP, z_t15, tid);
    lds_barrier();
    store_rows<NT>(arena + A_T15 + (p0 + 1) * T15_ROW, frames + L23_T15::OFF, K23_BP * T15_ROW, tid);
    if (p0 == 0) fill_dwords<NT>(arena + A_T15, z_t15, T15_ROW, tid);
  }
}

// ---- K4: the 20x20 tail ------------------------------------------------------------------------------------------------
// Two 8-wave workgroups per CU (78 KB each) instead of one 16-wave workgroup with the whole T15 (41 KB) in LDS: the tail's
// stages are latency chains with few jobs, so two independent frames per CU with twice the jobs per wave are faster, and two
// workgroups in different phases profit from the priority ladder.  T15 is consumed in two halves of rows (pool_25 and
// conv2d_27 for output rows 0-9, then 10-19) through the slot that later holds T19 and the small tensors; T17 sits on T20's
// slot (dead before conv2d_32 writes it), T33 on T19's.
constexpr int K4_HALF = G3 / 2, K4_ROWS0 = 2 * K4_HALF + 2, K4_ROWS1 = (G2 + 1) - 2 * K4_HALF;      // T15 halo'd rows of the halves
static_assert(G3 % 2 == 0 && K4_HALF >= 4 && K4_ROWS1 <= K4_ROWS0, "two halves of output rows");
constexpr int K4_T19_BYTES = ((G3 + 2) * ((G3 + 2) * 40 + YF_BAND_SKEW) + 15) & ~15;
constexpr int K4_R1 = K4_T19_BYTES + 3 * G3 * G3 * 8;                                     // T19 | T18 | T22 | T26
// T15's rows are 1008 bytes in HBM (16-byte rows for the band copies) = 252 dwords: conv2d_27's stride-2 tap reads (lanes 12 dwords apart, tile rows
// 504 apart) put all 64 lanes on the eight banks 4k -- the probe's worst pattern.  In LDS the rows are one dword longer (pf_commit_rows): tile rows land on
// different bank classes, two lanes per bank instead of eight.
#ifndef YF_K4_T15_SKEW
#define YF_K4_T15_SKEW 4
#endif
constexpr int K4_T15_PITCH = T15_ROW + YF_K4_T15_SKEW;
static_assert(K4_ROWS0 * K4_T15_PITCH <= K4_R1, "a T15 half fits the slot of T19 and the small tensors");
typedef Buf<LB,                                   G2, K4_ROWS0,   24, T15_RS, 0, 1, BUF_FS, YF_K4_T15_SKEW> L4_T15H;  // halo'd rows of one half (halo'd row 0 at OFF)
typedef Buf<LB,                                   G3, G3, 40, G3 + 2, 1, 1, BUF_FS, YF_BAND_SKEW> L4_T19;
typedef Buf<LB + K4_T19_BYTES,                    G3, G3,  8, G3,     0, 0> L4_T18;
typedef Buf<L4_T18::OFF + G3 * G3 * 8,            G3, G3,  8, G3,     0, 0> L4_T22;
typedef Buf<L4_T22::OFF + G3 * G3 * 8,            G3, G3,  8, G3,     0, 0> L4_T26;
typedef Buf<LB + K4_R1,                           G3, G3, 48, G3,     0, 0> L4_T20;
typedef Buf<L4_T20::OFF,                          G3, G3, 32, G3,     0, 0> L4_T17;   // aliases T20
typedef Buf<L4_T20::OFF,                          G3, K4_HALF, 32, G3, 0, 0> L4_T17A; // rows 0 .. HALF-1 (conv2d_27 writes one half at a time)
typedef Buf<L4_T20::OFF + K4_HALF * G3 * 32,      G3, K4_HALF, 32, G3, 0, 0> L4_T17B; // rows HALF .. G3-1
typedef Buf<L4_T20::OFF + G3 * G3 * 48,           G3, G3, 48, G3,     0, 0> L4_T30;
typedef Buf<LB,                                   G3, G3, 32, G3,     0, 0> L4_T33;   // aliases T19 (dead after conv2d_49)
constexpr int K4_BUFS_END = L4_T30::OFF + G3 * G3 * 48;
// lean stage forms in band_k4 (round 3): a frame's thirteen stages need 19 KB of constants -- not resident, but through TWO RING SLOTS in
// LUT-area bytes the tail never indexes: even const-stages (largest: conv2d_47, 2240 B) on LUTs 0-8 [0, 2304), odd ones (largest: a
// depthwise conv, 1760 B) on eltwise_18's add tables [4864, 6912).  conv2d_34 / 40 read their add tables from the resident area instead of
// from their blocks (which would not fit).  Zeros and the three depthwise job tables sit behind the buffers.
struct LayK4 {
  static constexpr int ZERO = K4_BUFS_END, JT = K4_BUFS_END + v2::ZERO_B, JT_BYTES = 256;
  static constexpr int slot(int cs) { return (cs & 1) ? YF_N_LUT * 256 : 0; }
};
constexpr bool k4_ring_ok() {
  for (int cs = 11; cs <= 23; ++cs) {
    const int b = PLAN.vb_bytes[cs] - (yf_cs_add[cs] >= 0 ? 2048 : 0);
    if (b > ((cs & 1) ? 2048 : 2304)) return false;
  }
  return true;
}
static_assert(k4_ring_ok(), "every tail block (without its add tables) fits its ring slot");
constexpr int K4_LDS = K4_BUFS_END + v2::ZERO_B + LayK4::JT_BYTES;
static_assert(K4_LDS <= 81920, "two workgroups per CU");

// pool_25 for output rows [oy0, oy0 + K4_HALF) from a T15 half whose first halo'd row is h0
template <int NT>
YF_STAGE_FN void pool25_half(char* frames, int oy0, int h0, int tid) {
  constexpr int OW = G3, LIM = G2 - 1;
  for (int i = tid; i < K4_HALF * OW * 6; i += NT) {
    const int cg = i % 6; const int p = i / 6;
    const int oy = oy0 + p / OW, ox = p % OW;
    SplitB m;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
        m = m.mx(SplitB(lds_u32(frames + L4_T15H::OFF + (clampi(2 * oy - 1 + ky, 0, LIM) + 1 - h0) * K4_T15_PITCH + (clampi(2 * ox - 1 + kx, 0, LIM) + 1) * 24 + 4 * cg)));
    *reinterpret_cast<uint32_t*>(frames + L4_T30::at(oy, ox) + 4 * cg) = lut4_raw<YF_L_Q45>(m);
  }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 4) band_k4(const Params prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64, F = 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t* __restrict__ tab = prm.tab;
  load_luts<NT>(reinterpret_cast<uint8_t*>(smem), tab, tid);
  const AddK no_add = {};
  auto addctx = [&](int k) {
    const uint8_t* a = tab + offsetof(yf_table_index, add) + k * sizeof(yf_add);
    return AddK{uniform_u32(a + offsetof(yf_add, mo2)), uniform_u32(a + offsetof(yf_add, zro)),
                (unsigned long)uniform_u32(a + offsetof(yf_add, c64o)) | ((unsigned long)uniform_u32(a + offsetof(yf_add, c64o) + 4) << 32),
                (int)uniform_u32(a + offsetof(yf_add, rso))};
  };
  // zeros and the depthwise job tables behind the buffers (the ring slots themselves are filled per stage)
  constexpr int JT_A = 0, JT_B = JT_A + 8 * v2::DwGeo<1, 2, L4_T15H, L4_T17A>::JPG, JT_32 = JT_B + 8 * v2::DwGeo<1, 2, L4_T15H, L4_T17B>::JPG;
  static_assert(JT_32 + 8 * v2::DwGeo<1, 1, L4_T19, L4_T20>::JPG <= LayK4::JT_BYTES, "job tables fit");
  for (int i = tid; i < v2::ZERO_B / 16; i += NT) reinterpret_cast<uint4*>(smem + LayK4::ZERO)[i] = uint4{0, 0, 0, 0};
  v2::fill_jobtab<1, 2, L4_T15H, L4_T17A, JT_A, LayK4>(smem, tid);
  v2::fill_jobtab<1, 2, L4_T15H, L4_T17B, JT_B, LayK4>(smem, tid);
  v2::fill_jobtab<1, 1, L4_T19, L4_T20, JT_32, LayK4>(smem, tid);
  constexpr int LA35 = YF_N_LUT * 256 + YF_A_ADD35 * 2048, LA41 = YF_N_LUT * 256 + YF_A_ADD41 * 2048;      // add tables: resident with the LUTs
  // a stage's constants arrive by LDS-DMA one stage ahead; the barrier that ends a stage waits for the transfer first
#define K4_FETCH(CS) v2::fetch_consts<CS, LayK4, PLAN.vb_bytes[CS] - (yf_cs_add[CS] >= 0 ? 2048 : 0)>(tab, wave, lane)
#define K4_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lds_barrier(); } while (0)
#define K4_DENSE(TPJ, KS, BW, IN, OUT, CH0, COUT, EPI, LUT, ADDB, AD, CS, LAABS) \
  v2::dense2_stage<F, NW, TPJ, KS, BW, IN, OUT, CH0, COUT, EPI, LUT, ADDB, CS, -1, 0, -1, LayK4, LAABS>(frames, out_all, tab, AD, wave, lane)
  char* frames = smem;
  constexpr int N0 = K4_ROWS0 * T15_ROW / 16, N1 = K4_ROWS1 * T15_ROW / 16, H1 = 2 * K4_HALF;     // halves: vectors, first halo'd row of the second
  Prefetch<NT, N0> pre;
  long fr = blockIdx.x;
  if (fr < prm.n) pf_fetch(pre, prm.arena + fr * (long)ARENA_BYTES + A_T15, N0, tid);
  for (; fr < prm.n; fr += gridDim.x) {
    char* out_all = reinterpret_cast<char*>(prm.out) + fr * (long)OUT_FRAME_BYTES;
    const char* t15 = prm.arena + fr * (long)ARENA_BYTES + A_T15;
    YF_BAND_PRIO(3);
    lds_barrier();                                                                    // every wave has left the previous frame's head stage (ring slots, buffers)
    K4_FETCH(11);
    pf_commit_rows<T15_ROW / 16, K4_T15_PITCH>(pre, frames + L4_T15H::OFF, N0, tid);    // halo'd rows 0 .. ROWS0-1
    K4_SYNC();
    pf_fetch(pre, t15 + H1 * T15_ROW, N1, tid);                                       // second half, behind the first half's compute
    pool25_half<NT>(frames, 0, 0, tid);
    v2::dw2_stage<F, NW, 2, L4_T15H, L4_T17A, 24, YF_L_LEAKY28, 11, JT_A, LayK4>(frames, tab, wave, lane);
    lds_barrier();
    pf_commit_rows<T15_ROW / 16, K4_T15_PITCH>(pre, frames + L4_T15H::OFF, N1, tid);    // halo'd rows H1 .. G2
    lds_barrier();
    if (fr + gridDim.x < prm.n) pf_fetch(pre, prm.arena + (fr + gridDim.x) * (long)ARENA_BYTES + A_T15, N0, tid);
    K4_FETCH(12);
    pool25_half<NT>(frames, K4_HALF, H1, tid);
    v2::dw2_stage<F, NW, 2, L4_T15H, L4_T17B, 24, YF_L_LEAKY28, 11, JT_B, LayK4>(frames, tab, wave, lane);
    K4_SYNC();
    YF_BAND_PRIO(2);
    K4_FETCH(13);
    K4_DENSE(1, 2, 16, L4_T17, L4_T18, 0, 8, EPI_RAW, 0, L4_T18, no_add, 12, -1);                              // conv2d_29
    K4_SYNC();
    K4_FETCH(14);
    fill_halo<L4_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW32), tid);
    K4_DENSE(3, 1, 8, L4_T18, L4_T19, 0, 40, EPI_LUT, YF_L_LEAKY31, L4_T19, no_add, 13, -1);                   // conv2d_30
    K4_SYNC();
    K4_FETCH(15);
    v2::dw2_stage<F, NW, 1, L4_T19, L4_T20, 40, YF_L_LEAKY33, 14, JT_32, LayK4>(frames, tab, wave, lane);      // conv2d_32
    K4_SYNC();
    YF_BAND_PRIO(1);
    K4_FETCH(16);
    K4_DENSE(1, 3, 16, L4_T20, L4_T22, 0, 8, EPI_ADD, YF_A_ADD35, L4_T18, addctx(YF_A_ADD35), 15, LA35);       // conv2d_34 + eltwise_35
    K4_SYNC();
    K4_FETCH(17);
    fill_halo<L4_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW38), tid);
    K4_DENSE(3, 1, 8, L4_T22, L4_T19, 0, 40, EPI_LUT, YF_L_LEAKY37, L4_T19, no_add, 16, -1);                   // conv2d_36
    K4_SYNC();
    K4_FETCH(18);
    v2::dw2_stage<F, NW, 1, L4_T19, L4_T20, 40, YF_L_LEAKY39, 17, JT_32, LayK4>(frames, tab, wave, lane);      // conv2d_38
    K4_SYNC();
    K4_FETCH(19);
    K4_DENSE(1, 3, 16, L4_T20, L4_T26, 0, 8, EPI_ADD, YF_A_ADD41, L4_T22, addctx(YF_A_ADD41), 18, LA41);       // conv2d_40 + eltwise_41
    K4_SYNC();
    YF_BAND_PRIO(0);
    K4_FETCH(20);
    K4_DENSE(2, 1, 8, L4_T26, L4_T30, 24, 24, EPI_LUT, YF_L_L43Q44, L4_T30, no_add, 19, -1);                   // conv2d_42 -> concat_46
    K4_SYNC();
    K4_FETCH(21);
    fill_halo<L4_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW49), tid);
    K4_DENSE(2, 3, 16, L4_T30, L4_T19, 0, 40, EPI_LUT, YF_L_LEAKY48, L4_T19, no_add, 20, -1);                  // conv2d_47
    K4_SYNC();
    K4_FETCH(22);
    v2::dw2_stage<F, NW, 1, L4_T19, L4_T20, 40, YF_L_LEAKY50, 21, JT_32, LayK4>(frames, tab, wave, lane);      // conv2d_49
    K4_SYNC();
    K4_FETCH(23);
    K4_DENSE(2, 3, 16, L4_T20, L4_T33, 0, 32, EPI_LUT, YF_L_LEAKY52, L4_T33, no_add, 22, -1);                  // conv2d_51
    K4_SYNC();
    K4_DENSE(1, 2, 16, L4_T33, L4_T33, 0, 18, EPI_HEAD, 0, L4_T33, no_add, 23, -1);                            // conv2d_53 -> head
  }
#undef K4_FETCH
#undef K4_SYNC
#undef K4_DENSE
}
}  // namespace band
