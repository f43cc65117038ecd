// Fused int8 yoloface forward for gfx950 (MI355X): device code.
//
// One workgroup walks groups of F frames through all 31 reference c-layers (reference
// stm32/X-CUBE-AI/App/network.c:2204-2927; ai_network_run -> ai_platform_network_process, network.c:3402-3407)
// with every activation resident in LDS; HBM is touched for the 9408-byte input frame and the 882-byte head only.
//
//   dense 3x3 / 1x1 conv  : v_mfma_i32_16x16x64_i8, weights as the A operand (rows = output channels), pixels as
//                           the B operand (columns), block-diagonal packing (BD pixel sets per MFMA) for the skinny
//                           layers, so every lane ends up owning 4 consecutive output channels of ONE pixel
//   depthwise 3x3         : the same MFMA with one-hot tap packing (lane-private: every lane owns one pixel)
//   bias+requantize       : TFLite MultiplyByQuantizedMultiplier + zero point in four VALU instructions per output:
//                           v_mad_u64_u32 (bias and both rounding constants in its 64-bit addend, its carry-out stands in
//                           for TFLite's sign term), v_addc_co_u32, v_ashrrev, v_med3  (yf_tables.h, yf_pass)
//   LeakyReLU / QUANTIZE  : 256-entry LDS byte LUTs built on the host with TFLite's fixed-point arithmetic
//   max-pool              : separable, packed 2x int16 max on the byte lanes, clamped coordinates
//   residual add          : TFLite int8 ADD arithmetic in the producing conv's epilogue
//   concat                : producers write straight into the concat buffer (no copy)
// Included twice by yf_engine.hip: namespace yf (56x56: the fused kernel) and namespace yf160 (YF_H0 160: the banded kernels).
// This file holds what both share -- the LDS plan, the arithmetic, input staging, pools and the stage forms (namespace v2) -- and includes, from inside
// the namespace:   yf_fused56.hip.h   the fused 56x56 kernel            (namespace yf)
//                  yf_band160.hip.h   the three banded 160x160 kernels  (namespace yf160)
// -DYF_LAB (make lab -> lib_lab/) adds what only tools and two debugging tests use: the other fused shapes, and yf_lab_stages.hip.h /
// yf_lab_layerwise.hip.h = the layer-by-layer 160x160 form with the round-2 stage forms it is written in.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <type_traits>
#include "yf_tables.h"

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"   // absolute LDS addresses (device); the host pass only parses them
#if !defined(YF_LAB) || !defined(YF_LAUNDER)
#undef YF_LAUNDER
#define YF_LAUNDER 0       /* laboratory builds may set it (profiles/r04_whatif.txt): stage groups whose thread index is recomputed per group instead of parked */
#endif
#define YF_ROW_SKEW 4      /* bytes of row skew in the depthwise inputs with 8- / 40-byte pixels (T1, T19: Buf::SK) */
namespace YF_NS {

// Stage functions are inlined (measured: real calls remove the scratch spills of the 128-VGPR builds but cost
// more than they save: 12.9 vs 14.5 M frames/s).  -DYF_NOINLINE_STAGES makes them calls again.
#ifndef YF_NOINLINE_STAGES
#define YF_STAGE_FN __device__ __forceinline__
#else
#define YF_STAGE_FN __device__ __attribute__((noinline))
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef short v2s __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------ LDS plan
// Per-frame arena (bytes).  Buffers alias by lifetime; see DESIGN.md "LDS plan" for the liveness table.
#if !defined(YF_H0) || YF_H0 == 56
constexpr int FRAME_BYTES = 34176;
#endif
constexpr int LUT_BYTES = YF_N_LUT * 256 + YF_ADDLUT_BYTES;    // byte LUTs, then the int32 add tables


// Buf: OFF byte offset in the frame arena, logical W x H, S bytes per pixel, RS pixels per row (incl. halo),
// PT/PL halo rows/cols in front of logical pixel (0,0); FS = bytes between the arenas of consecutive frames of a
// workgroup (the 7x7 tail of the 56x56 kernel packs its frames at half the stride of the front stages).
#if !defined(YF_H0) || YF_H0 == 56
// Timing-only what-if (lab builds, WRONG results; profiles/r04_whatif.txt): the frames of a workgroup YF_WHATIF_FS bytes apart instead of FRAME_BYTES -- their
// arenas overlap, every LDS access stays in range and a third workgroup fits the CU's LDS.
#if defined(YF_LAB) && defined(YF_WHATIF_FS)
constexpr int FRAME_STRIDE = YF_WHATIF_FS;
#else
constexpr int FRAME_STRIDE = FRAME_BYTES;
#endif
constexpr int BUF_FS = FRAME_STRIDE;
#else
constexpr int BUF_FS = 0;                      // one frame per workgroup
#endif
// SK: extra bytes per row (row pitch ROWB = RS * S + SK).  The depthwise tap reads put lane (g, c) of a 32-lane half on row g, column c:
// with 8- or 40-byte pixels a row's 16 lanes cover the 16 even LDS banks, and an even row pitch (in dwords) puts the next row on the
// same banks -- a 2-way conflict on every tap read.  One dword of skew per row makes the pitch odd: rows alternate between the even
// and the odd banks and the reads are conflict-free (T1: conv2d_3's input; T19: the input of conv2d_32 / 38 / 49).
template <int OFF_, int W_, int H_, int S_, int RS_, int PT_, int PL_, int FS_ = BUF_FS, int SK_ = 0>
struct Buf {
  static constexpr int OFF = OFF_, W = W_, H = H_, S = S_, RS = RS_, PT = PT_, PL = PL_, FS = FS_, SK = SK_;
  static constexpr int P = W_ * H_, ROWB = RS_ * S_ + SK_;
  __device__ static __forceinline__ int at(int y, int x) { return OFF_ + (y + PT_) * ROWB + (x + PL_) * S_; }
  __device__ static __forceinline__ int at_p(int p) {
    if constexpr (RS_ == W_ && PT_ == 0 && PL_ == 0) return OFF_ + p * S_;
    else { const int y = p / W_; return at(y, p - y * W_); }
  }
};

#ifndef YF_H0
#define YF_H0 56                               // input height = width; the network is fully convolutional
#endif
constexpr int G0 = YF_H0, G1 = G0 / 2, G2 = G0 / 4, G3 = G0 / 8;   // grids: input, after conv2d_1, after pool_8, after pool_25
static_assert(G0 % 8 == 0 && G3 >= 4, "input size must be a multiple of 8 and at least 32");

#if YF_H0 == 56
// LDS-resident plan for 56x56 (hand-placed, buffers alias by lifetime)
//                 OFF    W   H   S  RS PT PL
typedef Buf<    0, 56, 56,  4, 60, 1, 4> B_IN;    // RGBX dwords, top halo row, halo column at dword 3
typedef Buf<13696, 28, 28,  8, 30, 1, 1, BUF_FS, YF_ROW_SKEW> B_T1;    // conv2d_1 out (+LeakyReLU), halo ring for dw3; 30 rows of 244 bytes
typedef Buf<21024, 28, 28,  8, 28, 0, 0> B_T2;    // conv2d_3 out
typedef Buf<16832, 28, 28,  4, 28, 0, 0> B_T3;    // conv2d_5 out
typedef Buf<    0, 28, 28, 20, 29, 1, 1> B_T4;    // conv2d_6 out, top/left halo for dw10
typedef Buf<16832, 14, 28, 20, 14, 0, 0> B_HB;    // pool_8 horizontal pass [28 rows][14]
typedef Buf<24672, 14, 14, 48, 14, 0, 0> B_T14;   // concat_22: pool [0,18) | conv [20,38)
typedef Buf<16832, 14, 14, 32, 14, 0, 0> B_T6;    // conv2d_10 out
typedef Buf<    0, 14, 14,  8, 14, 0, 0> B_T7;    // conv2d_12 out
typedef Buf< 1568, 14, 14, 36, 16, 1, 1> B_T8;    // conv2d_13 out, halo ring
typedef Buf<10784, 14, 14, 48, 14, 0, 0> B_T9;    // conv2d_15 out
typedef Buf<20192, 14, 14,  8, 14, 0, 0> B_T11;   // eltwise_18 out
typedef Buf<    0, 14, 14, 24, 15, 1, 1> B_T15;   // conv2d_23 out, top/left halo
typedef Buf< 5408,  7,  7, 48,  7, 0, 0> B_T30;   // concat_46: pool [0,24) | conv [24,48)
typedef Buf< 7760,  7,  7, 32,  7, 0, 0> B_T17;   // conv2d_27 out
typedef Buf< 9328,  7,  7,  8,  7, 0, 0> B_T18;   // conv2d_29 out
typedef Buf< 9728,  7,  7, 40,  9, 1, 1, BUF_FS, YF_ROW_SKEW> B_T19;   // conv2d_30/36/47 out, halo ring (three lifetimes); 9 rows of 364 bytes
typedef Buf<13008,  7,  7, 48,  7, 0, 0> B_T20;   // conv2d_32/38/49 out
typedef Buf<15360,  7,  7,  8,  7, 0, 0> B_T22;   // eltwise_35 out
typedef Buf<15752,  7,  7,  8,  7, 0, 0> B_T26;   // eltwise_41 out
typedef Buf<16144,  7,  7, 32,  7, 0, 0> B_T33;   // conv2d_51 out
// The 7x7 tail (pool_25 ... conv2d_53) works on one 17 KB SET per frame: the same offsets as above, T33 on T17's slot
// (dead after conv2d_29) and the staged head behind T26.  FS = FRAME_BYTES addresses the sets at the start of each frame's
// arena; FS = FRAME_BYTES / 2 packs two sets per arena (tail batching: see the kernel).
template <int FS_>
struct TailBufs {
  typedef Buf<    0, 14, 14, 24, 15, 1, 1, FS_> T15;
  typedef Buf< 5408,  7,  7, 48,  7, 0, 0, FS_> T30;
  typedef Buf< 7760,  7,  7, 32,  7, 0, 0, FS_> T17;
  typedef Buf< 9328,  7,  7,  8,  7, 0, 0, FS_> T18;
  typedef Buf< 9728,  7,  7, 40,  9, 1, 1, FS_, YF_ROW_SKEW> T19;
  typedef Buf<13008,  7,  7, 48,  7, 0, 0, FS_> T20;
  typedef Buf<15360,  7,  7,  8,  7, 0, 0, FS_> T22;
  typedef Buf<15752,  7,  7,  8,  7, 0, 0, FS_> T26;
  typedef Buf< 7760,  7,  7, 32,  7, 0, 0, FS_> T33;
  typedef Buf<16160,  7,  7, 18,  7, 0, 0, FS_> HEAD;
  static constexpr int END = 16160 + 882, T15_BYTES = 5408;
  static_assert(T19::OFF + 9 * T19::ROWB <= T20::OFF && T20::OFF + 49 * 48 <= T22::OFF && T26::OFF + 49 * 8 <= HEAD::OFF, "tail buffers do not overlap");
#if !(defined(YF_LAB) && defined(YF_WHATIF_FS))
  static_assert(END <= FS_, "a set fits its stride");
#endif
};
#else
// Any other size: the same buffers laid out one after another in a per-frame HBM arena (nothing aliases; 64 bytes
// of slack behind each buffer absorb the depthwise stage's harmless over-reads on masked lanes).
constexpr int arena_next(int off, int bytes) { return (off + bytes + 64 + 63) & ~63; }
#define YF_SEQ(NAME, PREV_END, W_, H_, S_, RS_, PT_, PL_, ROWS_)                      \
  constexpr int NAME##_OFF = PREV_END;                                                \
  typedef Buf<NAME##_OFF, W_, H_, S_, RS_, PT_, PL_> NAME;                            \
  constexpr int NAME##_END = arena_next(NAME##_OFF, (ROWS_) * (RS_) * (S_));
YF_SEQ(B_IN,  0,         G0, G0,  4, G0 + 4, 1, 4, G0 + 1)
YF_SEQ(B_T1,  B_IN_END,  G1, G1,  8, G1 + 2, 1, 1, G1 + 2)
YF_SEQ(B_T2,  B_T1_END,  G1, G1,  8, G1,     0, 0, G1)
YF_SEQ(B_T3,  B_T2_END,  G1, G1,  4, G1,     0, 0, G1)
YF_SEQ(B_T4,  B_T3_END,  G1, G1, 20, G1 + 1, 1, 1, G1 + 1)
YF_SEQ(B_HB,  B_T4_END,  G2, G1, 20, G2,     0, 0, G1)
YF_SEQ(B_T14, B_HB_END,  G2, G2, 48, G2,     0, 0, G2)
YF_SEQ(B_T6,  B_T14_END, G2, G2, 32, G2,     0, 0, G2)
YF_SEQ(B_T7,  B_T6_END,  G2, G2,  8, G2,     0, 0, G2)
YF_SEQ(B_T8,  B_T7_END,  G2, G2, 36, G2 + 2, 1, 1, G2 + 2)
YF_SEQ(B_T9,  B_T8_END,  G2, G2, 48, G2,     0, 0, G2)
YF_SEQ(B_T11, B_T9_END,  G2, G2,  8, G2,     0, 0, G2)
YF_SEQ(B_T15, B_T11_END, G2, G2, 24, G2 + 1, 1, 1, G2 + 1)
YF_SEQ(B_T30, B_T15_END, G3, G3, 48, G3,     0, 0, G3)
YF_SEQ(B_T17, B_T30_END, G3, G3, 32, G3,     0, 0, G3)
YF_SEQ(B_T18, B_T17_END, G3, G3,  8, G3,     0, 0, G3)
YF_SEQ(B_T19, B_T18_END, G3, G3, 40, G3 + 2, 1, 1, G3 + 2)
YF_SEQ(B_T20, B_T19_END, G3, G3, 48, G3,     0, 0, G3)
YF_SEQ(B_T22, B_T20_END, G3, G3,  8, G3,     0, 0, G3)
YF_SEQ(B_T26, B_T22_END, G3, G3,  8, G3,     0, 0, G3)
YF_SEQ(B_T33, B_T26_END, G3, G3, 32, G3,     0, 0, G3)
#undef YF_SEQ
#endif

#if YF_H0 != 56
constexpr int FRAME_BYTES = B_T33_END;
#endif
constexpr int OUT_FRAME_BYTES = G3 * G3 * 18;
constexpr int IN_FRAME_BYTES = G0 * G0 * 3;

enum { EPI_LUT = 0, EPI_RAW = 1, EPI_ADD = 2, EPI_HEAD = 3, EPI_HEAD_LDS = 4 };   // HEAD: 18-byte pixels at out_all; HEAD_LDS: in the frame's arena (OUT)

// ------------------------------------------------------------------------------------------------ arithmetic
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned long v4ul __attribute__((ext_vector_type(4)));
// TFLite MultiplyByQuantizedMultiplier (shift <= -1) fused with "+ zero point + 128" for the four channels of a pass
// (derivation: yf_tables.h, yf_pass).  acc = O + sum w*x_raw straight out of the MFMA (C operand = inline constant 2.0).
//   {carry, d} = acc * 2M + C64      v_mad_u64_u32: multiplier in a VGPR, 64-bit addend in an SGPR pair, carry to an SGPR pair
//   t = hi32(d) + ZR + carry         v_addc_co_u32
// and the caller finishes with  idx = med3(t >> rshift, 0, 255).  Inline assembly because the carry-out of the multiply-add
// has no C++ spelling; hipcc does not pad hazards for an asm statement (cdna_hip_programming.md 5.7), so the block that
// consumes MFMA results opens with the wait states an MFMA result needs before a VALU read (the compiler emits 8 here).
//
// SIGNLESS (round 6): the roundings without a sign term -- the right shift breaks ties UPWARD (ruy's vector kernels, ARM srshl) or the product is rounded ONCE
// (ruy's portable path): yf_network_set_requant_rounding -- need neither the carry nor a separate ZR: the host folds ZR into C64's high word (yf_host_prep.c,
// build_c64), hi32 of the multiply-add IS t, and the epilogue is THREE VALU instructions per output: v_mad_u64_u32, v_ashrrev, v_med3.  The kernels of namespace
// yfu / yf160u (yf_engine.hip) build their DENSE convolutions this way (DENSE_SIGNLESS below); depthwise convolutions and the residual adds keep the four-instruction
// form, which serves every rounding (under "ties upward on the dense convs" they keep TFLite's reference rounding, which has the sign term).
template <bool AFTER_MFMA, bool SIGNLESS = false>
__device__ __forceinline__ void rq4(const v4i acc, const v4u m2, const v4u zr, const v4ul c64, int (&t)[4]) {
  v2u d0, d1, d2, d3;
  if constexpr (SIGNLESS) {     // the carry-out lands in vcc and is not read: no SGPR pairs held for it
    (void)zr;
#define YF_RQ3_MADS "v_mad_u64_u32 %0, vcc, %4, %8, %12\n\tv_mad_u64_u32 %1, vcc, %5, %9, %13\n\t" \
                    "v_mad_u64_u32 %2, vcc, %6, %10, %14\n\tv_mad_u64_u32 %3, vcc, %7, %11, %15"
#define YF_RQ3_OPS : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) \
                   : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(m2[0]), "v"(m2[1]), "v"(m2[2]), "v"(m2[3]), \
                     "s"(c64[0]), "s"(c64[1]), "s"(c64[2]), "s"(c64[3]) : "vcc"
    if constexpr (AFTER_MFMA) asm("s_nop 7\n\ts_nop 1\n\t" YF_RQ3_MADS YF_RQ3_OPS);
    else asm(YF_RQ3_MADS YF_RQ3_OPS);
#undef YF_RQ3_MADS
#undef YF_RQ3_OPS
    t[0] = (int)d0[1]; t[1] = (int)d1[1]; t[2] = (int)d2[1]; t[3] = (int)d3[1];
    return;
  }
  unsigned long cy0, cy1, cy2, cy3;
#define YF_RQ4_MADS "v_mad_u64_u32 %0, %4, %8, %12, %16\n\tv_mad_u64_u32 %1, %5, %9, %13, %17\n\t" \
                    "v_mad_u64_u32 %2, %6, %10, %14, %18\n\tv_mad_u64_u32 %3, %7, %11, %15, %19"
#define YF_RQ4_OPS : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&s"(cy0), "=&s"(cy1), "=&s"(cy2), "=&s"(cy3) \
                   : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(m2[0]), "v"(m2[1]), "v"(m2[2]), "v"(m2[3]), \
                     "s"(c64[0]), "s"(c64[1]), "s"(c64[2]), "s"(c64[3])
  if constexpr (AFTER_MFMA) asm("s_nop 7\n\ts_nop 1\n\t" YF_RQ4_MADS YF_RQ4_OPS);
  else asm(YF_RQ4_MADS YF_RQ4_OPS);
#undef YF_RQ4_MADS
#undef YF_RQ4_OPS
  asm("v_addc_co_u32_e64 %0, vcc, %4, %8, %12\n\tv_addc_co_u32_e64 %1, vcc, %5, %9, %13\n\t"
      "v_addc_co_u32_e64 %2, vcc, %6, %10, %14\n\tv_addc_co_u32_e64 %3, vcc, %7, %11, %15"
      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
      : "v"(zr[0]), "v"(zr[1]), "v"(zr[2]), "v"(zr[3]), "v"(d0[1]), "v"(d1[1]), "v"(d2[1]), "v"(d3[1]),
        "s"(cy0), "s"(cy1), "s"(cy2), "s"(cy3)
      : "vcc");
}
// the four requantised channels of a pass as LUT indices / unsigned bytes (q + 128)
template <bool AFTER_MFMA, bool SIGNLESS = false>
__device__ __forceinline__ void requant4(const v4i acc, const v4u m2, const v4u zr, const v4ul c64, const v4i rs, int (&idx)[4]) {
  int t[4];
  rq4<AFTER_MFMA, SIGNLESS>(acc, m2, zr, c64, t);
#pragma unroll
  for (int j = 0; j < 4; ++j) idx[j] = min(max(t[j] >> rs[j], 0), 255);     // v_ashrrev, v_med3_i32
}
// the same for TWO channels: the last pass of a layer with 4k + 2 output channels (6, 18) carries two padding channels whose
// requantisation, LUT reads and packing would be thrown away
template <bool AFTER_MFMA, bool SIGNLESS = false>
__device__ __forceinline__ void requant2(const v4i acc, const v4u m2, const v4u zr, const v4ul c64, const v4i rs, int (&idx)[2]) {
  v2u d0, d1;
  if constexpr (SIGNLESS) {
    (void)zr;
    if constexpr (AFTER_MFMA)
      asm("s_nop 7\n\ts_nop 1\n\tv_mad_u64_u32 %0, vcc, %2, %4, %6\n\tv_mad_u64_u32 %1, vcc, %3, %5, %7"
          : "=&v"(d0), "=&v"(d1) : "v"(acc[0]), "v"(acc[1]), "v"(m2[0]), "v"(m2[1]), "s"(c64[0]), "s"(c64[1]) : "vcc");
    else
      asm("v_mad_u64_u32 %0, vcc, %2, %4, %6\n\tv_mad_u64_u32 %1, vcc, %3, %5, %7"
          : "=&v"(d0), "=&v"(d1) : "v"(acc[0]), "v"(acc[1]), "v"(m2[0]), "v"(m2[1]), "s"(c64[0]), "s"(c64[1]) : "vcc");
    idx[0] = min(max((int)d0[1] >> rs[0], 0), 255);
    idx[1] = min(max((int)d1[1] >> rs[1], 0), 255);
    return;
  }
  unsigned long cy0, cy1;
  int t0, t1;
  if constexpr (AFTER_MFMA)
    asm("s_nop 7\n\ts_nop 1\n\tv_mad_u64_u32 %0, %2, %4, %6, %8\n\tv_mad_u64_u32 %1, %3, %5, %7, %9"
        : "=&v"(d0), "=&v"(d1), "=&s"(cy0), "=&s"(cy1) : "v"(acc[0]), "v"(acc[1]), "v"(m2[0]), "v"(m2[1]), "s"(c64[0]), "s"(c64[1]));
  else
    asm("v_mad_u64_u32 %0, %2, %4, %6, %8\n\tv_mad_u64_u32 %1, %3, %5, %7, %9"
        : "=&v"(d0), "=&v"(d1), "=&s"(cy0), "=&s"(cy1) : "v"(acc[0]), "v"(acc[1]), "v"(m2[0]), "v"(m2[1]), "s"(c64[0]), "s"(c64[1]));
  asm("v_addc_co_u32_e64 %0, vcc, %2, %4, %6\n\tv_addc_co_u32_e64 %1, vcc, %3, %5, %7"
      : "=&v"(t0), "=&v"(t1) : "v"(zr[0]), "v"(zr[1]), "v"(d0[1]), "v"(d1[1]), "s"(cy0), "s"(cy1) : "vcc");
  idx[0] = min(max(t0 >> rs[0], 0), 255);
  idx[1] = min(max(t1 >> rs[1], 0), 255);
}
__device__ __forceinline__ uint32_t join2(uint32_t b0, uint32_t b1) {
  uint32_t v;
  asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(v) : "v"(b1), "v"(b0));
  return v;
}
constexpr int ACC0 = YF_ACC_OFFSET;            // MFMA C operand: the inline constant 2.0 (no v_mov)
#ifdef YF_RQ3_DENSE
constexpr bool DENSE_SIGNLESS = true;          // this namespace's dense convolutions requantise in the sign-free three-instruction form (rq4)
#else
constexpr bool DENSE_SIGNLESS = false;
#endif
// A register with no particular content and no instruction behind it: the k-slots of an MFMA B operand whose weights are
// zero may hold anything (integer arithmetic: 0 * x = 0), so they are not cleared.
__device__ __forceinline__ int any_value() { int u; asm volatile("" : "=v"(u)); return u; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }   // v_med3_i32
// four zero-extended bytes -> one dword with three v_lshl_or_b32 (from a|b<<8|c<<16|d<<24 the compiler selects four ops)
__device__ __forceinline__ uint32_t join4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
  uint32_t lo, hi, v;
  asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(lo) : "v"(b1), "v"(b0));
  asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(hi) : "v"(b3), "v"(b2));
  asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(v) : "v"(hi), "v"(lo));
  return v;
}
// Byte LUTs live at LDS offset LUT_ID*256 (the dynamic LDS segment starts at 0: the kernel has no static LDS; the host
// checks it).  Absolute LDS addressing lets the table base ride in the ds_read immediate offset.
typedef const __attribute__((address_space(3))) uint8_t* lds_u8_ptr;
template <int LUT_ID>
__device__ __forceinline__ uint32_t lutb(int idx) {
  return *(lds_u8_ptr)(uint32_t)(LUT_ID * 256 + idx);
}
__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b) {          // v_pk_max_i16
  v2s x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
  v2s r = __builtin_elementwise_max(x, y);
  uint32_t o; __builtin_memcpy(&o, &r, 4); return o;
}
// Per-byte signed max of packed int8x4 on the packed-int16 unit.  "o" compares bytes 3 and 1 where they stand (the high
// byte of each int16 lane; the even byte below it only breaks ties between equal high bytes, which leaves the result's
// high byte unchanged), "e" holds bytes 2 and 0 lifted into the high bytes by one v_pk_lshlrev_b16.  One shift per loaded
// dword, two v_pk_max_i16 per max, one v_perm_b32 to merge.
__device__ __forceinline__ uint32_t pk_shl8(uint32_t d) {                    // v_pk_lshlrev_b16 8
  v2s x; __builtin_memcpy(&x, &d, 4);
  x = x << 8;
  uint32_t o; __builtin_memcpy(&o, &x, 4); return o;
}
struct SplitB {
  uint32_t o, e;
  __device__ __forceinline__ SplitB() : o(0x80008000u), e(0x80008000u) {}
  __device__ __forceinline__ explicit SplitB(uint32_t d) : o(d), e(pk_shl8(d)) {}
  __device__ __forceinline__ SplitB mx(const SplitB& b) const { SplitB r; r.o = pkmax(o, b.o); r.e = pkmax(e, b.e); return r; }
  __device__ __forceinline__ uint32_t merge() const { return __builtin_amdgcn_perm(o, e, 0x07030501u); }   // {o.b3, e.b3, o.b1, e.b1}
  // the four bytes as zero-extended values (two's complement bit patterns): indices of a raw-indexed byte LUT
  __device__ __forceinline__ uint32_t b3() const { return o >> 24; }
  __device__ __forceinline__ uint32_t b2() const { return e >> 24; }
  __device__ __forceinline__ uint32_t b1() const { return (o >> 8) & 255u; }
  __device__ __forceinline__ uint32_t b0() const { return (e >> 8) & 255u; }
};
// QUANTIZE of a pooled value through a RAW-indexed byte LUT (index = the int8 bit pattern; yf_host_prep.c stores the two
// pool LUTs that way, so no ^0x80 is needed here)
template <int LUT_ID>
__device__ __forceinline__ uint32_t lut4_raw(const SplitB& v) {
  return join4(lutb<LUT_ID>(v.b0()), lutb<LUT_ID>(v.b1()), lutb<LUT_ID>(v.b2()), lutb<LUT_ID>(v.b3()));
}

__device__ __forceinline__ uint32_t lds_u32(const char* p) { return *reinterpret_cast<const uint32_t*>(p); }
// wave-uniform table reads: the constant address space makes the compiler use scalar loads (SGPR results)
typedef const __attribute__((address_space(4))) v4i* cv4i_ptr;
typedef const __attribute__((address_space(4))) uint32_t* cu32_ptr;
typedef const __attribute__((address_space(4))) v4ul* cv4ul_ptr;
struct I4 { int x, y, z, w; };
__device__ __forceinline__ I4 uniform_int4(const void* p) { const v4i v = *(cv4i_ptr)(uintptr_t)p; return I4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ uint32_t uniform_u32(const void* p) { return *(cu32_ptr)(uintptr_t)p; }
// yf_pass (80 B): mult2[4] and zr[4] go to VGPRs (vz is a zero the compiler cannot see through, which keeps the load a
// vector load), c64[4] and rshift[4] to SGPRs
struct PassV { v4u m2, zr; };
#ifdef YF_LAB
__device__ __forceinline__ PassV load_pass_v(const uint8_t* pass, int vz) {
  const v4u* p = reinterpret_cast<const v4u*>(pass + vz);
  return PassV{p[0], p[1]};
}
__device__ __forceinline__ v4i load_wfrag(const uint8_t* p, int vz) {
  (void)vz;
  return *reinterpret_cast<const v4i*>(p);
}
#endif
struct PassS { v4ul c64; v4i rs; };
#ifdef YF_LAB
__device__ __forceinline__ PassS load_pass_s(const uint8_t* pass) {
  return PassS{*(cv4ul_ptr)(uintptr_t)(pass + 32), *(cv4i_ptr)(uintptr_t)(pass + 64)};
}
#endif
// stage descriptors out of the index at the head of the table blob, as scalar loads (offsets stay in SGPRs)
// Table layout, compiled in.  The blob is laid out by yf_prepare_tables (yf_host_prep.c) stage by stage with 16-byte
// alignment, and every size in it follows from the network's architecture alone -- so the byte offsets are constants of the
// build.  Kernels address tab + constant (no descriptor fetch in front of every stage's first table load); the engine
// compares this plan with the index the host preparation produced and refuses to start on any difference.
struct TablePlan { int w_off[YF_N_DENSE], c_off[YF_N_DENSE], g_off[YF_N_DW], lut_off, total;
                   int vb_off[YF_N_CS], vb_bytes[YF_N_CS], sb_off[YF_N_CS]; };     // constant blocks of the fused kernel (yf_tables.h)
constexpr int PLAN_COUT[YF_N_DENSE] = {8, 4, 18, 6, 36, 6, 18, 24, 8, 40, 8, 40, 8, 24, 40, 32, 18};
constexpr int PLAN_KROW[YF_N_DENSE] = {YF_CONV1_KROW, 16, 16, 32, 16, 48, 16, 48, 32, 16, 48, 16, 48, 16, 48, 48, 32};
constexpr int PLAN_DWC[YF_N_DW] = {8, 18, 36, 24, 40, 40, 40};
constexpr int plan_passes(int cs) { return yf_cs_dense[cs] >= 0 ? (PLAN_COUT[yf_cs_dense[cs]] + 3) / 4 : (PLAN_DWC[yf_cs_dw[cs]] + 3) / 4; }
constexpr int plan_wbytes(int cs) { return yf_cs_dense[cs] >= 0 ? ((PLAN_COUT[yf_cs_dense[cs]] + 3) & ~3) * PLAN_KROW[yf_cs_dense[cs]] : 0; }
constexpr TablePlan make_plan() {
  TablePlan p = {};
  int off = YF_INDEX_RESERVED;
  for (int i = 0; i < YF_N_DENSE; ++i) {
    const int cp = (PLAN_COUT[i] + 3) & ~3;
    off = (off + 15) & ~15; p.w_off[i] = off; off += cp * PLAN_KROW[i];
    off = (off + 15) & ~15; p.c_off[i] = off; off += (cp / 4) * (int)sizeof(yf_pass);
  }
  for (int i = 0; i < YF_N_DW; ++i) { off = (off + 15) & ~15; p.g_off[i] = off; off += ((PLAN_DWC[i] + 3) / 4) * YF_DW_GROUP_BYTES; }
  off = (off + 15) & ~15; p.lut_off = off; off += YF_N_LUT * 256 + YF_ADDLUT_BYTES + YF_DBG_LUT_BYTES;
  for (int cs = 0; cs < YF_N_CS; ++cs) {
    const int bytes = yf_cs_dense[cs] >= 0 ? plan_wbytes(cs) + plan_passes(cs) * (int)sizeof(yf_pass_v) + (yf_cs_add[cs] >= 0 ? 2048 : 0)
                                           : plan_passes(cs) * YF_DWV_GROUP_BYTES;
    off = (off + 15) & ~15; p.vb_off[cs] = off; p.vb_bytes[cs] = (bytes + 15) & ~15; off += p.vb_bytes[cs];
  }
  for (int cs = 0; cs < YF_N_CS; ++cs) { off = (off + 15) & ~15; p.sb_off[cs] = off; off += plan_passes(cs) * (int)sizeof(yf_pass_s); }
  off = (off + 15) & ~15; off += 64;            // zeroed tail (16-byte reads past the last row stay in bounds)
  p.total = off;
  return p;
}
constexpr TablePlan PLAN = make_plan();
#ifdef YF_LAB
__device__ __forceinline__ yf_dense load_dense(const uint8_t*, int i) {
  yf_dense d = {};
  d.w_off = (uint32_t)PLAN.w_off[i]; d.c_off = (uint32_t)PLAN.c_off[i];
  return d;
}
__device__ __forceinline__ yf_dw load_dw(const uint8_t*, int i) {
  yf_dw d = {};
  d.g_off = (uint32_t)PLAN.g_off[i];
  return d;
}
#endif
__device__ __forceinline__ int load_halo_zp(const uint8_t* tab, int i) {
  return (int)uniform_u32(tab + offsetof(yf_table_index, halo_zp) + 4 * i);
}

// Contiguous job range of a wave: JOBS / NW each, the first JOBS % NW waves take one more.  Waves w and w + 4 share
// a SIMD, so the surplus jobs land on different SIMDs (ceil(JOBS / NW) blocks leave the last waves -- and their SIMDs --
// idle: 49 jobs on 8 waves would be 7,7,7,7,7,7,7,0).
template <int JOBS, int NW>
__device__ __forceinline__ void job_range(int wave, int& j0, int& j1) {
  constexpr int BASE = JOBS / NW, REM = JOBS % NW;
  j0 = wave * BASE + min(wave, REM);
  j1 = j0 + BASE + (wave < REM ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------ halo fill
// RING: 1-pixel border all round (SAME 3x3 stride 1); otherwise top row + left column (explicit PAD, stride 2)
template <class B, bool RING, int F, int NT>
YF_STAGE_FN void fill_halo(char* frames, int zp, int tid) {
  const uint32_t v = (uint32_t)(zp & 255) * 0x01010101u;
  constexpr int DW = B::S / 4;
  constexpr int HR = B::H + B::PT + (RING ? 1 : 0), WR = B::RS;        // halo'd rows / cols
  constexpr int NPIX = RING ? (2 * WR + 2 * (HR - 2)) : (WR + HR - 1);
  for (int i = tid; i < F * NPIX * DW; i += NT) {
    const int d = i % DW, t = i / DW, k = t % NPIX, f = t / NPIX;
    int r, c;
    if constexpr (RING) {
      if (k < WR) { r = 0; c = k; }
      else if (k < 2 * WR) { r = HR - 1; c = k - WR; }
      else { const int m = k - 2 * WR; r = 1 + (m >> 1); c = (m & 1) ? WR - 1 : 0; }
    } else {
      if (k < WR) { r = 0; c = k; } else { r = 1 + (k - WR); c = 0; }
    }
    *reinterpret_cast<uint32_t*>(frames + f * B::FS + B::OFF + r * B::ROWB + c * B::S + 4 * d) = v;
  }
}

// ------------------------------------------------------------------------------------------------ input staging
// NHWC int8 frames (G0*G0*3 B) -> RGBX dwords with halo.  One item = 4 pixels: one 12-byte load, three funnel shifts,
// one 16-byte LDS store.  The X byte of a pixel is whatever byte follows it (its weights are zero, yf_tables.h), so no
// masking is needed.
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, int sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }  // ({hi,lo} >> sh)[31:0]
template <int F, int NT>
YF_STAGE_FN void stage_input(char* frames, const int8_t* __restrict__ in, long first_frame, long n_frames,
                                            int zp, int tid) {
  const uint32_t hv = (uint32_t)(zp & 255) * 0x01010101u;
  constexpr int RSW = B_IN::RS, HH = B_IN::H, WQ = B_IN::W / 4;    // dwords per halo'd row, rows, 4-pixel items per row
  constexpr int PER_FRAME = HH * WQ, TOTAL = F * PER_FRAME, ITERS = (TOTAL + NT - 1) / NT;
  // halo: row 0 (RSW dwords) and dword column 3 of rows 1..HH
  for (int i = tid; i < F * (RSW + HH); i += NT) {
    const int f = i / (RSW + HH), k = i - f * (RSW + HH);
    const int idx = k < RSW ? k : (k - RSW + 1) * RSW + 3;
    *reinterpret_cast<uint32_t*>(frames + f * B_IN::FS + B_IN::OFF + idx * 4) = hv;
  }
  // frames past the end of the batch re-read the last one; everything per item is 32-bit arithmetic off one scalar base
  const long rest = n_frames - 1 - first_frame;
  const int lastf = __builtin_amdgcn_readfirstlane((int)(rest < (long)(F - 1) ? rest : (long)(F - 1)));
  const int8_t* base = in + first_frame * IN_FRAME_BYTES;
  static_assert(RSW == B_IN::W + 4 && B_IN::S == 4, "dst = OFF + 16 * (r + y + RSW/4 + 1) relies on rows of W + 4 dwords");
  // ALL loads first (items past the end re-read the last one), then the shuffles and stores.  One loop with an early exit made every
  // iteration wait for its own load before the next one was issued -- three to four global-load latencies in a row per group (-1.1 %).
  uint32_t ld[ITERS][3];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int i = min(tid + it * NT, TOTAL - 1);
    int f = 0;
#pragma unroll
    for (int k = 1; k < F; ++k) f += (i >= k * PER_FRAME) ? 1 : 0;
    const int r = i - f * PER_FRAME;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(base + (uint32_t)(min(f, lastf) * PER_FRAME + r) * 12u);
    ld[it][0] = src[0]; ld[it][1] = src[1]; ld[it][2] = src[2];
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int i = tid + it * NT;
    if (ITERS * NT == TOTAL || i < TOTAL) {
      int f = 0;
#pragma unroll
      for (int k = 1; k < F; ++k) f += (i >= k * PER_FRAME) ? 1 : 0;
      const int r = i - f * PER_FRAME;
      uint4 px;
      px.x = ld[it][0];
      px.y = funnel(ld[it][1], ld[it][0], 24);
      px.z = funnel(ld[it][2], ld[it][1], 16);
      px.w = ld[it][2] >> 8;
      const int y = (int)((uint32_t)r / (uint32_t)WQ);
      // halo'd dword index ((y + 1) * RSW + 4 * xq + 4) with xq = r - y * WQ, RSW = 4 * WQ + 4  ->  4 * (r + y) + RSW + 4
      *reinterpret_cast<uint4*>(frames + f * B_IN::FS + B_IN::OFF + (RSW + 4) * 4 + 16 * (r + y)) = px;
    }
  }
}

// Camera-format input (the frame preparation fused into conv2d_1's load).  The firmware's
// resize_rgb565_uint8_112_to_56_direct + prepare_yolo_data (yoloface.c:26-93): 112x112 big-endian RGB565 -> 2x2 box average per
// 5/6/5 field (sum of four >> 2) -> re-packed RGB565 -> shift-expanded to 8 bits -> value - 128 as int8.  One item = four
// output pixels of one row = two 16-byte loads (source rows 2y, 2y+1), packed-field arithmetic on pixel PAIRS, one 16-byte
// LDS store of RGBX dwords.  (v - 128) as int8 is v ^ 0x80; ((sum >> 2) << 3) is (sum & 0x7C) << 1; green (sum >> 2) << 2 is sum & 0xFC.
constexpr int CAM_FRAME_BYTES = 112 * 112 * 2;
__device__ __forceinline__ uint32_t cam_pixel(uint32_t s0, uint32_t s1) {     // s0, s1: the pixel pair of rows 2y / 2y+1, halves byte-swapped to values
  const uint32_t r = ((s0 >> 11) & 0x001F001Fu) + ((s1 >> 11) & 0x001F001Fu);
  const uint32_t g = ((s0 >> 5) & 0x003F003Fu) + ((s1 >> 5) & 0x003F003Fu);
  const uint32_t b = (s0 & 0x001F001Fu) + (s1 & 0x001F001Fu);
  const uint32_t rs = r + (r >> 16), gs = g + (g >> 16), bs = b + (b >> 16);   // low halves: sums of the four pixels
  return (((rs & 0x7Cu) << 1) | ((gs & 0xFCu) << 8) | ((bs & 0x7Cu) << 17)) ^ 0x00808080u;
}
template <int F, int NT>
YF_STAGE_FN void stage_input_cam(char* frames, const uint8_t* __restrict__ cam, long first_frame, long n_frames, int zp, int tid) {
  const uint32_t hv = (uint32_t)(zp & 255) * 0x01010101u;
  constexpr int RSW = B_IN::RS, HH = B_IN::H, WQ = B_IN::W / 4;
  constexpr int PER_FRAME = HH * WQ, TOTAL = F * PER_FRAME;
  for (int i = tid; i < F * (RSW + HH); i += NT) {                 // halo: row 0 and dword column 3 of rows 1..HH
    const int f = i / (RSW + HH), k = i - f * (RSW + HH);
    const int idx = k < RSW ? k : (k - RSW + 1) * RSW + 3;
    *reinterpret_cast<uint32_t*>(frames + f * B_IN::FS + B_IN::OFF + idx * 4) = hv;
  }
  const long rest = n_frames - 1 - first_frame;
  const int lastf = __builtin_amdgcn_readfirstlane((int)(rest < (long)(F - 1) ? rest : (long)(F - 1)));
  const uint8_t* base = cam + first_frame * CAM_FRAME_BYTES;
  // items in PAIRS: the four 16-byte loads of two items are issued before the first is used (one loop iteration per item waited for its own
  // two loads before the next item's were issued: up to four global-load latencies in a row per group, as in stage_input before round 3)
  constexpr int ITERS = (TOTAL + NT - 1) / NT;
  auto item = [&](int i, int& f, int& r, int& y, const uint8_t*& src) {
    f = 0;
#pragma unroll
    for (int k = 1; k < F; ++k) f += (i >= k * PER_FRAME) ? 1 : 0;
    r = i - f * PER_FRAME;
    y = (int)((uint32_t)r / (uint32_t)WQ);
    const int xq = r - y * WQ;
    src = base + (uint32_t)(min(f, lastf) * CAM_FRAME_BYTES + (2 * y) * 224 + 16 * xq);
  };
  auto commit = [&](int f, int r, int y, const uint4& a, const uint4& c) {
    constexpr uint32_t SWAP = 0x02030001u;                           // bytes of each 16-bit half swapped: big-endian pairs -> values
    uint4 px;
    px.x = cam_pixel(__builtin_amdgcn_perm(a.x, a.x, SWAP), __builtin_amdgcn_perm(c.x, c.x, SWAP));
    px.y = cam_pixel(__builtin_amdgcn_perm(a.y, a.y, SWAP), __builtin_amdgcn_perm(c.y, c.y, SWAP));
    px.z = cam_pixel(__builtin_amdgcn_perm(a.z, a.z, SWAP), __builtin_amdgcn_perm(c.z, c.z, SWAP));
    px.w = cam_pixel(__builtin_amdgcn_perm(a.w, a.w, SWAP), __builtin_amdgcn_perm(c.w, c.w, SWAP));
    *reinterpret_cast<uint4*>(frames + f * B_IN::FS + B_IN::OFF + (RSW + 4) * 4 + 16 * (r + y)) = px;
  };
#pragma unroll
  for (int it = 0; it < ITERS; it += 2) {
    const int i0 = tid + it * NT, i1 = i0 + NT;
    int f0, r0, y0, f1, r1, y1;
    const uint8_t *s0, *s1;
    item(min(i0, TOTAL - 1), f0, r0, y0, s0);
    item(min(i1, TOTAL - 1), f1, r1, y1, s1);
    const uint4 a0 = *reinterpret_cast<const uint4*>(s0), c0 = *reinterpret_cast<const uint4*>(s0 + 224);
    const uint4 a1 = *reinterpret_cast<const uint4*>(s1), c1 = *reinterpret_cast<const uint4*>(s1 + 224);
    if (i0 < TOTAL) commit(f0, r0, y0, a0, c0);
    if (it + 1 < ITERS && i1 < TOTAL) commit(f1, r1, y1, a1, c1);
  }
}

// residual add (tflite ADD): the final requantisation's constants are the same for every channel (yf_add, device form)
struct AddK { uint32_t mo2, zro; unsigned long c64o; int rso; };

#ifdef YF_LAB
#include "yf_lab_stages.hip.h"      // the round-2 stage forms (laboratory build only)
#endif

// ------------------------------------------------------------------------------------------------ max-pools
// pool_8: 8x8 stride 2 pad 3 on T4 (28x28x18) -> separable; the vertical pass applies QUANTIZE#21 and writes the
// pool half of concat_22.  Out-of-range taps are handled by clamping the coordinate (max is idempotent).
// One item sweeps NO consecutive outputs along the pooled axis.  The 8-wide stride-2 window [2o-3, 2o+4] is exactly
// four ODD pairs R[j] = max(c[2j+1], c[2j+2]), j = o-2 .. o+1, so with S[j] = max(R[j], R[j+1])
//   out[o] = max(S[o-2], S[o])                     (coordinates clamped into [0,LIM]: max is idempotent)
// -- 2*NO+6 loads and 3*NO+5 packed maxima per NO outputs (direct form: 8 loads, 7 maxima per output).  The last
// chunk is shifted inwards (recomputing a few outputs) so every chunk has exactly NO outputs.
// LOADC(k): packed dword at clamped coordinate k along the pooled axis; STORE(o, v): write output o (v: SplitB).
template <int NO, int LIM, class LOADC, class STORE>
__device__ __forceinline__ void pool8_sweep(int o0, LOADC loadc, STORE store) {
  constexpr int NR = NO + 3;                       // odd pairs o0-2 .. o0+NO
  SplitB r[NR];
#pragma unroll
  for (int jj = 0; jj < NR; ++jj) {
    const int j = o0 - 2 + jj;
    r[jj] = SplitB(loadc(clampi(2 * j + 1, 0, LIM))).mx(SplitB(loadc(clampi(2 * j + 2, 0, LIM))));
  }
  SplitB q[NR - 1];                                // q[jj] = S[o0-2+jj]
#pragma unroll
  for (int jj = 0; jj < NR - 1; ++jj) q[jj] = r[jj].mx(r[jj + 1]);
#pragma unroll
  for (int n = 0; n < NO; ++n) store(o0 + n, q[n].mx(q[n + 2]));
}
template <int F, int NT>
YF_STAGE_FN void pool8_h(char* frames, int tid) {
  constexpr int NO = 5, OW = B_HB::W, IH = B_T4::H, NCH = (OW + NO - 1) / NO;   // output chunks, the last shifted left
  for (int i = tid; i < F * IH * NCH * 5; i += NT) {
    const int cg = i % 5; int t = i / 5;
    const int k = t % NCH; t /= NCH;
    const int y = t % IH; const int f = t / IH;
    char* fbase = frames + f * B_T4::FS;
    const char* row = fbase + B_T4::at(y, 0) + 4 * cg;
    char* dst = fbase + B_HB::OFF + (y * OW) * 20 + 4 * cg;
    pool8_sweep<NO, B_T4::W - 1>(min(k * NO, OW - NO),
                                 [&](int x) { return lds_u32(row + x * B_T4::S); },
                                 [&](int ox, const SplitB& v) { *reinterpret_cast<uint32_t*>(dst + ox * 20) = v.merge(); });
  }
}
template <int F, int NT, bool STASH = false>      // STASH (debug builds): the raw pooled value goes to the still unwritten conv half of concat_22
YF_STAGE_FN void pool8_v(char* frames, int tid) {
  constexpr int NO = 5, OW = B_HB::W, OH = B_T14::H, NCH = (OH + NO - 1) / NO;   // the last chunk shifted up
  static_assert(OH >= NO, "column shorter than one sweep");
  for (int i = tid; i < F * OW * NCH * 5; i += NT) {
    const int cg = i % 5; int t = i / 5;
    const int k = t % NCH; t /= NCH;
    const int ox = t % OW; const int f = t / OW;
    char* fbase = frames + f * B_HB::FS;
    const char* col = fbase + B_HB::OFF + ox * 20 + 4 * cg;
    char* dst = fbase + B_T14::OFF + ox * B_T14::S + 4 * cg;
    pool8_sweep<NO, B_HB::H - 1>(min(k * NO, OH - NO),
                                 [&](int r) { return lds_u32(col + r * (OW * 20)); },
                                 [&](int oy, const SplitB& v) {
                                   *reinterpret_cast<uint32_t*>(dst + oy * (OW * B_T14::S)) = lut4_raw<YF_L_Q21>(v);
                                   if constexpr (STASH) *reinterpret_cast<uint32_t*>(dst + oy * (OW * B_T14::S) + YF_T14_CONV_BASE) = v.merge();
                                 });
  }
}

// ================================================================================================ lean stages (round 3)
// The 56x56 fused kernel's own forms of the dense and depthwise stages.  What changes against dense_stage / dw_mfma_stage
// (which the 160x160 kernels keep using):
//   * a stage's constants come from an LDS RING SLOT, not from global memory: one LDS-DMA per stage (global_load_lds_dwordx4, no
//     registers), issued one stage ahead, brings the stage's whole vector-side block (weights in A-fragment order, {2M, ZR} per
//     pass, the residual-add tables); every wave then reads its fragments with ds_read_b128 (~100 cycles instead of a global
//     load's ~1-2 k).  The scalar side ({C64, shift} per pass) is a compact 6.7 KB array read with scalar loads.
//   * lanes that hold an all-zero A fragment read it from a zeroed LDS region at the same immediate offsets: no exec masking, no
//     zero-filling moves per chunk.
//   * stages on the 7x7 grid take ONE FRAME PER TILE (49 of 64 lanes, the same efficiency as 196 pixels in 4 linear tiles):
//     a lane's pixel offsets are per-lane constants and a tile adds a scalar frame offset -- no per-job index arithmetic.
//   * the depthwise stages read the offsets of a job (row block, column segment, frame pair) from a small LDS table built once
//     per workgroup instead of deriving them with ~35 scalar instructions per job.
namespace v2 {
constexpr int LUT_B = YF_N_LUT * 256;                         // byte LUTs at LDS offset 0 (absolute addressing, as before)
constexpr int JT = LUT_B, JT_B = 896;                         // depthwise job tables
constexpr int ZERO = JT + JT_B, ZERO_B = 640;                 // zeros: the A fragments of the lanes outside a row block
constexpr int SLOT0 = ZERO + ZERO_B, SLOT_B = 2816;           // two ring slots for the constant blocks of consecutive const-stages
// behind the ring slots: the halo tables (size depends on the kernel shape), then the frame arenas (pre_bytes)
constexpr int slot(int cs) { return SLOT0 + (cs & 1) * SLOT_B; }
// Where a kernel keeps the lean stages' LDS-resident pieces.  The 56x56 kernel: the regions above, constants through the two ring slots.
// The 160x160 band kernels have their own layouts (constants RESIDENT for the kernel's lifetime, in LUT / add-table bytes they do not use).
struct Lay56 {
  static constexpr int ZERO = v2::ZERO, JT = v2::JT, JT_BYTES = v2::JT_B;
  static constexpr int slot(int cs) { return v2::slot(cs); }
};
constexpr int max_block() { int m = 0; for (int i = 0; i < YF_N_CS; ++i) m = PLAN.vb_bytes[i] > m ? PLAN.vb_bytes[i] : m; return m; }
static_assert(max_block() <= SLOT_B && SLOT0 % 16 == 0 && SLOT_B % 16 == 0, "a constant block fits a ring slot");

typedef const __attribute__((address_space(3))) v4i* lds_v4i_ptr;
typedef const __attribute__((address_space(3))) v4u* lds_v4u_ptr;
typedef const __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
typedef const __attribute__((address_space(3))) v2u* lds_u2_ptr;
__device__ __forceinline__ v4i lds_v4i(uint32_t a) { return *(lds_v4i_ptr)a; }
__device__ __forceinline__ v4u lds_v4u(uint32_t a) { return *(lds_v4u_ptr)a; }

// LDS-DMA of const-stage CS's block into its ring slot: wave w moves bytes [1024 w, 1024 w + 1024) -- one wave-instruction of
// 64 x 16 bytes, no registers.  The compiler does not see the transfer (inline assembly): whoever reads the slot does so behind
// an explicit s_waitcnt vmcnt(0) + barrier (YF_SYNC in the kernel).  M0 carries the LDS destination and is restored.
template <int CS, class LAY = Lay56, int BYTES_ = -1>      // BYTES_: only the first BYTES_ bytes of the block (a residual-add stage whose add tables are resident elsewhere)
__device__ __forceinline__ void fetch_consts(const uint8_t* __restrict__ tab, int wave, int lane) {
  constexpr int BYTES = BYTES_ >= 0 ? BYTES_ : PLAN.vb_bytes[CS], NCHUNK = (BYTES + 1023) / 1024;
  if (wave < NCHUNK) {
    asm volatile("" : "+v"(lane));          // the source address is two instructions: recomputed here, not parked in (spilled) VGPRs for the whole kernel
    const int off = wave * 1024 + lane * 16;
    if (off < BYTES) {
      const uint8_t* src = tab + PLAN.vb_off[CS] + off;
      const uint32_t dst = (uint32_t)(LAY::slot(CS) + wave * 1024);
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  }
}

// floor(p / W) for p < 2048 by one multiply and one shift (checked exhaustively at compile time)
template <int W, int SH = (W > 32 ? 20 : 16)> struct DivW {      // wider grids (the 160x160 bands: 80, 40 columns) need the longer reciprocal
  static constexpr uint32_t M = ((1u << SH) + W - 1) / W;
  static constexpr bool ok() { for (uint32_t p = 0; p < 2048; ++p) if (((p * M) >> SH) != p / W) return false; return true; }
  static_assert(ok(), "multiply-shift division is exact on [0, 2048)");
  __device__ static __forceinline__ int div(int p) { return (int)(((uint32_t)p * M) >> SH); }
};

// residual-add context of the lean stages: the two 256-entry int32 tables sit in the stage's ring slot at LA / LA + 1024
template <int EPI, int LUT_ID, int LA, int STASH_LUT = -1>
__device__ __forceinline__ void epilogue2(char* dstpix, const char* addpix, char* headpix, int chq, int hq, const int (&idx)[4], const AddK& ad,
                                          char* stashpix = nullptr) {
  if constexpr (EPI == EPI_LUT) {
    *reinterpret_cast<uint32_t*>(dstpix + chq) = join4(lutb<LUT_ID>(idx[0]), lutb<LUT_ID>(idx[1]), lutb<LUT_ID>(idx[2]), lutb<LUT_ID>(idx[3]));
    if constexpr (STASH_LUT >= 0) {          // debug builds: the same indices through another table (absolute LDS address) to the stash
      if (stashpix) *reinterpret_cast<uint32_t*>(stashpix + chq) =
          join4(*(lds_u8_ptr)(uint32_t)(STASH_LUT + idx[0]), *(lds_u8_ptr)(uint32_t)(STASH_LUT + idx[1]),
                *(lds_u8_ptr)(uint32_t)(STASH_LUT + idx[2]), *(lds_u8_ptr)(uint32_t)(STASH_LUT + idx[3]));
    }
  } else if constexpr (EPI == EPI_RAW) {
    *reinterpret_cast<uint32_t*>(dstpix + chq) = join4(idx[0], idx[1], idx[2], idx[3]) ^ 0x80808080u;
  } else if constexpr (EPI == EPI_ADD) {
    typedef const __attribute__((address_space(3))) int* lds_i32_ptr;
    if (stashpix) *reinterpret_cast<uint32_t*>(stashpix + chq) = join4(idx[0], idx[1], idx[2], idx[3]) ^ 0x80808080u;   // debug builds: the convolution's own output
    const uint32_t o = lds_u32(addpix + chq) ^ 0x80808080u;
    v4i sum;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      sum[j] = *(lds_i32_ptr)(uint32_t)(LA + 4 * ((o >> (8 * j)) & 255)) + *(lds_i32_ptr)(uint32_t)(LA + 1024 + 4 * idx[j]);
    int r[4];
    requant4<false>(sum, v4u{ad.mo2, ad.mo2, ad.mo2, ad.mo2}, v4u{ad.zro, ad.zro, ad.zro, ad.zro},
                    v4ul{ad.c64o, ad.c64o, ad.c64o, ad.c64o}, v4i{ad.rso, ad.rso, ad.rso, ad.rso}, r);
    *reinterpret_cast<uint32_t*>(dstpix + chq) = join4(r[0], r[1], r[2], r[3]) ^ 0x80808080u;
  } else {  // head: 18 channels per pixel, 2-byte aligned
    const uint32_t v = join4(idx[0], idx[1], idx[2], idx[3]) ^ 0x80808080u;
    uint16_t* dst = reinterpret_cast<uint16_t*>(headpix + hq);
    dst[0] = (uint16_t)v;
    if (hq + 2 < 18) dst[1] = (uint16_t)(v >> 16);
  }
}

// the last pass of a layer with 4k + 2 output channels: two channels requantised, looked up and packed (the two padding bytes of the
// pixel's dword are written as zero; nothing reads them with a non-zero weight)
template <int EPI, int LUT_ID, int LA>
__device__ __forceinline__ void epilogue2_half(char* dstpix, const char* addpix, int chq, const int (&idx)[2], const AddK& ad) {
  static_assert(EPI == EPI_LUT || EPI == EPI_RAW || EPI == EPI_ADD, "half passes: LUT, raw and residual-add epilogues");
  if constexpr (EPI == EPI_LUT) {
    *reinterpret_cast<uint32_t*>(dstpix + chq) = join2(lutb<LUT_ID>(idx[0]), lutb<LUT_ID>(idx[1]));
  } else if constexpr (EPI == EPI_RAW) {
    *reinterpret_cast<uint32_t*>(dstpix + chq) = join2(idx[0], idx[1]) ^ 0x8080u;
  } else {
    typedef const __attribute__((address_space(3))) int* lds_i32_ptr;
    const uint32_t o = lds_u32(addpix + chq) ^ 0x80808080u;
    v4i sum = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 2; ++j)
      sum[j] = *(lds_i32_ptr)(uint32_t)(LA + 4 * ((o >> (8 * j)) & 255)) + *(lds_i32_ptr)(uint32_t)(LA + 1024 + 4 * idx[j]);
    int r[2];
    requant2<false>(sum, v4u{ad.mo2, ad.mo2, ad.mo2, ad.mo2}, v4u{ad.zro, ad.zro, ad.zro, ad.zro},
                    v4ul{ad.c64o, ad.c64o, ad.c64o, ad.c64o}, v4i{ad.rso, ad.rso, ad.rso, ad.rso}, r);
    *reinterpret_cast<uint32_t*>(dstpix + chq) = join2(r[0], r[1]) ^ 0x8080u;
  }
}
#if !defined(YF_LAB) || !defined(YF_CHUNK_UNROLL)
#undef YF_CHUNK_UNROLL
#define YF_CHUNK_UNROLL 0       /* laboratory builds may set it: dense stages with at most this many channel chunks unroll their chunk loop at compile time (the form that gave the fp16
                                   kernel 3 %; here the per-chunk values get parked: scalar instructions -13 %, eight more spills, +0.3 %) */
#endif
#define YF_HALF_PASS 1          /* the last pass of a layer with 4k + 2 output channels requantises its two real channels only */

// ---- dense 1x1 (lane-private MFMA, see dense_stage), constants from ring slot CS
// STASH_OFF >= 0 (debug builds, residual-add stages): the convolution's own requantised output of pixel p also goes to byte
// STASH_OFF + p * STASH_S of the frame's arena (the per-node observer wants the tensor the fused add never materialises)
template <int F, int NW, int TPJ, int KS, int BW, class IN, class OUT, int OUT_CH0, int COUT, int EPI, int LUT_ID, class ADDB, int CS,
          int STASH_OFF = -1, int STASH_S = 0, int STASH_LUT = -1, class LAY = Lay56, int LA_ABS = -1>      // LA_ABS: the stage's add tables at an absolute LDS address instead of inside its block
YF_STAGE_FN void dense2_stage(char* frames, char* out_all, const uint8_t* __restrict__ tab, const AddK ad, int wave, int lane) {
  constexpr int NP = (COUT + 3) / 4, NCH = (NP + TPJ - 1) / TPJ;
  constexpr int P = IN::P, TOT = F * P;
  constexpr bool FRAME_TILES = P <= 64;                       // one frame per tile (7x7 grid): no per-job index arithmetic
  constexpr int MT = FRAME_TILES ? F : (TOT + 63) / 64;
  constexpr int JOBS = NCH * MT, KROW = 16 * KS;
  constexpr int SLOT = LAY::slot(CS), WB = plan_wbytes(CS), PV = SLOT + WB, LA = LA_ABS >= 0 ? LA_ABS : PV + NP * (int)sizeof(yf_pass_v);
  static_assert(yf_cs_dense[CS] >= 0 && KROW == PLAN_KROW[yf_cs_dense[CS]] && NP == plan_passes(CS), "stage and constant block agree");
  static_assert((EPI == EPI_ADD) == (yf_cs_add[CS] >= 0), "residual-add tables travel with their stage");
  static_assert(OUT::P == P || EPI == EPI_HEAD || EPI == EPI_HEAD_LDS, "1x1 conv keeps the grid");
  static_assert(IN::FS == OUT::FS && IN::FS == ADDB::FS, "one frame stride per stage");
  static_assert(IN::S >= 16 * (KS - 1) + BW && (BW == 4 || BW == 8 || BW == 16), "the pixel vector must cover all k-steps");
  static_assert(IN::RS == IN::W && IN::PT == 0 && IN::PL == 0, "dense inputs are plain buffers");
  const int g = lane >> 4, c = lane & 15;
  const bool a_on = (c >> 2) == g;
  const uint32_t a_lane = a_on ? (uint32_t)(SLOT + (c & 3) * KROW) : (uint32_t)LAY::ZERO;      // A fragments: row 4*pass + (c&3), or zeros
  const uint32_t a_step = a_on ? (uint32_t)(TPJ * 4 * KROW) : 0u;
  const uint8_t* sc = tab + PLAN.sb_off[CS];
  // per-lane pixel offsets of the frame-per-tile form
  int in_c = 0, out_c = 0, add_c = 0, stash_c = 0;
  static_assert(STASH_OFF < 0 || EPI == EPI_ADD || (EPI == EPI_LUT && STASH_LUT >= 0), "a stash is the pre-add convolution output or a second LUT's view");
  if constexpr (FRAME_TILES) {
    const int p = min(lane, P - 1);                            // surplus lanes redo the last pixel (same value, same address)
    in_c = IN::OFF + p * IN::S;
    if constexpr (EPI == EPI_HEAD || EPI == EPI_HEAD_LDS) out_c = p * 18;
    else if constexpr (OUT::RS == OUT::W && OUT::PT == 0 && OUT::PL == 0) out_c = OUT::OFF + p * OUT::S + OUT_CH0;
    else { const int y = DivW<OUT::W>::div(p); out_c = OUT::at(y, p - y * OUT::W) + OUT_CH0; }
    add_c = ADDB::OFF + p * ADDB::S;
    if constexpr (STASH_OFF >= 0) stash_c = STASH_OFF + p * STASH_S;
  }
  int j0, j1;
  job_range<JOBS, NW>(wave, j0, j1);
  // one chunk of the wave's job range: jobs mt .. mt + n - 1 of channel chunk `chunk`
  auto run_chunk = [&](const int chunk, int mt, const int n) __attribute__((always_inline)) {
    // ---- the chunk's constants: TPJ passes of {A fragments, 2M, ZR} from the ring slot, {C64, shift} by scalar loads
    v4i a[TPJ][KS];
    PassV pv[TPJ];
    PassS ksr[TPJ];
    const uint32_t a_addr = a_lane + (uint32_t)chunk * a_step;
#pragma unroll
    for (int t = 0; t < TPJ; ++t) {
      const int ps = chunk * TPJ + t;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[t][ks] = lds_v4i(a_addr + (uint32_t)(t * 4 * KROW + 16 * ks));
      pv[t].m2 = lds_v4u((uint32_t)(PV + ps * 32));
      pv[t].zr = lds_v4u((uint32_t)(PV + ps * 32 + 16));
      const uint8_t* s = sc + min(ps, NP - 1) * (int)sizeof(yf_pass_s);
      ksr[t] = PassS{*(cv4ul_ptr)(uintptr_t)s, *(cv4i_ptr)(uintptr_t)(s + 32)};
    }
    const int cq = chunk * TPJ * 4;                              // first channel of the chunk: rides in the pixel bases, the pass in the immediates
    const int out_cc = out_c + cq, add_cc = add_c + cq;
    for (int k = 0; k < n; ++k, ++mt) {
      const char* src; char* dstpix = nullptr; const char* addpix = nullptr; char* headpix = nullptr; char* stashpix = nullptr;
      if constexpr (FRAME_TILES) {
        char* fbase = frames + mt * IN::FS;
        src = fbase + in_c;
        dstpix = fbase + out_cc; addpix = fbase + add_cc;
        if constexpr (STASH_OFF >= 0) stashpix = fbase + stash_c + cq;
        if constexpr (EPI == EPI_HEAD) headpix = out_all + mt * OUT_FRAME_BYTES + out_c;
        if constexpr (EPI == EPI_HEAD_LDS) headpix = fbase + OUT::OFF + out_c;
      } else {
        const int q = min(mt * 64 + lane, TOT - 1);
        int f = 0;
#pragma unroll
        for (int i = 1; i < F; ++i) f += (q >= i * P) ? 1 : 0;
        const int p = q - f * P;
        char* fbase = frames + f * IN::FS;
        src = fbase + IN::OFF + p * IN::S;
        if constexpr (EPI == EPI_HEAD) headpix = out_all + f * OUT_FRAME_BYTES + p * 18;
        else if constexpr (EPI == EPI_HEAD_LDS) headpix = fbase + OUT::OFF + p * 18;
        else if constexpr (OUT::RS == OUT::W && OUT::PT == 0 && OUT::PL == 0) dstpix = fbase + cq + OUT::OFF + p * OUT::S + OUT_CH0;
        else { const int y = DivW<OUT::W>::div(p); dstpix = fbase + cq + OUT::at(y, p - y * OUT::W) + OUT_CH0; }
        if constexpr (EPI == EPI_ADD) addpix = fbase + cq + ADDB::OFF + p * ADDB::S;
        if constexpr (STASH_OFF >= 0) stashpix = fbase + cq + STASH_OFF + p * STASH_S;
      }
      v4i b[KS];
#pragma unroll
      for (int ks = 0; ks < KS - 1; ++ks) b[ks] = *reinterpret_cast<const v4i*>(src + 16 * ks);
      const char* last = src + 16 * (KS - 1);
      if constexpr (BW == 16) b[KS - 1] = *reinterpret_cast<const v4i*>(last);
      else if constexpr (BW == 8) { const int2 t2 = *reinterpret_cast<const int2*>(last); b[KS - 1] = v4i{t2.x, t2.y, any_value(), any_value()}; }
      else b[KS - 1] = v4i{*reinterpret_cast<const int*>(last), any_value(), any_value(), any_value()};
#pragma unroll
      for (int t = 0; t < TPJ; ++t) {
        const int ps = chunk * TPJ + t;
        if (ps < NP) {                                          // uniform
          v4i acc = {ACC0, ACC0, ACC0, ACC0};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][ks], b[ks], acc, 0, 0, 0);
          // the layer's last pass holds two padding channels when COUT = 4k + 2: requantise the two real ones only
          constexpr bool HALF_L = YF_HALF_PASS && COUT % 4 == 2 && STASH_OFF < 0 && (EPI == EPI_LUT || EPI == EPI_RAW || EPI == EPI_ADD);
          const bool half = HALF_L && t == (NP - 1) % TPJ && ps == NP - 1;      // uniform; t is a constant of the unrolled pass
          if (half) {
            if constexpr (HALF_L) {
              int idx2[2];
              requant2<true, DENSE_SIGNLESS>(acc, pv[t].m2, pv[t].zr, ksr[t].c64, ksr[t].rs, idx2);
              epilogue2_half<EPI, LUT_ID, LA>(dstpix, addpix, t * 4, idx2, ad);
            }
          } else {
            int idx[4];
            requant4<true, DENSE_SIGNLESS>(acc, pv[t].m2, pv[t].zr, ksr[t].c64, ksr[t].rs, idx);
            epilogue2<EPI, LUT_ID, LA, STASH_LUT>(dstpix, addpix, headpix, t * 4, ps * 4, idx, ad, stashpix);
          }
        }
      }
    }
  };
  // Stages with at most YF_CHUNK_UNROLL chunks walk them in a loop unrolled at compile time: a chunk's pass numbers, the addresses of its scalar constants,
  // its `pass exists` / `half pass` tests and its channel offset are constants of its copy instead of scalar arithmetic and branches on the wave's path.
  if constexpr (NCH <= YF_CHUNK_UNROLL) {
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
      const int lo = max(j0, chunk * MT), hi = min(j1, (chunk + 1) * MT);
      if (lo < hi) run_chunk(chunk, lo - chunk * MT, hi - lo);
    }
  } else {
    int chunk = (int)((uint32_t)j0 / (uint32_t)MT), mt = j0 - chunk * MT, left = j1 - j0;
    while (left > 0) {
      const int n = min(left, MT - mt);
      run_chunk(chunk, mt, n);
      left -= n; ++chunk; mt = 0;
    }
  }
}

// ---- conv2d_1 (see conv1_stage), constants from ring slot CS
template <int F, int NW, int CS, class IN = B_IN, class OUT = B_T1, class LAY = Lay56>
YF_STAGE_FN void conv1_2_stage(char* frames, const uint8_t* __restrict__ tab, int wave, int lane) {
  constexpr int P = OUT::P, W1 = OUT::W, RSW = IN::RS, TOT = F * P;
  constexpr int MT = (TOT + 63) / 64;
  constexpr int SLOT = LAY::slot(CS), WB = plan_wbytes(CS), PV = SLOT + WB;
  const int g = lane >> 4, c = lane & 15;
  const bool a_on = (c >> 2) == g;
  const uint32_t a_addr = a_on ? (uint32_t)(SLOT + (c & 3) * YF_CONV1_KROW) : (uint32_t)LAY::ZERO;
  const uint8_t* sc = tab + PLAN.sb_off[CS];
  v4i a[2][3];
  PassV pv[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) a[ps][ks] = lds_v4i(a_addr + (uint32_t)(ps * 4 * YF_CONV1_KROW + 16 * ks));
    pv[ps].m2 = lds_v4u((uint32_t)(PV + ps * 32));
    pv[ps].zr = lds_v4u((uint32_t)(PV + ps * 32 + 16));
  }
  int j0, j1;
  job_range<MT, NW>(wave, j0, j1);
  const AddK ad = {};
  for (int mt = j0; mt < j1; ++mt) {
    const int q = min(mt * 64 + lane, TOT - 1);
    int f = 0;
#pragma unroll
    for (int i = 1; i < F; ++i) f += (q >= i * P) ? 1 : 0;
    const int p = q - f * P;
    const int oy = DivW<W1>::div(p), ox = p - oy * W1;
    char* fbase = frames + f * IN::FS;
    // tap (ky,kx) of output (oy,ox) = IN[2oy-1+ky][2ox-1+kx] = halo'd dword (2oy+ky)*RSW + 2ox+kx+3
    const uint32_t* src = reinterpret_cast<const uint32_t*>(fbase + IN::OFF) + (2 * oy * RSW + 2 * ox + 3);
    const v4i b0 = {(int)src[0], (int)src[1], (int)src[2], (int)src[RSW]};
    const v4i b1 = {(int)src[RSW + 1], (int)src[RSW + 2], (int)src[2 * RSW], (int)src[2 * RSW + 1]};
    const v4i b2 = {(int)src[2 * RSW + 2], any_value(), any_value(), any_value()};
    char* dstpix = fbase + OUT::at(oy, ox);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const uint8_t* s = sc + ps * (int)sizeof(yf_pass_s);
      const PassS k = PassS{*(cv4ul_ptr)(uintptr_t)s, *(cv4i_ptr)(uintptr_t)(s + 32)};
      v4i acc = {ACC0, ACC0, ACC0, ACC0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][0], b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][1], b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][2], b2, acc, 0, 0, 0);
      int idx[4];
      requant4<true, DENSE_SIGNLESS>(acc, pv[ps].m2, pv[ps].zr, k.c64, k.rs, idx);
      epilogue2<EPI_LUT, YF_L_LEAKY2, 0>(dstpix, nullptr, nullptr, 4 * ps, 0, idx, ad);
    }
  }
}

// ---- depthwise 3x3 (one-hot lane-private MFMA, see dw_mfma_stage): geometry of a stage instance
template <int F, int STRIDE, class IN, class OUT>
struct DwGeo {
  static constexpr int W = OUT::W, H = OUT::H;
  static constexpr int FL = (W <= 8 && F % 2 == 0) ? 2 : 1;
  static constexpr int NSEG = (W + 15) / 16, NRB = (H + 3) / 4, NFP = F / FL;
  static constexpr int JPG = NFP * NRB * NSEG;                 // jobs per channel group
  static_assert(OUT::RS == W && OUT::PT == 0 && OUT::PL == 0, "depthwise outputs are plain buffers");
  static_assert(IN::FS == OUT::FS, "one frame stride per stage");
  static_assert(H >= 4 && (W >= 16 || W * FL <= 16), "tile shape");
  // offsets (relative to the workgroup's frame arenas) of job jj: the top-left tap of lane (0,0) and its output pixel
  __device__ static __forceinline__ uint2 job(int jj) {
    const int fp = jj / (NRB * NSEG); int rem = jj - fp * (NRB * NSEG);
    const int rb = rem / NSEG, seg = rem - rb * NSEG;
    const int oy0 = min(rb * 4, H - 4);
    const int x0 = (W >= 16) ? min(seg * 16, W - 16) : 0;
    const int fb = fp * FL * IN::FS;
    return uint2{(uint32_t)(fb + IN::OFF + (oy0 * STRIDE) * IN::ROWB + x0 * STRIDE * IN::S), (uint32_t)(fb + OUT::OFF + (oy0 * W + x0) * OUT::S)};
  }
};
#if YF_H0 == 56
// job tables of the five depthwise geometries (8 bytes per job of one channel group), laid out one after another
template <int F, bool BATCH>
struct JobTabs {
  typedef TailBufs<BATCH ? FRAME_STRIDE / 2 : FRAME_STRIDE> U;
  static constexpr int FT = BATCH ? 2 * F : F;
  static constexpr int JT_DW3 = 0;
  static constexpr int JT_DW10 = JT_DW3 + 8 * DwGeo<F, 1, B_T1, B_T2>::JPG;
  static constexpr int JT_DW15 = JT_DW10 + 8 * DwGeo<F, 2, B_T4, B_T6>::JPG;
  static constexpr int JT_DW27 = JT_DW15 + 8 * DwGeo<F, 1, B_T8, B_T9>::JPG;
  static constexpr int JT_DW32 = JT_DW27 + 8 * DwGeo<FT, 2, typename U::T15, typename U::T17>::JPG;      // conv2d_32 / 38 / 49
  static constexpr int END = JT_DW32 + 8 * DwGeo<FT, 1, typename U::T19, typename U::T20>::JPG;
  static_assert(END <= JT_B, "job tables fit");
};
#endif
// ---- halo fills from a table.  The halo pixels of a depthwise input (ring or top row + left column, every frame of the workgroup)
// are a fixed list of LDS offsets: written once per workgroup as uint16 (offset / 4) tables, so that a fill is "thread i < N:
// read entry i, store one pixel of zero points" -- two VALU instructions instead of ~20 of index arithmetic per dword.
template <class B, bool RING, int F>
struct HaloGeo {
  static constexpr int HR = B::H + B::PT + (RING ? 1 : 0), WR = B::RS;
  static constexpr int NPIX = RING ? (2 * WR + 2 * (HR - 2)) : (WR + HR - 1);
  static constexpr int N = F * NPIX;
  static_assert(B::S % 4 == 0 && B::OFF % 4 == 0 && B::FS % 4 == 0, "dword pixels");
  __device__ static __forceinline__ uint32_t entry(int i) {       // byte offset (from the frame arenas) of halo pixel i
    const int f = i / NPIX, k = i - f * NPIX;
    int r, c;
    if constexpr (RING) {
      if (k < WR) { r = 0; c = k; }
      else if (k < 2 * WR) { r = HR - 1; c = k - WR; }
      else { const int m = k - 2 * WR; r = 1 + (m >> 1); c = (m & 1) ? WR - 1 : 0; }
    } else {
      if (k < WR) { r = 0; c = k; } else { r = 1 + (k - WR); c = 0; }
    }
    return (uint32_t)(f * B::FS + B::OFF + r * B::ROWB + c * B::S);
  }
};
#if YF_H0 == 56
template <int F, bool BATCH>
struct HaloTabs {
  typedef TailBufs<BATCH ? FRAME_STRIDE / 2 : FRAME_STRIDE> U;
  static constexpr int FT = BATCH ? 2 * F : F;
  typedef HaloGeo<B_T1, true, F> G1; typedef HaloGeo<B_T4, false, F> G4; typedef HaloGeo<B_T8, true, F> G8;
  typedef HaloGeo<B_T15, false, F> G15; typedef HaloGeo<typename U::T19, true, FT> G19;
  static constexpr int H_T1 = 0, H_T4 = H_T1 + 2 * G1::N, H_T8 = H_T4 + 2 * G4::N, H_T15 = H_T8 + 2 * G8::N, H_T19 = H_T15 + 2 * G15::N;
  static constexpr int BYTES = (H_T19 + 2 * G19::N + 15) & ~15;
};
constexpr int HT = SLOT0 + 2 * SLOT_B;                        // halo tables, then the frame arenas
template <int F, bool BATCH> constexpr int pre_bytes() { return HT + HaloTabs<F, BATCH>::BYTES; }
template <class G, int HOFF, int NT>
__device__ __forceinline__ void build_halotab(char* smem, int tid) {
  for (int i = tid; i < G::N; i += NT) *reinterpret_cast<uint16_t*>(smem + HT + HOFF + 2 * i) = (uint16_t)(G::entry(i) >> 2);
}
template <class B, class G, int HOFF>
YF_STAGE_FN void fill_halo_t(char* frames, int zp, int tid) {
  typedef const __attribute__((address_space(3))) uint16_t* lds_u16_ptr;
  if (tid < G::N) {
    const uint32_t v = (uint32_t)(zp & 255) * 0x01010101u;
    char* p = frames + 4u * (uint32_t)*(lds_u16_ptr)(uint32_t)(HT + HOFF + 2 * tid);
    if constexpr (B::S % 8 == 0 && B::OFF % 8 == 0 && B::FS % 8 == 0) {
#pragma unroll
      for (int d = 0; d < B::S / 8; ++d) *reinterpret_cast<uint2*>(p + 8 * d) = uint2{v, v};
    } else {
#pragma unroll
      for (int d = 0; d < B::S / 4; ++d) *reinterpret_cast<uint32_t*>(p + 4 * d) = v;
    }
  }
}
#endif

// the job table of one stage geometry, written once per workgroup (kernel prologue)
template <int F, int STRIDE, class IN, class OUT, int JTOFF, class LAY = Lay56>
__device__ __forceinline__ void fill_jobtab(char* smem, int tid) {
  typedef DwGeo<F, STRIDE, IN, OUT> G;
  static_assert(JTOFF % 8 == 0 && JTOFF + G::JPG * 8 <= LAY::JT_BYTES, "job table in bounds");
  if (tid < G::JPG) *reinterpret_cast<uint2*>(smem + LAY::JT + JTOFF + 8 * tid) = G::job(tid);
}

template <int F, int NW, int STRIDE, class IN, class OUT, int C, int LUT_ID, int CS, int JTOFF, class LAY = Lay56>
YF_STAGE_FN void dw2_stage(char* frames, const uint8_t* __restrict__ tab, int wave, int lane) {
  typedef DwGeo<F, STRIDE, IN, OUT> G;
  constexpr int W = G::W, FL = G::FL, JPG = G::JPG;
  constexpr int NG = (C + 3) / 4, JOBS = NG * JPG;
  constexpr int DROW = STRIDE * IN::ROWB;                      // input bytes between consecutive output rows
  constexpr int TS = IN::S, TR = IN::ROWB;                     // tap strides: +1 column, +1 row
  constexpr int SLOT = LAY::slot(CS);
  static_assert(yf_cs_dw[CS] >= 0 && NG == plan_passes(CS), "stage and constant block agree");
  const int g = lane >> 4, c = lane & 15;
  const int fl = (FL == 2) ? (c >> 3) : 0;
  const int xl = (FL == 2) ? min(c & 7, W - 1) : min(c, W - 1);      // surplus lanes duplicate the last column (idempotent)
  const char* lane_in = frames + fl * IN::FS + g * DROW + xl * STRIDE * IN::S;     // this lane's pixel: row oy0+g, col x0+xl
  char* lane_out = frames + fl * IN::FS + (g * W + xl) * OUT::S;
  const bool a_on = (c >> 2) == g;
  const uint32_t a_lane = a_on ? (uint32_t)(SLOT + 4 * (c & 3)) : (uint32_t)LAY::ZERO;   // masked weight dwords of channel c&3: +16*tap
  const uint32_t a_step = a_on ? (uint32_t)YF_DWV_GROUP_BYTES : 0u;
  const uint8_t* sc = tab + PLAN.sb_off[CS];
  int j, j1;
  job_range<JOBS, NW>(wave, j, j1);
  int cg = (int)((uint32_t)j / (uint32_t)JPG), jj = j - cg * JPG, left = j1 - j;
  while (left > 0) {
    const uint32_t wa = a_lane + (uint32_t)cg * a_step;
    v4i a0, a1, a2 = {0, 0, 0, 0};
    a0 = v4i{(int)*(lds_u32_ptr)(wa), (int)*(lds_u32_ptr)(wa + 16), (int)*(lds_u32_ptr)(wa + 32), (int)*(lds_u32_ptr)(wa + 48)};
    a1 = v4i{(int)*(lds_u32_ptr)(wa + 64), (int)*(lds_u32_ptr)(wa + 80), (int)*(lds_u32_ptr)(wa + 96), (int)*(lds_u32_ptr)(wa + 112)};
    a2[0] = (int)*(lds_u32_ptr)(wa + 128);
    PassV pv;
    pv.m2 = lds_v4u((uint32_t)(SLOT + cg * YF_DWV_GROUP_BYTES + 144));
    pv.zr = lds_v4u((uint32_t)(SLOT + cg * YF_DWV_GROUP_BYTES + 160));
    const uint8_t* s = sc + cg * (int)sizeof(yf_pass_s);
    const PassS k = PassS{*(cv4ul_ptr)(uintptr_t)s, *(cv4i_ptr)(uintptr_t)(s + 32)};
    const char* lin = lane_in + 4 * cg;
    char* lout = lane_out + 4 * cg;
    auto entry = [&](int job) { return *(lds_u2_ptr)(uint32_t)(LAY::JT + JTOFF + 8 * min(job, JPG - 1)); };    // {src, dst} offsets of a job
    auto taps = [&](const v2u e, v4i& b0, v4i& b1, v4i& b2, char*& dst) {
      const char* src = lin + e[0];
#if defined(YF_LAB) && defined(YF_WHATIF_DW_ROWWIN)
      // Timing-only what-if (WRONG results; profiles/EXPERIMENTS.md round 5): the bound for taps by KERNEL ROW with a three-row register window in the two
      // stride-1 stages on the big grids (conv2d_3, conv2d_15) -- a lane walking down a column strip would read 3 new dwords per output row instead of 9.
      // Here: three of the nine reads are issued, the other six operands reuse them; MFMAs, epilogue and stores unchanged.
      if constexpr (STRIDE == 1 && G::W >= 14) {
        b0[0] = (int)lds_u32(src + 2 * TR);    b0[1] = (int)lds_u32(src + 2 * TR + TS); b0[2] = (int)lds_u32(src + 2 * TR + 2 * TS);
        b0[3] = b0[0]; b1[0] = b0[1]; b1[1] = b0[2]; b1[2] = b0[0]; b1[3] = b0[1]; b2[0] = b0[2];
        dst = lout + e[1];
        return;
      }
#endif
      b0[0] = (int)lds_u32(src);               b0[1] = (int)lds_u32(src + TS);          b0[2] = (int)lds_u32(src + 2 * TS);
      b0[3] = (int)lds_u32(src + TR);          b1[0] = (int)lds_u32(src + TR + TS);     b1[1] = (int)lds_u32(src + TR + 2 * TS);
      b1[2] = (int)lds_u32(src + 2 * TR);      b1[3] = (int)lds_u32(src + 2 * TR + TS); b2[0] = (int)lds_u32(src + 2 * TR + 2 * TS);
      dst = lout + e[1];
    };
    auto conv = [&](const v4i& b0, const v4i& b1, const v4i& b2) {
      v4i acc = {ACC0, ACC0, ACC0, ACC0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc, 0, 0, 0);
      return __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, acc, 0, 0, 0);
    };
    auto finish = [&](const v4i& acc, char* dst) {
      if (YF_HALF_PASS && C % 4 == 2 && cg == NG - 1) {          // uniform: the last channel group of an 18-channel layer has two real channels
        int idx2[2];
        requant2<true>(acc, pv.m2, pv.zr, k.c64, k.rs, idx2);
        *reinterpret_cast<uint32_t*>(dst) = join2(lutb<LUT_ID>(idx2[0]), lutb<LUT_ID>(idx2[1]));
      } else {
        int idx[4];
        requant4<true>(acc, pv.m2, pv.zr, k.c64, k.rs, idx);
        *reinterpret_cast<uint32_t*>(dst) = join4(lutb<LUT_ID>(idx[0]), lutb<LUT_ID>(idx[1]), lutb<LUT_ID>(idx[2]), lutb<LUT_ID>(idx[3]));
      }
    };
    const int n = min(left, JPG - jj);
    int i = 0;
    v2u e0 = entry(jj), e1 = entry(jj + 1);     // the next pair's table entries are read one iteration ahead (behind this pair's tap reads)
    for (; i + 1 < n; i += 2) {               // two jobs in flight: the second job's tap reads and MFMAs overlap the first one's epilogue
      v4i p0, p1, p2 = {0, any_value(), any_value(), any_value()}, q0, q1, q2 = {0, any_value(), any_value(), any_value()};
      char *dp, *dq;
      taps(e0, p0, p1, p2, dp);
      taps(e1, q0, q1, q2, dq);
      e0 = entry(jj + i + 2); e1 = entry(jj + i + 3);
      const v4i ap = conv(p0, p1, p2);
      const v4i aq = conv(q0, q1, q2);
      finish(ap, dp);
      finish(aq, dq);
    }
    if (i < n) {
      v4i b0, b1, b2 = {0, any_value(), any_value(), any_value()};
      char* dst;
      taps(e0, b0, b1, b2, dst);
      finish(conv(b0, b1, b2), dst);
    }
    left -= n; ++cg; jj = 0;
  }
}
// ---- pool_25 (4x4 stride 2 pad 1 on T15, QUANTIZE#45 -> pool half of concat_46) by COLUMNS: one item = (frame, output column,
// channel dword) walks the 14 rows of T15 once -- per row the horizontal 4-tap maximum (clamped columns), pairs of rows
// R[j] = max(h[2j-1], h[2j]), out[oy] = max(R[oy], R[oy+1]) -- and writes its 7 outputs.  230 VALU instructions per item instead
// of 7 x 60 for the direct 4x4 window; FT x 42 items, i.e. a few waves' worth: the stage gives pool_25 to the first POOL25_WAVES
// waves and conv2d_27 (which reads the same T15) to the others.
template <int FT> constexpr int pool25_waves() { return (FT * 42 + 63) / 64; }
template <int FT, class T15, class T30, bool STASH = false>      // STASH (debug builds): raw pooled value -> the conv half of concat_46
YF_STAGE_FN void pool25_cols(char* frames, int item) {
  static_assert(T15::W == 14 && T15::H == 14 && T30::W == 7 && T15::FS == T30::FS, "pool_25 geometry");
  const int t = DivW<6>::div(item), cg = item - 6 * t;
  const int f = DivW<7>::div(t), ox = t - 7 * f;
  if (f >= FT) return;
  char* fbase = frames + f * T15::FS;
  const char* base = fbase + T15::at(0, 0) + 4 * cg;
  constexpr int S = T15::S, ROW = T15::ROWB;
  const int c0 = max(2 * ox - 1, 0) * S, c1 = 2 * ox * S, c2 = c1 + S, c3 = min(2 * ox + 2, 13) * S;
  auto hrow = [&](int r) {
    const char* p = base + r * ROW;
    return SplitB(lds_u32(p + c0)).mx(SplitB(lds_u32(p + c1))).mx(SplitB(lds_u32(p + c2))).mx(SplitB(lds_u32(p + c3)));
  };
  char* dst = fbase + T30::OFF + ox * T30::S + 4 * cg;
  SplitB prev = hrow(0);                                       // R[0] = max(h[-1 -> 0], h[0])
#pragma unroll
  for (int oy = 0; oy < 7; ++oy) {
    SplitB next = hrow(2 * oy + 1);                            // R[oy+1] = max(h[2oy+1], h[2oy+2 -> 13])
    if (2 * oy + 2 <= 13) next = next.mx(hrow(2 * oy + 2));
    const SplitB mxv = prev.mx(next);
    *reinterpret_cast<uint32_t*>(dst + oy * (7 * T30::S)) = lut4_raw<YF_L_Q45>(mxv);
    if constexpr (STASH) *reinterpret_cast<uint32_t*>(dst + oy * (7 * T30::S) + 24) = mxv.merge();
    prev = next;
  }
}
}  // namespace v2
#ifndef YF_GENERIC
#include "yf_fused56.hip.h"         // the fused 56x56 kernel
#else   // YF_GENERIC
#ifdef YF_LAB
#include "yf_lab_layerwise.hip.h"   // layer-by-layer 160x160 form (laboratory build only)
#endif
#include "yf_band160.hip.h"         // the three banded 160x160 kernels
#endif  // YF_GENERIC

}  // namespace YF_NS
#pragma clang diagnostic pop
