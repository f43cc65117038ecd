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
// -DYF_LAB (make lab -> lib_lab/) adds what only tools and two debugging tests use: the other fused shapes and the layer-by-layer 160x160 form with
// the round-2 stage forms it is written in.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "yf_tables.h"

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"   // absolute LDS addresses (device); the host pass only parses them
#ifndef YF_LAUNDER
#define YF_LAUNDER 0
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
template <bool AFTER_MFMA>
__device__ __forceinline__ void rq4(const v4i acc, const v4u m2, const v4u zr, const v4ul c64, int (&t)[4]) {
  v2u d0, d1, d2, d3;
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
template <bool AFTER_MFMA>
__device__ __forceinline__ void requant4(const v4i acc, const v4u m2, const v4u zr, const v4ul c64, const v4i rs, int (&idx)[4]) {
  int t[4];
  rq4<AFTER_MFMA>(acc, m2, zr, c64, t);
#pragma unroll
  for (int j = 0; j < 4; ++j) idx[j] = min(max(t[j] >> rs[j], 0), 255);     // v_ashrrev, v_med3_i32
}
// the same for TWO channels: the last pass of a layer with 4k + 2 output channels (6, 18) carries two padding channels whose
// requantisation, LUT reads and packing would be thrown away
template <bool AFTER_MFMA>
__device__ __forceinline__ void requant2(const v4i acc, const v4u m2, const v4u zr, const v4ul c64, const v4i rs, int (&idx)[2]) {
  v2u d0, d1;
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
// ==== round-2 stage forms (constants from global memory, per-job index arithmetic): what the layer-by-layer 160x160 kernels are written in
// ------------------------------------------------------------------------------------------------ epilogue store

// idx[4]: the pass's four requantised channels as unsigned bytes q + 128 (= LUT indices) of pixel p of frame f
template <int EPI, int LUT_ID, class OUT, int OUT_CH0, class ADDB>
__device__ __forceinline__ void epilogue_store(char* fbase /*frame arena*/, char* out_all, int f, int p, int chq,
                                               const int (&idx)[4], const AddK& ad) {
  if constexpr (EPI == EPI_LUT) {
    *reinterpret_cast<uint32_t*>(fbase + OUT::at_p(p) + OUT_CH0 + chq) =
        join4(lutb<LUT_ID>(idx[0]), lutb<LUT_ID>(idx[1]), lutb<LUT_ID>(idx[2]), lutb<LUT_ID>(idx[3]));
  } else if constexpr (EPI == EPI_RAW) {
    *reinterpret_cast<uint32_t*>(fbase + OUT::at_p(p) + OUT_CH0 + chq) = join4(idx[0], idx[1], idx[2], idx[3]) ^ 0x80808080u;
  } else if constexpr (EPI == EPI_ADD) {
    // tflite ADD (LUT_ID = add index): in1 = stored tensor (ADDB) -> table A, in2 = this conv's output -> table B (which
    // carries the accumulator offset), then one fused requantisation of the sum.
    typedef const __attribute__((address_space(3))) int* lds_i32_ptr;
    constexpr uint32_t LA = YF_N_LUT * 256 + LUT_ID * 2048, LB = LA + 1024;
    const uint32_t o = lds_u32(fbase + ADDB::at_p(p) + chq) ^ 0x80808080u;
    v4i sum;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      sum[j] = *(lds_i32_ptr)(uint32_t)(LA + 4 * ((o >> (8 * j)) & 255)) + *(lds_i32_ptr)(uint32_t)(LB + 4 * idx[j]);
    int r[4];
    requant4<false>(sum, v4u{ad.mo2, ad.mo2, ad.mo2, ad.mo2}, v4u{ad.zro, ad.zro, ad.zro, ad.zro},
                    v4ul{ad.c64o, ad.c64o, ad.c64o, ad.c64o}, v4i{ad.rso, ad.rso, ad.rso, ad.rso}, r);
    *reinterpret_cast<uint32_t*>(fbase + OUT::at_p(p) + OUT_CH0 + chq) = join4(r[0], r[1], r[2], r[3]) ^ 0x80808080u;
  } else {  // head: 18 channels per pixel, 2-byte aligned, staged for one coalesced copy to HBM
    static_assert(EPI == EPI_HEAD || EPI == EPI_HEAD_LDS, "epilogue kind");
    const uint32_t v = join4(idx[0], idx[1], idx[2], idx[3]) ^ 0x80808080u;
    uint16_t* dst = reinterpret_cast<uint16_t*>((EPI == EPI_HEAD ? out_all + f * OUT_FRAME_BYTES : fbase + OUT::OFF) + p * 18 + chq);
    dst[0] = (uint16_t)v;
    if (chq + 2 < 18) dst[1] = (uint16_t)(v >> 16);
  }
}

// ------------------------------------------------------------------------------------------------ dense 1x1
// Lane-private MFMA: the lane's own pixel supplies KS fragments of 16 bytes (k-steps; the last one BW = 4, 8 or 16 bytes
// wide), the A operand of k-step ks carries W[4*pass + (r&3)][16*ks ..] in row r's own slot group only, and KS MFMAs
// accumulate the 4 channels of one pass for 64 pixels.  Every lane owns ONE pixel: no lane is wasted when Cout is not a
// multiple of 16 (6, 8, 18, 24, 40), the constants of a pass are wave-uniform, and the pixel math is shared by the TPJ
// passes of a job.  MFMA count grows (KS per 4 channels) but the matrix pipe is idle anyway.
template <int F, int NW, int TPJ, int KS, int BW, class IN, class OUT, int OUT_CH0, int COUT, int EPI, int LUT_ID, class ADDB>
YF_STAGE_FN void dense_stage(char* frames, char* out_all, const uint8_t* __restrict__ tab, const yf_dense d, const AddK ad,
                             int wave, int lane, int vz) {
  constexpr int NP = (COUT + 3) / 4;                        // passes of 4 output channels
  constexpr int NCH = (NP + TPJ - 1) / TPJ;
  constexpr int P = IN::P, TOT = F * P;
  constexpr int MT = (TOT + 63) / 64;
  constexpr int JOBS = NCH * MT;
  constexpr int KROW = 16 * KS;
  static_assert(OUT::P == P || EPI == EPI_HEAD || EPI == EPI_HEAD_LDS, "1x1 conv keeps the grid");
  static_assert(IN::FS == OUT::FS && IN::FS == ADDB::FS, "one frame stride per stage");
  static_assert(IN::S >= 16 * (KS - 1) + BW && (BW == 4 || BW == 8 || BW == 16), "the pixel vector must cover all k-steps");
  const int g = lane >> 4, c = lane & 15;
  int j0, j1;
  job_range<JOBS, NW>(wave, j0, j1);
  const uint8_t* pp = tab + d.c_off;
  const bool a_on = (c >> 2) == g;
  int cur_chunk = -1;
  v4i a[TPJ][KS];
  PassV pv[TPJ];
  PassS ksr[TPJ];                           // scalar constants stay resident per chunk (one load per tile costs a wait per tile)
  for (int j = j0; j < j1; ++j) {
    const int chunk = j / MT, mt = j - chunk * MT;
    if (chunk != cur_chunk) {
      cur_chunk = chunk;
#pragma unroll
      for (int t = 0; t < TPJ; ++t) {
        const int ps = min(chunk * TPJ + t, NP - 1);
        pv[t] = load_pass_v(pp + ps * (int)sizeof(yf_pass), vz);
        ksr[t] = load_pass_s(pp + ps * (int)sizeof(yf_pass));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a[t][ks] = v4i{0, 0, 0, 0};
          if (a_on) a[t][ks] = load_wfrag(tab + d.w_off + (ps * 4 + (c & 3)) * KROW + 16 * ks, vz);
        }
      }
    }
    const int q = mt * 64 + lane;
    const int qc = min(q, TOT - 1);
    const int f = qc / P, p = qc - f * P;
    char* fbase = frames + f * IN::FS;
    v4i b[KS];
    {
      const char* src = fbase + IN::at_p(p);
      {
#pragma unroll
      for (int ks = 0; ks < KS - 1; ++ks) b[ks] = *reinterpret_cast<const v4i*>(src + 16 * ks);
      const char* last = src + 16 * (KS - 1);
      if constexpr (BW == 16) b[KS - 1] = *reinterpret_cast<const v4i*>(last);
      else if constexpr (BW == 8) { const int2 t2 = *reinterpret_cast<const int2*>(last); b[KS - 1] = v4i{t2.x, t2.y, any_value(), any_value()}; }
      else b[KS - 1] = v4i{*reinterpret_cast<const int*>(last), any_value(), any_value(), any_value()};
      }
    }
#pragma unroll
    for (int t = 0; t < TPJ; ++t) {
      const int ps = chunk * TPJ + t;
      if (ps < NP) {                                          // uniform
        const PassS k = ksr[t];
        v4i acc = {ACC0, ACC0, ACC0, ACC0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][ks], b[ks], acc, 0, 0, 0);
        int idx[4];                         // no exec mask: surplus lanes redo pixel TOT-1 (same value, same address)
        requant4<true>(acc, pv[t].m2, pv[t].zr, k.c64, k.rs, idx);
        epilogue_store<EPI, LUT_ID, OUT, OUT_CH0, ADDB>(fbase, out_all, f, p, ps * 4, idx, ad);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ conv2d_1
// 3x3 stride 2, Cin 3 -> 8 on RGBX dwords, lane-private like the 1x1 stages: the lane's pixel gathers its nine taps
// (nine aligned dwords of the staged frame) into three k-steps, both 4-channel passes share them.
template <int F, int NW, class IN = B_IN, class OUT = B_T1>
YF_STAGE_FN void conv1_stage(char* frames, const uint8_t* __restrict__ tab, const yf_dense d, int wave, int lane, int vz) {
  constexpr int P = OUT::P, W1 = OUT::W, RSW = IN::RS, TOT = F * P;
  constexpr int MT = (TOT + 63) / 64;
  const int g = lane >> 4, c = lane & 15;
  const bool a_on = (c >> 2) == g;
  const uint8_t* pp = tab + d.c_off;
  v4i a[2][3];
  PassV pv[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    pv[ps] = load_pass_v(pp + ps * (int)sizeof(yf_pass), vz);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      a[ps][ks] = v4i{0, 0, 0, 0};
      if (a_on) a[ps][ks] = load_wfrag(tab + d.w_off + (ps * 4 + (c & 3)) * YF_CONV1_KROW + 16 * ks, vz);
    }
  }
  int j0, j1;
  job_range<MT, NW>(wave, j0, j1);
  const AddK ad = {};
  for (int mt = j0; mt < j1; ++mt) {
    const int q = mt * 64 + lane;
    const int qc = min(q, TOT - 1);
    const int f = qc / P, p = qc - f * P;
    const int oy = p / W1, ox = p - oy * W1;
    char* fbase = frames + f * IN::FS;
    // tap (ky,kx) of output (oy,ox) = IN[2oy-1+ky][2ox-1+kx] = halo'd dword (2oy+ky)*RSW + 2ox+kx+3
    const uint32_t* src = reinterpret_cast<const uint32_t*>(fbase + IN::OFF) + (2 * oy * RSW + 2 * ox + 3);
    const v4i b0 = {(int)src[0], (int)src[1], (int)src[2], (int)src[RSW]};
    const v4i b1 = {(int)src[RSW + 1], (int)src[RSW + 2], (int)src[2 * RSW], (int)src[2 * RSW + 1]};
    const v4i b2 = {(int)src[2 * RSW + 2], any_value(), any_value(), any_value()};
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const PassS k = load_pass_s(pp + ps * (int)sizeof(yf_pass));
      v4i acc = {ACC0, ACC0, ACC0, ACC0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][0], b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][1], b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][2], b2, acc, 0, 0, 0);
      int idx[4];                           // no exec mask (surplus lanes redo pixel TOT-1)
      requant4<true>(acc, pv[ps].m2, pv[ps].zr, k.c64, k.rs, idx);
      epilogue_store<EPI_LUT, YF_L_LEAKY2, OUT, 0, OUT>(fbase, nullptr, f, p, 4 * ps, idx, ad);
    }
  }
}

// ------------------------------------------------------------------------------------------------ depthwise on MFMA
// Lane-private one-hot packing.  The 64 k-slots of v_mfma_i32_16x16x64_i8 are supplied by four lane groups of 16
// slots each; rows 4g..4g+3 of the A operand are non-zero only in group g's slots.  D[4g+j][c] is then a 16-long dot
// product over data that lane (g,c) itself supplied -- 64 independent pixels per MFMA, each lane working on ITS OWN
// pixel.  One k-step carries 4 taps x 4 channels (4 aligned dwords of the pixel's halo'd neighbourhood), so the
// 9 taps of a 3x3 depthwise filter take 3 k-steps; A holds w[tap][channel j] at byte j of tap's dword in row 4g+j.
// Per 64 pixels x 4 channels: 9 ds_read_b32 off one address register, 3 MFMAs, no VALU multiply at all.
// IN has a halo holding its zero point; the zero point itself is folded into the requantisation constant.
// A job = 4 output rows x 16 columns (2 frames side by side for the 7x7 grids); border blocks are shifted inwards so
// every lane's neighbourhood address is in range.
template <int F, int NW, int STRIDE, class IN, class OUT, int C, int LUT_ID>
YF_STAGE_FN void dw_mfma_stage(char* frames, const uint8_t* __restrict__ tab, const yf_dw d, int wave, int lane, int vz) {
  constexpr int W = OUT::W, H = OUT::H;
  constexpr int FL = (W <= 8 && F % 2 == 0) ? 2 : 1;       // frames side by side in the 16 lanes of a row tile
  constexpr int NSEG = (W + 15) / 16;                       // 16-column segments, the last one shifted left (28 -> x0 in {0, 12})
  constexpr int NRB = (H + 3) / 4;                          // 4-row blocks (last one shifted up)
  constexpr int NG = (C + 3) / 4;
  constexpr int NFP = F / FL;
  constexpr int JPG = NFP * NRB * NSEG;                     // jobs per channel group
  constexpr int JOBS = NG * JPG;
  constexpr int DROW = STRIDE * IN::ROWB;                   // input bytes between consecutive output rows
  constexpr int TS = IN::S, TR = IN::ROWB;                  // tap strides: +1 column, +1 row
  static_assert(OUT::RS == W && OUT::PT == 0 && OUT::PL == 0, "depthwise outputs are plain buffers");
  static_assert(IN::FS == OUT::FS, "one frame stride per stage");
  static_assert(H >= 4 && (W >= 16 || W * FL <= 16), "tile shape");
  const int g = lane >> 4, c = lane & 15;
  const int fl = (FL == 2) ? (c >> 3) : 0;
  const int xl = (FL == 2) ? min(c & 7, W - 1) : min(c, W - 1);      // surplus lanes duplicate the last column (idempotent)
  const int lane_in = fl * IN::FS + g * DROW + xl * STRIDE * IN::S;      // this lane's pixel: row oy0+g, col x0+xl
  const int lane_out = fl * IN::FS + (g * W + xl) * OUT::S;
  const bool a_on = (c >> 2) == g;                          // A row r = c belongs to row block r>>2
  int j, j1;
  job_range<JOBS, NW>(wave, j, j1);
  while (j < j1) {
    const int cg = j / JPG;
    const int jend = min(j1, (cg + 1) * JPG);
    const uint8_t* grp = tab + d.g_off + cg * YF_DW_GROUP_BYTES;
    const uint32_t* wg = reinterpret_cast<const uint32_t*>(grp);
    v4i a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;                // k-steps: taps 0-3, 4-7, 8
    if (a_on) {
      const uint32_t* wl = wg + (c & 3);                    // masked weight dwords of channel c&3: wl[4*tap]
      a0 = v4i{(int)wl[0], (int)wl[4], (int)wl[8], (int)wl[12]};
      a1 = v4i{(int)wl[16], (int)wl[20], (int)wl[24], (int)wl[28]};
      a2[0] = (int)wl[32];
    }
    const PassV pv = load_pass_v(grp + 144, vz);
    const PassS k = load_pass_s(grp + 144);
    // one job: 9 tap dwords -> 3 MFMAs -> requantise -> LUT -> packed store
    auto taps = [&](int jj, v4i& b0, v4i& b1, v4i& b2, char*& dst) {
      int rem = jj - cg * JPG;
      const int fp = rem / (NRB * NSEG); rem -= fp * (NRB * NSEG);
      const int rb = rem / NSEG, seg = rem - rb * NSEG;
      const int oy0 = min(rb * 4, H - 4);
      const int x0 = (W >= 16) ? min(seg * 16, W - 16) : 0;
      char* fb = frames + fp * FL * IN::FS;
      const char* src = fb + IN::OFF + (oy0 * STRIDE) * IN::ROWB + x0 * STRIDE * IN::S + 4 * cg + lane_in;
      b0[0] = (int)lds_u32(src);               b0[1] = (int)lds_u32(src + TS);          b0[2] = (int)lds_u32(src + 2 * TS);
      b0[3] = (int)lds_u32(src + TR);          b1[0] = (int)lds_u32(src + TR + TS);     b1[1] = (int)lds_u32(src + TR + 2 * TS);
      b1[2] = (int)lds_u32(src + 2 * TR);      b1[3] = (int)lds_u32(src + 2 * TR + TS); b2[0] = (int)lds_u32(src + 2 * TR + 2 * TS);
      dst = fb + OUT::OFF + (oy0 * W + x0) * OUT::S + lane_out + 4 * cg;
    };
    auto conv = [&](const v4i& b0, const v4i& b1, const v4i& b2) {
      v4i acc = {ACC0, ACC0, ACC0, ACC0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc, 0, 0, 0);
      return __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, acc, 0, 0, 0);
    };
    auto finish = [&](const v4i& acc, char* dst) {
      int idx[4];
      requant4<true>(acc, pv.m2, pv.zr, k.c64, k.rs, idx);
      *reinterpret_cast<uint32_t*>(dst) = join4(lutb<LUT_ID>(idx[0]), lutb<LUT_ID>(idx[1]), lutb<LUT_ID>(idx[2]), lutb<LUT_ID>(idx[3]));
    };
    // two jobs in flight per iteration: the second job's tap reads and MFMAs overlap the first one's epilogue chain
    for (; j + 1 < jend; j += 2) {
      v4i p0, p1, p2 = {0, any_value(), any_value(), any_value()}, q0, q1, q2 = {0, any_value(), any_value(), any_value()};
      char *dp, *dq;
      taps(j, p0, p1, p2, dp);
      taps(j + 1, q0, q1, q2, dq);
      const v4i ap = conv(p0, p1, p2);
      const v4i aq = conv(q0, q1, q2);
      finish(ap, dp);
      finish(aq, dq);
    }
    for (; j < jend; ++j) {
      v4i b0, b1, b2 = {0, any_value(), any_value(), any_value()};
      char* dst;
      taps(j, b0, b1, b2, dst);
      finish(conv(b0, b1, b2), dst);
    }
  }
}

#endif   // YF_LAB

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
#ifdef YF_LAB   // (the direct 4x4 form: the layer-by-layer 160x160 kernels)
// pool_25: 4x4 stride 2 pad 1 on T15 (14x14x24) -> QUANTIZE#45 -> pool half of concat_46
template <int F, int NT, class T15 = B_T15, class T30 = B_T30>
YF_STAGE_FN void pool25(char* frames, int tid) {
  constexpr int PP = T30::P, OW = T30::W, LIM = T15::W - 1;
  static_assert(T15::FS == T30::FS, "one frame stride per stage");
  for (int i = tid; i < F * PP * 6; i += NT) {
    const int cg = i % 6; int t = i / 6;
    const int p = t % PP; const int f = t / PP;
    const int oy = p / OW, ox = p - oy * OW;
    char* fbase = frames + f * T15::FS;
    SplitB m;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
        m = m.mx(SplitB(lds_u32(fbase + T15::at(clampi(2 * oy - 1 + ky, 0, LIM), clampi(2 * ox - 1 + kx, 0, LIM)) + 4 * cg)));
    *reinterpret_cast<uint32_t*>(fbase + T30::at_p(p) + 4 * cg) = lut4_raw<YF_L_Q45>(m);
  }
}

#endif   // YF_LAB

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
// an explicit s_waitcnt vmcnt(0) + barrier (V2_SYNC in the kernel).  M0 carries the LDS destination and is restored.
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
#ifndef YF_CHUNK_UNROLL
#define YF_CHUNK_UNROLL 0       /* dense stages with at most this many channel chunks unroll their chunk loop at compile time (0: never) */
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
              requant2<true>(acc, pv[t].m2, pv[t].zr, ksr[t].c64, ksr[t].rs, idx2);
              epilogue2_half<EPI, LUT_ID, LA>(dstpix, addpix, t * 4, idx2, ad);
            }
          } else {
            int idx[4];
            requant4<true>(acc, pv[t].m2, pv[t].zr, ksr[t].c64, ksr[t].rs, idx);
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
      requant4<true>(acc, pv[ps].m2, pv[ps].zr, k.c64, k.rs, idx);
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

// ------------------------------------------------------------------------------------------------ debug dump
// Observer-style per-stage dump (reference observer API, ai_platform_interface.h:684-731): logical NHWC bytes.
template <class B, int C, int F, int NT>
__device__ __forceinline__ void dump_buf(const char* frames, int8_t* dump, long stride, long off, long first_frame,
                                         long n_frames, int tid, int ch0 = 0, int split = 1 << 30, int gap = 0) {
  if (!dump) return;
  for (int i = tid; i < F * B::P * C; i += NT) {
    const int ch = i % C; const int t = i / C;
    const int p = t % B::P; const int f = t / B::P;
    if (first_frame + f >= n_frames) continue;
    const int phys = ch0 + ch + (ch >= split ? gap : 0);
    dump[(first_frame + f) * stride + off + (long)p * C + ch] = (int8_t)frames[f * B::FS + B::at_p(p) + phys];
  }
}

struct DumpOffsets {   // byte offsets of each fused stage's tensor inside one frame's dump record
  enum { T1 = 0, T2 = T1 + 6272, T3 = T2 + 6272, T4 = T3 + 3136, Q21 = T4 + 14112, T6 = Q21 + 3528, T7 = T6 + 3528,
         T8 = T7 + 1176, T9 = T8 + 7056, T11 = T9 + 7056, T14 = T11 + 1176, T15 = T14 + 7056, Q45 = T15 + 4704,
         T17 = Q45 + 1176, T18 = T17 + 1176, T19 = T18 + 392, T20 = T19 + 1960, T22 = T20 + 1960, T23 = T22 + 392,
         T24 = T23 + 1960, T26 = T24 + 1960, T30 = T26 + 392, T31 = T30 + 2352, T32 = T31 + 1960, T33 = T32 + 1960,
         // tensors that the fused stages never materialise, dumped for the per-node observer (platform_abi.c): the raw max-pools (ST's
         // pool nodes carry their input's quantisation; the kernel applies QUANTIZE in the same pass) and the convolutions in front of
         // the residual adds (the add is part of their epilogue).  Debug builds park them in bytes of the concat buffers that are still
         // unwritten at that point (the conv halves) and dump them from there.
         // ... and LEAKY_RELU #43's output (production composes it with QUANTIZE #44 into one LUT)
         P8 = T33 + 1568, C17 = P8 + 3528, P25 = C17 + 1176, C34 = P25 + 1176, C40 = C34 + 392, L43 = C40 + 392,
         TOTAL = L43 + 1176 };
};

// ------------------------------------------------------------------------------------------------ the kernel
struct NetParams {
  const int8_t* in;       // [n][56][56][3] int8
  int8_t* out;            // [n][7][7][18] int8
  long n;
  const uint8_t* tab;     // device table blob (yf_host_prep.c)
  int8_t* dump;           // optional per-stage dump, [n][DumpOffsets::TOTAL]
  int stop_stage;         // debug kernel only: leave the group after this many stages (stage timing); <0 = run all
  // optional fused box decode (heads are decoded while still in LDS): dets == nullptr -> heads only
  yf_det* dets;           // [n][cap] detection records
  int* counts;            // [n] candidates per frame (may exceed cap)
  int cap, mode;          // YF_DECODE_PY / YF_DECODE_FW
  int q_thr;              // smallest quantised confidence that passes the mode's threshold (the sigmoid table is monotonic): set by the engine
  float w_scale, h_scale;
  char* scratch;          // tail batching: gridDim.x * F * TailBufs::T15_BYTES bytes (a workgroup parks one group's T15 there)
};
static_assert(sizeof(yf_table_index) <= YF_INDEX_RESERVED, "index does not fit its reserved slot");

// Issue priority per stage (s_setprio, 0..3), stage order: staging, conv2d_1, 3, 5, 6, pool_8 h, pool_8 v, conv2d_10, 12, 13, 15,
// 17, 19, 23, then the thirteen tail stages.  See the kernel: a workgroup's priority FALLS as its group advances.
#ifndef YF_PRIO_LIST
#define YF_PRIO_LIST 3,3,3,3,3,3,3,3,3,3, 2,2,2,2, 1,1,1,1,1,1, 0,0,0,0,0,0,0
#endif
constexpr int STAGE_PRIO[27] = {YF_PRIO_LIST};
template <int K> __device__ __forceinline__ void stage_prio() {
  if constexpr (K == 0 || STAGE_PRIO[K] != STAGE_PRIO[K - 1]) __builtin_amdgcn_s_setprio(STAGE_PRIO[K]);
}
template <bool DUMP> constexpr bool tail_batch() { return !DUMP; }

// CAM: prm.in holds 112x112 RGB565 camera frames (25 088 B each) instead of int8 56x56x3 frames: the firmware's frame
// preparation runs inside the input staging (stage_input_cam).
template <int F, int NW, bool DUMP, bool CAM = false>
// passes per job of conv2d_13 / conv2d_23 (lab builds may override: the register-hungry settings that spill at the 128-VGPR cap)
#if !defined(YF_LAB) || !defined(YF_TPJ13)
#undef YF_TPJ13
#define YF_TPJ13 3
#endif
#if !defined(YF_LAB) || !defined(YF_TPJ23)
#undef YF_TPJ23
#define YF_TPJ23 2
#endif
#if defined(YF_LAB) && defined(YF_WHATIF_WPE)      // what-if: another register budget (waves per SIMD the kernel must fit)
#define YF_WPE(NW) YF_WHATIF_WPE
#else
#define YF_WPE(NW) ((NW) == 12 ? 6 : (NW) >= 8 ? 4 : ((NW) == 6 ? 3 : 2))
#endif
__global__ void __launch_bounds__(NW * 64, YF_WPE(NW)) yoloface56_fused(const NetParams prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64;
  constexpr bool BATCH = tail_batch<DUMP>();             // tail on two groups at a time (production builds)
  constexpr int FT = BATCH ? 2 * F : F;                  // frames per tail run
  typedef TailBufs<BATCH ? FRAME_STRIDE / 2 : FRAME_STRIDE> U;
  constexpr int OUT_ALL_BYTES = BATCH ? 0 : (F * OUT_FRAME_BYTES + 15) & ~15;      // BATCH stages the heads inside the tail sets
  uint8_t* luts = reinterpret_cast<uint8_t*>(smem);      // LUTs are addressed absolutely: the host checks that the kernel has no static LDS
  constexpr int PRE = v2::pre_bytes<F, tail_batch<DUMP>()>();   // LUTs | depthwise job tables | zeros | two constant ring slots | halo tables
  char* out_all = smem + PRE;
  char* frames = smem + PRE + OUT_ALL_BYTES;
  long parked_first = -1;                                // first frame of the group whose T15 waits in the scratch
  const int tid0 = threadIdx.x;
  const uint8_t* __restrict__ tab = prm.tab;
  int vz = 0;
  asm volatile("" : "+v"(vz));              // a zero the compiler cannot see through: keeps the pass constants' loads vector loads
  // Issue priorities.  A static priority for the first-dispatched half of a workgroup was worth -1.9 % in round 1 and costs
  // 1.7 % with the tail on four frames.  What pays is a priority
  // LADDER over a group's stages (YF_PRIO_LIST, s_setprio before a stage whenever the level changes): 3 up to conv2d_13, 2 up
  // to conv2d_23, 1 for the first six tail stages, 0 for the rest.  The two workgroups of a CU are in different phases; the one
  // in the VALU-bound front stages then issues ahead of the one in the latency-bound tail, which only needs the slots left
  // over.  -6.8 % kernel time in-run (A/B 1.073 against no ladder); every placement of the three steps tried gave 6.0-7.3 %.
  for (int i = tid0; i < v2::LUT_B / 16; i += NT)
    reinterpret_cast<uint4*>(luts)[i] = reinterpret_cast<const uint4*>(tab + PLAN.lut_off)[i];
  for (int i = tid0; i < v2::ZERO_B / 16; i += NT) reinterpret_cast<uint4*>(smem + v2::ZERO)[i] = uint4{0, 0, 0, 0};
  constexpr int DBG_LUT = PRE + OUT_ALL_BYTES + FRAME_BYTES + (F - 1) * FRAME_STRIDE;      // debug builds: LEAKY_RELU #43 alone, behind the frame arenas
  if constexpr (DUMP) { if (tid0 < YF_DBG_LUT_BYTES / 16) reinterpret_cast<uint4*>(smem + DBG_LUT)[tid0] = reinterpret_cast<const uint4*>(tab + PLAN.lut_off + YF_N_LUT * 256 + YF_ADDLUT_BYTES)[tid0]; }
  {   // job tables of the five depthwise geometries (offsets relative to the frame arenas)
    typedef v2::JobTabs<F, tail_batch<DUMP>()> JTS;
    typedef typename JTS::U UT;
    v2::fill_jobtab<F, 1, B_T1, B_T2, JTS::JT_DW3>(smem, tid0);
    v2::fill_jobtab<F, 2, B_T4, B_T6, JTS::JT_DW10>(smem, tid0);
    v2::fill_jobtab<F, 1, B_T8, B_T9, JTS::JT_DW15>(smem, tid0);
    v2::fill_jobtab<JTS::FT, 2, typename UT::T15, typename UT::T17, JTS::JT_DW27>(smem, tid0);
    v2::fill_jobtab<JTS::FT, 1, typename UT::T19, typename UT::T20, JTS::JT_DW32>(smem, tid0);
    typedef v2::HaloTabs<F, tail_batch<DUMP>()> HTS;       // halo pixel lists of the five depthwise inputs
    v2::build_halotab<typename HTS::G1, HTS::H_T1, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G4, HTS::H_T4, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G8, HTS::H_T8, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G15, HTS::H_T15, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G19, HTS::H_T19, NT>(smem, tid0);
  }

  const long n_groups = (prm.n + F - 1) / F;
  const AddK no_add = {};
  auto addctx = [&](int k) {
    const uint8_t* a = tab + offsetof(yf_table_index, add) + k * sizeof(yf_add);
    return AddK{uniform_u32(a + offsetof(yf_add, mo2)), uniform_u32(a + offsetof(yf_add, zro)),
                (unsigned long)uniform_u32(a + offsetof(yf_add, c64o)) | ((unsigned long)uniform_u32(a + offsetof(yf_add, c64o) + 4) << 32),
                (int)uniform_u32(a + offsetof(yf_add, rso))};
  };
  constexpr long DS = DumpOffsets::TOTAL;
#ifdef YF_BARPROF
  // Stage timeline (tools/barrier_profile.py): for ONE group of every workgroup (its second: steady state) each wave stores
  // the cycle counter on arrival at and on release from every __syncthreads(): [wg][wave][40][2] in prm.dump.
  bool prof_on = false;
  int bar_no = 0;
  long long* prof_out = reinterpret_cast<long long*>(prm.dump) + ((long)blockIdx.x * NW + __builtin_amdgcn_readfirstlane(tid0 >> 6)) * 80;
#define YF_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* the LDS-DMA of the next stage's constants, as in the shipped form of YF_SYNC: the stamp follows it */ \
                       if (prof_on && (tid0 & 63) == 0 && bar_no < 40) prof_out[2 * bar_no] = __builtin_readcyclecounter(); __syncthreads(); \
                       if (prof_on && (tid0 & 63) == 0 && bar_no < 40) prof_out[2 * bar_no + 1] = __builtin_readcyclecounter(); ++bar_no; } while (0)
#else
  // the barrier behind a stage also publishes the LDS-DMA of the NEXT stage's constants, which the compiler does not see
#define YF_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#endif
  // Stage calls: constants from an LDS ring slot, fetched one stage ahead by YF_FETCH
#define YF_HALO(B, RING, FR, G, HOFF, WI, TID) \
  v2::fill_halo_t<B, typename v2::HaloTabs<F, BATCH>::G, v2::HaloTabs<F, BATCH>::HOFF>(frames, load_halo_zp(tab, WI), TID)
#define YF_FETCH(CS, WV, LN) v2::fetch_consts<CS>(tab, WV, LN)
#define YF_CONV1(WV, LN, CS) v2::conv1_2_stage<F, NW, CS>(frames, tab, WV, LN)
#define YF_DENSE(FR, TPJ, KS, BW, IN, OUT, CH0, COUT, EPI, LUT, ADDB, DI, AD, WV, LN, CS) \
  v2::dense2_stage<FR, NW, TPJ, KS, BW, IN, OUT, CH0, COUT, EPI, LUT, ADDB, CS>(frames, out_all, tab, AD, WV, LN)
#define YF_DW(FR, STRIDE, IN, OUT, C, LUT, WI, WV, LN, CS, JTOFF) v2::dw2_stage<FR, NW, STRIDE, IN, OUT, C, LUT, CS, v2::JobTabs<F, BATCH>::JTOFF>(frames, tab, WV, LN)
#define YF_DUMP(BUF, C, OFF, ...) \
  if constexpr (DUMP) { if (prm.dump) { dump_buf<BUF, C, F, NT>(frames, prm.dump, DS, DumpOffsets::OFF, first, prm.n, tid, ##__VA_ARGS__); YF_SYNC(); } }
  int stage_no = 0;
#define YF_STAGE_END() if constexpr (DUMP) { if (++stage_no == prm.stop_stage) continue; }
#define YF_PRIO(K) stage_prio<K>()

  // Fused box decode: the staged heads of group g stay in out_all until conv2d_53 of group g+1, so they are decoded by
  // the last F waves DURING conv2d_29 of the next group (a 4-job stage: those waves are idle there), off the critical
  // path; the workgroup's last group is decoded after the loop.
  long prev_first = -1;
  auto decode_prev = [&](int w, int ln) {
    if (prm.dets != nullptr && prev_first >= 0 && w >= NW - F && prev_first + (w - (NW - F)) < prm.n) {
      int dl = ln;
      asm volatile("" : "+v"(dl));              // keep the decode's per-lane index arithmetic out of the kernel-wide hoisted set
      yfdec::decode_frame(reinterpret_cast<const int8_t*>(out_all) + (w - (NW - F)) * OUT_FRAME_BYTES, prev_first + (w - (NW - F)), dl,
                          prm.mode, prm.w_scale, prm.h_scale, prm.dets, prm.counts, prm.cap);
    }
  };
  for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long first = grp * F;
#ifdef YF_BARPROF
    bar_no = 0;
    prof_on = !DUMP && prm.dump != nullptr && grp == (long)blockIdx.x + gridDim.x;
#endif
    // Loop-invariant code motion hoists the per-lane index arithmetic of every stage out of this loop and parks the
    // results in VGPRs for the whole kernel.  YF_LAUNDER selects stage groups (1 front 28x28, 2 middle 14x14, 4 tail
    // 7x7) whose thread index is laundered once per group, i.e. recomputed instead of parked.
    int tid = tid0, tid_f = tid0, tid_m = tid0, tid_t = tid0;
    if constexpr ((YF_LAUNDER & 1) != 0) asm volatile("" : "+v"(tid_f));
    if constexpr ((YF_LAUNDER & 2) != 0) asm volatile("" : "+v"(tid_m));
    if constexpr ((YF_LAUNDER & 4) != 0) asm volatile("" : "+v"(tid_t));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L_f = tid_f & 63, L_m = tid_m & 63, L_t = tid_t & 63;
    const int W_f = __builtin_amdgcn_readfirstlane(tid_f >> 6), W_m = __builtin_amdgcn_readfirstlane(tid_m >> 6), W_t = __builtin_amdgcn_readfirstlane(tid_t >> 6);
    (void)lane; (void)wave;
#if !defined(YF_BARPROF)
    // previous group's arena is dead.  LDS-only barrier: __syncthreads() would also wait for the acknowledgements of the
    // previous group's head / detection / parking stores (vmcnt), which nothing in this group depends on
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    YF_SYNC();
#endif
    stage_no = 0;
    YF_PRIO(0);
    int tid_s = tid0;
    asm volatile("" : "+v"(tid_s));         // the staging offsets are cheap: recomputed per group instead of parked in VGPRs for the whole kernel
    if constexpr (CAM) stage_input_cam<F, NT>(frames, reinterpret_cast<const uint8_t*>(prm.in), first, prm.n, (int)uniform_u32(tab + offsetof(yf_table_index, in_zp)), tid_s);
    else stage_input<F, NT>(frames, prm.in, first, prm.n, (int)uniform_u32(tab + offsetof(yf_table_index, in_zp)), tid_s);
    YF_FETCH(0, W_f, L_f);                                                                            // conv2d_1's constants -> ring slot 0
    YF_HALO(B_T1, true, F, G1, H_T1, YF_W_DW3, tid_f);
    YF_SYNC();
    YF_STAGE_END()
    YF_PRIO(1);
    YF_FETCH(1, W_f, L_f);
    YF_CONV1(W_f, L_f, 0);                                                                            // conv2d_1
    YF_SYNC(); YF_DUMP(B_T1, 8, T1)
    YF_STAGE_END()
    YF_PRIO(2);
    YF_FETCH(2, W_f, L_f);
    YF_DW(F, 1, B_T1, B_T2, 8, YF_L_LEAKY4, YF_W_DW3, W_f, L_f, 1, JT_DW3);                              // conv2d_3
    YF_SYNC(); YF_DUMP(B_T2, 8, T2)
    YF_STAGE_END()
    YF_PRIO(3);
    YF_FETCH(3, W_f, L_f);
    YF_DENSE(F, 1, 1, 8, B_T2, B_T3, 0, 4, EPI_RAW, 0, B_T3, YF_D_C5, no_add, W_f, L_f, 2);              // conv2d_5
    YF_SYNC(); YF_DUMP(B_T3, 4, T3)
    YF_STAGE_END()
    YF_PRIO(4);
    YF_HALO(B_T4, false, F, G4, H_T4, YF_W_DW10, tid_f);
    YF_FETCH(4, W_f, L_f);
    YF_DENSE(F, 5, 1, 4, B_T3, B_T4, 0, 18, EPI_LUT, YF_L_LEAKY7, B_T4, YF_D_C6, no_add, W_f, L_f, 3);   // conv2d_6: all five passes per job (25 jobs per two frames instead of 50; experimental bit 1024: three)
    YF_SYNC(); YF_DUMP(B_T4, 18, T4)
    YF_STAGE_END()
    YF_PRIO(5);
#if !(defined(YF_LAB) && defined(YF_WHATIF_NO_POOL8H))   // what-if (WRONG results): the horizontal pass and its barrier gone -- the bound for folding it into conv2d_6's epilogue
    pool8_h<F, NT>(frames, tid_f);                                                                   // pool_8 (h)
    YF_SYNC();
#endif
    YF_STAGE_END()
    YF_PRIO(6);
    pool8_v<F, NT, DUMP>(frames, tid_m);                                                       // pool_8 (v) + QUANTIZE#21
    YF_SYNC();                                    // T6 (written next) aliases HB (read by pool_8 v)
    YF_PRIO(7);
    YF_FETCH(5, W_m, L_m);
    YF_DW(F, 2, B_T4, B_T6, 18, YF_L_LEAKY11, YF_W_DW10, W_m, L_m, 4, JT_DW10);                          // conv2d_10
    YF_SYNC(); YF_DUMP(B_T14, 18, Q21) YF_DUMP(B_T14, 18, P8, YF_T14_CONV_BASE) YF_DUMP(B_T6, 18, T6)
    YF_STAGE_END()
    YF_PRIO(8);
    YF_FETCH(6, W_m, L_m);
    YF_DENSE(F, 1, 2, 16, B_T6, B_T7, 0, 6, EPI_RAW, 0, B_T7, YF_D_C12, no_add, W_m, L_m, 5);            // conv2d_12
    YF_SYNC(); YF_DUMP(B_T7, 6, T7)
    YF_STAGE_END()
    YF_PRIO(9);
    YF_HALO(B_T8, true, F, G8, H_T8, YF_W_DW15, tid_m);
    YF_FETCH(7, W_m, L_m);
    YF_DENSE(F, YF_TPJ13, 1, 8, B_T7, B_T8, 0, 36, EPI_LUT, YF_L_LEAKY14, B_T8, YF_D_C13, no_add, W_m, L_m, 6); // conv2d_13
    YF_SYNC(); YF_DUMP(B_T8, 36, T8)
    YF_STAGE_END()
    YF_PRIO(10);
    YF_FETCH(8, W_m, L_m);
    YF_DW(F, 1, B_T8, B_T9, 36, YF_L_LEAKY16, YF_W_DW15, W_m, L_m, 7, JT_DW15);                          // conv2d_15
    YF_SYNC(); YF_DUMP(B_T9, 36, T9)
    YF_STAGE_END()
    YF_PRIO(11);
    YF_FETCH(9, W_m, L_m);
    if constexpr (DUMP)   // debug builds: conv2d_17's own output is parked in the (still unwritten) conv half of concat_22 for the dump
      v2::dense2_stage<F, NW, 1, 3, 16, B_T9, B_T11, 0, 6, EPI_ADD, YF_A_ADD18, B_T7, 8, B_T14::OFF + YF_T14_CONV_BASE, B_T14::S>(frames, out_all, tab, addctx(YF_A_ADD18), W_m, L_m);
    else
    YF_DENSE(F, 1, 3, 16, B_T9, B_T11, 0, 6, EPI_ADD, YF_A_ADD18, B_T7, YF_D_C17, addctx(YF_A_ADD18), W_m, L_m, 8);   // conv2d_17 + eltwise_18
    YF_SYNC(); YF_DUMP(B_T11, 6, T11) YF_DUMP(B_T14, 6, C17, YF_T14_CONV_BASE)
    YF_STAGE_END()
    YF_PRIO(12);
    YF_FETCH(10, W_m, L_m);
    YF_DENSE(F, 3, 1, 8, B_T11, B_T14, YF_T14_CONV_BASE, 18, EPI_LUT, YF_L_LEAKY20, B_T14, YF_D_C19, no_add, W_m, L_m, 9);  // conv2d_19 -> concat_22
    YF_SYNC(); YF_DUMP(B_T14, 36, T14, 0, 18, 2)
    YF_STAGE_END()
    YF_PRIO(13);
    // BATCH: the parked group's T15 goes from the scratch straight into the odd tail sets by LDS-DMA (no registers), issued
    // here -- their bytes (T9/T11's old slots) are dead once conv2d_19 is through.  The wait that guards it is the explicit
    // s_waitcnt vmcnt(0) in front of the barrier behind conv2d_23 (this toolchain also waits at conv2d_23's first LDS access: its
    // alias analysis cannot tell the DMA's destination from the stage's buffers, so the transfer overlaps less than it could).
    // One wave-instruction moves 64 x 16 contiguous bytes.
    if constexpr (BATCH) {
      if (parked_first >= 0) {
        constexpr int PV = TailBufs<FRAME_BYTES>::T15_BYTES / 16, WI = (PV + 63) / 64;      // vectors / wave-instructions per frame
        const char* park = prm.scratch + (long)blockIdx.x * (F * PV * 16);
        for (int j = W_m; j < F * WI; j += NW) {
          const int f = j / WI, k0 = (j - f * WI) * 64;
          if (k0 + L_m < PV)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)(park + (f * PV + k0 + L_m) * 16),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(PRE + (2 * f + 1) * U::T15::FS + 16 * k0),
                                             16, 0, 0);
        }
      }
    }
    YF_HALO(B_T15, false, F, G15, H_T15, YF_W_DW27, tid_m);
    YF_FETCH(11, W_m, L_m);
    YF_DENSE(F, YF_TPJ23, 3, 16, B_T14, B_T15, 0, 24, EPI_LUT, YF_L_LEAKY24, B_T15, YF_D_C23, no_add, W_m, L_m, 10);   // conv2d_23
    if constexpr (BATCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the parked T15's LDS-DMA must have landed before the barrier that publishes the odd sets
    YF_SYNC(); YF_DUMP(B_T15, 24, T15)
    YF_STAGE_END()
    // ---- the 7x7 tail.  BATCH: it runs once per PAIR of groups on FT = 2F frames.  Its thirteen stages are latency chains
    // (98 pixels per group: one or two jobs per wave), so twice the jobs per stage cost far less than twice the time.  The
    // first group of a pair parks its T15 (5.4 KB per frame) in a per-workgroup HBM scratch and skips the tail; the second
    // group fetches it back into the odd tail sets -- set f of the tail sits at f * FRAME_BYTES / 2, so the even sets ARE the
    // arenas' own T15 -- and runs the tail for both.  A workgroup's last group runs the tail alone when it has no partner.
    long odd_first = -1;                          // first frame of the odd sets (the parked group), -1: none
    if constexpr (BATCH) {
      constexpr int V = U::T15_BYTES / 16;
      uint4* park = reinterpret_cast<uint4*>(prm.scratch) + (long)blockIdx.x * (F * V);
      if (parked_first < 0 && grp + gridDim.x < n_groups) {
        for (int i = tid_t; i < F * V; i += NT) {
          const int f = i / V, k = i - f * V;
          park[i] = *reinterpret_cast<const uint4*>(frames + f * FRAME_STRIDE + 16 * k);
        }
        parked_first = first;
        continue;
      }
      if (parked_first >= 0) {                  // its T15 is already in the odd sets (LDS-DMA issued before conv2d_23)
        odd_first = parked_first;
        parked_first = -1;
      }
    }
    // frame number of tail set f (BATCH: even sets = this group, odd sets = the parked one), -1 = nothing to write
    auto frame_of = [&](int f) -> long {
      long id = first + f;
      if constexpr (BATCH) id = (f & 1) ? (odd_first >= 0 ? odd_first + (f >> 1) : -1) : first + (f >> 1);
      return id < prm.n ? id : -1;
    };
#define YF_DUMP_T(BUF, C, OFF, ...) \
  if constexpr (DUMP) { if (prm.dump) { dump_buf<BUF, C, FT, NT>(frames, prm.dump, DS, DumpOffsets::OFF, first, prm.n, tid, ##__VA_ARGS__); YF_SYNC(); } }
    YF_PRIO(14);
    YF_FETCH(12, W_t, L_t);
    {   // pool_25 + QUANTIZE#45 on the first waves (by columns), conv2d_27 on the others: both only read T15
      constexpr int PW = v2::pool25_waves<FT>();
      static_assert(PW < NW, "waves left for conv2d_27");
      if (W_t < PW) v2::pool25_cols<FT, typename U::T15, typename U::T30, DUMP>(frames, W_t * 64 + L_t);
      else v2::dw2_stage<FT, NW - PW, 2, typename U::T15, typename U::T17, 24, YF_L_LEAKY28, 11, v2::JobTabs<F, BATCH>::JT_DW27>(frames, tab, W_t - PW, L_t);
    }
    YF_SYNC(); YF_DUMP_T(typename U::T30, 24, Q45) YF_DUMP_T(typename U::T30, 24, P25, 24) YF_DUMP_T(typename U::T17, 24, T17)
    YF_STAGE_END()
    YF_PRIO(15);
    YF_FETCH(13, W_t, L_t);
    YF_DENSE(FT, 1, 2, 16, typename U::T17, typename U::T18, 0, 8, EPI_RAW, 0, typename U::T18, YF_D_C29, no_add, W_t, L_t, 12);   // conv2d_29
    if constexpr (!BATCH) decode_prev(W_t, L_t);                                                    // previous group's boxes
    YF_SYNC(); YF_DUMP_T(typename U::T18, 8, T18)
    YF_STAGE_END()
    YF_PRIO(16);
    YF_HALO(typename U::T19, true, FT, G19, H_T19, YF_W_DW32, tid_t);
    YF_FETCH(14, W_t, L_t);
    YF_DENSE(FT, 5, 1, 8, typename U::T18, typename U::T19, 0, 40, EPI_LUT, YF_L_LEAKY31, typename U::T19, YF_D_C30, no_add, W_t, L_t, 13);  // conv2d_30
    YF_SYNC(); YF_DUMP_T(typename U::T19, 40, T19)
    YF_STAGE_END()
    YF_PRIO(17);
    YF_FETCH(15, W_t, L_t);
    YF_DW(FT, 1, typename U::T19, typename U::T20, 40, YF_L_LEAKY33, YF_W_DW32, W_t, L_t, 14, JT_DW32);    // conv2d_32
    YF_SYNC(); YF_DUMP_T(typename U::T20, 40, T20)
    YF_STAGE_END()
    YF_PRIO(18);
    YF_FETCH(16, W_t, L_t);
    if constexpr (DUMP)   // debug builds: conv2d_34's own output -> the conv half of concat_46 (written by conv2d_42 only)
      v2::dense2_stage<FT, NW, 1, 3, 16, typename U::T20, typename U::T22, 0, 8, EPI_ADD, YF_A_ADD35, typename U::T18, 15, U::T30::OFF + 24, U::T30::S>(frames, out_all, tab, addctx(YF_A_ADD35), W_t, L_t);
    else
    YF_DENSE(FT, 1, 3, 16, typename U::T20, typename U::T22, 0, 8, EPI_ADD, YF_A_ADD35, typename U::T18, YF_D_C34, addctx(YF_A_ADD35), W_t, L_t, 15);   // conv2d_34 + eltwise_35
    YF_SYNC(); YF_DUMP_T(typename U::T22, 8, T22) YF_DUMP_T(typename U::T30, 8, C34, 24)
    YF_STAGE_END()
    YF_PRIO(19);
    YF_HALO(typename U::T19, true, FT, G19, H_T19, YF_W_DW38, tid_t);
    YF_FETCH(17, W_t, L_t);
    YF_DENSE(FT, 5, 1, 8, typename U::T22, typename U::T19, 0, 40, EPI_LUT, YF_L_LEAKY37, typename U::T19, YF_D_C36, no_add, W_t, L_t, 16);  // conv2d_36
    YF_SYNC(); YF_DUMP_T(typename U::T19, 40, T23)
    YF_STAGE_END()
    YF_PRIO(20);
    YF_FETCH(18, W_t, L_t);
    YF_DW(FT, 1, typename U::T19, typename U::T20, 40, YF_L_LEAKY39, YF_W_DW38, W_t, L_t, 17, JT_DW32);    // conv2d_38
    YF_SYNC(); YF_DUMP_T(typename U::T20, 40, T24)
    YF_STAGE_END()
    YF_PRIO(21);
    YF_FETCH(19, W_t, L_t);
    if constexpr (DUMP)
      v2::dense2_stage<FT, NW, 1, 3, 16, typename U::T20, typename U::T26, 0, 8, EPI_ADD, YF_A_ADD41, typename U::T22, 18, U::T30::OFF + 24, U::T30::S>(frames, out_all, tab, addctx(YF_A_ADD41), W_t, L_t);
    else
    YF_DENSE(FT, 1, 3, 16, typename U::T20, typename U::T26, 0, 8, EPI_ADD, YF_A_ADD41, typename U::T22, YF_D_C40, addctx(YF_A_ADD41), W_t, L_t, 18);   // conv2d_40 + eltwise_41
    YF_SYNC(); YF_DUMP_T(typename U::T26, 8, T26) YF_DUMP_T(typename U::T30, 8, C40, 24)
    YF_STAGE_END()
    YF_PRIO(22);
    YF_FETCH(20, W_t, L_t);
    if constexpr (DUMP)   // debug builds: LEAKY_RELU #43's output (through the debug LUT) -> T20's slot, dead since conv2d_40
      v2::dense2_stage<FT, NW, 2, 1, 8, typename U::T26, typename U::T30, 24, 24, EPI_LUT, YF_L_L43Q44, typename U::T30, 19, U::T20::OFF, U::T20::S, DBG_LUT>(frames, out_all, tab, no_add, W_t, L_t);
    else
    YF_DENSE(FT, 3, 1, 8, typename U::T26, typename U::T30, 24, 24, EPI_LUT, YF_L_L43Q44, typename U::T30, YF_D_C42, no_add, W_t, L_t, 19);  // conv2d_42 -> concat_46
    YF_SYNC(); YF_DUMP_T(typename U::T30, 48, T30) YF_DUMP_T(typename U::T20, 24, L43)
    YF_STAGE_END()
    YF_PRIO(23);
    YF_HALO(typename U::T19, true, FT, G19, H_T19, YF_W_DW49, tid_t);
    YF_FETCH(21, W_t, L_t);
    YF_DENSE(FT, 2, 3, 16, typename U::T30, typename U::T19, 0, 40, EPI_LUT, YF_L_LEAKY48, typename U::T19, YF_D_C47, no_add, W_t, L_t, 20);   // conv2d_47
    YF_SYNC(); YF_DUMP_T(typename U::T19, 40, T31)
    YF_STAGE_END()
    YF_PRIO(24);
    YF_FETCH(22, W_t, L_t);
    YF_DW(FT, 1, typename U::T19, typename U::T20, 40, YF_L_LEAKY50, YF_W_DW49, W_t, L_t, 21, JT_DW32);    // conv2d_49
    YF_SYNC(); YF_DUMP_T(typename U::T20, 40, T32)
    YF_STAGE_END()
    YF_PRIO(25);
    YF_FETCH(23, W_t, L_t);
    // the decode's two look-up tables (2 KB) -> the first bytes of the frame arenas, dead since conv2d_47 (T15 / T30 of set 0), two stages ahead of
    // the decode: the barrier behind this stage waits for the transfer, the one behind conv2d_53 would do so on the critical path
    if (BATCH && prm.dets != nullptr && W_t < 2) {
      int dl = L_t;
      asm volatile("" : "+v"(dl));
      const uint8_t* src = reinterpret_cast<const uint8_t*>(W_t == 0 ? yfdec::d_sig_bits : yfdec::d_exp_bits) + 16 * dl;
      const uint32_t dst = (uint32_t)(PRE + OUT_ALL_BYTES + 1024 * W_t);
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    YF_DENSE(FT, 2, 3, 16, typename U::T20, typename U::T33, 0, 32, EPI_LUT, YF_L_LEAKY52, typename U::T33, YF_D_C51, no_add, W_t, L_t, 22);   // conv2d_51
    YF_SYNC(); YF_DUMP_T(typename U::T33, 32, T33)
    YF_STAGE_END()
    YF_PRIO(26);
    if constexpr (!BATCH) {
      YF_DENSE(FT, 1, 2, 16, typename U::T33, typename U::T33, 0, 18, EPI_HEAD, 0, typename U::T33, YF_D_C53, no_add, W_t, L_t, 23);   // conv2d_53
      YF_SYNC();
      // head: F*882 contiguous bytes -> HBM, 2-byte granules (882 is not a multiple of 4)
      const long valid = min((long)F, prm.n - first);
      const int n16 = (int)(valid * (OUT_FRAME_BYTES / 2));
      uint16_t* dst = reinterpret_cast<uint16_t*>(prm.out + first * OUT_FRAME_BYTES);
      const uint16_t* srcp = reinterpret_cast<const uint16_t*>(out_all);
      for (int i = tid; i < n16; i += NT) dst[i] = srcp[i];
      prev_first = first;
    } else {
      YF_DENSE(FT, 1, 2, 16, typename U::T33, typename U::HEAD, 0, 18, EPI_HEAD_LDS, 0, typename U::T33, YF_D_C53, no_add, W_t, L_t, 23);   // conv2d_53
      YF_SYNC();
      // heads: 882 bytes per frame from its set -> HBM, 2-byte granules; the boxes of set w are decoded by wave w meanwhile
      constexpr int H16 = OUT_FRAME_BYTES / 2;
      // with a decode the first FT waves decode (one frame each) while the other waves copy the heads; without one every wave copies
      const bool split = prm.dets != nullptr && FT < NW;
      const int c0 = split ? tid_t - FT * 64 : tid_t, cstep = split ? NT - FT * 64 : NT;
      if (!split || W_t >= FT) {
        for (int i = c0; i < FT * H16; i += cstep) {
          const int f = i / H16, k = i - f * H16;
          const long id = frame_of(f);
          if (id >= 0) reinterpret_cast<uint16_t*>(prm.out + id * OUT_FRAME_BYTES)[k] = *reinterpret_cast<const uint16_t*>(frames + f * U::HEAD::FS + U::HEAD::OFF + 2 * k);
        }
      }
      for (int f = W_t; f < FT; f += NW) {
        const long id = frame_of(f);
        if (prm.dets != nullptr && id >= 0) {
          int dl = L_t;
          asm volatile("" : "+v"(dl));
          yfdec::decode_frame_lds(reinterpret_cast<const int8_t*>(frames + f * U::HEAD::FS + U::HEAD::OFF), id, dl, prm.mode, prm.w_scale, prm.h_scale, prm.dets, prm.counts, prm.cap,
                                  (uint32_t)(PRE + OUT_ALL_BYTES), prm.q_thr);
        }
      }
    }
#undef YF_DUMP_T
  }
  if constexpr (!BATCH) {   // boxes of this workgroup's last group
    const int tid = tid0, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    decode_prev(wave, lane);
  }
#undef YF_HALO
#undef YF_FETCH
#undef YF_CONV1
#undef YF_DENSE
#undef YF_DW
#undef YF_DUMP
#undef YF_STAGE_END
#undef YF_PRIO
#undef YF_SYNC
}

template <int F, int NW, bool DUMP>
constexpr size_t lds_bytes() { return (size_t)v2::pre_bytes<F, tail_batch<DUMP>()>() + (tail_batch<DUMP>() ? 0 : (F * OUT_FRAME_BYTES + 15) & ~15) + (size_t)FRAME_BYTES + (size_t)(F - 1) * FRAME_STRIDE + (DUMP ? YF_DBG_LUT_BYTES : 0); }
template <bool DUMP>
constexpr size_t scratch_bytes_per_frame_slot() { return tail_batch<DUMP>() ? (size_t)TailBufs<FRAME_BYTES>::T15_BYTES : 0; }

#else   // YF_GENERIC
#ifdef YF_LAB
// ------------------------------------------------------------------------------------------------ layer-by-layer form
// For input sizes whose activations do not fit in LDS (160x160: conv2d_6's output alone is 131 KB) the SAME stage
// functions run one kernel per fused stage over a per-frame arena in HBM (one workgroup per frame and stage, frames
// grid-strided).  Results are bit-identical to the oracle at that size; HBM traffic is no longer the algorithmic
// minimum -- fusing this variant with spatial tiles is later work.
struct GenParams {
  const int8_t* in;       // [n][G0][G0][3]
  int8_t* out;            // [n][G3][G3][18]
  long n;                 // frames in this launch (<= arena capacity)
  const uint8_t* tab;
  char* arena;            // n * FRAME_BYTES bytes of HBM scratch
};
constexpr int GEN_STAGES = 27;

template <int ST, int NW>
__global__ void __launch_bounds__(NW * 64, 2) generic_stage_kernel(const GenParams prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64, F = 1;
  uint8_t* luts = reinterpret_cast<uint8_t*>(smem);      // addressed absolutely (host-checked: no static LDS)
  char* out_all = nullptr;
  int vz = 0;
  asm volatile("" : "+v"(vz));              // a zero the compiler cannot see through: keeps the pass constants' loads vector loads
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t* __restrict__ tab = prm.tab;
  for (int i = tid; i < LUT_BYTES / 16; i += NT)
    reinterpret_cast<uint4*>(luts)[i] = reinterpret_cast<const uint4*>(tab + PLAN.lut_off)[i];
  __syncthreads();
  const AddK no_add = {};
  auto addctx = [&](int k) {
    const uint8_t* a = tab + offsetof(yf_table_index, add) + k * sizeof(yf_add);
    return AddK{uniform_u32(a + offsetof(yf_add, mo2)), uniform_u32(a + offsetof(yf_add, zro)),
                (unsigned long)uniform_u32(a + offsetof(yf_add, c64o)) | ((unsigned long)uniform_u32(a + offsetof(yf_add, c64o) + 4) << 32),
                (int)uniform_u32(a + offsetof(yf_add, rso))};
  };
  for (long fr = blockIdx.x; fr < prm.n; fr += gridDim.x) {
    char* frames = prm.arena + fr * (long)FRAME_BYTES;
    out_all = reinterpret_cast<char*>(prm.out) + fr * (long)OUT_FRAME_BYTES;
    if constexpr (ST == 0) {
      stage_input<F, NT>(frames, prm.in, fr, prm.n, (int)uniform_u32(tab + offsetof(yf_table_index, in_zp)), tid);
      fill_halo<B_T1, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW3), tid);
    } else if constexpr (ST == 1) {
      conv1_stage<F, NW>(frames, tab, load_dense(tab, YF_D_CONV1), wave, lane, vz);
    } else if constexpr (ST == 2) {
      dw_mfma_stage<F, NW, 1, B_T1, B_T2, 8, YF_L_LEAKY4>(frames, tab, load_dw(tab, YF_W_DW3), wave, lane, vz);
    } else if constexpr (ST == 3) {
      dense_stage<F, NW, 1, 1, 8, B_T2, B_T3, 0, 4, EPI_RAW, 0, B_T3>(frames, out_all, tab, load_dense(tab, YF_D_C5), no_add, wave, lane, vz);
    } else if constexpr (ST == 4) {
      fill_halo<B_T4, false, F, NT>(frames, load_halo_zp(tab, YF_W_DW10), tid);
      dense_stage<F, NW, 3, 1, 4, B_T3, B_T4, 0, 18, EPI_LUT, YF_L_LEAKY7, B_T4>(frames, out_all, tab, load_dense(tab, YF_D_C6), no_add, wave, lane, vz);
    } else if constexpr (ST == 5) {
      pool8_h<F, NT>(frames, tid);
    } else if constexpr (ST == 6) {
      pool8_v<F, NT>(frames, tid);
    } else if constexpr (ST == 7) {
      dw_mfma_stage<F, NW, 2, B_T4, B_T6, 18, YF_L_LEAKY11>(frames, tab, load_dw(tab, YF_W_DW10), wave, lane, vz);
    } else if constexpr (ST == 8) {
      dense_stage<F, NW, 1, 2, 16, B_T6, B_T7, 0, 6, EPI_RAW, 0, B_T7>(frames, out_all, tab, load_dense(tab, YF_D_C12), no_add, wave, lane, vz);
    } else if constexpr (ST == 9) {
      fill_halo<B_T8, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW15), tid);
      dense_stage<F, NW, 3, 1, 8, B_T7, B_T8, 0, 36, EPI_LUT, YF_L_LEAKY14, B_T8>(frames, out_all, tab, load_dense(tab, YF_D_C13), no_add, wave, lane, vz);
    } else if constexpr (ST == 10) {
      dw_mfma_stage<F, NW, 1, B_T8, B_T9, 36, YF_L_LEAKY16>(frames, tab, load_dw(tab, YF_W_DW15), wave, lane, vz);
    } else if constexpr (ST == 11) {
      dense_stage<F, NW, 1, 3, 16, B_T9, B_T11, 0, 6, EPI_ADD, YF_A_ADD18, B_T7>(frames, out_all, tab, load_dense(tab, YF_D_C17), addctx(YF_A_ADD18), wave, lane, vz);
    } else if constexpr (ST == 12) {
      dense_stage<F, NW, 2, 1, 8, B_T11, B_T14, YF_T14_CONV_BASE, 18, EPI_LUT, YF_L_LEAKY20, B_T14>(frames, out_all, tab, load_dense(tab, YF_D_C19), no_add, wave, lane, vz);
    } else if constexpr (ST == 13) {
      fill_halo<B_T15, false, F, NT>(frames, load_halo_zp(tab, YF_W_DW27), tid);
      dense_stage<F, NW, 2, 3, 16, B_T14, B_T15, 0, 24, EPI_LUT, YF_L_LEAKY24, B_T15>(frames, out_all, tab, load_dense(tab, YF_D_C23), no_add, wave, lane, vz);
    } else if constexpr (ST == 14) {
      pool25<F, NT>(frames, tid);
      dw_mfma_stage<F, NW, 2, B_T15, B_T17, 24, YF_L_LEAKY28>(frames, tab, load_dw(tab, YF_W_DW27), wave, lane, vz);
    } else if constexpr (ST == 15) {
      dense_stage<F, NW, 1, 2, 16, B_T17, B_T18, 0, 8, EPI_RAW, 0, B_T18>(frames, out_all, tab, load_dense(tab, YF_D_C29), no_add, wave, lane, vz);
    } else if constexpr (ST == 16) {
      fill_halo<B_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW32), tid);
      dense_stage<F, NW, 3, 1, 8, B_T18, B_T19, 0, 40, EPI_LUT, YF_L_LEAKY31, B_T19>(frames, out_all, tab, load_dense(tab, YF_D_C30), no_add, wave, lane, vz);
    } else if constexpr (ST == 17) {
      dw_mfma_stage<F, NW, 1, B_T19, B_T20, 40, YF_L_LEAKY33>(frames, tab, load_dw(tab, YF_W_DW32), wave, lane, vz);
    } else if constexpr (ST == 18) {
      dense_stage<F, NW, 1, 3, 16, B_T20, B_T22, 0, 8, EPI_ADD, YF_A_ADD35, B_T18>(frames, out_all, tab, load_dense(tab, YF_D_C34), addctx(YF_A_ADD35), wave, lane, vz);
    } else if constexpr (ST == 19) {
      fill_halo<B_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW38), tid);
      dense_stage<F, NW, 3, 1, 8, B_T22, B_T19, 0, 40, EPI_LUT, YF_L_LEAKY37, B_T19>(frames, out_all, tab, load_dense(tab, YF_D_C36), no_add, wave, lane, vz);
    } else if constexpr (ST == 20) {
      dw_mfma_stage<F, NW, 1, B_T19, B_T20, 40, YF_L_LEAKY39>(frames, tab, load_dw(tab, YF_W_DW38), wave, lane, vz);
    } else if constexpr (ST == 21) {
      dense_stage<F, NW, 1, 3, 16, B_T20, B_T26, 0, 8, EPI_ADD, YF_A_ADD41, B_T22>(frames, out_all, tab, load_dense(tab, YF_D_C40), addctx(YF_A_ADD41), wave, lane, vz);
    } else if constexpr (ST == 22) {
      dense_stage<F, NW, 2, 1, 8, B_T26, B_T30, 24, 24, EPI_LUT, YF_L_L43Q44, B_T30>(frames, out_all, tab, load_dense(tab, YF_D_C42), no_add, wave, lane, vz);
    } else if constexpr (ST == 23) {
      fill_halo<B_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW49), tid);
      dense_stage<F, NW, 2, 3, 16, B_T30, B_T19, 0, 40, EPI_LUT, YF_L_LEAKY48, B_T19>(frames, out_all, tab, load_dense(tab, YF_D_C47), no_add, wave, lane, vz);
    } else if constexpr (ST == 24) {
      dw_mfma_stage<F, NW, 1, B_T19, B_T20, 40, YF_L_LEAKY50>(frames, tab, load_dw(tab, YF_W_DW49), wave, lane, vz);
    } else if constexpr (ST == 25) {
      dense_stage<F, NW, 2, 3, 16, B_T20, B_T33, 0, 32, EPI_LUT, YF_L_LEAKY52, B_T33>(frames, out_all, tab, load_dense(tab, YF_D_C51), no_add, wave, lane, vz);
    } else {
      static_assert(ST == 26, "stage index");
      dense_stage<F, NW, 1, 2, 16, B_T33, B_T33, 0, 18, EPI_HEAD, 0, B_T33>(frames, out_all, tab, load_dense(tab, YF_D_C53), no_add, wave, lane, vz);
    }
  }
}

#endif   // YF_LAB

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
  uint32_t pre[PER][3];
  // input rows of a band: local row l <-> input row 2(a-1)+l-1, out of range = zero point
  auto fetch = [&](long job) {
    const long fr = job / K1_BANDS;
    const int a = (int)(job - fr * K1_BANDS) * K1_BH;
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
    const long fr = job / K1_BANDS;
    const int a = (int)(job - fr * K1_BANDS) * K1_BH;              // first T4 row of the band
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
    store_rows<NT>(arena + A_T4 + (a + 1) * T4_ROW, frames + L1_T4::OFF, K1_BH * T4_ROW, tid);              // halo'd rows a+1 ..
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
typedef Buf<K23_RH,                              G2, K23_NM, 32, G2,     0, 0> L23_T6;       // aliases HB (dead after the vertical pool pass)
typedef Buf<K23_RH + K23_NM * G2 * 32,           G2, K23_NM,  8, G2,     0, 0> L23_T7;       // rows p0-1 .. p0+8
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
static_assert(K23_NM * G2 * 32 + K23_NM * G2 * 8 <= K23_RH_BYTES && K23_NM * K23_T8_ROW + K23_BP * G2 * 48 <= K23_RA_BYTES, "aliases fit");
static_assert(K23_BP * G2 * 8 + K23_BP * T15_ROW <= K23_NM * G2 * 32, "T11 | T15 fit T6's bytes (T7 behind them stays alive until conv2d_17)");
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
  Prefetch<NT, K23_NR * T4_ROW / 16> pre;
  // T4 halo'd rows [2p0-2, 2p0-2+NR) that exist (0 .. G1): contiguous in the arena
  auto range = [&](long job, const char*& src, int& lo_local, int& n16) {
    const long fr = job / K23_BANDS;
    const int p0 = (int)(job - fr * K23_BANDS) * K23_BP;
    const int h0 = 2 * p0 - 2, lo = max(h0, 0), hi = min(h0 + K23_NR, G1 + 1);
    src = prm.arena + fr * (long)ARENA_BYTES + A_T4 + lo * T4_ROW;
    lo_local = lo - h0; n16 = (hi - lo) * (T4_ROW / 16);
  };
  long job = blockIdx.x;
  if (job < jobs) { const char* src; int ll, n16; range(job, src, ll, n16); pf_fetch(pre, src, n16, tid); }
  for (; job < jobs; job += gridDim.x) {
    const long fr = job / K23_BANDS;
    const int p0 = (int)(job - fr * K23_BANDS) * K23_BP;           // first 40x40 row of the band
    char* arena = prm.arena + fr * (long)ARENA_BYTES;
    lds_barrier();
    YF_BAND_PRIO(3);
    { const char* src; int ll, n16; range(job, src, ll, n16); pf_commit(pre, frames + L23_T4::OFF + ll * T4_ROW, n16, tid); }
    lds_barrier();
    if (job + gridDim.x < jobs) { const char* src; int ll, n16; range(job + gridDim.x, src, ll, n16); pf_fetch(pre, src, n16, tid); }
    {   // pool_8 horizontal pass over every band row (rows outside the image are never read back)
      constexpr int NO = 5, NCH = G2 / NO;
      static_assert(G2 % NO == 0, "sweeps of 5 outputs");
      for (int i = tid; i < K23_NR * NCH * 5; i += NT) {
        const int cg = i % 5; int t = i / 5;
        const int k = t % NCH; const int l = t / NCH;
        const char* row = frames + L23_T4::OFF + l * T4_ROW + 20 + 4 * cg;                // pixel 0 sits behind the halo column
        char* dst = frames + L23_HB::OFF + l * (G2 * 20) + 4 * cg;
        pool8_sweep<NO, G1 - 1>(k * NO, [&](int x) { return lds_u32(row + x * 20); },
                                [&](int ox, const SplitB& v) { *reinterpret_cast<uint32_t*>(dst + ox * 20) = v.merge(); });
      }
    }
    lds_barrier();
    {   // vertical pass + QUANTIZE#21 straight into the pooled half of the band's concat_22 rows
      constexpr int NO = 4, NSW = K23_BP / NO;
      for (int i = tid; i < NSW * G2 * 5; i += NT) {
        const int cg = i % 5; int t = i / 5;
        const int ox = t % G2; const int sw = t / G2;
        const char* col = frames + L23_HB::OFF + ox * 20 + 4 * cg;
        char* dst = frames + L23_T14::OFF + ox * 48 + 4 * cg;
        pool8_sweep<NO, G1 - 1>(p0 + sw * NO, [&](int r) { return lds_u32(col + (r - (2 * p0 - 3)) * (G2 * 20)); },
                                [&](int oy, const SplitB& v) { *reinterpret_cast<uint32_t*>(dst + (oy - p0) * (G2 * 48)) = lut4_raw<YF_L_Q21>(v); });
      }
    }
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
#endif  // YF_GENERIC

}  // namespace YF_NS
#pragma clang diagnostic pop
