// fp16 configuration (BASELINE configs[3]): the reference's fp32 ONNX export (yoloface/pytorch/yoloface-50k.onnx; the same
// graph as yoloface/pytorch/yoloface.py:83-119) with fp16 weights and activations and fp32 accumulation, as ONE fused
// kernel on the LDS-resident plan of the int8 engine: a workgroup of 8 waves walks one frame through all 31 layers, every
// activation stays in LDS as fp16 (76 KB per frame -> two workgroups per CU), HBM is touched for the 18.8 KB fp16 frame and
// the 3.5 KB of fp32 head logits.
//
//   every conv (3x3 dense, 3x3 depthwise, 1x1): v_mfma_f32_16x16x32_f16 in the LANE-PRIVATE form of the int8 engine -- lane
//     group g supplies the k-slots 8g..8g+7 from ITS OWN pixel (16 bytes: 8 channels of a 1x1 conv, or 2 taps x 4 channels
//     of a 3x3 one), rows 4g..4g+3 of the A operand are non-zero only in those slots, so D[4g+j][c] is pixel (g,c)'s dot
//     product with output channel j: 64 pixels x 4 channels per MFMA, every lane owns one pixel, no lane idles when Cout is
//     not a multiple of 16 (4, 6, 8, 18, 24, 40)
//   epilogue: bias rides in the accumulator's initial value; LeakyReLU = max(x, 0.1 x); two v_cvt_pk_f16_f32; one 8-byte
//     LDS store (residual add: the stored fp16 operand is added in fp32 first)
//   max-pools: v_pk_max_f16 on channel pairs, separable 8x8, clamped coordinates (padding never wins)
// Tolerance-checked against an fp32 numpy evaluation of the same graph (tests/test_gpu_parity.py, atol/rtol 2e-2).  gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "yf_fp16.h"
#include "yf_stream_scratch.h"

#ifndef YF16_NW
#define YF16_NW 8          /* waves per workgroup (one frame per workgroup, two workgroups per CU) */
#endif
namespace yf16 {

typedef _Float16 half_t;
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ LDS plan (bytes, one frame)
// Buf: OFF byte offset, logical W x H, S bytes per pixel (fp16 channels, padded), RS pixels per row incl. halo, PT/PL halo
// rows / columns in front of logical pixel (0,0).  Halos hold 0 (the ONNX graph pads with zeros).  Buffers alias by lifetime.
// FS: bytes between consecutive frames of a stage (only the 7x7 tail runs on two frames: tail batching, see the kernel).
// SK: extra bytes per row (row pitch ROWB = RS * S + SK).  A depthwise tap read (ds_read_b64, 64 banks) puts the 16 lanes of a tile row on
// the banks 4k and 4k+1 when the pixel is 16 or 80 bytes; with a row pitch that is a multiple of 4 dwords the next tile row lands on the
// same banks.  Eight bytes of skew move it to 4k+2 / 4k+3 (T1, T8, T19: the inputs of conv2d_3, conv2d_15 and conv2d_32 / 38 / 49).
#ifndef YF16_ROW_SKEW
#define YF16_ROW_SKEW 8
#endif
template <int OFF_, int W_, int H_, int S_, int RS_, int PT_, int PL_, int FS_ = 0, int SK_ = 0>
struct Buf {
  static constexpr int OFF = OFF_, W = W_, H = H_, S = S_, RS = RS_, PT = PT_, PL = PL_, FS = FS_, SK = SK_;
  static constexpr int P = W_ * H_, ROWB = RS_ * S_ + SK_;
  __device__ static __forceinline__ int at(int y, int x) { return OFF_ + (y + PT_) * ROWB + (x + PL_) * S_; }
  __device__ static __forceinline__ int at_p(int p) {
    if constexpr (RS_ == W_ && PT_ == 0 && PL_ == 0) return OFF_ + p * S_;
    else { const int y = p / W_; return at(y, p - y * W_); }
  }
};
//              OFF    W   H   S  RS PT PL
// The input frame in COLUMN-PARITY PLANES: halo'd column hx = x + 1 of a row sits in plane hx & 1 at index hx >> 1 (plane 0: 29 pixels from byte 0, plane 1:
// 28 pixels from byte IN_PLANE1), so the stride-2 taps of conv2d_1 -- tap kx of output column ox is halo'd column 2 ox + kx = {plane 0 [ox], plane 1 [ox],
// plane 0 [ox + 1]} -- are reads of CONSECUTIVE 8-byte pixels across a tile row (16 lanes on 32 banks) instead of every other pixel (16 lanes on the 16 bank
// pairs 4k, 4k+1, the next tile row on the same ones: 5.1 pipe cycles per tap read in the per-stage counters, 444 of the stage's 1128 LDS cycles conflicts).
// Rows are 480 bytes apart and conv2d_1's tile rows go to the lane groups in the order 0, 2, 1, 3: the two rows of a 32-lane half are then 2 x 960 bytes =
// 32 banks (mod 64) apart, and every tap read touches each bank once.  at() does not apply to this buffer.
typedef Buf<     0, 56, 56,  8, 58, 1, 1, 0, 16> B_IN;    // RGBX fp16, top halo row, halo column = plane 0 [0]; 57 rows of 480 bytes
constexpr int IN_PLANE1 = 29 * 8;
typedef Buf< 27360, 28, 28, 16, 30, 1, 1, 0, YF16_ROW_SKEW> B_T1;    // conv1 out, 8 ch, halo ring; 30 rows of 488 bytes
typedef Buf<     0, 28, 28, 16, 28, 0, 0> B_T2;    // dw3 out, 8 ch (on the dead input frame: conv2d_5 -> conv2d_6 reads it while it writes T4)
typedef Buf< 12544, 28, 28, 40, 29, 1, 1> B_T4;    // c6 out, 18 ch (stride 20), top/left halo for dw10
typedef Buf< 46184, 14, 28, 36, 14, 0, 0> B_HB;    // pool_8 horizontal pass, 18 ch
typedef Buf< 60304, 14, 14, 80, 14, 0, 0> B_T14;   // concat: pool [0,18) | conv [20,38) (8-byte aligned start), stride 40; 16-byte aligned pixels (conv2d_23 reads them with ds_read_b128)
typedef Buf<     0, 14, 14, 48, 14, 0, 0> B_T6;    // dw10 out, 18 ch (stride 24)
typedef Buf<  9408, 14, 14, 16, 14, 0, 0> B_T7;    // c12 out, 6 ch
typedef Buf< 12544, 14, 14, 80, 16, 1, 1, 0, YF16_ROW_SKEW> B_T8;    // c13 out, 36 ch, halo ring; 16 rows of 1288 bytes
typedef Buf< 33152, 14, 14, 80, 14, 0, 0> B_T9;    // dw15 out
typedef Buf< 48832, 14, 14, 16, 14, 0, 0> B_T11;   // c17 + add out, 6 ch
typedef Buf<     0, 14, 14, 48, 15, 1, 1, 0, YF16_ROW_SKEW> B_T15;   // c23 out, 24 ch, top/left halo; 15 rows of 728 bytes
constexpr int LDS_BYTES = 76032;                    // end of B_T14, rounded up to 64
constexpr int ZSLACK = 75984;                       // the 48 bytes between the end of B_T14 and LDS_BYTES: never written after the arena clear, i.e. always zero
static_assert(ZSLACK == B_T14::OFF + 14 * 14 * 80 && ZSLACK + 48 <= LDS_BYTES && ZSLACK % 16 == 0, "zero spot of the dense stages with at most three k-steps");
static_assert(B_T8::OFF + 16 * B_T8::ROWB <= B_T9::OFF && B_T9::OFF + 196 * 80 <= B_T11::OFF && B_T11::OFF + 196 * 16 <= B_T14::OFF && B_T15::OFF + 15 * B_T15::ROWB <= B_T8::OFF, "skewed buffers do not run into their neighbours");
static_assert(B_T14::OFF + 14 * 14 * 80 <= LDS_BYTES && B_HB::OFF + 28 * 14 * 36 <= B_T14::OFF && B_T4::OFF + 29 * 29 * 40 <= B_HB::OFF, "plan");
static_assert(B_IN::OFF + 57 * B_IN::ROWB <= B_T1::OFF && B_IN::ROWB == 480 && B_T1::OFF % 16 == 0 && B_T1::OFF + 30 * B_T1::ROWB <= B_HB::OFF && B_T2::OFF + 28 * 28 * 16 <= B_T4::OFF && B_T2::OFF + 28 * 28 * 16 <= B_T1::OFF, "plan: conv2d_3 reads T1 and writes T2; conv2d_5 -> conv2d_6 reads T2 and writes T4");

// ---- weight ring.  A conv's A-operand rows come from LDS, not from global memory: stage k's first act is ONE LDS-DMA of stage
// k+1's rows (global_load_lds_dwordx4: 64 x 16 bytes per wave-instruction, no registers), so that the next stage's waves read
// their fragments with ds_read_b128 (~100 cycles) instead of waiting 1.5-2.5 k cycles for a global load behind every barrier
// and every channel-group switch (what-if without those loads: -9 % kernel time at one frame per workgroup).  Blocks k and k+1
// are the only ones alive together, so even blocks grow up from the bottom of a 5.9 KB region behind the arena and odd blocks
// down from its top.
// (a dense 1x1 block carries its fp32 biases behind its rows: 16 bytes per pass of four output channels; the 3x3 stages read theirs with scalar loads)
constexpr int WBYTES[24] = {640, 640, 80, 400, 1600, 416, 720, 2880, 672, 400, 2016, 1920, 416, 800, 3200, 672, 800, 3200, 672, 480, 4000, 3200, 2688, 1360};
constexpr int RING0 = LDS_BYTES, LDS_TOTAL = 81920, RING_BYTES = LDS_TOTAL - LDS_BYTES;
constexpr int COUT_[24] = {8, 8, 4, 18, 18, 6, 36, 36, 6, 18, 24, 24, 8, 40, 40, 8, 40, 40, 8, 24, 40, 40, 32, 18};
constexpr bool IS_3X3_[24] = {1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 1, 0, 0};
// The ring serves the FRONT stages (convs 0..11; the tail's blocks are resident during the tail phase).  A UNIT is what one stage needs: one block, or
// the two blocks of the fused 1x1 pair conv2d_5 -> conv2d_6 (adjacent in the blob: ONE DMA).  Units k and k+1 are the only ones alive together, so even units grow up from the
// bottom of the region and odd units down from its top.
constexpr int UNIT_OF[12] = {0, 1, 2, 2, 3, 4, 5, 6, 7, 8, 9, 10};
constexpr int NUNITS = 11;
constexpr int unit_first(int u) { for (int k = 0; k < 12; ++k) if (UNIT_OF[k] == u) return k; return -1; }
constexpr int unit_bytes(int u) { int b = 0; for (int k = 0; k < 12; ++k) if (UNIT_OF[k] == u) b += WBYTES[k] + (IS_3X3_[k] ? ((COUT_[k] + 3) / 4) * 16 : 0); return b; }
constexpr int unit_base(int u) { return u % 2 == 0 ? RING0 : RING0 + RING_BYTES - unit_bytes(u); }
constexpr int woff(int k) { int o = unit_base(UNIT_OF[k]); for (int i = unit_first(UNIT_OF[k]); i < k; ++i) o += WBYTES[i] + (IS_3X3_[i] ? ((COUT_[i] + 3) / 4) * 16 : 0); return o; }
constexpr bool ring_ok() {
  for (int u = 0; u < NUNITS; ++u) {
    if (unit_bytes(u) % 16 != 0) return false;
    if (u + 1 < NUNITS && unit_bytes(u) + unit_bytes(u + 1) > RING_BYTES) return false;
  }
  return RING0 % 16 == 0 && RING_BYTES % 16 == 0;
}
static_assert(ring_ok(), "adjacent units fit the ring side by side");
// LDS-DMA of ring unit U (rows and biases of its convs, contiguous in the blob at byte w_off) into its place in the ring: a wave moves 1 KB per
// instruction (64 x 16 bytes, no registers).  The compiler does not see the transfer (inline assembly): the barrier behind every stage is preceded by
// an explicit s_waitcnt vmcnt(0) (SYNC in the kernel).  The LAST waves of the workgroup issue it: the job split gives a stage's surplus jobs to the first ones.
template <int U, int NW>
__device__ __forceinline__ void fetch_unit(const uint8_t* __restrict__ tab, uint32_t w_off, int wave, int lane) {
  constexpr int BYTES = unit_bytes(U), NCHUNK = (BYTES + 1023) / 1024;
  static_assert(NCHUNK <= NW, "one DMA instruction per wave");
  const int ch = NW - 1 - wave;
  if (ch < NCHUNK) {
    const int off = ch * 1024 + lane * 16;
    if (off < BYTES) {
      const uint8_t* src = tab + w_off + off;
      const uint32_t dst = (uint32_t)(unit_base(U) + ch * 1024);
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  }
}

enum { EPI_ACT = 0, EPI_LINEAR = 1, EPI_ADD = 2, EPI_HEAD = 3 };

// ------------------------------------------------------------------------------------------------ table blob
// Per conv: f16 A-operand rows and f32 biases, laid out for the lane-private form (built by yf_fp16_create):
//   dense 1x1 (and conv1): rows [cout_pad4][KS*8] f16                   (KS k-steps of 8 input channels / 2 taps)
//   depthwise: per 4-channel group [5 k-steps][4 rows][8] f16           (row j: w[tap 2ks][j] at slot j, w[tap 2ks+1][j] at slot 4+j)
//   bias [cout_pad4] f32 behind the rows (16-byte aligned)
struct ConvT { uint32_t w_off, b_off; };
struct Tables { ConvT conv[24]; };
// The blob's layout is fixed by the graph, so the kernel compiles it in (no descriptor load in front of a stage's weight DMA or bias loads):
// [Tables][conv 0 rows | biases][conv 1 rows | biases] ...; every piece is a multiple of 16 bytes.  yf_fp16_create builds the blob and checks
// that it arrives at the same offsets.
#define COUT COUT_
#define IS_3X3 IS_3X3_
constexpr int bias_bytes(int k) { return ((COUT[k] + 3) / 4) * 16; }
constexpr int rows_bytes(int k) { return IS_3X3[k] ? WBYTES[k] : WBYTES[k] - bias_bytes(k); }
constexpr ConvT conv_at(int k) {
  uint32_t off = sizeof(Tables);
  for (int i = 0; i < k; ++i) off += rows_bytes(i) + bias_bytes(i);
  return ConvT{off, off + (uint32_t)rows_bytes(k)};
}

__device__ __forceinline__ uint32_t lds_u32(const char* p) { return *reinterpret_cast<const uint32_t*>(p); }
__device__ __forceinline__ uint2 lds_u64(const char* p) { return *reinterpret_cast<const uint2*>(p); }
// A depthwise tap: eight bytes read as ONE ds_read_b64.  Left to itself the compiler merges two taps into a ds_read2_b64, and that
// instruction runs at HALF the pipe rate of two ds_read_b64 (tools/probe/lds_banks.hip: 8.0 cycles against 2 x 2.2 -- the 256-byte-per-
// clock mode of 64-bit reads is lost); a volatile access is not merged and still gets its s_waitcnt from the compiler.
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 lds_tap64(const char* p) {
  const v2u_t v = *(const volatile __attribute__((address_space(3))) v2u_t*)p;      // explicit LDS address space: a volatile access through a generic pointer becomes a flat load
  return uint2{v.x, v.y};
}
typedef const __attribute__((address_space(4))) v4f* cv4f_ptr;
__device__ __forceinline__ v4f uniform_f4(const void* p) { return *(cv4f_ptr)(uintptr_t)p; }

// absolute LDS addresses (the kernel has no static LDS: the dynamic segment starts at 0; `lds + offset` costs a v_add of the segment base per address)
typedef __attribute__((address_space(3))) v4i* lds_v4i_p;
typedef __attribute__((address_space(3))) v2u_t* lds_v2u_p;
__device__ __forceinline__ v4i ld128(int a) { return *(const lds_v4i_p)(uintptr_t)(uint32_t)a; }
__device__ __forceinline__ uint2 ld64(int a) { const v2u_t v = *(const lds_v2u_p)(uintptr_t)(uint32_t)a; return uint2{v.x, v.y}; }
__device__ __forceinline__ void st64(int a, uint2 v) { *(lds_v2u_p)(uintptr_t)(uint32_t)a = v2u_t{v.x, v.y}; }
template <int JOBS, int NW>
__device__ __forceinline__ void job_range(int wave, int& j0, int& j1) {
  constexpr int BASE = JOBS / NW, REM = JOBS % NW;
  j0 = wave * BASE + min(wave, REM);
  j1 = j0 + BASE + (wave < REM ? 1 : 0);
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float a, float b) {          // v_cvt_pk_f16_f32 (gfx950): round to nearest even, like a
  const v2h h = __builtin_convertvector(v2f{a, b}, v2h);               // plain cast; the round-toward-zero form (v_cvt_pkrtz) biases
  uint32_t u; __builtin_memcpy(&u, &h, 4); return u;                   // every layer the same way and misses the tolerance
}
// LeakyReLU of two channels, then one v_cvt_pk_f16_f32.  (On the packed halfs AFTER the conversion -- v_pk_mul_f16 by 0.1, v_pk_max_f16:
// three instructions per pair instead of five -- the extra fp16 rounding of 0.1 h misses the tolerance and the kernel is 0.5 % faster.)
// 0.1 x for both channels as ONE v_pk_mul_f32 (same rate as a single v_mul_f32), then the two v_max_f32 and the conversion: four VALU instructions
// per channel pair instead of five, results bit for bit the same
__device__ __forceinline__ uint32_t leaky_pack2(float a, float b) {
  const v2f x = {a, b};
  const v2f t = x * 0.1f;                 // compiler-selected (it pads the MFMA -> VALU hazard only for instructions it selects itself); the inline v_max reads t, so it stays behind
  float r0, r1;
  asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(x[0]), "v"(t[0]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(x[1]), "v"(t[1]));
  return pack2(r0, r1);
}
// ------------------------------------------------------------------------------------------------ dense 1x1, lane-private
// KS k-steps of 8 input channels (16 bytes of the pixel's fp16 vector each); TPJ passes of 4 output channels share a job's B fragments; the bias is
// the accumulator's initial value.  A job's 64 lanes are a TILE:
//   28x28 / 14x14: 64 CONSECUTIVE pixels p0 + lane (the inputs of every dense stage are halo-free: consecutive pixels are consecutive bytes, and sixteen
//     consecutive 16-, 48-, 80- or 112-byte pixels cover the 64 banks of a ds_read_b128 lane group exactly once); the last tile starts at P - 64 and redoes
//     some pixels of the one before (same values, same addresses).  A lane's input offset is a per-stage constant plus a wave-uniform (scalar) tile offset;
//     so is its output offset when the output has no halo, and with a halo it is one compare-and-select more (the lane's row / column inside the first
//     tile are per-stage constants, a tile adds a scalar row / column and at most one carry into the next row): one to three VALU instructions per job
//     (round 3: ~14 of pixel arithmetic -- a min, two divisions by multiplication, two multiply-adds)
// A fragments: the lanes whose fragment is all zero (48 of 64) read it too -- from ZB, a spot of the arena that holds zeros while the stage runs (the
// kernel names one per stage) -- so a chunk set-up is ONE address instruction and KS unmasked ds_read_b128 per pass instead of four zero moves, an exec
// mask and a masked read per fragment.  The biases come with the block (LDS-DMA), are read into VGPRs once per chunk with one broadcast ds_read_b128 per
// pass and enter the MFMA as its C operand: no scalar load, no move per pass and job.
constexpr int dense_rows_bytes(int cout, int ks) { return ((cout + 3) / 4) * 4 * 8 * ks * 2; }
template <int K, int NW, int TPJ, int KS, class IN, class OUT, int OUT_CH0, int COUT, int EPI, class ADDB, int ZB>
__device__ __forceinline__ void dense_tile_stage(int wave, int lane) {
  constexpr int NP = (COUT + 3) / 4, NCH = (NP + TPJ - 1) / TPJ, KROW = 8 * KS;
  constexpr int W = IN::W, P = W * W, NT = (P + 63) / 64, JOBS = NCH * NT;
  constexpr bool OUT_LINEAR = (OUT::RS == W && OUT::PT == 0 && OUT::PL == 0 && OUT::SK == 0);
  constexpr int ROWS = NP * 4 * KROW * 2;                     // the block: A-operand rows, then the fp32 biases (16 bytes per pass)
  static_assert(WBYTES[K] == ROWS + NP * 16, "stage and weight block agree");
  static_assert(IN::S >= 16 * KS, "the pixel vector must cover every k-step");
  static_assert(IN::OFF % 16 == 0 && IN::S % 16 == 0 && IN::FS % 16 == 0, "B fragments are aligned ds_read_b128 (a misaligned one is several times slower)");
  static_assert(IN::RS == W && IN::PT == 0 && IN::PL == 0 && IN::SK == 0 && (EPI != EPI_ADD || (ADDB::RS == W && ADDB::PT == 0 && ADDB::PL == 0 && ADDB::SK == 0)), "inputs are halo-free");
  static_assert((W == 28 || W == 14) && IN::H == W && OUT::W == W && (EPI != EPI_ADD || ADDB::W == W) && EPI != EPI_HEAD, "the front stages' grids (the 7x7 layers and the head run in tail_chain)");
  static_assert(ZB % 16 == 0 && ZB >= 0, "zero fragments are aligned reads too");
  const int g = lane >> 4, c = lane & 15;
  const int pl = lane;                                       // the lane's pixel inside the first tile
  const int ly = (W == 14) ? (pl * 74) >> 10 : (pl * 37) >> 10;     // pl / W for pl < 64 (14: 74/1024, 28: 37/1024; checked exhaustively)
  const int lx = pl - W * ly;
  const int in_lane = IN::OFF + pl * IN::S, add_lane = ADDB::OFF + pl * ADDB::S;
  const int out_lane = OUT::at(ly, lx) + 2 * OUT_CH0;
  // tile t of a 28x28 / 14x14 stage starts at pixel p0 = min(64 t, P - 64) = (Y, X): output row Y + ly, column X + lx, minus one row's worth of columns on a carry
  auto tile_p0 = [&](int tile) { return min(64 * tile, P - 64); };
  const bool a_on = (c >> 2) == g;
  const int fr_base = a_on ? woff(K) + (c & 3) * (KROW * 2) : ZB;
  int fr_scale = a_on ? 1 : 0;
  asm("" : "+v"(fr_scale));                 // opaque: the fragment address stays ONE multiply-add (otherwise the compiler selects between two sums)
  // The chunk loop is unrolled at compile time: a chunk's pass numbers, fragment and bias offsets and the `pass exists` tests are constants of its copy
  // (as a run-time loop every chunk set-up spent ~35 scalar instructions on them, and every pass of a job a scalar branch)
  int j0, j1;
  job_range<JOBS, NW>(wave, j0, j1);
#pragma unroll
  for (int chunk = 0; chunk < NCH; ++chunk) {
    int j = max(j0, chunk * NT);
    const int jend = min(j1, (chunk + 1) * NT);
    if (j >= jend) continue;
    v4i a[TPJ][KS];
    v4f bias[TPJ];
#pragma unroll
    for (int tt = 0; tt < TPJ; ++tt) {
      const int ps = min(chunk * TPJ + tt, NP - 1);
      const int fa = fr_base + __mul24(fr_scale, ps * (4 * KROW * 2));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[tt][ks] = ld128(fa + 16 * ks);
      bias[tt] = __builtin_bit_cast(v4f, ld128(woff(K) + ROWS + 16 * ps));        // one address for all lanes
    }
    for (; j < jend; ++j) {
      const int tile = j - chunk * NT;
      int src, dst, addp;
      {
        const int p0 = tile_p0(tile);
        src = in_lane + p0 * IN::S;
        addp = add_lane + (p0 * ADDB::S + 8 * TPJ * chunk);
        if constexpr (OUT_LINEAR) dst = out_lane + (p0 * OUT::S + 8 * TPJ * chunk);
        else {
          const int Y = (W == 14) ? (p0 * 4682) >> 16 : (p0 * 2341) >> 16;          // p0 / W for p0 <= P - 64 (scalar)
          const int X = p0 - W * Y;
          dst = out_lane + (Y * OUT::ROWB + X * OUT::S + 8 * TPJ * chunk) + (lx >= W - X ? OUT::ROWB - W * OUT::S : 0);
        }
      }
      v4i b[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) b[ks] = ld128(src + 16 * ks);
      {
#pragma unroll
        for (int tt = 0; tt < TPJ; ++tt) {
          if (chunk * TPJ + tt < NP) {
            v4f acc = bias[tt];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a[tt][ks]), __builtin_bit_cast(v8h, b[ks]), acc, 0, 0, 0);
            if constexpr (EPI == EPI_ADD) {
              const uint2 r = ld64(addp + 8 * tt);
              v2h r0, r1; __builtin_memcpy(&r0, &r.x, 4); __builtin_memcpy(&r1, &r.y, 4);
              acc[0] += (float)r0[0]; acc[1] += (float)r0[1]; acc[2] += (float)r1[0]; acc[3] += (float)r1[1];
            }
            uint2 v;
            if constexpr (EPI == EPI_ACT) { v.x = leaky_pack2(acc[0], acc[1]); v.y = leaky_pack2(acc[2], acc[3]); }
            else { v.x = pack2(acc[0], acc[1]); v.y = pack2(acc[2], acc[3]); }
            st64(dst + 8 * tt, v);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ two 1x1 layers in one stage
// Layer A's four (or eight) output channels of a lane's pixel land in the lane's own accumulators: packed to fp16 they ARE layer B's B operand (KS = 1:
// at most 8 input channels).  conv2d_5 -> conv2d_6 run as ONE stage that way: conv2d_5's output makes no LDS round trip and one barrier-separated
// latency chain of ~1.3 k cycles per frame is gone (the two other candidates, conv2d_12 -> 13 and conv2d_17 -> 19, have two chunks of layer B's passes per
// tile -- layer A would run twice -- and measured slower).  A job = (chunk of layer B's passes, tile); STA optionally stores A's output from the jobs of
// chunk 0.  Arithmetic and its order are those of the separate stages.
struct NoBuf { static constexpr int OFF = 0, S = 0, W = 0, RS = 0, PT = 0, PL = 0, SK = 0, FS = 0; };
template <int KA, int KSA, int COUTA, int EPIA, class ADDA, class STA, int ZBA, int KB, int NW, int TPJ, int COUTB, class IN, class OUT, int OUT_CH0, int ZBB>
__device__ __forceinline__ void dense_pair_stage(int wave, int lane) {
  constexpr int NPA = (COUTA + 3) / 4, KROWA = 8 * KSA, ROWSA = NPA * 4 * KROWA * 2;
  constexpr int NPB = (COUTB + 3) / 4, NCH = (NPB + TPJ - 1) / TPJ, ROWSB = NPB * 4 * 8 * 2;
  constexpr int W = IN::W, P = W * W, NT = (P + 63) / 64, JOBS = NCH * NT;
  constexpr bool OUT_LINEAR = (OUT::RS == W && OUT::PT == 0 && OUT::PL == 0 && OUT::SK == 0), STORE_A = STA::S != 0;
  static_assert(NPA <= 2 && COUTA <= 8 && (EPIA == EPI_LINEAR || EPIA == EPI_ADD), "layer A: at most eight linear output channels");
  static_assert(WBYTES[KA] == ROWSA + NPA * 16 && WBYTES[KB] == ROWSB + NPB * 16 && UNIT_OF[KA] == UNIT_OF[KB] && KB == KA + 1, "the two blocks of one ring unit");
  static_assert(IN::S >= 16 * KSA && IN::OFF % 16 == 0 && IN::S % 16 == 0 && IN::RS == W && IN::PT == 0 && IN::PL == 0 && IN::SK == 0, "halo-free input, aligned B fragments");
  static_assert((W == 28 || W == 14) && IN::H == W && OUT::W == W && ZBA % 16 == 0 && ZBB % 16 == 0, "tile geometries");
  const int g = lane >> 4, c = lane & 15;
  const int ly = (W == 14) ? (lane * 74) >> 10 : (lane * 37) >> 10, lx = lane - W * ly;         // the lane's pixel inside the first tile
  const int in_lane = IN::OFF + lane * IN::S, out_lane = OUT::at(ly, lx) + 2 * OUT_CH0;
  const bool a_on = (c >> 2) == g;
  const int fa_base = a_on ? woff(KA) + (c & 3) * (KROWA * 2) : ZBA, fb_base = a_on ? woff(KB) + (c & 3) * 16 : ZBB;
  int fr_scale = a_on ? 1 : 0;
  asm("" : "+v"(fr_scale));
  int j0, j1;
  job_range<JOBS, NW>(wave, j0, j1);
#pragma unroll
  for (int chunk = 0; chunk < NCH; ++chunk) {                  // unrolled at compile time (see dense_tile_stage)
    int j = max(j0, chunk * NT);
    const int jend = min(j1, (chunk + 1) * NT);
    if (j >= jend) continue;
    v4i aA[NPA][KSA], aB[TPJ];
    v4f bA[NPA], bB[TPJ];
#pragma unroll
    for (int pa = 0; pa < NPA; ++pa) {
      const int fa = fa_base + __mul24(fr_scale, pa * (4 * KROWA * 2));
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) aA[pa][ks] = ld128(fa + 16 * ks);
      bA[pa] = __builtin_bit_cast(v4f, ld128(woff(KA) + ROWSA + 16 * pa));
    }
#pragma unroll
    for (int tt = 0; tt < TPJ; ++tt) {
      const int ps = min(chunk * TPJ + tt, NPB - 1);
      aB[tt] = ld128(fb_base + __mul24(fr_scale, ps * 64));
      bB[tt] = __builtin_bit_cast(v4f, ld128(woff(KB) + ROWSB + 16 * ps));
    }
    for (; j < jend; ++j) {
      const int p0 = min(64 * (j - chunk * NT), P - 64);
      const int src = in_lane + p0 * IN::S;
      v4i b[KSA];
#pragma unroll
      for (int ks = 0; ks < KSA; ++ks) b[ks] = ld128(src + 16 * ks);
      v4i mid = {0, 0, 0, 0};                                   // layer A's packed outputs = layer B's k-slots (channels past COUTA: zero, as the arena's padding was)
#pragma unroll
      for (int pa = 0; pa < NPA; ++pa) {
        v4f acc = bA[pa];
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, aA[pa][ks]), __builtin_bit_cast(v8h, b[ks]), acc, 0, 0, 0);
        if constexpr (EPIA == EPI_ADD) {
          const uint2 r = ld64(ADDA::OFF + (p0 + lane) * ADDA::S + 8 * pa);
          v2h r0, r1; __builtin_memcpy(&r0, &r.x, 4); __builtin_memcpy(&r1, &r.y, 4);
          acc[0] += (float)r0[0]; acc[1] += (float)r0[1]; acc[2] += (float)r1[0]; acc[3] += (float)r1[1];
        }
        mid[2 * pa] = (int)pack2(acc[0], acc[1]); mid[2 * pa + 1] = (int)pack2(acc[2], acc[3]);
      }
      if constexpr (STORE_A) {
        if (chunk == 0) {
#pragma unroll
          for (int pa = 0; pa < NPA; ++pa) st64(STA::OFF + (p0 + lane) * STA::S + 8 * pa, uint2{(uint32_t)mid[2 * pa], (uint32_t)mid[2 * pa + 1]});
        }
      }
      int dst;
      if constexpr (OUT_LINEAR) dst = out_lane + (p0 * OUT::S + 8 * TPJ * chunk);
      else {
        const int Y = (W == 14) ? (p0 * 4682) >> 16 : (p0 * 2341) >> 16;          // p0 / W (scalar)
        const int X = p0 - W * Y;
        dst = out_lane + (Y * OUT::ROWB + X * OUT::S + 8 * TPJ * chunk) + (lx >= W - X ? OUT::ROWB - W * OUT::S : 0);
      }
#pragma unroll
      for (int tt = 0; tt < TPJ; ++tt) {
        if (chunk * TPJ + tt < NPB) {
          v4f acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, aB[tt]), __builtin_bit_cast(v8h, mid), bB[tt], 0, 0, 0);
          uint2 v; v.x = leaky_pack2(acc[0], acc[1]); v.y = leaky_pack2(acc[2], acc[3]);
          st64(dst + 8 * tt, v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ 3x3 convs, lane-private
// conv1 (RGBX pixels, 8 bytes per tap) and the depthwise convs (4 channels = 8 bytes per tap and group): a k-step carries
// two taps, nine taps take five k-steps (the last slot pair is empty: its weights are zero, its data whatever tap 8 was).
// Jobs: 4 output rows x 16 columns (border blocks shifted inwards) x channel group.
// GOUT: the results go to global memory (gout + OUT's offsets + OUT_B0) instead of LDS -- conv2d_27 writes the frame's park slot.
typedef __attribute__((address_space(1))) v2u_t* glb_v2u_p;
typedef __attribute__((address_space(1))) uint32_t* glb_u32_p;
template <int K, int NW, int STRIDE, class IN, class OUT, int C, bool DEPTHWISE, bool GOUT = false, int OUT_B0 = 0>
__device__ __forceinline__ void conv3x3_stage(char* lds, const uint8_t* __restrict__ tab, ConvT t, int wave, int lane, char* gout = nullptr) {
  constexpr int W = OUT::W, H = OUT::H;
  static_assert(WBYTES[K] == ((C + 3) / 4) * 320, "stage and weight block agree");
  constexpr int NSEG = (W + 15) / 16, NRB = (H + 3) / 4;
  constexpr int NG = (C + 3) / 4;                                  // output-channel groups of 4
  constexpr int JPG = NRB * NSEG;
  constexpr int DROW = STRIDE * IN::ROWB, TS = IN::S, TR = IN::ROWB;
  const int g = lane >> 4, c = lane & 15;
  const int xl = min(c, W - 1);                                    // surplus lanes redo the last column (same value, same address)
  const int lane_in = g * DROW + xl * STRIDE * IN::S;
  const bool a_on = (c >> 2) == g;
  if constexpr (!DEPTHWISE) {
    // a dense 3x3 (conv2d_1): every output-channel group reads the SAME taps, so a job is a tile with ALL its groups -- the nine tap reads and
    // the pixel arithmetic once instead of once per group, NG independent MFMA chains in flight
    static_assert(NG == 2, "conv2d_1: eight output channels");
    v4i a[NG][5];
    v4f bias[NG];
    {
      const char* abase = lds + (a_on ? woff(K) + (c & 3) * 16 : IN::OFF);        // zero fragments from the zero halo row (see below)
      const int gstep = a_on ? 320 : 0;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) a[q][ks] = *reinterpret_cast<const v4i*>(abase + q * gstep + 64 * ks);
        bias[q] = uniform_f4(tab + t.b_off + 16 * q);
        asm volatile("" : "+v"(bias[q]));
      }
    }
    // the input is in column-parity planes (B_IN): the three taps of a row are plane 0 [ox], plane 1 [ox], plane 0 [ox + 1]; tile row gp of lane group g
    static_assert(STRIDE == 2 && IN::S == 8 && IN::RS == 58, "conv2d_1 reads the plane layout of B_IN");
    const int gp = ((g & 1) << 1) | (g >> 1);
    const int lane_pl = gp * DROW + xl * IN::S;
    int jt, jt1;
    job_range<JPG, NW>(wave, jt, jt1);
    for (; jt < jt1; ++jt) {
      const int rb = jt / NSEG, seg = jt - rb * NSEG;
      const int oy0 = min(rb * 4, H - 4);
      const int x0 = (W >= 16) ? min(seg * 16, W - 16) : 0;
      const char* src = lds + IN::OFF + (oy0 * STRIDE) * IN::ROWB + x0 * IN::S + lane_pl;
      uint2 tp[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) tp[k] = lds_tap64(src + (k / 3) * TR + (k % 3 == 0 ? 0 : k % 3 == 1 ? IN_PLANE1 : IN::S));
      char* dst = lds + OUT::at(oy0 + gp, x0 + xl);
      v4f acc[NG];
#pragma unroll
      for (int q = 0; q < NG; ++q) acc[q] = bias[q];
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const uint2 lo = tp[2 * ks], hi = tp[ks < 4 ? 2 * ks + 1 : 8];
        const v4i u = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
#pragma unroll
        for (int q = 0; q < NG; ++q)
          acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a[q][ks]), __builtin_bit_cast(v8h, u), acc[q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        uint2 v; v.x = leaky_pack2(acc[q][0], acc[q][1]); v.y = leaky_pack2(acc[q][2], acc[q][3]);
        *reinterpret_cast<uint2*>(dst + 8 * q) = v;
      }
    }
    return;
  }
  // Jobs of the depthwise stages.  A two-segment grid (conv2d_3: 28 columns = segments at x0 = 0 and 12) hands out ROW BLOCKS: the two tiles of a block are the
  // two jobs in flight of an iteration, so a job's coordinates are its block's row and a compile-time column (the per-tile formulation spent 34 scalar
  // instructions per iteration on two divisions by NSEG, their remainders and clamps).
  constexpr bool BLOCKS = (NSEG == 2 && NW <= 8);
  constexpr int JU = BLOCKS ? NRB : JPG;                          // job units per channel group
  int j, j1;
  job_range<NG * JU, NW>(wave, j, j1);
  while (j < j1) {
    const int cg = j / JU;
    const int jend = min(j1, (cg + 1) * JU);
    v4i a[5];                        // A fragments as dword vectors (bit-cast at the MFMA): as half vectors the compiler re-packs
                                     // every already loaded fragment behind each conditional load (~120 VALU instructions per group)
    // the lanes whose fragment is all zero read it too -- from the input buffer's top halo row, which holds zeros while the stage runs --
    // at the same immediate offsets: no zero-filling moves, no exec masking per channel group (the int8 kernel's zero region)
    static_assert(IN::PT == 1 && IN::RS * IN::S >= 5 * 64 && IN::OFF % 16 == 0, "a zero halo row of at least five fragments");
    {
      const char* abase = lds + (a_on ? woff(K) + cg * 320 + (c & 3) * 16 : IN::OFF);
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) a[ks] = *reinterpret_cast<const v4i*>(abase + 64 * ks);
    }
    v4f bias = uniform_f4(tab + t.b_off + 16 * cg);
    asm volatile("" : "+v"(bias));     // in VGPRs before the job loop: as SGPRs the accumulators' initial moves wait for the scalar load INSIDE it (lgkmcnt(0): every LDS read with it)
    // one job: nine tap reads -> five MFMAs -> LeakyReLU -> fp16 -> one 8-byte store.  TWO jobs run in flight per iteration: all
    // eighteen tap reads are issued before the first MFMA and the two accumulator chains interleave (the default schedule paired
    // every MFMA with its own reads: five LDS round trips in a row per job, 1250 cycles per job in the stage timeline).
    auto taps_at = [&](int oy0, int x0, uint2 (&tp)[9], char*& dst) {
      // tap (ky,kx) of output (oy,ox) sits at halo'd row oy*STRIDE+ky, column ox*STRIDE+kx; depthwise: channel group cg
      const char* src = lds + IN::OFF + (oy0 * STRIDE) * IN::ROWB + x0 * STRIDE * IN::S + (DEPTHWISE ? 8 * cg : 0) + lane_in;
#pragma unroll
      for (int k = 0; k < 9; ++k) tp[k] = lds_tap64(src + (k / 3) * TR + (k % 3) * TS);
      dst = (GOUT ? gout : lds) + OUT::at(oy0 + g, x0 + xl) + 8 * cg + OUT_B0;
    };
    auto taps = [&](int jj, uint2 (&tp)[9], char*& dst) {
      const int rem = jj - cg * JPG;
      const int rb = rem / NSEG, seg = rem - rb * NSEG;
      taps_at(min(rb * 4, H - 4), (W >= 16) ? min(seg * 16, W - 16) : 0, tp, dst);
    };
    auto kstep = [&](const uint2 (&tp)[9], int ks, v4f acc) {
      const uint2 lo = tp[2 * ks], hi = tp[ks < 4 ? 2 * ks + 1 : 8];
      const v4i u = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a[ks]), __builtin_bit_cast(v8h, u), acc, 0, 0, 0);
    };
    auto finish = [&](v4f acc, char* dst) {
      uint2 v; v.x = leaky_pack2(acc[0], acc[1]); v.y = leaky_pack2(acc[2], acc[3]);
      if constexpr (GOUT) *(glb_v2u_p)(uintptr_t)dst = v2u_t{v.x, v.y};
      else *reinterpret_cast<uint2*>(dst) = v;
    };
    auto pair = [&](const uint2 (&tp0)[9], const uint2 (&tp1)[9], char* d0, char* d1) {
      __builtin_amdgcn_sched_group_barrier(0x100, 18, 0);         // the eighteen tap reads ...
      __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);         // ... then the ten MFMAs
      v4f acc0 = bias, acc1 = bias;
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) { acc0 = kstep(tp0, ks, acc0); acc1 = kstep(tp1, ks, acc1); }
      finish(acc0, d0);
      finish(acc1, d1);
    };
    if constexpr (BLOCKS) {
      for (; j < jend; ++j) {
        const int oy0 = min((j - cg * NRB) * 4, H - 4);
        uint2 tp0[9], tp1[9];
        char *d0, *d1;
        taps_at(oy0, 0, tp0, d0);
        taps_at(oy0, W - 16, tp1, d1);
        pair(tp0, tp1, d0, d1);
      }
    } else {
      for (; NW <= 8 && j + 1 < jend; j += 2) {
        uint2 tp0[9], tp1[9];
        char *d0, *d1;
        taps(j, tp0, d0);
        taps(j + 1, tp1, d1);
        pair(tp0, tp1, d0, d1);
      }
      for (; j < jend; ++j) {
        uint2 tp[9];
        char* dst;
        taps(j, tp, dst);
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        v4f acc = bias;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) acc = kstep(tp, ks, acc);
        finish(acc, dst);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ zero fills
// Halo of a buffer as CONTIGUOUS RUNS of 8- or 16-byte granules.  RING (1-pixel border all round): the top row and the first
// pixel of row 1 are one run, the last pixel of row r and the first of row r+1 are adjacent (HR-3 runs of two pixels), the last
// pixel of row HR-2 and the bottom row are one run.  Otherwise (top row + left column): the top row is one run, then one pixel per
// row.  An item is one granule: a compare or two and a multiply-shift instead of the five divisions of a per-dword formulation
// (the halo fills were 25 % of the kernel's VALU instructions).
template <class B, bool RING, int NT>
__device__ __forceinline__ void fill_halo(char* lds0, int tid) {
  constexpr int S = B::S, WR = B::RS, HR = B::H + B::PT + (RING ? 1 : 0), SK = B::SK, ROWB = B::ROWB;
  constexpr int G = (S % 16 == 0 && B::OFF % 16 == 0 && SK % 16 == 0) ? 16 : 8, PG = S / G, SG = SK / G;       // granule bytes, granules per pixel / per row skew
  static_assert(S % G == 0 && B::OFF % G == 0 && SK % G == 0, "granules");
  // the skew bytes between a row's last pixel and the next row's first one are unused: a run that crosses a row boundary clears them too
  constexpr int NA = RING ? (WR + 1) * PG + SG : WR * PG;             // first run
  constexpr int RUN = RING ? 2 * PG + SG : PG;                        // middle runs (two pixels / one pixel each)
  constexpr int NB = RING ? (HR - 3) * RUN : (HR - 1) * RUN;
  constexpr int NC = RING ? (WR + 1) * PG + SG : 0;                   // last run
  constexpr int N1 = NA + NB + NC;
  for (int i = NT - 1 - tid; i < N1; i += NT) {              // the last waves first (see fetch_unit)
    const int k = i;
    int off;
    if (k < NA) off = k * G;
    else if (k < NA + NB) {
      const int j = k - NA;
      const int r = j / RUN, g = j - r * RUN;                         // compile-time divisor
      off = RING ? (r + 1) * ROWB + (WR - 1) * S + g * G : (r + 1) * ROWB + g * G;
    } else off = (HR - 2) * ROWB + (WR - 1) * S + (k - NA - NB) * G;
    char* dst = lds0 + B::OFF + off;
    if constexpr (G == 16) *reinterpret_cast<uint4*>(dst) = uint4{0u, 0u, 0u, 0u};
    else *reinterpret_cast<uint2*>(dst) = uint2{0u, 0u};
  }
}

// ------------------------------------------------------------------------------------------------ max-pools
__device__ __forceinline__ uint32_t pkmaxh(uint32_t a, uint32_t b) {      // v_pk_max_f16
  v2h x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
  const v2h r = __builtin_elementwise_max(x, y);
  uint32_t o; __builtin_memcpy(&o, &r, 4); return o;
}
// 8-wide stride-2 window [2o-3, 2o+4] = four odd pairs R[j] = max(c[2j+1], c[2j+2]); S[j] = max(R[j], R[j+1]);
// out[o] = max(S[o-2], S[o]); coordinates clamped into [0, LIM] (max is idempotent).  NO outputs from O0 on per sweep; O0 is a compile-time
// constant, so every clamped coordinate is one and the loads' offsets are immediates (round 3 computed ~85 address instructions per item).
// V: the packed value type (one or two dwords of fp16 pairs).
template <int O0, int NO, int LIM, class V, class MAX, class LOADC, class STORE>
__device__ __forceinline__ void pool8_sweep(MAX mx, LOADC loadc, STORE store) {
  constexpr int NR = NO + 3;
  V r[NR];
#pragma unroll
  for (int jj = 0; jj < NR; ++jj) {
    constexpr int dummy = 0; (void)dummy;
    const int j = O0 - 2 + jj;
    const int xa = 2 * j + 1 < 0 ? 0 : (2 * j + 1 > LIM ? LIM : 2 * j + 1), xb = 2 * j + 2 < 0 ? 0 : (2 * j + 2 > LIM ? LIM : 2 * j + 2);
    r[jj] = xa == xb ? loadc(xa) : mx(loadc(xa), loadc(xb));
  }
  V q[NR - 1];
#pragma unroll
  for (int jj = 0; jj < NR - 1; ++jj) q[jj] = mx(r[jj], r[jj + 1]);
#pragma unroll
  for (int n = 0; n < NO; ++n) store(O0 + n, mx(q[n], q[n + 2]));
}
typedef __attribute__((address_space(3))) uint32_t* lds_u32w_p;
__device__ __forceinline__ uint32_t ld32(int a) { return *(const lds_u32w_p)(uintptr_t)(uint32_t)a; }
__device__ __forceinline__ void st32(int a, uint32_t v) { *(lds_u32w_p)(uintptr_t)(uint32_t)a = v; }
// pool_8, horizontal pass: T4 [28][28] x 18 ch -> HB [28 rows][14].  An item = (row, 8-byte chunk of the 40-byte pixel: four channels) sweeps its WHOLE row
// (fourteen outputs, twenty-eight ds_read_b64 at immediate offsets): 140 items = three waves, which run beside conv2d_10 on the other five -- both only
// read T4 (round 4; as a stage of its own the pass swept half rows on six waves and cost a barrier interval of 2.4 k cycles per frame).
constexpr int POOL8H_WAVES = 3;
__device__ __forceinline__ void pool8_h(int pw, int lane) {
  static_assert(B_T4::S == 40 && B_T4::OFF % 8 == 0 && B_T4::ROWB % 8 == 0 && B_HB::S == 36, "8-byte chunks of T4's pixels, dword stores into HB's");
  const int it = min(pw * 64 + lane, 139);                       // surplus lanes redo the last item (same values, same addresses)
  const int q = (it * 2341) >> 16, y = it - 28 * q;              // chunk = it / 28 (rows vary fastest: 34-dword row pitch mod 64 spreads the lanes over the banks)
  const int row = B_T4::at(y, 0) + 8 * q, dst = B_HB::OFF + y * (14 * B_HB::S) + 8 * q;
  const bool two = q < 4;                                        // the fifth chunk holds channels 16, 17 and two padding channels
  auto mx = [](uint2 a, uint2 b) { return uint2{pkmaxh(a.x, b.x), pkmaxh(a.y, b.y)}; };
  pool8_sweep<0, 14, 27, uint2>(mx, [&](int x) { return ld64(row + x * B_T4::S); },
                                [&](int ox, uint2 v) { st32(dst + ox * B_HB::S, v.x); if (two) st32(dst + ox * B_HB::S + 4, v.y); });
}
// pool_8, vertical pass: HB -> pool half of concat_22 (T14 channels 0..17).  An item = (column, channel dword) sweeps its whole column (fourteen outputs,
// twenty-eight loads at immediate offsets): 126 items = two waves; the other waves of the stage run conv2d_12 (T6 -> T7), which does not touch HB or T14.
constexpr int POOL8V_WAVES = 2;
__device__ __forceinline__ void pool8_v(int item) {
  const int it = min(item, 14 * 9 - 1);
  const int ox = (it * 7282) >> 16, d = it - 9 * ox;             // it / 9
  const int col = B_HB::OFF + 4 * it, dst = B_T14::OFF + ox * B_T14::S + 4 * d;
  pool8_sweep<0, 14, 27, uint32_t>([](uint32_t a, uint32_t b) { return pkmaxh(a, b); }, [&](int r) { return ld32(col + r * (14 * B_HB::S)); },
                                   [&](int oy, uint32_t v) { st32(dst + oy * (14 * B_T14::S), v); });
}
// pool_25 by COLUMNS (as in the int8 kernel): one item = (output column, channel dword) walks the 14 rows of T15 once -- per row
// the horizontal 4-tap maximum (clamped columns), pairs of rows R[j] = max(h[2j-1], h[2j]), out[oy] = max(R[oy], R[oy+1]) -- and writes
// its 7 outputs: 56 loads per item instead of 7 x 16, no per-tap clamping.  84 items: the first two waves of the stage take them, the
// others run conv2d_27, which reads the same T15.
constexpr int POOL25_WAVES = (84 + 63) / 64;
template <class T15, class T30>
__device__ __forceinline__ void pool25_cols(char* lds, int item, char* gout) {
  static_assert(T15::W == 14 && T15::H == 14 && T30::W == 7, "pool_25 geometry");
  if (item >= 84) return;
  const int ox = item / 12, d = item - 12 * ox;
  const char* base = lds + T15::at(0, 0) + 4 * d;
  constexpr int S = T15::S, ROW = T15::ROWB;
  const int c0 = max(2 * ox - 1, 0) * S, c1 = 2 * ox * S, c2 = c1 + S, c3 = min(2 * ox + 2, 13) * S;
  auto hrow = [&](int r) {
    const char* p = base + r * ROW;
    return pkmaxh(pkmaxh(lds_u32(p + c0), lds_u32(p + c1)), pkmaxh(lds_u32(p + c2), lds_u32(p + c3)));
  };
  char* dst = gout + T30::OFF + ox * T30::S + 4 * d;          // the pooled half goes straight to the frame's park slot (global memory)
  uint32_t prev = hrow(0);                                     // R[0] = max(h[-1 -> 0], h[0])
#pragma unroll
  for (int oy = 0; oy < 7; ++oy) {
    uint32_t next = hrow(2 * oy + 1);                          // R[oy+1] = max(h[2oy+1], h[2oy+2 -> 13])
    if (2 * oy + 2 <= 13) next = pkmaxh(next, hrow(2 * oy + 2));
    *(glb_u32_p)(uintptr_t)(dst + oy * (7 * T30::S)) = pkmaxh(prev, next);
    prev = next;
  }
}

// ------------------------------------------------------------------------------------------------ the 7x7 tail: one frame per WAVE, no barriers
// Behind conv2d_27 every map of the network is 7 x 7: 49 pixels, ONE lane-private tile.  As barrier-separated workgroup stages the twelve layers
// conv2d_29 .. conv2d_53 were twelve latency chains of 0.7-2.5 k cycles with 2-20 jobs for 8 waves (28 k cycles per PAIR of frames in round 3's tail
// batching, a third of the kernel).  Here a wave owns a whole frame: lane = pixel, and because a lane-private pass leaves 4 output channels of the
// lane's pixel in the lane's own accumulator, a 1x1 layer's packed fp16 outputs ARE the next 1x1 layer's B operand -- the chain runs in registers,
// with no LDS round trip and no barrier.  Only the three depthwise 3x3 layers need other pixels: the wave writes its 40-channel input into a
// PRIVATE halo'd exchange buffer (6.4 KB) and reads the nine taps back; LDS operations of one wave execute in order, so no barrier either.
// The workgroup runs the front stages of up to NW frames one after the other (each leaves {pool_25 | conv2d_27} = 49 x 96 bytes in its park slot in
// HBM), then ONE tail phase: the twelve tail layers' weights and biases (21.5 KB) become resident in LDS behind the NW exchange buffers, and wave w
// runs the tail of the batch's frame w.  Arithmetic and its order are those of the staged form: results bit for bit equal.
constexpr int PARK_PX = 96, PARK_BYTES = 49 * PARK_PX;        // a frame's park slot: 49 pixels x (24 pool_25 + 24 conv2d_27) fp16 channels
typedef Buf<0, 7, 7, PARK_PX, 7, 0, 0> PARK;
// Exchange buffer: T19's geometry (40 channels, halo ring) with a row pitch of 198 dwords.  A tap is a ds_read_b64 (two banks per lane, 32 lanes per
// group): pixel x of a row starts at bank 20 x mod 64 -- slots 5 x mod 16 of the sixteen 4-bank slots of one parity class -- and with 198 dwords per row
// rows y and y + 2 are three slots apart and rows y, y + 1 in different classes, so the lane -> pixel map of tail_chain (lanes 0-31: rows 0-3 and the first
// two pixels of rows 4 and 5) puts the 32 lanes of each group on 64 different banks for every tap: 2.2 cycles per read instead of ~5.5 with the 182-dword
// pitch of the staged T19 (the chain is bound by the LDS pipe: 33.9 k -> 30.9 k cycles per tail phase).
typedef Buf<0, 7, 7, 80, 9, 1, 1, 0, 72> XT;
constexpr int XB = (9 * XT::ROWB + 15) & ~15;                  // bytes per wave
constexpr int TAILW0 = (int)conv_at(12).w_off, TAILW_BYTES = (int)conv_at(23).b_off + bias_bytes(23) - TAILW0;
template <int NW> constexpr int tw(int k) { return NW * XB + (int)conv_at(k).w_off - TAILW0; }     // LDS address of tail conv k's rows / biases
template <int NW> constexpr int tb(int k) { return NW * XB + (int)conv_at(k).b_off - TAILW0; }
static_assert(XB % 16 == 0 && TAILW0 % 16 == 0 && TAILW_BYTES % 16 == 0 && 5 * 64 + 16 <= XT::ROWB && 16 * 6 <= XT::ROWB, "tail plan: aligned blocks, a zero halo row that covers every zero fragment");
// PART 0: the 1 KB chunks that end below the weight ring (their destination -- concat_22 / T11 territory -- is dead once conv2d_23 is through: they are issued during
// the batch's last {pool_25 || conv2d_27} stage); PART 1: the chunks on the ring itself, where conv2d_27's block is still being read then (issued in front of the
// tail phase's first barrier); PART 2: all of them.
template <int NW> constexpr int tailw_early_chunks() { const int c = (RING0 - NW * XB) / 1024; return c < 0 ? 0 : c; }
template <int NW, int PART = 2>
__device__ __forceinline__ void fetch_tailw(const uint8_t* __restrict__ tab, int wave, int lane) {
  static_assert(NW * XB + TAILW_BYTES <= LDS_TOTAL, "exchange buffers and resident tail weights fit the workgroup's LDS (arena and ring are dead in the tail phase; the arena is cleared before the next batch)");
  constexpr int NCHUNK = (TAILW_BYTES + 1023) / 1024, EARLY = tailw_early_chunks<NW>() < NCHUNK ? tailw_early_chunks<NW>() : NCHUNK;
  constexpr int J0 = PART == 1 ? EARLY : 0, J1 = PART == 0 ? EARLY : NCHUNK;
  // PART 0 lands while the batch's last {pool_25 || conv2d_27} stage still READS T15 and the zero spot: its destination [NW * XB, NW * XB + EARLY KB) must lie
  // above T15's last row and below the zero spot (and below the ring, which tailw_early_chunks bounds) -- whatever XB, NW or the arena plan become
  static_assert(NW * XB >= B_T15::OFF + 15 * B_T15::ROWB, "the early tail-weight DMA must not land on T15, which the stage it is issued in still reads");
  static_assert(NW * XB + EARLY * 1024 <= ZSLACK && NW * XB + EARLY * 1024 <= RING0, "the early tail-weight DMA must stay below the zero spot and the weight ring");
  for (int j = J0 + wave; j < J1; j += NW) {
    const int off = j * 1024 + lane * 16;
    if (off < TAILW_BYTES) {
      const uint8_t* src = tab + TAILW0 + off;
      const uint32_t dst = (uint32_t)(NW * XB + j * 1024);
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  }
}
__device__ __forceinline__ uint2 tap64(int a) {               // one ds_read_b64 (volatile: not merged into a ds_read2_b64, see lds_tap64)
  const v2u_t v = *(const volatile __attribute__((address_space(3))) v2u_t*)(uintptr_t)(uint32_t)a;
  return uint2{v.x, v.y};
}
struct TailLane { int scale, c3, zb, tapb; bool a_on; int base32, scale32; };       // per-lane constants of the tail: fragment selectors (16x16 and 32x32 forms), zero spot, tap base
typedef float v16f __attribute__((ext_vector_type(16)));
// 1x1 layer K on the lane's pixel: in = NI packed dwords (8 channels per k-step), out = the layer's packed outputs from dword O0 on.
// The chain is bound by the LDS pipe -- every MFMA of the sixteen waves of a CU wants its own 1 KB A fragment -- so the 1x1 layers use the 32x32x16 form
// here: lane (kb = lane / 32, i = lane % 32) supplies k-slots 8 kb .. 8 kb + 7 from its own pixel, A row i carries W[16 q + 4 (i >> 3) + (i & 3)] in the slots
// of block ((i >> 2) & 1) only, and the lane's sixteen accumulators are output channels 16 q .. 16 q + 15 of its pixel: SIXTEEN channels per fragment read
// instead of four (conv2d_47: 18 fragments and MFMAs instead of 60).  Same products, same k-step order: bit for bit the results of the 16x16x32 form
// (tools/probe/mfma_f16_shapes.hip: 0 of 2 048 000 results differ).  Rows past the layer's channels read whatever follows the block: they only reach
// accumulators that are never stored.
template <int NW, int K, int KS, int COUT, int EPI, int O0, int NI, int NO, int NA>
__device__ __forceinline__ void dense_reg(const TailLane& L, const uint32_t (&in)[NI], uint32_t (&out)[NO], const uint32_t (&add)[NA], float* __restrict__ head) {
  constexpr int NP = (COUT + 3) / 4, KROW = 8 * KS, NQ = (COUT + 15) / 16;
  static_assert(NI == 4 * KS && (EPI == EPI_HEAD || O0 + 2 * NP <= NO) && (EPI != EPI_ADD || NA == 2 * NP), "operand sizes");
  static_assert(WBYTES[K] == NP * 4 * KROW * 2 + NP * 16, "layer and weight block agree");
  const int base = L.base32 >= 0 ? tw<NW>(K) + L.base32 * (KROW * 2) : L.zb;        // row 4 (i >> 3) + (i & 3) of the block, or the zero spot
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int fa = base + __mul24(L.scale32, q * (16 * KROW * 2));
    v16f acc;
#pragma unroll
    for (int v4 = 0; v4 < 4; ++v4) {
      const v4f bq = __builtin_bit_cast(v4f, ld128(tb<NW>(K) + 64 * q + 16 * v4));
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[4 * v4 + e] = bq[e];
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const v4i b = {(int)in[4 * ks], (int)in[4 * ks + 1], (int)in[4 * ks + 2], (int)in[4 * ks + 3]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, ld128(fa + 16 * ks)), __builtin_bit_cast(v8h, b), acc, 0, 0, 0);
    }
#pragma unroll
    for (int v4 = 0; v4 < 4; ++v4) {
      const int ps = 4 * q + v4;
      if (ps < NP) {
        float c0 = acc[4 * v4], c1 = acc[4 * v4 + 1], c2 = acc[4 * v4 + 2], c3 = acc[4 * v4 + 3];
        if constexpr (EPI == EPI_HEAD) {
          if (4 * ps + 0 < COUT) head[4 * ps + 0] = c0;
          if (4 * ps + 1 < COUT) head[4 * ps + 1] = c1;
          if (4 * ps + 2 < COUT) head[4 * ps + 2] = c2;
          if (4 * ps + 3 < COUT) head[4 * ps + 3] = c3;
        } else {
          if constexpr (EPI == EPI_ADD) {
            v2h r0, r1; __builtin_memcpy(&r0, &add[2 * ps], 4); __builtin_memcpy(&r1, &add[2 * ps + 1], 4);
            c0 += (float)r0[0]; c1 += (float)r0[1]; c2 += (float)r1[0]; c3 += (float)r1[1];
          }
          if constexpr (EPI == EPI_ACT) { out[O0 + 2 * ps] = leaky_pack2(c0, c1); out[O0 + 2 * ps + 1] = leaky_pack2(c2, c3); }
          else { out[O0 + 2 * ps] = pack2(c0, c1); out[O0 + 2 * ps + 1] = pack2(c2, c3); }
        }
      }
    }
  }
}
// depthwise 3x3 (stride 1, 40 channels) K: the lane's pixel goes into the wave's exchange buffer, the nine taps come back per 4-channel group
template <int NW, int K>
__device__ __forceinline__ void dw_reg(const TailLane& L, const uint32_t (&in)[20], uint32_t (&out)[20]) {
  static_assert(WBYTES[K] == 10 * 320, "a 40-channel depthwise block");
#pragma unroll
  for (int q = 0; q < 10; ++q) st64(L.tapb + XT::ROWB + XT::S + 8 * q, uint2{in[2 * q], in[2 * q + 1]});
  const int base = L.a_on ? tw<NW>(K) + L.c3 * 16 : L.zb;
#pragma unroll
  for (int cg = 0; cg < 10; ++cg) {
    const int fa = base + __mul24(L.scale, cg * 320);
    v4f acc = __builtin_bit_cast(v4f, ld128(tb<NW>(K) + 16 * cg));
    uint2 tp[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) tp[k] = tap64(L.tapb + (k / 3) * XT::ROWB + (k % 3) * XT::S + 8 * cg);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const uint2 lo = tp[2 * ks], hi = tp[ks < 4 ? 2 * ks + 1 : 8];
      const v4i u = {(int)lo.x, (int)lo.y, (int)hi.x, (int)hi.y};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, ld128(fa + 64 * ks)), __builtin_bit_cast(v8h, u), acc, 0, 0, 0);
    }
    out[2 * cg] = leaky_pack2(acc[0], acc[1]); out[2 * cg + 1] = leaky_pack2(acc[2], acc[3]);
  }
}
// the tail of ONE frame on ONE wave: xb = the wave's exchange buffer (zeroed: its halo ring stays zero), park = the frame's park slot, head = its logits
template <int NW>
__device__ __forceinline__ void tail_chain(int xb, const char* __restrict__ park, float* __restrict__ head, int lane) {
  const int g = lane >> 4, c = lane & 15;
  // lane -> pixel: rows 0-3 on lanes 0-27, then (4,0) (4,1) (5,0) (5,1) on lanes 28-31, the rest of rows 4 and 5 and row 6 on lanes 32-48 (see XT: every
  // tap read is then free of bank conflicts); lanes 49..63 redo pixel 48
  const int p = lane < 30 ? lane : lane < 32 ? lane + 5 : lane < 37 ? lane - 2 : min(lane, 48);
  const int y = (p * 37) >> 8, x = p - 7 * y;
  TailLane L;
  L.a_on = (c >> 2) == g; L.c3 = c & 3; L.zb = xb; L.tapb = xb + y * XT::ROWB + x * XT::S;
  L.scale = L.a_on ? 1 : 0;
  asm("" : "+v"(L.scale));
  {
    const int i = lane & 31, kb = lane >> 5;
    const bool on32 = ((i >> 2) & 1) == kb;
    L.base32 = on32 ? 4 * (i >> 3) + (i & 3) : -1;
    L.scale32 = on32 ? 1 : 0;
    asm("" : "+v"(L.scale32));
  }
  uint32_t cat[24], t17[12];                                                          // concat_46: [pool_25 | conv2d_42]; conv2d_27's output
  {
    typedef const __attribute__((address_space(1))) v4i* glb_v4i_p;
    const glb_v4i_p src = (glb_v4i_p)(uintptr_t)(park + p * PARK_PX);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const v4i v = __builtin_nontemporal_load(src + q);                              // written by other waves of this workgroup: past the vector L1
#pragma unroll
      for (int e = 0; e < 4; ++e) { if (q < 3) cat[4 * q + e] = (uint32_t)v[e]; else t17[4 * (q - 3) + e] = (uint32_t)v[e]; }
    }
  }
  const uint32_t none[1] = {0u};
  uint32_t t18[4], t19[20], t20[20], t22[4], t26[4], t33[16];
  dense_reg<NW, 12, 3,  8, EPI_LINEAR, 0>(L, t17, t18, none, nullptr);                // conv2d_29
  dense_reg<NW, 13, 1, 40, EPI_ACT,    0>(L, t18, t19, none, nullptr);                // conv2d_30
  dw_reg<NW, 14>(L, t19, t20);                                                        // conv2d_32 (dw)
  dense_reg<NW, 15, 5,  8, EPI_ADD,    0>(L, t20, t22, t18, nullptr);                 // conv2d_34 + eltwise_35
  dense_reg<NW, 16, 1, 40, EPI_ACT,    0>(L, t22, t19, none, nullptr);                // conv2d_36
  dw_reg<NW, 17>(L, t19, t20);                                                        // conv2d_38 (dw)
  dense_reg<NW, 18, 5,  8, EPI_ADD,    0>(L, t20, t26, t22, nullptr);                 // conv2d_40 + eltwise_41
  dense_reg<NW, 19, 1, 24, EPI_ACT,   12>(L, t26, cat, none, nullptr);                // conv2d_42 -> concat_46[24,48)
  dense_reg<NW, 20, 6, 40, EPI_ACT,    0>(L, cat, t19, none, nullptr);                // conv2d_47
  dw_reg<NW, 21>(L, t19, t20);                                                        // conv2d_49 (dw)
  dense_reg<NW, 22, 5, 32, EPI_ACT,    0>(L, t20, t33, none, nullptr);                // conv2d_51
  dense_reg<NW, 23, 4, 18, EPI_HEAD,   0>(L, t33, t33, none, head + p * 18);          // conv2d_53: fp32 logits -> HBM
}

// ------------------------------------------------------------------------------------------------ the kernel
struct Params { const half_t* in; float* out; long n; const uint8_t* tab; char* scratch; long long* prof; int stop; };   // scratch: gridDim.x * NW * PARK_BYTES; prof: stage timeline (YF16_BARPROF builds); stop: leave a frame behind barrier `stop` (YF16_STAGEPMC builds, 0 = never)

template <int NW>
__global__ void __launch_bounds__(NW * 64, NW / 2) yoloface56_f16_fused(const Params prm) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NT = NW * 64;
  const int tid0 = threadIdx.x;
  const uint8_t* __restrict__ tab0 = prm.tab;
  // the barrier behind a stage also publishes the LDS-DMA of the next stage's weights, which the compiler does not see
#ifdef YF16_BARPROF
  // stage timeline (tools/fp16_timeline.py): in the workgroup's second frame (the one that closes a pair and runs the tail) every wave
  // stamps the cycle counter on arrival at and on release from each barrier: prof[wg][wave][40][2]
  bool prof_on = false; int bar_no = 0;
  long long* prof_out = prm.prof ? prm.prof + ((long)blockIdx.x * NW + __builtin_amdgcn_readfirstlane(tid0 >> 6)) * 80 : nullptr;
#define SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    if (prof_on && (tid0 & 63) == 0 && bar_no < 40) prof_out[2 * bar_no] = __builtin_readcyclecounter(); __syncthreads(); \
    if (prof_on && (tid0 & 63) == 0 && bar_no < 40) prof_out[2 * bar_no + 1] = __builtin_readcyclecounter(); ++bar_no; } while (0)
#elif defined(YF16_STAGEPMC)
  // per-stage counters (tools/fp16_stage_pmc.py): one launch per value of prm.stop = 1 .. 12, every frame is abandoned behind its barrier number `stop`
  // and the tail phase is skipped; stop = 0 runs everything.  The differences between consecutive launches are the stages' instruction counts.  Results
  // are wrong by construction.
  int stage_no = 0;
#define SYNC() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (++stage_no == prm.stop) continue; }
#define SYNC_BATCH() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#elif defined(YF16_WHATIF_NO_FRONT_BARRIERS)
  // Timing-only what-if (WRONG results; profiles/r05_fp16/whatif.txt): the per-frame barriers of the front chain become waits for the wave's OWN memory
  // operations -- no wave ever waits for another inside a batch; the two batch-level barriers stay.  The bound for any barrier-free form of the 28x28 / 14x14 stages.
#define SYNC() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
#define SYNC_BATCH() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#else
#define SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#endif
#if !defined(YF16_STAGEPMC) && !defined(YF16_WHATIF_NO_FRONT_BARRIERS)
#define SYNC_BATCH() SYNC()
#endif
#if defined(YF16_BARPROF) || defined(YF16_STAGEPMC)
#define SYNC_LDS() SYNC()
#elif defined(YF16_WHATIF_NO_FRONT_BARRIERS)
#define SYNC_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define SYNC_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
#endif
// SYNC_PIX: the barriers whose consumer stage is PIXELWISE in its producer's output (a 1x1 layer behind any layer: conv2d_3 -> 5, 12 -> 13, 15 -> 17, 17 -> 19,
// 19 -> 23) -- the ones a register-chained merge of the two stages could remove.  YF16_WHATIF_NO_PIXELWISE_BARRIERS (timing only, WRONG results) turns exactly
// those five into waits for the wave's own operations: the bound for every such merge at zero cost (profiles/r05_fp16/whatif.txt).
#if defined(YF16_WHATIF_NO_PIXELWISE_BARRIERS)
#define SYNC_PIX() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define SYNC_PIX() SYNC()
#endif
#define FETCH(U) fetch_unit<U, NW>(tab, conv_at(unit_first(U)).w_off, wave, lane)
  // the barrier behind a stage that issued prefetch_in() AFTER its weight DMA: the IN_ITERS youngest loads (global_load_dwordx3 each, checked
  // in the ISA) may stay in flight.  The profiling builds keep the plain barrier.
#if defined(YF16_BARPROF) || defined(YF16_STAGEPMC)
#define SYNC_KEEP_PREFETCH() SYNC()
#elif defined(YF16_WHATIF_NO_FRONT_BARRIERS)
#define SYNC_KEEP_PREFETCH() do { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((56 * 28 + NW * 64 - 1) / (NW * 64)) : "memory"); } while (0)
#else
#define SYNC_KEEP_PREFETCH() do { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((56 * 28 + NW * 64 - 1) / (NW * 64)) : "memory"); __syncthreads(); } while (0)
#endif
  // The next frame's input (12 bytes per item, IN_ITERS items per thread) is loaded into registers while a long stage of the current
  // frame runs -- conv2d_23 when the frame only parks its T15, conv2d_51 when it runs the tail -- instead of behind the barrier of the
  // staging stage, where the HBM latency was the stage's whole time (2.9 k cycles in the stage timeline).
  constexpr int IN_ITERS = (56 * 28 + NT - 1) / NT;
  uint32_t pin[IN_ITERS][3];
  // ALWAYS IN_ITERS load instructions (a frame index past the batch re-reads the last frame): the barrier behind the stage that issues
  // them waits with vmcnt(IN_ITERS) -- for everything older than these loads, i.e. for the weight DMA, but not for the HBM latency of the
  // prefetch itself (SYNC_KEEP_PREFETCH below; vector-memory loads return in order)
  auto prefetch_in = [&](long frame) {
    frame = frame < prm.n ? frame : prm.n - 1;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(prm.in + frame * (56 * 56 * 3));
#pragma unroll
    for (int k = 0; k < IN_ITERS; ++k) {
      const int i = min(tid0 + k * NT, 56 * 28 - 1);
      pin[k][0] = src[3 * i]; pin[k][1] = src[3 * i + 1]; pin[k][2] = src[3 * i + 2];
    }
  };
  const long G = gridDim.x;
  // a BATCH: up to NW of the workgroup's frames (base + k G): their front stages one after the other, then one tail phase with a wave per frame
  for (long base = blockIdx.x; base < prm.n; base += NW * G) {
    int nb = 1;                                                 // frames of this batch: base + k G < n (counted: a 64-bit division costs ~120 scalar instructions per wave)
#pragma unroll
    for (int k = 1; k < NW; ++k) nb += (base + k * G < prm.n) ? 1 : 0;
    // Padding channels and k-slots whose weights are zero may hold stale data: fine as long as it is FINITE (0 * NaN = NaN).  Everything the
    // stages store is finite fp16, but the tail phase leaves fp32 biases in the arena: it is cleared once per batch.
    for (int i = tid0; i < LDS_BYTES / 16; i += NT) reinterpret_cast<uint4*>(lds)[i] = uint4{0u, 0u, 0u, 0u};
    prefetch_in(base);
#ifdef YF16_BARPROF
    prof_on = false;
#endif
    SYNC_BATCH();
    for (int k = 0; k < nb; ++k) {
    const long fr = base + k * G;
    int tid = tid0;
    asm volatile("" : "+v"(tid));       // per-lane index arithmetic is recomputed per frame instead of parked in VGPRs for the whole kernel
    const uint8_t* tab = tab0;
    asm volatile("" : "+s"(tab));       // likewise the stages' table addresses (base + compile-time offset): not 24 hoisted SGPR pairs
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef YF16_BARPROF
    bar_no = 0; prof_on = prof_out != nullptr && base == (long)blockIdx.x && k == 1;
#endif
#ifdef YF16_STAGEPMC
    stage_no = 0;
#endif
    FETCH(0);
    {   // input: fp16 [56][56][3] -> RGBX pixels with a zero top row and left column.  Two pixels (12 bytes) per item; the frame's
        // dwords were loaded into registers one long stage earlier (prefetch_in)
#pragma unroll
      for (int kk = 0; kk < IN_ITERS; ++kk) {
        const int i = tid0 + kk * NT;
        if (IN_ITERS * NT != 56 * 28 && i >= 56 * 28) break;
        const uint32_t d0 = pin[kk][0], d1 = pin[kk][1], d2 = pin[kk][2];
        const int y = i / 28, xh = i - y * 28;
        uint4 px = {d0, d1 & 0xFFFFu, (d1 >> 16) | (d2 << 16), d2 >> 16};
        // pixels 2 xh and 2 xh + 1 of row y = halo'd columns 2 xh + 1 (plane 1 [xh]) and 2 xh + 2 (plane 0 [xh + 1]) of halo'd row y + 1
        char* row = lds + B_IN::OFF + (y + 1) * B_IN::ROWB + xh * 8;
        *reinterpret_cast<uint2*>(row + IN_PLANE1) = uint2{px.x, px.y};
        *reinterpret_cast<uint2*>(row + 8) = uint2{px.z, px.w};
      }
      fill_halo<B_IN, false, NT>(lds, tid);
      fill_halo<B_T1, true, NT>(lds, tid);
    }
    SYNC();
    FETCH(1);
    conv3x3_stage<0, NW, 2, B_IN, B_T1, 8, false>(lds, tab, conv_at(0), wave, lane);                    // conv2d_1
    SYNC();
    FETCH(2);
    conv3x3_stage<1, NW, 1, B_T1, B_T2, 8, true>(lds, tab, conv_at(1), wave, lane);                     // conv2d_3 (dw)
    SYNC_PIX();
    FETCH(3);
    fill_halo<B_T4, false, NT>(lds, tid);
    dense_pair_stage<2, 1, 4, EPI_LINEAR, NoBuf, NoBuf, ZSLACK, 3, NW, 5, 18, B_T2, B_T4, 0, ZSLACK>(wave, lane);            // conv2d_5 -> conv2d_6
    SYNC();
    // pool_8 and the branch beside it share two stages: {horizontal pass || conv2d_10} both only read T4, {vertical pass || conv2d_12} touch disjoint buffers
    // (HB -> concat_22's pool half; T6 -> T7).  One barrier interval less per frame than {h}, {v || conv2d_10}, {conv2d_12}.
    FETCH(4);
    if (wave < POOL8H_WAVES) pool8_h(wave, lane);                                                     // pool_8 h: T4 -> HB ...
    else conv3x3_stage<4, NW - POOL8H_WAVES, 2, B_T4, B_T6, 18, true>(lds, tab, conv_at(4), wave - POOL8H_WAVES, lane);   // ... beside conv2d_10 (dw, stride 2): T4 -> T6
    SYNC();
    FETCH(5);
    if (wave < POOL8V_WAVES) pool8_v(wave * 64 + lane);                                               // pool_8 v: HB -> concat_22[0,18) ...
    else dense_tile_stage<5, NW - POOL8V_WAVES, 1, 3, B_T6, B_T7, 0, 6, EPI_LINEAR, B_T7, ZSLACK>(wave - POOL8V_WAVES, lane);   // ... beside conv2d_12: T6 -> T7
    SYNC_PIX();
    FETCH(6);
    fill_halo<B_T8, true, NT>(lds, tid);
    dense_tile_stage<6, NW, (NW > 8 ? 3 : 5), 1, B_T7, B_T8, 0, 36, EPI_ACT, B_T8, ZSLACK>(wave, lane);     // conv2d_13
    SYNC();
    FETCH(7);
    conv3x3_stage<7, NW, 1, B_T8, B_T9, 36, true>(lds, tab, conv_at(7), wave, lane);                     // conv2d_15 (dw)
    SYNC_PIX();
    FETCH(8);
    dense_tile_stage<8, NW, 1, 5, B_T9, B_T11, 0, 6, EPI_ADD, B_T7, B_T8::OFF>(wave, lane);     // conv2d_17 + eltwise_18
    SYNC_PIX();
    FETCH(9);
    dense_tile_stage<9, NW, (NW > 8 ? 2 : 3), 1, B_T11, B_T14, 20, 18, EPI_ACT, B_T14, ZSLACK>(wave, lane); // conv2d_19 -> concat_22 conv half
    SYNC_PIX();
    FETCH(10);
    const bool more = k + 1 < nb;
    if (more) prefetch_in(fr + G);                                                                    // the next frame's input, behind this stage's weight DMA
    fill_halo<B_T15, false, NT>(lds, tid);
    dense_tile_stage<10, NW, (NW > 8 ? 1 : 2), 5, B_T14, B_T15, 0, 24, EPI_ACT, B_T15, B_T8::OFF>(wave, lane);  // conv2d_23
    if (more) { SYNC_KEEP_PREFETCH(); } else { SYNC(); }
    {   // pool_25 (by columns, on the first waves) and conv2d_27 (dw, stride 2, on the others) both only read T15; their outputs -- the two
        // inputs of the tail -- go straight to the frame's park slot in HBM (49 pixels x {24 | 24} channels)
      constexpr int PW = POOL25_WAVES;
      static_assert(PW < NW, "waves left for conv2d_27");
      char* slot = prm.scratch + ((long)blockIdx.x * NW + k) * PARK_BYTES;
#if !defined(YF16_STAGEPMC)
      // the batch's last frame: the tail phase's resident weights (21.5 KB) start their way into LDS here -- their place behind the exchange buffers is dead since
      // conv2d_23 (concat_22 / T11 territory), and the tail phase's first barrier waits for them -- instead of in front of that barrier with nothing to hide behind
      if (!more) fetch_tailw<NW, 0>(tab0, wave, lane);
#endif
      if (wave < PW) pool25_cols<B_T15, PARK>(lds, wave * 64 + lane, slot);
      else conv3x3_stage<11, NW - PW, 2, B_T15, PARK, 24, true, true, 48>(lds, tab, conv_at(11), wave - PW, lane, slot);
    }
    // This barrier orders LDS only (the next frame's staging overwrites what this stage still reads): it does not wait for the acknowledgements of the park
    // slot's global stores -- ~1 k cycles per frame at the end of a short stage (round 4: -1.5 %).  The slots are read in the tail phase, behind a barrier that
    // does wait (SYNC_BATCH); no weight DMA is in flight here (conv2d_27's block landed behind conv2d_23's barrier, the next one is issued by the next stage).
    SYNC_LDS();
    }
    {   // ---- tail phase: weights resident, exchange buffers zeroed, then wave w runs the whole tail of the batch's frame w
      int tid = tid0;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63;
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef YF16_BARPROF
      bar_no = 20; prof_on = prof_out != nullptr && base == (long)blockIdx.x;
#endif
#ifdef YF16_STAGEPMC
      if (prm.stop > 0) continue;                                 // counting a front stage: no tail phase
#endif
#if defined(YF16_STAGEPMC)
      fetch_tailw<NW, 2>(tab0, wave, lane);
#else
      fetch_tailw<NW, 1>(tab0, wave, lane);
#endif
      for (int i = tid; i < NW * XB / 16; i += NT) reinterpret_cast<uint4*>(lds)[i] = uint4{0u, 0u, 0u, 0u};
      SYNC_BATCH();
      if (wave < nb) tail_chain<NW>(wave * XB, prm.scratch + ((long)blockIdx.x * NW + wave) * PARK_BYTES, prm.out + (base + wave * G) * (7 * 7 * 18), lane);
      SYNC_LDS();                                                 // the next batch's clear overwrites the buffers (LDS order only: nothing here reads the logits just stored)
    }
  }
#undef SYNC
#undef SYNC_BATCH
#undef SYNC_LDS
#undef SYNC_KEEP_PREFETCH
#undef SYNC_PIX
#undef FETCH
}

}  // namespace yf16

// ---------------------------------------------------------------------------------------------- host side
namespace {
#define HIPCHK(ctx, call) do { hipError_t rc_ = (call); if (rc_ != hipSuccess) { \
    (ctx)->err = std::string(#call) + ": " + hipGetErrorString(rc_); return -1; } } while (0)
}

struct yf_fp16 {
  int device = 0;
  int cus = 0, wgs_per_cu = 2;
  uint8_t* d_tab = nullptr;
  yf_stream_scratch park; size_t park_region = 0;      // tail scratch: one region per launch stream (yf_stream_scratch.h)
  std::string err;
};

extern "C" {

const char* yf_fp16_error(const yf_fp16* c) { return c ? c->err.c_str() : "null context"; }

void yf_fp16_destroy(yf_fp16* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();                  // a launch still in flight finishes before its tables and park slots are freed
  if (c->d_tab) (void)hipFree(c->d_tab);
  c->park.release();
  delete c;
}

// yfw: the file written by tools/gen_fp16_model.py ('YFW1', 24 convs, fp32 weights OHWI / HWC + bias)
int yf_fp16_create(int device, const void* yfw, size_t bytes, yf_fp16** out, char* err, size_t errlen) {
  using yf16::half_t;
  auto fail = [&](const std::string& m) { if (err && errlen) snprintf(err, errlen, "%s", m.c_str()); return -1; };
  if (!yfw || bytes < 8 || !out || memcmp(yfw, "YFW1", 4)) return fail("not a YFW1 weight pack");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail("no HIP device");
  if (device < 0 || device >= ndev) return fail("device index out of range");
  const uint8_t* p = (const uint8_t*)yfw;
  uint32_t n; memcpy(&n, p + 4, 4);
  if (n != 24) return fail("expected 24 convolutions");
  static const int expect[24][5] = {   // depthwise, cin, cout, k, stride: the graph the fused kernel implements
      {0, 3, 8, 3, 2}, {1, 8, 8, 3, 1}, {0, 8, 4, 1, 1}, {0, 4, 18, 1, 1}, {1, 18, 18, 3, 2}, {0, 18, 6, 1, 1}, {0, 6, 36, 1, 1},
      {1, 36, 36, 3, 1}, {0, 36, 6, 1, 1}, {0, 6, 18, 1, 1}, {0, 36, 24, 1, 1}, {1, 24, 24, 3, 2}, {0, 24, 8, 1, 1}, {0, 8, 40, 1, 1},
      {1, 40, 40, 3, 1}, {0, 40, 8, 1, 1}, {0, 8, 40, 1, 1}, {1, 40, 40, 3, 1}, {0, 40, 8, 1, 1}, {0, 8, 24, 1, 1}, {0, 48, 40, 1, 1},
      {1, 40, 40, 3, 1}, {0, 40, 32, 1, 1}, {0, 32, 18, 1, 1}};
  // k-steps the kernel instantiates per dense layer (8 input channels each; the INPUT buffer's channel layout decides)
  static const int ksteps[24] = {5, 5, 1, 1, 5, 3, 1, 5, 5, 1, 5, 5, 3, 1, 5, 5, 1, 5, 5, 1, 6, 5, 5, 4};
  std::vector<uint8_t> blob(sizeof(yf16::Tables), 0);
  yf16::Tables T;
  auto alloc = [&](size_t nbytes) { size_t off = (blob.size() + 15) & ~(size_t)15; blob.resize(off + nbytes, 0); return off; };
  size_t off = 8;
  for (int i = 0; i < 24; ++i) {
    if (off + 24 > bytes) return fail("truncated weight pack");
    uint32_t h[6]; memcpy(h, p + off, 24); off += 24;
    for (int k = 0; k < 5; ++k) if ((int)h[k] != expect[i][k]) return fail("weight pack does not describe the yoloface graph");
    const int dw = (int)h[0], cin = (int)h[1], cout = (int)h[2], k = (int)h[3];
    const size_t nw = h[5];
    if (off + 4 * (nw + cout) > bytes) return fail("truncated weight pack");
    const float* wf = (const float*)(p + off); off += 4 * nw;
    const float* bf = (const float*)(p + off); off += 4 * (size_t)cout;
    const int cp = (cout + 3) & ~3;
    if (dw || k == 3) {
      // 3x3: per 4-output-channel group [5 k-steps][4 rows][8 halfs].  Depthwise: row j, tap t -> slot (t&1)*4 + j of k-step t>>1.
      // conv1 (dense, Cin 3 as RGBX): row j (output channel 4g+j), tap t, colour c -> slot (t&1)*4 + c.
      const int ng = cp / 4;
      const size_t w_off = alloc((size_t)ng * 5 * 4 * 8 * 2);
      half_t* W = (half_t*)(blob.data() + w_off);
      for (int g = 0; g < ng; ++g) for (int t = 0; t < 9; ++t) for (int j = 0; j < 4; ++j) {
        const int ch = 4 * g + j;
        if (ch >= cout) continue;
        half_t* row = W + (((size_t)g * 5 + (t >> 1)) * 4 + j) * 8;
        if (dw) row[(t & 1) * 4 + j] = (half_t)wf[(size_t)t * cout + ch];                               // HWC
        else for (int c = 0; c < cin; ++c) row[(t & 1) * 4 + c] = (half_t)wf[((size_t)ch * 9 + t) * cin + c];   // OHWI
      }
      T.conv[i].w_off = (uint32_t)w_off;
    } else {
      const int krow = 8 * ksteps[i];
      // input-channel position k of the weight row = channel k of the input buffer; concat_22's buffer holds the pool half at
      // [0,18) and the conv half at [20,38) (8-byte aligned stores), concat_46 at [0,24) | [24,48)
      const size_t w_off = alloc((size_t)cp * krow * 2);
      half_t* W = (half_t*)(blob.data() + w_off);
      for (int o = 0; o < cout; ++o) for (int c = 0; c < cin; ++c) {
        const int pos = (i == 10 && c >= 18) ? c + 2 : c;            // conv2d_23 reads concat_22's buffer: conv half at channel 20
        W[(size_t)o * krow + pos] = (half_t)wf[(size_t)o * cin + c];
      }
      T.conv[i].w_off = (uint32_t)w_off;
    }
    const size_t b_off = alloc((size_t)cp * 4);
    memcpy(blob.data() + b_off, bf, 4 * (size_t)cout);
    T.conv[i].b_off = (uint32_t)b_off;
    if (T.conv[i].w_off != yf16::conv_at(i).w_off || T.conv[i].b_off != yf16::conv_at(i).b_off) return fail("blob layout differs from the one compiled into the kernel");
  }
  blob.resize((blob.size() + 15 + 64) & ~(size_t)15, 0);          // zeroed tail: 16-byte reads of the last row stay in bounds
  memcpy(blob.data(), &T, sizeof T);
  yf_fp16* c = new yf_fp16();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) { delete c; return fail("hipSetDevice failed"); }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete c; return fail(std::string("unsupported GPU ") + prop.gcnArchName); }
  c->cus = prop.multiProcessorCount;
  c->wgs_per_cu = (int)(prop.maxSharedMemoryPerMultiProcessor / (size_t)yf16::LDS_TOTAL);      // two 82 KB workgroups in gfx950's 160 KB
  if (c->wgs_per_cu < 1) { delete c; return fail("the device's LDS per CU is smaller than one workgroup of the fp16 kernel"); }
  if (hipMalloc((void**)&c->d_tab, blob.size()) != hipSuccess || hipMemcpy(c->d_tab, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess ||
      hipFuncSetAttribute((const void*)yf16::yoloface56_f16_fused<YF16_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, yf16::LDS_TOTAL) != hipSuccess) {
    yf_fp16_destroy(c); return fail("uploading the fp16 tables failed");
  }
  c->park_region = (size_t)c->cus * c->wgs_per_cu * YF16_NW * yf16::PARK_BYTES;    // a park slot per wave of every workgroup; allocated per stream on first use
  *out = c;
  return 0;
}

int yf_fp16_release_stream(yf_fp16* c, void* stream) {
  if (!c) return -2;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->park.release_stream((hipStream_t)stream));
  return 0;
}
size_t yf_fp16_scratch_bytes(yf_fp16* c) { return c ? c->park.bytes_held() : 0; }
void yf_fp16_scratch_stats(yf_fp16* c, unsigned long long out[6]) {          // added to out (yf_engine_scratch_stats has the order)
  if (!c) return;
  const yf_stream_scratch::Stats s = c->park.stats();
  out[0] += s.events_recorded; out[1] += s.events_skipped; out[2] += s.event_waits; out[3] += s.device_syncs; out[4] += s.acquire_waits; out[5] += s.regions;
}

// d_in: fp16 [n][56][56][3] (pixel / 255), d_out: fp32 logits [n][7][7][18]
int yf_fp16_run_device(yf_fp16* c, const void* d_in, void* d_out, long n, void* stream) {
  if (!c || !d_in || !d_out || n < 0) return -2;
  if (n == 0) return 0;
  if (((uintptr_t)d_in & 3) != 0) { c->err = "fp16 input must be 4-byte aligned"; return -2; }
  HIPCHK(c, hipSetDevice(c->device));
  yf16::Params prm;
  prm.in = (const yf16::half_t*)d_in; prm.out = (float*)d_out; prm.n = n; prm.tab = c->d_tab; prm.prof = nullptr; prm.stop = 0;
#ifdef YF16_STAGEPMC
  if (getenv("YF16_STOP_STAGE")) prm.stop = atoi(getenv("YF16_STOP_STAGE"));
#endif
  yf_stream_scratch::Lease lease;                            // marks its region on every way out (yf_stream_scratch.h)
  HIPCHK(c, c->park.get((hipStream_t)stream, c->park_region, &lease));
  prm.scratch = lease.ptr;
  long grid = (long)c->cus * c->wgs_per_cu;                  // two workgroups per CU (LDS), persistent over the frames
  if (grid > n) grid = n;
#ifdef YF16_BARPROF
  if (getenv("YF16_ONE_WG_PER_CU")) grid = c->cus < n ? c->cus : n;       // dev builds: one workgroup per CU (how much do two share?)
  const char* prof_path = getenv("YF16_PROF_OUT");               // dev builds only: the stage timeline of this launch goes to a file
  const size_t prof_bytes = (size_t)grid * 8 * 80 * sizeof(long long);
  if (prof_path) { HIPCHK(c, hipMalloc((void**)&prm.prof, prof_bytes)); HIPCHK(c, hipMemset(prm.prof, 0, prof_bytes)); }
#endif
  hipLaunchKernelGGL(yf16::yoloface56_f16_fused<YF16_NW>, dim3((unsigned)grid), dim3(YF16_NW * 64), yf16::LDS_TOTAL, (hipStream_t)stream, prm);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, lease.mark());                                    // the region is busy until this launch has completed (yf_stream_scratch.h)
#ifdef YF16_BARPROF
  if (prof_path) {
    std::vector<long long> h(prof_bytes / sizeof(long long));
    HIPCHK(c, hipMemcpy(h.data(), prm.prof, prof_bytes, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(prof_path, "wb")) { fwrite(h.data(), 1, prof_bytes, f); fclose(f); }
    (void)hipFree(prm.prof);
  }
#endif
  return 0;
}

}  // extern "C"
