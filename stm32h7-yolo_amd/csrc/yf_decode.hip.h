// Box decode shared by the stand-alone decode kernel and the fused network kernel (decode of the heads while they are
// still in LDS).  Included once by yf_engine.hip, before the kernel header.
//   YF_DECODE_PY: yoloface/tflite/tflite_prediction.py:42-63  (a, row, col), conf > 0.7
//   YF_DECODE_FW: stm32/X-CUBE-AI/App/yoloface.c:98-152       (cell, a), conf >= 0.7, LCD axis swap, clamp, x2
//                 float -> int saturates like the Cortex-M7's VCVT; YF_DECODE_FW_HOST converts like an x86-64 build
// All transcendental values come from the committed float32 tables; the remaining float32 operations are single
// IEEE operations (the engine is compiled with -ffp-contract=off).
#ifndef YF_DECODE_HIP_H
#define YF_DECODE_HIP_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/yf_network.h"   // yf_det, YF_DECODE_*

namespace yfdec {

__device__ __constant__ uint32_t d_sig_bits[256];
__device__ __constant__ uint32_t d_exp_bits[256];

// float -> int32 with the x86 convention of the reference's hosts (cvttss2si): truncate, out of range -> INT32_MIN
__device__ __forceinline__ int f2i_x86(float v) {
  return (v > -2147483904.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
}
// float -> int32 as the Cortex-M7 does it (VCVT.S32.F32): truncate, saturate, NaN -> 0
__device__ __forceinline__ int f2i_sat(float v) {
  return v != v ? 0 : (v >= 2147483648.0f ? 0x7FFFFFFF : (v <= -2147483648.0f ? (int)0x80000000 : (int)v));
}
__device__ __forceinline__ int dbl_wrap(int v) { return (int)((unsigned)v * 2u); }

// One wave per frame; candidates are visited in the reference order and compacted with ballots so that the record
// order equals the order the reference loops produce.  `head` may point to LDS or to global memory.
__device__ __forceinline__ void decode_frame(const int8_t* head, long frame, int lane, int mode, float w_scale, float h_scale,
                                             yf_det* __restrict__ dets, int* __restrict__ counts, int cap) {
  yf_det* out = dets + frame * cap;
  const float anc_w[3] = {9.f, 12.f, 22.f}, anc_h[3] = {14.f, 17.f, 21.f};
  int total = 0;
  for (int base = 0; base < 147; base += 64) {
    const int i = base + lane;
    bool keep = false;
    int a = 0, row = 0, col = 0;
    const int8_t* p = head;
    float conf = 0.f;
    if (i < 147) {
      if (mode == YF_DECODE_PY) { a = i / 49; const int cell = i - a * 49; row = cell / 7; col = cell - row * 7; }
      else { const int cell = i / 3; a = i - cell * 3; row = cell / 7; col = cell - row * 7; }
      p = head + (row * 7 + col) * 18 + a * 6;
      conf = __uint_as_float(d_sig_bits[p[4] + 128]);
      keep = (mode == YF_DECODE_PY) ? (conf > 0.7f) : ((double)conf >= 0.7);
    }
    const unsigned long long mask = __ballot(keep);
    const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
    if (keep && pos < cap) {
      const float sx = __uint_as_float(d_sig_bits[p[0] + 128]), sy = __uint_as_float(d_sig_bits[p[1] + 128]);
      const float ew = __uint_as_float(d_exp_bits[p[2] + 128]), eh = __uint_as_float(d_exp_bits[p[3] + 128]);
      yf_det d;
      d.frame = (int32_t)frame; d.anchor = (uint8_t)a; d.row = (uint8_t)row; d.col = (uint8_t)col;
      d.q_conf = p[4]; d.conf = conf;
      const float cx = (sx + (float)col) * 8.f, cy = (sy + (float)row) * 8.f;
      const float bw = ew * anc_w[a], bh = eh * anc_h[a];
      if (mode == YF_DECODE_PY) {
        float x1 = cx - bw / 2, y1 = cy - bh / 2, x2 = cx + bw / 2, y2 = cy + bh / 2;
        x1 *= w_scale; x2 *= w_scale; y1 *= h_scale; y2 *= h_scale;
        d.x1 = f2i_x86(x1); d.y1 = f2i_x86(y1); d.x2 = f2i_x86(x2); d.y2 = f2i_x86(y2);
      } else {
        const float fy2 = cx - bw / 2, fy1 = cx + bw / 2, fx1 = cy - bh / 2, fx2 = cy + bh / 2;
        const bool arm = mode == YF_DECODE_FW;
        int y2 = arm ? f2i_sat(fy2) : f2i_x86(fy2), y1 = arm ? f2i_sat(fy1) : f2i_x86(fy1);
        int x1 = arm ? f2i_sat(fx1) : f2i_x86(fx1), x2 = arm ? f2i_sat(fx2) : f2i_x86(fx2);
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > 55) x2 = 55;
        if (y2 > 55) y2 = 55;
        d.x1 = dbl_wrap(x1); d.y1 = dbl_wrap(y1); d.x2 = dbl_wrap(x2); d.y2 = dbl_wrap(y2);
      }
      out[pos] = d;
    }
    total += __popcll(mask);
  }
  if (lane == 0) counts[frame] = total;
}

// The same decode for the fused kernel's tail-batching build: both tables sit in LDS (2 KB copied by LDS-DMA into bytes of the arena that are
// dead by then: sig at lds_tabs, exp 1 KB behind it) and the three 64-candidate chunks of a frame run their first phase -- confidence byte,
// table look-up, ballot -- back to back before any record is assembled.  With the tables in global memory every chunk was two dependent
// global round trips, six in a row per frame on one wave (2.7 us per launch of 4096 frames; tools/probe/decode_cost.py).
typedef const __attribute__((address_space(3))) uint32_t* dec_lds_u32;
__device__ __forceinline__ void decode_frame_lds(const int8_t* head, long frame, int lane, int mode, float w_scale, float h_scale,
                                                 yf_det* __restrict__ dets, int* __restrict__ counts, int cap, uint32_t lds_tabs, int q_thr) {
  yf_det* out = dets + frame * cap;
  const float anc_w[3] = {9.f, 12.f, 22.f}, anc_h[3] = {14.f, 17.f, 21.f};
  auto sig = [&](int q) { return __uint_as_float(*(dec_lds_u32)(lds_tabs + 4u * (uint32_t)(q + 128))); };
  auto ex = [&](int q) { return __uint_as_float(*(dec_lds_u32)(lds_tabs + 1024u + 4u * (uint32_t)(q + 128))); };
  // phase 1: the confidence test on the QUANTISED value (the sigmoid table is monotonic: conf > 0.7 <=> q >= q_thr, the first table entry that
  // passes the mode's comparison -- found by the engine on the host, yf_decode_q_threshold) and the candidate's byte offset; (anchor, row, column)
  // and everything else only for the records that are written
  int pos[3], off[3], q4[3];
  bool keep[3];
  int total = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int i = min(64 * c + lane, 146);
    if (mode == YF_DECODE_PY) { const int a = (i >= 49) + (i >= 98); off[c] = (i - 49 * a) * 18 + 6 * a; }
    else off[c] = 6 * i;
    q4[c] = head[off[c] + 4];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    keep[c] = (64 * c + lane < 147) && q4[c] >= q_thr;
    const unsigned long long mask = __ballot(keep[c]);
    pos[c] = total + __popcll(mask & ((1ull << lane) - 1ull));
    total += __popcll(mask);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (keep[c] && pos[c] < cap) {
      const int i = 64 * c + lane;
      int a, row, col;
      if (mode == YF_DECODE_PY) { a = i / 49; const int cell = i - a * 49; row = cell / 7; col = cell - row * 7; }
      else { const int cell = i / 3; a = i - cell * 3; row = cell / 7; col = cell - row * 7; }
      const int8_t* q = head + off[c];
      const float sx = sig(q[0]), sy = sig(q[1]), ew = ex(q[2]), eh = ex(q[3]);
      yf_det d;
      d.frame = (int32_t)frame; d.anchor = (uint8_t)a; d.row = (uint8_t)row; d.col = (uint8_t)col;
      d.q_conf = (int8_t)q4[c]; d.conf = sig(q4[c]);
      const float cx = (sx + (float)col) * 8.f, cy = (sy + (float)row) * 8.f;
      const float bw = ew * anc_w[a], bh = eh * anc_h[a];
      if (mode == YF_DECODE_PY) {
        float x1 = cx - bw / 2, y1 = cy - bh / 2, x2 = cx + bw / 2, y2 = cy + bh / 2;
        x1 *= w_scale; x2 *= w_scale; y1 *= h_scale; y2 *= h_scale;
        d.x1 = f2i_x86(x1); d.y1 = f2i_x86(y1); d.x2 = f2i_x86(x2); d.y2 = f2i_x86(y2);
      } else {
        const float fy2 = cx - bw / 2, fy1 = cx + bw / 2, fx1 = cy - bh / 2, fx2 = cy + bh / 2;
        const bool arm = mode == YF_DECODE_FW;
        int y2 = arm ? f2i_sat(fy2) : f2i_x86(fy2), y1 = arm ? f2i_sat(fy1) : f2i_x86(fy1);
        int x1 = arm ? f2i_sat(fx1) : f2i_x86(fx1), x2 = arm ? f2i_sat(fx2) : f2i_x86(fx2);
        if (x1 < 0) x1 = 0;
        if (y1 < 0) y1 = 0;
        if (x2 > 55) x2 = 55;
        if (y2 > 55) y2 = 55;
        d.x1 = dbl_wrap(x1); d.y1 = dbl_wrap(y1); d.x2 = dbl_wrap(x2); d.y2 = dbl_wrap(y2);
      }
      out[pos[c]] = d;
    }
  }
  if (lane == 0) counts[frame] = total;
}

}  // namespace yfdec
#endif
