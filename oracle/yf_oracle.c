/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see yf_oracle.h).  PARITY UNPINNED (no reference goldens exist).
 *
 * Each kernel below restates one TFLite 2.10 builtin reference kernel (the third-party arithmetic behind
 * reference yoloface/tflite/tflite_prediction.py:23-41); the ST twin of each op is cited from
 * reference stm32/X-CUBE-AI/App/network.c.  Plain scalar C, no intrinsics: it doubles as the CPU baseline
 * ("port") that bench.py times next to the GPU.
 */
#include "yf_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { OP_ADD = 0, OP_CONCAT = 2, OP_CONV = 3, OP_DWCONV = 4, OP_MAXPOOL = 17, OP_PAD = 34, OP_LEAKY = 98,
       OP_QUANTIZE = 114 };

typedef struct {
  int32_t shape[4];
  uint32_t type;          /* 0 int8, 1 int32 */
  int32_t zero_point;
  uint32_t n_scales, scales_off;
  int32_t qdim;
  uint32_t data_off, data_bytes;
} yfm_tensor;

typedef struct {
  uint32_t opcode;
  int32_t in[3];
  int32_t out;
  int32_t padding, stride_w, stride_h, filter_w, filter_h, depth_multiplier, axis;
  uint32_t alpha_bits;
} yfm_op;

/* parameters prepared once per op (what TFLite computes in Prepare()) */
typedef struct {
  int32_t* mult;          /* per output channel (conv) */
  int* shift;
  float* fscale;          /* per output channel, float32 arithmetic: s_in * s_w / s_out (variant FP32 only) */
  int32_t m_id, m_alpha;  /* leaky */
  int s_id, s_alpha;
  int32_t m1, m2, mo;     /* add */
  int s1, s2, so;
  int32_t mq;             /* quantize */
  int sq;
} op_prep;

struct yfo_model {
  uint32_t n_tensors, n_ops, input, output, data_bytes;
  yfm_tensor* t;
  yfm_op* op;
  uint8_t* data;
  op_prep* prep;
};

/* ------------------------------------------------------------------ fixed-point primitives */
/* tensorflow/lite/kernels/internal/quantization_util.cc QuantizeMultiplier (default, non single-rounding build) */
void yfo_quantize_multiplier(double d, int32_t* m, int* shift) {
  if (d == 0.) { *m = 0; *shift = 0; return; }
  const double q = frexp(d, shift);
  int64_t q_fixed = (int64_t)round(q * (double)(1LL << 31));
  if (q_fixed == (1LL << 31)) { q_fixed /= 2; ++*shift; }
  if (*shift < -31) { *shift = 0; q_fixed = 0; }
  *m = (int32_t)q_fixed;
}

/* gemmlowp SaturatingRoundingDoublingHighMul == arm_nn_sat_doubling_high_mult (arm_nnsupportfunctions.h:210-235) */
int32_t yfo_srdhm(int32_t a, int32_t b) {
  if (a == INT32_MIN && b == INT32_MIN) return INT32_MAX;
  const int64_t ab = (int64_t)a * (int64_t)b;
  const int32_t nudge = ab >= 0 ? (1 << 30) : (1 - (1 << 30));
  return (int32_t)((ab + nudge) / (1LL << 31));       /* C division: truncates toward zero */
}

/* gemmlowp RoundingDivideByPOT == arm_nn_divide_by_power_of_two (arm_nnsupportfunctions.h:242-268) */
int32_t yfo_rdivpot(int32_t x, int exponent) {
  const int32_t mask = (int32_t)((1LL << exponent) - 1);
  const int32_t remainder = x & mask;
  const int32_t threshold = (mask >> 1) + (x < 0 ? 1 : 0);
  return (x >> exponent) + (remainder > threshold ? 1 : 0);
}

/* tensorflow/lite/kernels/internal/common.h MultiplyByQuantizedMultiplier */
int32_t yfo_mbqm(int32_t x, int32_t m, int shift) {
  const int left_shift = shift > 0 ? shift : 0;
  const int right_shift = shift > 0 ? 0 : -shift;
  return yfo_rdivpot(yfo_srdhm(x * (1 << left_shift), m), right_shift);
}

/* ---- rounding VARIANTS of the requantisation (yf_oracle.h, YFO_RV_*): other published arithmetics of the same step.
 * The restatement above (RoundingDivideByPOT: ties away from zero) is what TFLite's builtin REFERENCE kernels compute and
 * what SURVEY.md 8(c).3 fixes as "the tflite int8 reference".  tflite_prediction.py:23 builds tf.lite.Interpreter with
 * default arguments, i.e. the default (optimized) resolver, whose per-channel int8 CONV_2D GEMMs go through ruy:
 *   UP     : ruy's vector kernels = ARM sqrdmulh + srshl: the right shift breaks ties UPWARD, (s + 2^(e-1)) >> e;
 *   SINGLE : ruy's standard C++ path (ruy/apply_multiplier.cc): ONE rounding, (x*M + 2^(30-shift)) >> (31-shift);
 *   FP32   : XNNPACK's qs8 requantisation: lrintf((float)acc * (float)(s_in*s_w/s_out)).
 * None of them can be checked against an interpreter here; they exist to MEASURE how much the choice matters
 * (tests/test_oracle.py::test_rounding_variant_exposure) and as the checker of the library's selectable rounding. */
enum { MB_REF = 0, MB_UP = 1, MB_SINGLE = 2 };
int32_t yfo_mbqm_mode(int32_t x, int32_t m, int shift, int mode) {
  if (mode == MB_REF) return yfo_mbqm(x, m, shift);
  const int left_shift = shift > 0 ? shift : 0;
  const int right_shift = shift > 0 ? 0 : -shift;
  if (mode == MB_UP) {
    const int32_t s = yfo_srdhm(x * (1 << left_shift), m);
    return right_shift ? (int32_t)(((int64_t)s + (1LL << (right_shift - 1))) >> right_shift) : s;
  }
  const int total = 31 - shift;                            /* MB_SINGLE */
  return (int32_t)(((int64_t)x * (int64_t)m + (1LL << (total - 1))) >> total);
}
/* which form each op family takes under a variant: [dense conv, depthwise conv, element-wise (LEAKY_RELU, ADD, QUANTIZE)] */
static inline int mode_dense(int v) { return v == YFO_RV_UP_DENSE || v == YFO_RV_UP_ALL ? MB_UP : v == YFO_RV_SINGLE_DENSE ? MB_SINGLE : MB_REF; }
static inline int mode_dw(int v)    { return v == YFO_RV_UP_ALL ? MB_UP : MB_REF; }
static inline int mode_elt(int v)   { return v == YFO_RV_UP_ALL ? MB_UP : MB_REF; }

static inline int8_t clamp8(int32_t v) { return (int8_t)(v < -128 ? -128 : (v > 127 ? 127 : v)); }
static inline float bits2f(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline float tscale(const yfo_model* m, int ti, int k) {
  float f; memcpy(&f, m->data + m->t[ti].scales_off + 4 * (size_t)k, 4); return f;
}

/* ------------------------------------------------------------------ model pack */
yfo_model* yfo_load(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  char magic[4];
  uint32_t h[5];
  if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "YFM1", 4) || fread(h, 4, 5, f) != 5) { fclose(f); return NULL; }
  yfo_model* m = (yfo_model*)calloc(1, sizeof *m);
  m->n_tensors = h[0]; m->n_ops = h[1]; m->input = h[2]; m->output = h[3]; m->data_bytes = h[4];
  m->t = (yfm_tensor*)malloc(sizeof(yfm_tensor) * m->n_tensors);
  m->op = (yfm_op*)malloc(sizeof(yfm_op) * m->n_ops);
  m->data = (uint8_t*)malloc(m->data_bytes + 4);
  m->prep = (op_prep*)calloc(m->n_ops, sizeof(op_prep));
  int ok = fread(m->t, sizeof(yfm_tensor), m->n_tensors, f) == m->n_tensors &&
           fread(m->op, sizeof(yfm_op), m->n_ops, f) == m->n_ops &&
           fread(m->data, 1, m->data_bytes, f) == m->data_bytes;
  fclose(f);
  if (!ok) { yfo_free(m); return NULL; }

  for (uint32_t i = 0; i < m->n_ops; ++i) {
    const yfm_op* o = &m->op[i];
    op_prep* p = &m->prep[i];
    const float s_out = tscale(m, o->out, 0);
    if (o->opcode == OP_CONV || o->opcode == OP_DWCONV) {
      /* kernel_util.cc PopulateConvolutionQuantizationParams: per-channel effective scale in double */
      const int wt = o->in[1];
      const int n = (int)m->t[wt].n_scales;
      const float s_in = tscale(m, o->in[0], 0);
      p->mult = (int32_t*)malloc(sizeof(int32_t) * n);
      p->shift = (int*)malloc(sizeof(int) * n);
      p->fscale = (float*)malloc(sizeof(float) * n);
      for (int c = 0; c < n; ++c) {
        const double eff = (double)s_in * (double)tscale(m, wt, c) / (double)s_out;
        yfo_quantize_multiplier(eff, &p->mult[c], &p->shift[c]);
        const volatile float num = s_in * tscale(m, wt, c);                 /* two float32 operations, no contraction */
        p->fscale[c] = num / s_out;
      }
    } else if (o->opcode == OP_LEAKY) {
      /* activations.cc LeakyReluPrepare: float expressions widened to double */
      const float s_in = tscale(m, o->in[0], 0);
      const float alpha = bits2f(o->alpha_bits);
      const double alpha_multiplier = (double)(float)(s_in * alpha / s_out);
      const double identity_multiplier = (double)(float)(s_in / s_out);
      yfo_quantize_multiplier(alpha_multiplier, &p->m_alpha, &p->s_alpha);
      yfo_quantize_multiplier(identity_multiplier, &p->m_id, &p->s_id);
    } else if (o->opcode == OP_ADD) {
      /* add.cc Prepare (int8): left_shift 20, QuantizeMultiplierSmallerThanOneExp */
      const float s1 = tscale(m, o->in[0], 0), s2 = tscale(m, o->in[1], 0);
      const double twice_max = (double)(2 * (s1 > s2 ? s1 : s2));
      const double r1 = (double)s1 / twice_max, r2 = (double)s2 / twice_max;
      const double ro = twice_max / (double)((float)(1 << 20) * s_out);
      yfo_quantize_multiplier(r1, &p->m1, &p->s1);
      yfo_quantize_multiplier(r2, &p->m2, &p->s2);
      yfo_quantize_multiplier(ro, &p->mo, &p->so);
    } else if (o->opcode == OP_QUANTIZE) {
      /* quantize.cc Prepare: effective scale = s_in / s_out in double */
      const double eff = (double)tscale(m, o->in[0], 0) / (double)s_out;
      yfo_quantize_multiplier(eff, &p->mq, &p->sq);
    }
  }
  return m;
}

void yfo_free(yfo_model* m) {
  if (!m) return;
  if (m->prep) for (uint32_t i = 0; i < m->n_ops; ++i) { free(m->prep[i].mult); free(m->prep[i].shift); free(m->prep[i].fscale); }
  free(m->prep); free(m->t); free(m->op); free(m->data); free(m);
}

int yfo_num_ops(const yfo_model* m) { return (int)m->n_ops; }

/* ------------------------------------------------------------------ shape inference (fully convolutional) */
typedef struct { int h, w, c; } shp;

static void same_or_valid(int padding, int in, int k, int stride, int* out, int* pad) {
  if (padding == 0) {            /* SAME */
    *out = (in + stride - 1) / stride;
    int total = (*out - 1) * stride + k - in;
    if (total < 0) total = 0;
    *pad = total / 2;
  } else {                       /* VALID */
    *out = (in - k + stride) / stride;
    *pad = 0;
  }
}

static void infer_shapes(const yfo_model* m, int h, int w, shp* s) {
  for (uint32_t i = 0; i < m->n_tensors; ++i) { s[i].h = s[i].w = s[i].c = 0; }
  s[m->input].h = h; s[m->input].w = w; s[m->input].c = m->t[m->input].shape[3];
  for (uint32_t i = 0; i < m->n_ops; ++i) {
    const yfm_op* o = &m->op[i];
    const shp a = s[o->in[0]];
    shp r = a;
    int pad;
    switch (o->opcode) {
      case OP_PAD: {
        const int32_t* p = (const int32_t*)(m->data + m->t[o->in[1]].data_off);   /* [4][2] */
        r.h = a.h + p[2] + p[3]; r.w = a.w + p[4] + p[5];
      } break;
      case OP_CONV: {
        const yfm_tensor* wt = &m->t[o->in[1]];                                      /* OHWI */
        same_or_valid(o->padding, a.h, wt->shape[1], o->stride_h, &r.h, &pad);
        same_or_valid(o->padding, a.w, wt->shape[2], o->stride_w, &r.w, &pad);
        r.c = wt->shape[0];
      } break;
      case OP_DWCONV: {
        const yfm_tensor* wt = &m->t[o->in[1]];                                      /* 1HWC */
        same_or_valid(o->padding, a.h, wt->shape[1], o->stride_h, &r.h, &pad);
        same_or_valid(o->padding, a.w, wt->shape[2], o->stride_w, &r.w, &pad);
        r.c = wt->shape[3];
      } break;
      case OP_MAXPOOL:
        same_or_valid(o->padding, a.h, o->filter_h, o->stride_h, &r.h, &pad);
        same_or_valid(o->padding, a.w, o->filter_w, o->stride_w, &r.w, &pad);
        break;
      case OP_CONCAT:
        r.c = a.c + s[o->in[1]].c;
        break;
      default: break;
    }
    s[o->out] = r;
  }
}

long yfo_dump_bytes(const yfo_model* m, int h, int w) {
  shp* s = (shp*)malloc(sizeof(shp) * m->n_tensors);
  infer_shapes(m, h, w, s);
  long tot = 0;
  for (uint32_t i = 0; i < m->n_ops; ++i) { shp r = s[m->op[i].out]; tot += (long)r.h * r.w * r.c; }
  free(s);
  return tot;
}

void yfo_out_shape(const yfo_model* m, int h, int w, int* oh, int* ow, int* oc) {
  shp* s = (shp*)malloc(sizeof(shp) * m->n_tensors);
  infer_shapes(m, h, w, s);
  *oh = s[m->output].h; *ow = s[m->output].w; *oc = s[m->output].c;
  free(s);
}

/* ------------------------------------------------------------------ kernels (one per TFLite builtin) */

/* reference_ops::Pad, int8 without constant_values: fill with the output zero point.
 * ST folds this into conv2d_1/10/27 as filter_pad (1,1,0,0) (network.c:2592,2823,2925). */
static void k_pad(const int8_t* in, shp a, int8_t* out, shp r, int top, int left, int8_t pad_value) {
  memset(out, pad_value, (size_t)r.h * r.w * r.c);
  for (int y = 0; y < a.h; ++y)
    memcpy(out + ((size_t)(y + top) * r.w + left) * r.c, in + (size_t)y * a.w * a.c, (size_t)a.w * a.c);
}

/* the requantisation of one conv accumulator under a variant: mode = MB_* or -1 for FP32 */
static inline int32_t conv_requant(int32_t acc, int c, const op_prep* p, int mode) {
  if (mode >= 0) return yfo_mbqm_mode(acc, p->mult[c], p->shift[c], mode);
  return (int32_t)lrintf((float)acc * p->fscale[c]);                        /* round to nearest, ties to even */
}

/* reference_integer_ops::ConvPerChannel (int8).  ST: forward_conv2d_integer_SSSA_ch, conv2d_1/5/6/... */
static void k_conv(const int8_t* in, shp a, const int8_t* w, int kh, int kw, const int32_t* bias,
                   int stride, int pad_h, int pad_w, int32_t in_zp, int32_t out_zp,
                   const op_prep* p, int mode, int8_t* out, shp r) {
  for (int oy = 0; oy < r.h; ++oy)
    for (int ox = 0; ox < r.w; ++ox)
      for (int oc = 0; oc < r.c; ++oc) {
        int32_t acc = 0;
        for (int fy = 0; fy < kh; ++fy) {
          const int iy = oy * stride - pad_h + fy;
          if (iy < 0 || iy >= a.h) continue;
          for (int fx = 0; fx < kw; ++fx) {
            const int ix = ox * stride - pad_w + fx;
            if (ix < 0 || ix >= a.w) continue;
            const int8_t* ip = in + ((size_t)iy * a.w + ix) * a.c;
            const int8_t* wp = w + (((size_t)oc * kh + fy) * kw + fx) * a.c;
            for (int ic = 0; ic < a.c; ++ic) acc += (int32_t)wp[ic] * ((int32_t)ip[ic] - in_zp);
          }
        }
        acc += bias[oc];
        acc = conv_requant(acc, oc, p, mode) + out_zp;
        out[((size_t)oy * r.w + ox) * r.c + oc] = clamp8(acc);
      }
}

/* reference_integer_ops::DepthwiseConvPerChannel (int8, depth_multiplier 1).  ST: same fn with .groups=C. */
static void k_dwconv(const int8_t* in, shp a, const int8_t* w, int kh, int kw, const int32_t* bias,
                     int stride, int pad_h, int pad_w, int32_t in_zp, int32_t out_zp,
                     const op_prep* p, int mode, int8_t* out, shp r) {
  for (int oy = 0; oy < r.h; ++oy)
    for (int ox = 0; ox < r.w; ++ox)
      for (int c = 0; c < r.c; ++c) {
        int32_t acc = 0;
        for (int fy = 0; fy < kh; ++fy) {
          const int iy = oy * stride - pad_h + fy;
          if (iy < 0 || iy >= a.h) continue;
          for (int fx = 0; fx < kw; ++fx) {
            const int ix = ox * stride - pad_w + fx;
            if (ix < 0 || ix >= a.w) continue;
            acc += (int32_t)w[((size_t)fy * kw + fx) * r.c + c] * ((int32_t)in[((size_t)iy * a.w + ix) * a.c + c] - in_zp);
          }
        }
        acc += bias[c];
        acc = conv_requant(acc, c, p, mode) + out_zp;
        out[((size_t)oy * r.w + ox) * r.c + c] = clamp8(acc);
      }
}

/* reference_ops::QuantizeLeakyRelu<int8_t>.  ST uses a float-rounded LUT instead (nl_func_array_integer,
 * network.c:2218..2902) and differs in 11-22 entries per layer -- TFLite semantics are the oracle. */
static inline int8_t leaky1(int8_t q, int32_t in_zp, int32_t out_zp, const op_prep* p, int mode) {
  const int32_t v = (int32_t)q - in_zp;
  const int32_t u = v >= 0 ? yfo_mbqm_mode(v, p->m_id, p->s_id, mode) : yfo_mbqm_mode(v, p->m_alpha, p->s_alpha, mode);
  return clamp8(out_zp + u);
}

/* reference_integer_ops::MaxPool (int8): padding never wins.  ST: forward_mp_integer_INT8, pool_8/pool_25. */
static void k_maxpool(const int8_t* in, shp a, int k_h, int k_w, int stride, int pad_h, int pad_w,
                      int8_t* out, shp r) {
  for (int oy = 0; oy < r.h; ++oy)
    for (int ox = 0; ox < r.w; ++ox) {
      const int y0 = oy * stride - pad_h, x0 = ox * stride - pad_w;
      const int fy0 = y0 < 0 ? -y0 : 0, fx0 = x0 < 0 ? -x0 : 0;
      const int fy1 = (k_h < a.h - y0) ? k_h : a.h - y0, fx1 = (k_w < a.w - x0) ? k_w : a.w - x0;
      for (int c = 0; c < r.c; ++c) {
        int8_t mx = -128;
        for (int fy = fy0; fy < fy1; ++fy)
          for (int fx = fx0; fx < fx1; ++fx) {
            const int8_t v = in[((size_t)(y0 + fy) * a.w + (x0 + fx)) * a.c + c];
            if (v > mx) mx = v;
          }
        out[((size_t)oy * r.w + ox) * r.c + c] = mx;     /* activation range = full int8 */
      }
    }
}

/* reference_integer_ops::Add (int8).  ST: forward_eltwise_integer_INT8 with float scales (eltwise_18/35/41). */
static inline int8_t add1(int8_t q1, int8_t q2, int32_t zp1, int32_t zp2, int32_t zpo, const op_prep* p, int mode) {
  const int32_t a = ((int32_t)q1 - zp1) * (1 << 20);
  const int32_t b = ((int32_t)q2 - zp2) * (1 << 20);
  const int32_t sa = yfo_mbqm_mode(a, p->m1, p->s1, mode);
  const int32_t sb = yfo_mbqm_mode(b, p->m2, p->s2, mode);
  return clamp8(yfo_mbqm_mode(sa + sb, p->mo, p->so, mode) + zpo);
}

/* reference_ops::Requantize int8->int8 (QUANTIZE).  ST folds these into concat_22/46 ("conversion_21/44/45"). */
static inline int8_t requant1(int8_t q, int32_t in_zp, int32_t out_zp, const op_prep* p, int mode) {
  return clamp8(yfo_mbqm_mode((int32_t)q - in_zp, p->mq, p->sq, mode) + out_zp);
}

int yfo_leaky_lut(const yfo_model* m, int op_index, int8_t lut[256]) { return yfo_leaky_lut_variant(m, op_index, lut, YFO_RV_REF); }

int yfo_leaky_lut_variant(const yfo_model* m, int op_index, int8_t lut[256], int variant) {
  if (variant < 0 || variant >= YFO_RV_COUNT) return -2;
  if (op_index < 0 || op_index >= (int)m->n_ops || m->op[op_index].opcode != OP_LEAKY) return -1;
  const yfm_op* o = &m->op[op_index];
  for (int q = -128; q < 128; ++q)
    lut[q + 128] = leaky1((int8_t)q, m->t[o->in[0]].zero_point, m->t[o->out].zero_point, &m->prep[op_index], mode_elt(variant));
  return 0;
}

/* ------------------------------------------------------------------ graph runner (one frame) */
static void run_frame(const yfo_model* m, const shp* s, const int8_t* in, int8_t* out, int8_t* dump,
                      int8_t** buf, int variant) {
  const int fp32 = variant == YFO_RV_FP32;
  const int md = fp32 ? -1 : mode_dense(variant), mw = fp32 ? -1 : mode_dw(variant), me = mode_elt(variant);
  /* buf[t]: scratch for tensor t (allocated by caller, sized from s) */
  const shp si = s[m->input];
  memcpy(buf[m->input], in, (size_t)si.h * si.w * si.c);
  long doff = 0;
  for (uint32_t i = 0; i < m->n_ops; ++i) {
    const yfm_op* o = &m->op[i];
    const op_prep* p = &m->prep[i];
    const shp a = s[o->in[0]], r = s[o->out];
    const int8_t* x = buf[o->in[0]];
    int8_t* y = buf[o->out];
    const int32_t zi = m->t[o->in[0]].zero_point, zo = m->t[o->out].zero_point;
    switch (o->opcode) {
      case OP_PAD: {
        const int32_t* pd = (const int32_t*)(m->data + m->t[o->in[1]].data_off);
        k_pad(x, a, y, r, pd[2], pd[4], (int8_t)zo);
      } break;
      case OP_CONV: case OP_DWCONV: {
        const yfm_tensor* wt = &m->t[o->in[1]];
        const int8_t* w = (const int8_t*)(m->data + wt->data_off);
        const int32_t* b = (const int32_t*)(m->data + m->t[o->in[2]].data_off);
        const int kh = wt->shape[1], kw = wt->shape[2];
        int oh, ow, ph, pw;
        same_or_valid(o->padding, a.h, kh, o->stride_h, &oh, &ph);
        same_or_valid(o->padding, a.w, kw, o->stride_w, &ow, &pw);
        if (o->opcode == OP_CONV) k_conv(x, a, w, kh, kw, b, o->stride_h, ph, pw, zi, zo, p, md, y, r);
        else k_dwconv(x, a, w, kh, kw, b, o->stride_h, ph, pw, zi, zo, p, mw, y, r);
      } break;
      case OP_LEAKY: {
        int8_t lut[256];
        for (int q = -128; q < 128; ++q) lut[q + 128] = leaky1((int8_t)q, zi, zo, p, me);
        const size_t n = (size_t)r.h * r.w * r.c;
        for (size_t k = 0; k < n; ++k) y[k] = lut[(int)x[k] + 128];
      } break;
      case OP_MAXPOOL: {
        int oh, ow, ph, pw;
        same_or_valid(o->padding, a.h, o->filter_h, o->stride_h, &oh, &ph);
        same_or_valid(o->padding, a.w, o->filter_w, o->stride_w, &ow, &pw);
        k_maxpool(x, a, o->filter_h, o->filter_w, o->stride_h, ph, pw, y, r);
      } break;
      case OP_ADD: {
        const int8_t* x2 = buf[o->in[1]];
        const int32_t z2 = m->t[o->in[1]].zero_point;
        const size_t n = (size_t)r.h * r.w * r.c;
        for (size_t k = 0; k < n; ++k) y[k] = add1(x[k], x2[k], zi, z2, zo, p, me);
      } break;
      case OP_QUANTIZE: {
        const size_t n = (size_t)r.h * r.w * r.c;
        for (size_t k = 0; k < n; ++k) y[k] = requant1(x[k], zi, zo, p, me);
      } break;
      case OP_CONCAT: {   /* axis 3, equal quantisation on all operands: byte copy.  ST: forward_concat */
        const int8_t* x2 = buf[o->in[1]];
        const shp b = s[o->in[1]];
        for (size_t px = 0; px < (size_t)r.h * r.w; ++px) {
          memcpy(y + px * r.c, x + px * a.c, a.c);
          memcpy(y + px * r.c + a.c, x2 + px * b.c, b.c);
        }
      } break;
      default: break;
    }
    if (dump) { const size_t n = (size_t)r.h * r.w * r.c; memcpy(dump + doff, y, n); doff += (long)n; }
  }
  const shp so = s[m->output];
  memcpy(out, buf[m->output], (size_t)so.h * so.w * so.c);
}

typedef struct {
  const yfo_model* m; const shp* s; const int8_t* in; int8_t* out; int8_t* dump;
  int n0, n1; size_t in_bytes, out_bytes; long dump_bytes; int rc, variant;
} job;

static void* worker(void* arg) {
  job* j = (job*)arg;
  const yfo_model* m = j->m;
  int8_t** buf = (int8_t**)calloc(m->n_tensors, sizeof(int8_t*));
  for (uint32_t t = 0; t < m->n_tensors; ++t) {
    const size_t n = (size_t)j->s[t].h * j->s[t].w * j->s[t].c;
    if (n) buf[t] = (int8_t*)malloc(n);
  }
  for (int f = j->n0; f < j->n1; ++f)
    run_frame(m, j->s, j->in + (size_t)f * j->in_bytes, j->out + (size_t)f * j->out_bytes,
              j->dump ? j->dump + (size_t)f * j->dump_bytes : NULL, buf, j->variant);
  for (uint32_t t = 0; t < m->n_tensors; ++t) free(buf[t]);
  free(buf);
  return NULL;
}

int yfo_run(const yfo_model* m, const int8_t* in, int n, int h, int w, int8_t* out, int8_t* dump, int threads) {
  return yfo_run_variant(m, in, n, h, w, out, dump, threads, YFO_RV_REF);
}

int yfo_run_variant(const yfo_model* m, const int8_t* in, int n, int h, int w, int8_t* out, int8_t* dump, int threads, int variant) {
  if (!m || !in || !out || n < 0) return -1;
  if (variant < 0 || variant >= YFO_RV_COUNT) return -3;
  shp* s = (shp*)malloc(sizeof(shp) * m->n_tensors);
  infer_shapes(m, h, w, s);
  const shp so = s[m->output];
  if (so.h <= 0 || so.w <= 0) { free(s); return -2; }
  if (threads < 1) threads = 1;
  if (threads > n) threads = n > 0 ? n : 1;
  job* jobs = (job*)calloc(threads, sizeof(job));
  pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
  const long db = yfo_dump_bytes(m, h, w);
  for (int k = 0; k < threads; ++k) {
    job* j = &jobs[k];
    j->m = m; j->s = s; j->in = in; j->out = out; j->dump = dump; j->variant = variant;
    j->n0 = (int)((long)n * k / threads); j->n1 = (int)((long)n * (k + 1) / threads);
    j->in_bytes = (size_t)h * w * s[m->input].c; j->out_bytes = (size_t)so.h * so.w * so.c; j->dump_bytes = db;
  }
  if (threads == 1) worker(&jobs[0]);
  else {
    for (int k = 0; k < threads; ++k) pthread_create(&th[k], NULL, worker, &jobs[k]);
    for (int k = 0; k < threads; ++k) pthread_join(th[k], NULL);
  }
  free(jobs); free(th); free(s);
  return n;
}

/* ------------------------------------------------------------------ box decode */
/* float -> int32 as the reference's hosts do it: truncation; out-of-range or NaN gives INT32_MIN (x86 cvttss2si,
 * which is what numpy's astype(int32) and a C cast produce on the reference's PC).  Only reachable with exp()
 * of extreme logits (q >= ~100); stated explicitly so that the GPU decode can match it bit for bit. */
static inline int32_t f2i(float v) {
  if (!(v > -2147483904.0f && v < 2147483648.0f)) return INT32_MIN;
  return (int32_t)v;
}
/* float -> int32 on the firmware's Cortex-M7 (VCVT.S32.F32): truncation, saturation, NaN -> 0 */
static inline int32_t f2i_sat(float v) {
  if (v != v) return 0;
  if (v >= 2147483648.0f) return INT32_MAX;
  if (v <= -2147483648.0f) return INT32_MIN;
  return (int32_t)v;
}
static inline int32_t dbl(int32_t v) { return (int32_t)((uint32_t)v * 2u); }
static const float k_anchors[3][2] = {{9.f, 14.f}, {12.f, 17.f}, {22.f, 21.f}};  /* tflite_prediction.py:45-47, yoloface.c:20 */

/* tflite_prediction.py:42-63.  Every transcendental is a lookup in the committed float32 tables
 * (index q+128); everything else is single float32 operations in the script's order. */
int yfo_decode_py(const int8_t* head, int gh, int gw, int frame, const float* sig, const float* ex,
                  float w_scale, float h_scale, yfo_det* dets, int max_dets) {
  int n = 0;
  for (int a = 0; a < 3; ++a)
    for (int row = 0; row < gh; ++row)
      for (int col = 0; col < gw; ++col) {
        const int8_t* p = head + ((size_t)row * gw + col) * 18 + a * 6;
        const float conf = sig[p[4] + 128];
        if (!(conf > 0.7f)) continue;                              /* prediction[..., 4] > conf_thres */
        const float cx = (sig[p[0] + 128] + (float)col) * 8.f;     /* (sigmoid(t) + grid) * 8 */
        const float cy = (sig[p[1] + 128] + (float)row) * 8.f;
        const float bw = ex[p[2] + 128] * k_anchors[a][0];         /* exp(t) * anchors */
        const float bh = ex[p[3] + 128] * k_anchors[a][1];
        float x1 = cx - bw / 2, y1 = cy - bh / 2, x2 = cx + bw / 2, y2 = cy + bh / 2;   /* xywh2xyxy */
        x1 *= w_scale; x2 *= w_scale; y1 *= h_scale; y2 *= h_scale;
        if (n < max_dets) {
          yfo_det* d = &dets[n];
          d->frame = frame; d->anchor = (uint8_t)a; d->row = (uint8_t)row; d->col = (uint8_t)col;
          d->q_conf = p[4]; d->conf = conf;
          d->x1 = f2i(x1); d->y1 = f2i(y1); d->x2 = f2i(x2); d->y2 = f2i(y2);   /* astype(int32) */
        }
        ++n;
      }
  return n;
}

/* stm32/X-CUBE-AI/App/yoloface.c:98-152 (post_process): cell-major, conf >= 0.7, LCD axis swap, clamps, x2. */
int yfo_decode_c(const int8_t* head, int frame, const float* sig, const float* ex, yfo_det* dets, int max_dets, int host_x86) {
  int n = 0;
  for (int i = 0; i < 49; ++i)
    for (int j = 0; j < 3; ++j) {
      const int8_t* p = head + i * 18 + j * 6;
      const float conf = sig[p[4] + 128];
      if (!(conf >= 0.7)) continue;
      const int grid_x = i % 7, grid_y = (i - grid_x) / 7;
      float x = (sig[p[0] + 128] + grid_x) * 8;
      float y = (sig[p[1] + 128] + grid_y) * 8;
      float w = ex[p[2] + 128] * k_anchors[j][0];
      float h = ex[p[3] + 128] * k_anchors[j][1];
      /* yoloface.c:135-138: float expressions assigned to int; the MCU saturates, an x86-64 host build does not */
      int y2 = host_x86 ? f2i(x - w / 2) : f2i_sat(x - w / 2), y1 = host_x86 ? f2i(x + w / 2) : f2i_sat(x + w / 2);
      int x1 = host_x86 ? f2i(y - h / 2) : f2i_sat(y - h / 2), x2 = host_x86 ? f2i(y + h / 2) : f2i_sat(y + h / 2);
      if (x1 < 0) x1 = 0;
      if (y1 < 0) y1 = 0;
      if (x2 > 55) x2 = 55;
      if (y2 > 55) y2 = 55;
      if (n < max_dets) {
        yfo_det* d = &dets[n];
        d->frame = frame; d->anchor = (uint8_t)j; d->row = (uint8_t)grid_y; d->col = (uint8_t)grid_x;
        d->q_conf = p[4]; d->conf = conf;
        d->x1 = dbl(x1); d->y1 = dbl(y1); d->x2 = dbl(x2); d->y2 = dbl(y2);    /* as printed at yoloface.c:148 */
      }
      ++n;
    }
  return n;
}

/* stm32/X-CUBE-AI/App/yoloface.c:26-71 + 73-93 fused: big-endian RGB565 112x112 -> 2x2 box average in 5/6/5 bit
 * space -> re-pack -> shift-expand to 8 bit -> value - 128. */
void yfo_prepare_rgb565(const uint8_t* src, int8_t* out) {
  for (int y = 0; y < 56; ++y)
    for (int x = 0; x < 56; ++x) {
      uint32_t sr = 0, sg = 0, sb = 0;
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          const int o = ((2 * y + dy) * 112 + (2 * x + dx)) * 2;
          const uint16_t px = (uint16_t)(((uint16_t)src[o] << 8) | src[o + 1]);
          sr += (px >> 11) & 0x1F; sg += (px >> 5) & 0x3F; sb += px & 0x1F;
        }
      const uint16_t color = (uint16_t)((((sr >> 2) & 0x1F) << 11) | (((sg >> 2) & 0x3F) << 5) | ((sb >> 2) & 0x1F));
      const uint8_t r = (uint8_t)((color & 0xF800) >> 8), g = (uint8_t)((color & 0x07E0) >> 3), b = (uint8_t)((color & 0x001F) << 3);
      int8_t* o8 = out + (y * 56 + x) * 3;
      o8[0] = (int8_t)((int8_t)r - 128); o8[1] = (int8_t)((int8_t)g - 128); o8[2] = (int8_t)((int8_t)b - 128);
    }
}
