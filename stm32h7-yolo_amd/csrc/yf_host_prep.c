/* Host-side preparation of the device tables (plain C, no GPU needed).
 *
 * Runs once inside ai_network_init (reference stm32/X-CUBE-AI/App/network.c:3385-3399): the reference binds 48
 * weight/bias arrays at fixed offsets of the blob it is handed (network_configure_weights, network.c:3108-3267);
 * this build reads the same blob at the same offsets (gen/yf_model_gen.h) and turns it, together with the baked
 * quantisation tables (reference network.c:663-1341), into the packed constants the HIP kernels consume.
 *
 * Fixed-point preparation follows TFLite 2.10 (the arithmetic the metric names):
 *   conv / depthwise : QuantizeMultiplier((double)s_in * (double)s_w[c] / (double)s_out)
 *   LEAKY_RELU       : QuantizeMultiplier((double)(float)(s_in*alpha/s_out)), ...(s_in/s_out)
 *   ADD              : left_shift 20, QuantizeMultiplierSmallerThanOneExp x3
 *   QUANTIZE         : QuantizeMultiplier((double)s_in / (double)s_out)
 *
 * Which ROUNDING the requantisation's right shift takes is selectable (include/yf_network.h, yf_network_set_requant_rounding):
 * the default is TFLite's builtin REFERENCE kernels (RoundingDivideByPOT: ties away from zero) = SURVEY.md 8(c).3's definition
 * of "the tflite int8 reference"; the reference's script builds its interpreter with the default resolver
 * (yoloface/tflite/tflite_prediction.py:23), whose per-channel int8 CONV_2D goes through ruy -- ties upward, or one single
 * rounding on ruy's portable path.  Every form is a different set of CONSTANTS for the same kernels (yf_tables.h, yf_pass).
 */
#include "yf_host_prep.h"
#include "gen/yf_model_gen.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static float f32_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static float t_scale(int t) { return f32_from_bits(yf_tensor_scale_bits[t]); }
static int32_t t_zp(int t) { return yf_tensor_zero_point[t]; }

/* ---- TFLite fixed-point helpers (own statement; the oracle has an independent one) ---------------------- */
void yf_quantize_multiplier(double real, int32_t* mult, int* shift) {
  if (real == 0.0) { *mult = 0; *shift = 0; return; }
  int e;
  const double frac = frexp(real, &e);                  /* real = frac * 2^e, frac in [0.5, 1) */
  long long fixed = llround(frac * 2147483648.0);       /* half away from zero, like TfLiteRound */
  if (fixed == (1LL << 31)) { fixed >>= 1; ++e; }
  if (e < -31) { e = 0; fixed = 0; }
  *mult = (int32_t)fixed; *shift = e;
}

static int32_t sat_rounding_doubling_high_mul(int32_t a, int32_t b) {
  if (a == INT32_MIN && b == INT32_MIN) return INT32_MAX;
  const long long p = (long long)a * b;
  /* floor((p + 2^30) / 2^31): identical to gemmlowp's signed nudge + truncating division */
  return (int32_t)((p + (1LL << 30)) >> 31);
}

static int32_t rounding_div_pot(int32_t x, int e) {
  if (e == 0) return x;
  const int32_t half = (int32_t)1 << (e - 1);
  /* round half away from zero */
  return x >= 0 ? (int32_t)(((long long)x + half) >> e) : -(int32_t)((-(long long)x + half) >> e);
}

int32_t yf_mbqm(int32_t x, int32_t mult, int shift) {
  const int ls = shift > 0 ? shift : 0, rs = shift > 0 ? 0 : -shift;
  return rounding_div_pot(sat_rounding_doubling_high_mul(x * (1 << ls), mult), rs);
}

/* The three published forms of MultiplyByQuantizedMultiplier's rounding (own statement; the oracle has an independent one):
 * RQ_REF the above; RQ_UP the right shift breaks ties upward (ARM srshl, ruy's vector kernels); RQ_SINGLE one rounding of
 * the 64-bit product (ruy/apply_multiplier.cc). */
enum { RQ_REF = 0, RQ_UP = 1, RQ_SINGLE = 2 };
int32_t yf_mbqm_form(int32_t x, int32_t mult, int shift, int form) {
  if (form == RQ_REF) return yf_mbqm(x, mult, shift);
  const int ls = shift > 0 ? shift : 0, rs = shift > 0 ? 0 : -shift;
  if (form == RQ_UP) {
    const long long s = sat_rounding_doubling_high_mul(x * (1 << ls), mult);
    return rs ? (int32_t)((s + (1LL << (rs - 1))) >> rs) : (int32_t)s;
  }
  const int total = 31 - shift;
  return (int32_t)(((long long)x * mult + (1LL << (total - 1))) >> total);
}
/* the form each op family takes under a rounding mode of the public interface */
static int form_dense(int rounding) { return rounding == YF_ROUND_TIES_UP || rounding == YF_ROUND_TIES_UP_ALL ? RQ_UP : rounding == YF_ROUND_SINGLE ? RQ_SINGLE : RQ_REF; }
static int form_other(int rounding) { return rounding == YF_ROUND_TIES_UP_ALL ? RQ_UP : RQ_REF; }

static int8_t sat8(int32_t v) { return (int8_t)(v < -128 ? -128 : v > 127 ? 127 : v); }

/* ---- LUT builders ------------------------------------------------------------------------------------------ */
/* TFLite int8 LEAKY_RELU as a table over q in [-128,127] (index q+128). */
static void build_leaky_lut(int t_in, int t_out, uint8_t* lut, int form) {
  const float s_in = t_scale(t_in), s_out = t_scale(t_out);
  const float alpha = 0.1f;                              /* LeakyReluOptions.alpha of all 17 ops (0x3dcccccd) */
  int32_t m_a, m_i; int sh_a, sh_i;
  yf_quantize_multiplier((double)(float)(s_in * alpha / s_out), &m_a, &sh_a);
  yf_quantize_multiplier((double)(float)(s_in / s_out), &m_i, &sh_i);
  for (int q = -128; q < 128; ++q) {
    const int32_t v = q - t_zp(t_in);
    const int32_t u = v >= 0 ? yf_mbqm_form(v, m_i, sh_i, form) : yf_mbqm_form(v, m_a, sh_a, form);
    lut[q + 128] = (uint8_t)sat8(t_zp(t_out) + u);
  }
}

/* TFLite int8->int8 QUANTIZE (requantize) as a table. */
static void build_requant_lut(int t_in, int t_out, uint8_t* lut, int form) {
  int32_t m; int sh;
  yf_quantize_multiplier((double)t_scale(t_in) / (double)t_scale(t_out), &m, &sh);
  for (int q = -128; q < 128; ++q)
    lut[q + 128] = (uint8_t)sat8(yf_mbqm_form(q - t_zp(t_in), m, sh, form) + t_zp(t_out));
}

/* ---- per-channel requantisation constants (device form: yf_tables.h, yf_pass) -------------------------------- */
/* 64-bit constant of the fused requantisation and the matching ZR (yf_tables.h, yf_pass), off = bias' - O, Z = zp_out + 128:
 *   RQ_REF    C64 = off*2M + 2^31 + (2^(rs-1) - 1)*2^32 (mod 2^64),   ZR = Z << rs         carry-out = TFLite's sign term
 *   RQ_UP     C64 = off*2M + 2^31 + 2^(rs-1)*2^32 + 2^63,             ZR = (Z << rs) - 2^31
 *   RQ_SINGLE C64 = off*2M + 2^(31+rs) + 2^63,                        ZR = (Z << rs) - 2^31
 * The last two have no sign term: the 2^63 keeps acc_p*2M + C64 inside [2^61, 2^64) for every accumulator the host admits
 * (|acc| < 2^29), so the multiply-add never carries out, and its 2^31 in the high word is taken back by ZR (32-bit wrap).  The
 * kernels' instruction sequence -- v_mad_u64_u32, v_addc_co_u32, v_ashrrev, v_med3 -- is the same for all three.
 * FOLD (dense stages of the kernels with the sign-free epilogue, rq4 SIGNLESS: v_mad_u64_u32, v_ashrrev, v_med3): nothing reads the carry or ZR, so ZR rides in
 * C64's high word and hi32 of the multiply-add is t itself (mod 2^32; no 2^63 needed):
 *   RQ_UP     C64 = off*2M + 2^31 + (2^(rs-1) + (Z << rs))*2^32,      RQ_SINGLE C64 = off*2M + 2^(31+rs) + (Z << rs)*2^32,      ZR = 0 */
static void build_c64(long long off, int32_t m, int rs, int form, int z, uint32_t out[2], uint32_t* zr, int fold) {
  unsigned long long c = (unsigned long long)off * (2ull * (unsigned long long)(uint32_t)m);   /* wraps mod 2^64 as intended */
  if (form == RQ_REF) c += (1ull << 31) + ((((unsigned long long)1 << (rs - 1)) - 1ull) << 32);
  else if (form == RQ_UP) c += (1ull << 31) + (((unsigned long long)1 << (rs - 1)) << 32);
  else c += ((unsigned long long)1 << (31 + rs));
  *zr = (uint32_t)z << rs;
  if (form != RQ_REF) {
    if (fold) { c += (unsigned long long)*zr << 32; *zr = 0; }
    else { c += 1ull << 63; *zr -= 0x80000000u; }
  }
  out[0] = (uint32_t)c; out[1] = (uint32_t)(c >> 32);
}

/* Channel j of pass p.  abs_w = sum |w| (bounds the accumulator).  Returns 0 or an error code. */
static int build_chan(const yf_conv_desc* d, const uint8_t* blob, int ch, int32_t sum_w, int32_t abs_w, yf_pass* p, int j, int form, int fold) {
  const float s_in = t_scale(d->t_in), s_out = t_scale(d->t_out);
  const float s_w = f32_from_bits(d->wscale_bits[ch]);
  int32_t m; int sh;
  yf_quantize_multiplier((double)s_in * (double)s_w / (double)s_out, &m, &sh);
  if (sh > -1 || sh < -20) return YF_PREP_ERR_SHIFT_RANGE;      /* fused epilogue needs 1 <= rshift <= 20 */
  if (m <= (1 << 30)) return YF_PREP_ERR_SHIFT_RANGE;           /* normalised multiplier (frexp) is > 2^30 unless exact pow2 */
  int32_t bias;
  memcpy(&bias, blob + d->b_off + 4 * (size_t)ch, 4);
  const int rs = -sh;
  const long long bias2 = (long long)bias - (long long)t_zp(d->t_in) * sum_w;
  /* O + sum w*x_raw must stay a positive 32-bit multiplicand and the true accumulator below 2^29 in magnitude */
  if ((bias2 < 0 ? -bias2 : bias2) + 255ll * abs_w >= (1ll << 29)) return YF_PREP_ERR_SHIFT_RANGE;
  p->mult2[j] = (uint32_t)m << 1;
  build_c64(bias2 - (long long)YF_ACC_OFFSET, m, rs, form, t_zp(d->t_out) + 128, p->c64[j], &p->zr[j], fold);
  p->rshift[j] = rs;
  return 0;
}
/* padding channel of a pass (cout not a multiple of 4): harmless constants, the byte it produces is never read */
static void pad_chan(yf_pass* p, int j) {
  p->mult2[j] = ((1u << 30) + 1u) << 1; p->rshift[j] = 1;
  build_c64(-(long long)YF_ACC_OFFSET, (1 << 30) + 1, 1, RQ_REF, 128, p->c64[j], &p->zr[j], 0);
}

static const yf_conv_desc* find_conv(int tfl_op) {
  for (int i = 0; i < YF_N_CONVS; ++i) if (yf_convs[i].tfl_op == tfl_op) return &yf_convs[i];
  return NULL;
}

typedef struct { uint8_t* p; size_t size, cap; } blob_t;
static size_t blob_alloc(blob_t* b, size_t n) {
  size_t off = (b->size + 15) & ~(size_t)15;
  if (off + n > b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 65536;
    while (nc < off + n) nc *= 2;
    b->p = (uint8_t*)realloc(b->p, nc);
    memset(b->p + b->cap, 0, nc - b->cap);
    b->cap = nc;
  }
  b->size = off + n;
  return off;
}

/* stage descriptors: the tflite conv op each dense stage evaluates --------------------------------------------- */
typedef struct { int id; int tfl_op; } dense_plan;
static const dense_plan k_dense[YF_N_DENSE] = {
  {YF_D_CONV1, 1}, {YF_D_C5, 5}, {YF_D_C6, 6}, {YF_D_C12, 12}, {YF_D_C13, 13}, {YF_D_C17, 17},
  {YF_D_C19, 19}, {YF_D_C23, 23}, {YF_D_C29, 29}, {YF_D_C30, 30}, {YF_D_C34, 34}, {YF_D_C36, 36},
  {YF_D_C40, 40}, {YF_D_C42, 42}, {YF_D_C47, 47}, {YF_D_C51, 51}, {YF_D_C53, 53},
};
static const int k_dw_ops[YF_N_DW] = {3, 10, 15, 27, 32, 38, 49};

/* conv2d_1 tap -> (k-step, byte slot) map: RGBX pixels, see yf_tables.h */
static void conv1_slot(int ky, int kx, int c, int* step, int* slot) {
  const int pix = ky * 3 + kx;              /* 0..8 */
  *step = pix / 4; *slot = (pix % 4) * 4 + c;
}

/* 1 if the blob yf_prepare_tables_rounding builds for `rounding` is for the kernels with the sign-free dense epilogue (the engine launches those then) */
int yf_rounding_signless_dense(int rounding) {
  if (rounding & YF_ROUND_GENERIC_KERNELS) return 0;
  return rounding > YF_ROUND_TFLITE_REF && rounding < YF_ROUND_COUNT && form_dense(rounding) != RQ_REF;
}

int yf_prepare_tables(const uint8_t* weights_blob, size_t blob_bytes, uint8_t** out_blob, yf_table_index* ix) {
  return yf_prepare_tables_rounding(weights_blob, blob_bytes, YF_ROUND_TFLITE_REF, out_blob, ix);
}

int yf_prepare_tables_rounding(const uint8_t* weights_blob, size_t blob_bytes, int rounding, uint8_t** out_blob, yf_table_index* ix) {
  if (!weights_blob || blob_bytes < YF_WEIGHTS_BLOB_BYTES || !out_blob || !ix) return YF_PREP_ERR_ARGS;
  const int generic = (rounding & YF_ROUND_GENERIC_KERNELS) != 0;      /* constants for the four-instruction kernels (any rounding) instead of the sign-free dense form */
  rounding &= ~YF_ROUND_GENERIC_KERNELS;
  if (rounding < 0 || rounding >= YF_ROUND_COUNT) return YF_PREP_ERR_ARGS;
  const int fd = form_dense(rounding), fo = form_other(rounding);      /* dense CONV_2D | DEPTHWISE_CONV_2D, LEAKY_RELU, ADD, QUANTIZE */
  const int fold = fd != RQ_REF && !generic;                           /* == yf_rounding_signless_dense(rounding): the dense stages' ZR rides in C64 */
  memset(ix, 0, sizeof *ix);
  blob_t b = {0, 0, 0};
  int rc = 0;
  blob_alloc(&b, YF_INDEX_RESERVED);          /* the index itself is stored at offset 0 (filled in at the end) */

  /* ---------------- dense stages ---------------- */
  for (int s = 0; s < YF_N_DENSE && !rc; ++s) {
    const yf_conv_desc* d = find_conv(k_dense[s].tfl_op);
    if (!d || d->depthwise) { rc = YF_PREP_ERR_MODEL; break; }
    yf_dense* o = &ix->dense[s];
    const int kk = d->kh * d->kw * d->cin;
    o->cout = d->cout; o->cout_pad4 = (uint16_t)((d->cout + 3) & ~3); o->k = (uint16_t)kk;
    const int8_t* w = (const int8_t*)weights_blob + d->w_off;                      /* OHWI */
    if (s == YF_D_CONV1) {
      o->krow = YF_CONV1_KROW;
    } else if (s == YF_D_C23) {
      o->krow = 48;                                 /* k order = T14 channel order (38 slots) */
    } else {
      o->krow = (uint16_t)((kk + 15) & ~15);
    }
    o->w_off = (uint32_t)blob_alloc(&b, (size_t)o->cout_pad4 * o->krow);
    o->c_off = (uint32_t)blob_alloc(&b, (size_t)(o->cout_pad4 / 4) * sizeof(yf_pass));
    for (int ch = d->cout; ch < o->cout_pad4; ++ch) pad_chan((yf_pass*)(b.p + o->c_off) + ch / 4, ch & 3);
    for (int ch = 0; ch < d->cout; ++ch) {
      int8_t* row = (int8_t*)b.p + o->w_off + (size_t)ch * o->krow;
      int32_t sum_w = 0, abs_w = 0;
      for (int k = 0; k < kk; ++k) { const int wv = w[(size_t)ch * kk + k]; sum_w += wv; abs_w += wv < 0 ? -wv : wv; }
      if (s == YF_D_CONV1) {
        for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) for (int c = 0; c < 3; ++c) {
          int step, slot; conv1_slot(ky, kx, c, &step, &slot);
          row[step * 16 + slot] = w[(size_t)ch * 27 + (ky * 3 + kx) * 3 + c];
        }
      } else if (s == YF_D_C23) {
        for (int k = 0; k < 18; ++k) row[k] = w[(size_t)ch * 36 + k];                       /* pool branch */
        for (int k = 18; k < 36; ++k) row[YF_T14_CONV_BASE + (k - 18)] = w[(size_t)ch * 36 + k];  /* conv branch */
      } else {
        memcpy(row, w + (size_t)ch * kk, (size_t)kk);
      }
      rc = build_chan(d, weights_blob, ch, sum_w, abs_w, (yf_pass*)(b.p + o->c_off) + ch / 4, ch & 3, fd, fold);
      if (rc) break;
    }
  }

  /* ---------------- depthwise stages ---------------- */
  for (int s = 0; s < YF_N_DW && !rc; ++s) {
    const yf_conv_desc* d = find_conv(k_dw_ops[s]);
    if (!d || !d->depthwise) { rc = YF_PREP_ERR_MODEL; break; }
    yf_dw* o = &ix->dw[s];
    o->c = d->cout; o->ngroups = (uint16_t)((d->cout + 3) / 4);
    o->g_off = (uint32_t)blob_alloc(&b, (size_t)o->ngroups * YF_DW_GROUP_BYTES);
    ix->halo_zp[s] = t_zp(d->t_in);
    const int8_t* w = (const int8_t*)weights_blob + d->w_off;                      /* 1HWC */
    for (int g = 0; g < o->ngroups; ++g) {
      uint32_t* wd = (uint32_t*)(b.p + o->g_off + (size_t)g * YF_DW_GROUP_BYTES);  /* [9 taps][4 lanes] */
      yf_pass* cc = (yf_pass*)(wd + 36);
      for (int j = 0; j < 4; ++j) {
        const int ch = g * 4 + j;
        if (ch >= d->cout) {                 /* padding channel: harmless constants */
          for (int t = 0; t < 9; ++t) wd[t * 4 + j] = 0;
          pad_chan(cc, j);
          continue;
        }
        int32_t sum_w = 0, abs_w = 0;
        for (int t = 0; t < 9; ++t) {
          const int8_t wv = w[(size_t)t * d->cout + ch];
          sum_w += wv; abs_w += wv < 0 ? -wv : wv;
          wd[t * 4 + j] = ((uint32_t)(uint8_t)wv) << (8 * j);      /* byte j of the tap's dword carries channel j */
        }
        rc = build_chan(d, weights_blob, ch, sum_w, abs_w, cc, j, fo, 0);
        if (rc) break;
      }
      if (rc) break;
    }
  }

  /* ---------------- residual adds ---------------- */
  static const int add_t[YF_N_ADD][3] = {{62, 67, 68}, {78, 83, 84}, {84, 89, 90}};   /* tfl tensors in1,in2,out */
  for (int s = 0; s < YF_N_ADD && !rc; ++s) {
    const float s1 = t_scale(add_t[s][0]), s2 = t_scale(add_t[s][1]), so = t_scale(add_t[s][2]);
    const double twice_max = (double)(2 * (s1 > s2 ? s1 : s2));
    yf_add* a = &ix->add[s];
    int sh;
    yf_quantize_multiplier((double)s1 / twice_max, &a->m1, &sh); a->s1 = sh;
    yf_quantize_multiplier((double)s2 / twice_max, &a->m2, &sh); a->s2 = sh;
    yf_quantize_multiplier(twice_max / (double)((float)(1 << 20) * so), &a->mo, &sh); a->so = sh;
    if (a->s1 > 0 || a->s2 > 0 || a->so > -1 || a->so < -30 || a->mo <= (1 << 30)) rc = YF_PREP_ERR_SHIFT_RANGE;
    a->zp1 = t_zp(add_t[s][0]); a->zp2 = t_zp(add_t[s][1]); a->zpo = t_zp(add_t[s][2]);
    a->rso = -a->so;
    a->kco = ((int32_t)1 << (a->rso - 1)) + a->zpo * ((int32_t)1 << a->rso);
    if (!rc && a->rso > 20) rc = YF_PREP_ERR_SHIFT_RANGE;
    a->mo2 = (uint32_t)a->mo << 1;
    if (!rc) build_c64(-(long long)YF_ACC_OFFSET, a->mo, a->rso, fo, a->zpo + 128, a->c64o, &a->zro, 0);
  }

  /* ---------------- LUTs ---------------- */
  if (!rc) {
    ix->lut_off = (uint32_t)blob_alloc(&b, (size_t)YF_N_LUT * 256 + YF_ADDLUT_BYTES + YF_DBG_LUT_BYTES);
    uint8_t* L = b.p + ix->lut_off;
    int32_t* AL = (int32_t*)(L + YF_N_LUT * 256);            /* [add][A|B][256] */
    for (int s = 0; s < YF_N_ADD; ++s) {
      const yf_add* a = &ix->add[s];
      for (int q = -128; q < 128; ++q) {
        AL[(s * 2 + 0) * 256 + q + 128] = yf_mbqm_form((q - a->zp1) * (1 << 20), a->m1, a->s1, fo);
        AL[(s * 2 + 1) * 256 + q + 128] = yf_mbqm_form((q - a->zp2) * (1 << 20), a->m2, a->s2, fo) + YF_ACC_OFFSET;   /* |sa + sb| < 2^29 */
      }
    }
    static const int leaky[][3] = {       /* lut id, tensor in, tensor out (tflite LEAKY_RELU ops) */
      {YF_L_LEAKY2, 51, 52}, {YF_L_LEAKY4, 53, 54}, {YF_L_LEAKY7, 56, 57}, {YF_L_LEAKY11, 60, 61},
      {YF_L_LEAKY14, 63, 64}, {YF_L_LEAKY16, 65, 66}, {YF_L_LEAKY20, 69, 70}, {YF_L_LEAKY24, 72, 73},
      {YF_L_LEAKY28, 76, 77}, {YF_L_LEAKY31, 79, 80}, {YF_L_LEAKY33, 81, 82}, {YF_L_LEAKY37, 85, 86},
      {YF_L_LEAKY39, 87, 88}, {YF_L_LEAKY48, 94, 95}, {YF_L_LEAKY50, 96, 97}, {YF_L_LEAKY52, 98, 99},
    };
    for (unsigned i = 0; i < sizeof leaky / sizeof leaky[0]; ++i)
      build_leaky_lut(leaky[i][1], leaky[i][2], L + 256 * leaky[i][0], fo);
    /* the two pool LUTs are RAW-indexed (index = the int8 bit pattern, not q + 128): the pooling code extracts bytes
     * straight out of its packed maxima */
    uint8_t q21[256], q45[256];
    build_requant_lut(58, 103, q21, fo);                         /* QUANTIZE #21: pool_8 branch of concat_22 */
    build_requant_lut(74, 101, q45, fo);                         /* QUANTIZE #45: pool_25 branch of concat_46 */
    for (int i = 0; i < 256; ++i) { L[256 * YF_L_Q21 + i] = q21[i ^ 128]; L[256 * YF_L_Q45 + i] = q45[i ^ 128]; }
    uint8_t l43[256], q44[256];
    build_leaky_lut(91, 92, l43, fo);                            /* LEAKY_RELU #43 */
    build_requant_lut(92, 102, q44, fo);                         /* QUANTIZE #44 */
    for (int i = 0; i < 256; ++i) L[256 * YF_L_L43Q44 + i] = q44[(int)(int8_t)l43[i] + 128];
    memcpy(L + YF_N_LUT * 256 + YF_ADDLUT_BYTES, l43, 256);   /* LEAKY_RELU #43 alone: the debug builds' per-node dump (tensor 92 is never materialised otherwise) */
  }

  /* ---------------- constant blocks of the fused 56x56 kernel (yf_tables.h): the same numbers, regrouped ---------------- */
  for (int cs = 0; cs < YF_N_CS && !rc; ++cs) {
    const int di = yf_cs_dense[cs], wi = yf_cs_dw[cs], ai = yf_cs_add[cs];
    if (di >= 0) {
      const yf_dense* d = &ix->dense[di];
      const size_t wbytes = (size_t)d->cout_pad4 * d->krow, np = d->cout_pad4 / 4;
      const size_t bytes = (wbytes + np * sizeof(yf_pass_v) + (ai >= 0 ? 2048 : 0) + 15) & ~(size_t)15;
      const size_t off = blob_alloc(&b, bytes);
      memcpy(b.p + off, b.p + d->w_off, wbytes);
      for (size_t p = 0; p < np; ++p) {
        const yf_pass* src = (const yf_pass*)(b.p + d->c_off) + p;
        yf_pass_v* dst = (yf_pass_v*)(b.p + off + wbytes) + p;
        memcpy(dst->mult2, src->mult2, 16); memcpy(dst->zr, src->zr, 16);
      }
      if (ai >= 0) memcpy(b.p + off + wbytes + np * sizeof(yf_pass_v), b.p + ix->lut_off + YF_N_LUT * 256 + (size_t)ai * 2048, 2048);
      ix->cs_v_off[cs] = (uint32_t)off; ix->cs_v_bytes[cs] = (uint32_t)bytes;
    } else {
      const yf_dw* d = &ix->dw[wi];
      const size_t bytes = ((size_t)d->ngroups * YF_DWV_GROUP_BYTES + 15) & ~(size_t)15;
      const size_t off = blob_alloc(&b, bytes);
      for (int g = 0; g < d->ngroups; ++g) {
        const uint8_t* src = b.p + d->g_off + (size_t)g * YF_DW_GROUP_BYTES;
        uint8_t* dst = b.p + off + (size_t)g * YF_DWV_GROUP_BYTES;
        memcpy(dst, src, 144);
        const yf_pass* ps = (const yf_pass*)(src + 144);
        memcpy(dst + 144, ps->mult2, 16); memcpy(dst + 160, ps->zr, 16);
      }
      ix->cs_v_off[cs] = (uint32_t)off; ix->cs_v_bytes[cs] = (uint32_t)bytes;
    }
  }
  for (int cs = 0; cs < YF_N_CS && !rc; ++cs) {        /* scalar side: one compact array (stays in the scalar cache) */
    const int di = yf_cs_dense[cs], wi = yf_cs_dw[cs];
    const size_t np = di >= 0 ? ix->dense[di].cout_pad4 / 4 : ix->dw[wi].ngroups;
    const size_t off = blob_alloc(&b, np * sizeof(yf_pass_s));
    for (size_t p = 0; p < np; ++p) {
      const yf_pass* src = di >= 0 ? (const yf_pass*)(b.p + ix->dense[di].c_off) + p
                                   : (const yf_pass*)(b.p + ix->dw[wi].g_off + p * YF_DW_GROUP_BYTES + 144);
      yf_pass_s* dst = (yf_pass_s*)(b.p + off) + p;
      memcpy(dst->c64, src->c64, 32); memcpy(dst->rshift, src->rshift, 16);
    }
    ix->cs_s_off[cs] = (uint32_t)off;
  }

  ix->in_zp = t_zp(0);
  if (rc) { free(b.p); *out_blob = NULL; return rc; }
  blob_alloc(&b, 64);                       /* zeroed tail so 16-byte reads past the last row stay in bounds */
  ix->total_bytes = (uint32_t)b.size;
  memcpy(b.p, ix, sizeof *ix);                /* device-side copy of the index, read by the kernels with scalar loads */
  *out_blob = b.p;
  return 0;
}
