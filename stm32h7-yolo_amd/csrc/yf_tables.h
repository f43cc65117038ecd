/* Device constant tables of the fused yoloface int8 engine: shared by the host preparation (yf_host_prep.c),
 * the HIP kernels (yf_engine.hip) and the table-level CPU emulator used only by tests (tests/csrc/).
 *
 * One "stage" = one fused reference c-layer (reference stm32/X-CUBE-AI/App/network.c:2204-2927 lists the 31
 * c-layers; SURVEY.md Appendix A maps them to the 54 tflite ops).  Everything a stage needs at run time is
 * precomputed here once, at ai_network_init time (reference network.c:3385-3399 does the equivalent pointer
 * binding on the MCU).
 */
#ifndef YF_TABLES_H
#define YF_TABLES_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- requantisation constants of one PASS = 4 consecutive output channels (80 B) ------------------------------
 * TFLite:  acc = bias - zp_in*sum(w) + sum w*x_raw;  s1 = SRDHM(acc, M) = floor(acc*M / 2^31 + 1/2);
 *          y = RoundingDivideByPOT(s1, r) + Z = floor((s1 + h - [s1 < 0]) / 2^r) + Z,  h = 2^(r-1),  Z = zp_out + 128
 *          (every stage produces the unsigned byte q + 128: it indexes a LUT / an add table, or is flipped back with ^0x80).
 * Device form -- the MFMA accumulator starts at O = 2^30 (the inline constant 2.0 as the C operand: no v_mov), so
 * acc_p = O + sum w*x_raw is a positive 32-bit multiplicand, and with
 *          N = acc*2M + 2^31 + (h-1)*2^32 = acc_p*2M + C64,     C64 = (bias' - O)*2M + 2^31 + (h-1)*2^32  (mod 2^64)
 * one v_mad_u64_u32 yields  hi32 = s1 + h - 1  and, as its carry-out, [N >= 0] = [s1 > -h]; every accumulator with
 * -h < s1 < 0 rounds to 0 either way, so  s1 + h - [s1 < 0]  may be taken as  hi32 + carry:
 *          y + Z = (hi32 + ZR + carry) >> r,   ZR = Z << r      (one v_addc_co_u32 with the carry, one v_ashrrev)
 * followed by the clamp to [0, 255] (v_med3).  Four VALU instructions per output; bias, zero points and both rounding
 * constants live in C64 / ZR.  The host refuses a model with a channel outside 1 <= r <= 20, M <= 2^30 or
 * |bias'| + 255*sum|w| >= 2^29.
 * Layout is struct-of-arrays so that one vector load brings the four multipliers (VGPRs: the multiplicand and the 64-bit
 * addend cannot both come from SGPRs) and scalar loads bring the rest. */
#define YF_ACC_OFFSET 0x40000000          /* O: bit pattern of the inline constant 2.0f */
typedef struct {
  uint32_t mult2[4];   /* 2 * M                                  (vector registers) */
  uint32_t zr[4];      /* (zp_out + 128) << rshift               (vector registers) */
  uint32_t c64[4][2];  /* C64: low dword, high dword             (scalar register pairs) */
  int32_t  rshift[4];  /*                                        (scalar registers) */
} yf_pass;

/* ---- dense (MFMA) stage ----------------------------------------------------------------------------------
 * Weight rows are stored [cout_pad4][krow] int8, krow = K rounded up to 16, zero filled; the k order is the
 * channel order of the stage's INPUT buffer (which may be a permutation of the tflite order: concat buffers). */
typedef struct {
  uint32_t w_off;      /* byte offset of the weight rows in the table blob (16-B aligned) */
  uint32_t c_off;      /* byte offset of the yf_pass array (cout_pad4 / 4 entries) */
  uint16_t cout, cout_pad4, k, krow;
} yf_dense;

/* ---- depthwise stage: per group of 4 channels: 9 taps x 4 masked weight dwords, then one yf_pass --------- */
typedef struct {
  uint32_t g_off;      /* byte offset of group 0; each group is YF_DW_GROUP_BYTES */
  uint16_t c, ngroups;
} yf_dw;
#define YF_DW_GROUP_BYTES (36 * 4 + 80)

/* ---- residual add (tflite ADD, int8): out = clamp(zpo + MBQM(MBQM((a-zp1)<<20,m1,s1) + MBQM((b-zp2)<<20,m2,s2), mo, so)) */
typedef struct {
  int32_t zp1, zp2, zpo;
  int32_t m1, s1, m2, s2, mo, so;
  int32_t kco, rso;    /* plain form of the final requantisation of sa+sb: y = (s + kco + (s>>31)) >> rso, rso = -so >= 1 */
  uint32_t mo2, zro;   /* device form (see yf_pass): 2*mo, (zpo + 128) << rso */
  uint32_t c64o[2];    /* (-O)*2mo + 2^31 + (2^(rso-1) - 1)*2^32: the offset O rides in table B */
} yf_add;
/* Device form of an ADD: two 256-entry int32 tables, index q+128:
 *   A[q1] = MBQM((q1 - zp1) << 20, m1, s1)   (the stored operand)     B[q2] = MBQM((q2 - zp2) << 20, m2, s2) + O
 * laid out [YF_N_ADD][2][256] right behind the byte LUTs. */
#define YF_ADDLUT_BYTES (YF_N_ADD * 2 * 256 * 4)
#define YF_DBG_LUT_BYTES 256   /* behind the add tables: LEAKY_RELU #43 alone (production composes it with QUANTIZE #44); only the debug builds load it, for the per-node dump */

enum {
  /* dense stages in execution order */
  YF_D_CONV1 = 0, YF_D_C5, YF_D_C6, YF_D_C12, YF_D_C13, YF_D_C17, YF_D_C19, YF_D_C23, YF_D_C29, YF_D_C30,
  YF_D_C34, YF_D_C36, YF_D_C40, YF_D_C42, YF_D_C47, YF_D_C51, YF_D_C53, YF_N_DENSE
};
enum { YF_W_DW3 = 0, YF_W_DW10, YF_W_DW15, YF_W_DW27, YF_W_DW32, YF_W_DW38, YF_W_DW49, YF_N_DW };
enum { YF_A_ADD18 = 0, YF_A_ADD35, YF_A_ADD41, YF_N_ADD };

/* 256-entry byte LUTs (index q+128, value int8 stored as a byte), in execution order.
 * LEAKY_n = TFLite int8 LEAKY_RELU of tflite op n; Q21/Q45 = QUANTIZE ops; L43Q44 = QUANTIZE#44 o LEAKY#43.
 * Q21 and Q45 (the max-pool outputs) are RAW-indexed instead: index = the int8 bit pattern as an unsigned byte. */
enum {
  YF_L_LEAKY2 = 0, YF_L_LEAKY4, YF_L_LEAKY7, YF_L_Q21, YF_L_LEAKY11, YF_L_LEAKY14, YF_L_LEAKY16, YF_L_LEAKY20,
  YF_L_LEAKY24, YF_L_Q45, YF_L_LEAKY28, YF_L_LEAKY31, YF_L_LEAKY33, YF_L_LEAKY37, YF_L_LEAKY39, YF_L_L43Q44,
  YF_L_LEAKY48, YF_L_LEAKY50, YF_L_LEAKY52, YF_N_LUT
};

/* conv2d_1 packs its 27 taps into three 16-byte MFMA k-steps over RGBX pixels (4 bytes per pixel, X weight = 0):
 * step 0 carries the pixels (ky,kx) = (0,0)(0,1)(0,2)(1,0), step 1 (1,1)(1,2)(2,0)(2,1), step 2 (2,2) in its first
 * dword.  Row layout [8 cout][3 steps][16 B]. */
#define YF_CONV1_KROW 48

/* ---- constant blocks of the 56x56 fused kernel (round 3) -----------------------------------------------------------------
 * The 24 stages that have constants ("const-stages", execution order) each own ONE contiguous block of everything their
 * VECTOR side needs, so that a single LDS-DMA per stage brings it into an LDS ring slot one stage ahead of its use:
 *   dense stage : [cout_pad4 x krow weight rows][cout_pad4/4 x yf_pass_v][residual add: tables A and B, 2 x 256 x int32]
 *   depthwise   : per group of 4 channels [9 taps x 4 masked weight dwords (144 B)][yf_pass_v]          (YF_DWV_GROUP_BYTES)
 * and one compact array of the SCALAR side (yf_pass_s per pass / group), read with scalar loads: all of them together are
 * 6.7 KB and stay in the scalar cache.  Same numbers as yf_pass, split by the register file they are loaded into. */
typedef struct { uint32_t mult2[4]; uint32_t zr[4]; } yf_pass_v;                 /* 32 B */
typedef struct { uint32_t c64[4][2]; int32_t rshift[4]; } yf_pass_s;             /* 48 B */
#define YF_DWV_GROUP_BYTES (36 * 4 + 32)
#define YF_N_CS 24
/* const-stage -> index of its dense / depthwise / residual-add descriptor (-1: none) */
#ifdef __cplusplus
#define YF_CONST_TABLE constexpr
#else
#define YF_CONST_TABLE static const
#endif
YF_CONST_TABLE int8_t yf_cs_dense[YF_N_CS] = {0, -1, 1, 2, -1, 3, 4, -1, 5, 6, 7, -1, 8, 9, -1, 10, 11, -1, 12, 13, 14, -1, 15, 16};
YF_CONST_TABLE int8_t yf_cs_dw[YF_N_CS]    = {-1, 0, -1, -1, 1, -1, -1, 2, -1, -1, -1, 3, -1, -1, 4, -1, -1, 5, -1, -1, -1, 6, -1, -1};
YF_CONST_TABLE int8_t yf_cs_add[YF_N_CS]   = {-1, -1, -1, -1, -1, -1, -1, -1, 0, -1, -1, -1, -1, -1, -1, 1, -1, -1, 2, -1, -1, -1, -1, -1};

typedef struct {
  yf_dense dense[YF_N_DENSE];
  yf_dw    dw[YF_N_DW];
  yf_add   add[YF_N_ADD];
  uint32_t lut_off;          /* YF_N_LUT * 256 bytes of byte LUTs followed by YF_ADDLUT_BYTES of add tables and YF_DBG_LUT_BYTES */
  uint32_t total_bytes;
  int32_t  in_zp;            /* input zero point (-128): halo fill of the staged frame */
  int32_t  halo_zp[YF_N_DW]; /* zero point of each depthwise INPUT buffer: its halo fill value */
  uint32_t cs_v_off[YF_N_CS], cs_v_bytes[YF_N_CS];   /* vector block of a const-stage (16-byte aligned, bytes a multiple of 16) */
  uint32_t cs_s_off[YF_N_CS];                        /* its yf_pass_s array */
} yf_table_index;

/* The table blob starts with a copy of the index (so kernels fetch stage descriptors with scalar loads instead of
 * carrying ~120 dwords of kernel arguments in SGPRs). */
#define YF_INDEX_RESERVED 1024

/* Channel order of concat_22's buffer T14: pool branch at [0,18), conv branch at [20,38) (4-byte aligned
 * starts so that packed 4-channel stores stay aligned); conv2d_23's k order follows it. */
#define YF_T14_CONV_BASE 20

#ifdef __cplusplus
}
#endif
#endif
