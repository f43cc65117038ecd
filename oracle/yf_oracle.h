/* CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under stm32h7-yolo_amd/ may include, link or call this.
 *
 * Plain-C restatement of the arithmetic behind the reference's int8 path:
 *   reference call site : yoloface/tflite/tflite_prediction.py:23-41 (tf.lite.Interpreter on yoloface_int8.tflite)
 *   reference C twin    : stm32/X-CUBE-AI/App/network.c:3402-3407 (ai_network_run -> 31 c-layers; kernels are a
 *                         closed ARM-only .lib)
 * The arithmetic itself lives in a third-party dependency that is NOT under /root/reference:
 *   TensorFlow Lite, tensorflow==2.10.0 (yoloface/tensorflow/requirements.txt:2), builtin REFERENCE kernels.
 * Its published algorithm is restated here op by op.
 *
 * PARITY UNPINNED: the reference holds no golden vector, test or fixture for this path and the TFLite interpreter
 * cannot be run in the build container, so this restatement is not checked against reference outputs.  Partial pins
 * that ARE checked (tests/test_oracle.py): weights/bias/scales byte-identical to the reference's ST blob and tables,
 * the 17 ST LeakyReLU LUTs (known answers for the float formula and the quant params), an independent numpy
 * restatement, and a float evaluation of the dequantised graph.
 */
#ifndef YF_ORACLE_H
#define YF_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct yfo_model yfo_model;

/* gemmlowp / TFLite fixed-point primitives (the same two primitives are stated in C in the reference tree at
 * stm32/Drivers/CMSIS/NN/Include/arm_nnsupportfunctions.h:210-235 and :242-268). */
void    yfo_quantize_multiplier(double d, int32_t* m, int* shift);
int32_t yfo_srdhm(int32_t a, int32_t b);
int32_t yfo_rdivpot(int32_t x, int exponent);
int32_t yfo_mbqm(int32_t x, int32_t m, int shift);

yfo_model* yfo_load(const char* path);          /* .yfm pack written by tools/gen_model.py */
void       yfo_free(yfo_model* m);
int        yfo_num_ops(const yfo_model* m);

/* Bytes of every op output for one frame of h x w input, concatenated in op order (196199 at 56x56). */
long yfo_dump_bytes(const yfo_model* m, int h, int w);
/* Output grid of the head for an h x w input (7x7 at 56x56). */
void yfo_out_shape(const yfo_model* m, int h, int w, int* oh, int* ow, int* oc);

/* Run n frames, NHWC int8 [n][h][w][3] -> [n][oh][ow][18].  dump (optional) receives every op output of every
 * frame ([n][yfo_dump_bytes]).  threads<=1: single thread.  Returns n or <0. */
int yfo_run(const yfo_model* m, const int8_t* in, int n, int h, int w, int8_t* out, int8_t* dump, int threads);

/* LeakyReLU table of tflite op `op_index` as TFLite computes it (256 entries, index q+128). */
int yfo_leaky_lut(const yfo_model* m, int op_index, int8_t lut[256]);

/* ---- rounding variants (round 6): how much does the one unverifiable CHOICE under every parity claim matter? ----
 * yfo_run above is variant REF = TFLite's builtin REFERENCE kernels (SURVEY.md 8(c).3's definition of "the tflite int8
 * reference").  tflite_prediction.py:23 constructs tf.lite.Interpreter with default arguments = the default resolver,
 * whose per-channel int8 CONV_2D goes through ruy; the variants restate the other published roundings of the
 * requantisation step so that tests can measure the exposure of heads and boxes to that choice, and so that the
 * library's selectable rounding (yf_network_set_requant_rounding) has a checker.  All equally interpreter-unverified. */
enum {
  YFO_RV_REF = 0,           /* RoundingDivideByPOT everywhere: ties away from zero (common.h)                         */
  YFO_RV_UP_DENSE = 1,      /* dense CONV_2D: right shift breaks ties upward (ruy's vector kernels, ARM srshl);
                               DEPTHWISE_CONV_2D, LEAKY_RELU, ADD, QUANTIZE keep common.h's form                      */
  YFO_RV_UP_ALL = 2,        /* every op ties upward                                                                    */
  YFO_RV_FP32 = 3,          /* CONV_2D and DEPTHWISE_CONV_2D requantise in float32 (XNNPACK qs8): lrintf(acc * scale) */
  YFO_RV_SINGLE_DENSE = 4,  /* dense CONV_2D: single rounding (x*M + 2^(30-shift)) >> (31-shift) (ruy standard C++)    */
  YFO_RV_COUNT
};
int32_t yfo_mbqm_mode(int32_t x, int32_t m, int shift, int mode);   /* mode: 0 REF, 1 ties upward, 2 single rounding */
int yfo_run_variant(const yfo_model* m, const int8_t* in, int n, int h, int w, int8_t* out, int8_t* dump, int threads, int variant);
int yfo_leaky_lut_variant(const yfo_model* m, int op_index, int8_t lut[256], int variant);

typedef struct {
  int32_t frame;
  uint8_t anchor, row, col;
  int8_t  q_conf;
  float   conf;
  int32_t x1, y1, x2, y2;
} yfo_det;

/* Python decode (yoloface/tflite/tflite_prediction.py:42-63): anchor-major order, conf > 0.7, xyxy scaled by
 * (w_scale,h_scale), truncated to int32.  sig/ex are the committed 256-entry float32 tables. */
int yfo_decode_py(const int8_t* head, int gh, int gw, int frame, const float* sig, const float* ex,
                  float w_scale, float h_scale, yfo_det* dets, int max_dets);
/* Firmware decode (stm32/X-CUBE-AI/App/yoloface.c:98-152): cell-major order, conf >= 0.7, axis swap, clamp to
 * [0,55], x2.  Same tables. */
int yfo_decode_c(const int8_t* head, int frame, const float* sig, const float* ex, yfo_det* dets, int max_dets, int host_x86);

/* Frame preparation (stm32/X-CUBE-AI/App/yoloface.c:26-93): 112x112 big-endian RGB565 -> 56x56 box average in
 * 5/6/5 space -> int8 NHWC (value-128). */
void yfo_prepare_rgb565(const uint8_t* rgb565_112, int8_t* out_56x56x3);

#ifdef __cplusplus
}
#endif
#endif
