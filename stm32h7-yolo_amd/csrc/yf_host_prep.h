/* Host-side table preparation (plain C).  See yf_host_prep.c. */
#ifndef YF_HOST_PREP_H
#define YF_HOST_PREP_H
#include <stddef.h>
#include <stdint.h>
#include "yf_tables.h"
#ifdef __cplusplus
extern "C" {
#endif

enum {
  YF_PREP_OK = 0,
  YF_PREP_ERR_ARGS = 1,
  YF_PREP_ERR_MODEL = 2,        /* generated conv table does not match the fused stage plan */
  YF_PREP_ERR_SHIFT_RANGE = 3,  /* a channel's multiplier/shift is outside what the fused epilogue handles exactly */
};

/* Rounding of the requantisation's right shift (the values of yf_requant_rounding, include/yf_network.h). */
#ifndef YF_ROUND_ENUM
#define YF_ROUND_ENUM
enum { YF_ROUND_TFLITE_REF = 0, YF_ROUND_TIES_UP = 1, YF_ROUND_TIES_UP_ALL = 2, YF_ROUND_SINGLE = 3, YF_ROUND_COUNT, YF_ROUND_GENERIC_KERNELS = 0x100 };
#endif

/* Build the device table blob from the caller's weight blob (ST layout, 11304 B).  *out_blob is malloc'd.
 * yf_prepare_tables = rounding YF_ROUND_TFLITE_REF; every rounding gives a blob of the SAME layout (only constants differ). */
int yf_prepare_tables(const uint8_t* weights_blob, size_t blob_bytes, uint8_t** out_blob, yf_table_index* ix);
int yf_prepare_tables_rounding(const uint8_t* weights_blob, size_t blob_bytes, int rounding, uint8_t** out_blob, yf_table_index* ix);
/* `rounding` may carry YF_ROUND_GENERIC_KERNELS.  Without it the dense stages of a rounding that has no sign term get the FOLDED constants of the three-instruction
 * epilogue (ZR inside C64), and the engine must launch the kernels built for them: */
int yf_rounding_signless_dense(int rounding);
int32_t yf_mbqm_form(int32_t x, int32_t mult, int shift, int form);   /* form: 0 reference, 1 ties upward, 2 single rounding */

/* TFLite QuantizeMultiplier / MultiplyByQuantizedMultiplier (exposed for the CPU tests of the host logic). */
void    yf_quantize_multiplier(double real, int32_t* mult, int* shift);
int32_t yf_mbqm(int32_t x, int32_t mult, int shift);

#ifdef __cplusplus
}
#endif
#endif
