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

/* Build the device table blob from the caller's weight blob (ST layout, 11304 B).  *out_blob is malloc'd. */
int yf_prepare_tables(const uint8_t* weights_blob, size_t blob_bytes, uint8_t** out_blob, yf_table_index* ix);

/* TFLite QuantizeMultiplier / MultiplyByQuantizedMultiplier (exposed for the CPU tests of the host logic). */
void    yf_quantize_multiplier(double real, int32_t* mult, int* shift);
int32_t yf_mbqm(int32_t x, int32_t mult, int shift);

#ifdef __cplusplus
}
#endif
#endif
