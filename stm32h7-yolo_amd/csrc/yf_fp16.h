/* fp16 side configuration (BASELINE configs[3]): thin C FFI of yf_fp16.hip. */
#ifndef YF_FP16_H
#define YF_FP16_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct yf_fp16 yf_fp16;
int  yf_fp16_create(int device, const void* yfw, size_t bytes, yf_fp16** out, char* err, size_t errlen);
void yf_fp16_destroy(yf_fp16* c);
int  yf_fp16_run_device(yf_fp16* c, const void* d_in_f16, void* d_out_f32, long n, void* stream);
int  yf_fp16_release_stream(yf_fp16* c, void* stream);
size_t yf_fp16_scratch_bytes(yf_fp16* c);
void yf_fp16_scratch_stats(yf_fp16* c, unsigned long long out[6]);
const char* yf_fp16_error(const yf_fp16* c);
#ifdef __cplusplus
}
#endif
#endif
