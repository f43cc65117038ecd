/* Thin FFI between the C host layer (network_abi.c) and the HIP engine (yf_engine.hip).
 * Plain C types only; every function returns 0 on success or a negative code, with text in yf_engine_error(). */
#ifndef YF_ENGINE_H
#define YF_ENGINE_H
#include <stddef.h>
#include <stdint.h>
#include "yf_tables.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct yf_engine yf_engine;

enum { YF_ENG_OK = 0, YF_ENG_ERR_HIP = -1, YF_ENG_ERR_ARG = -2, YF_ENG_ERR_NO_DEVICE = -3, YF_ENG_ERR_VARIANT = -4 };

/* signless_dense: the table blob carries the sign-free form of the dense stages' constants (ZR folded into C64: yf_host_prep.c) and the launches take the kernels
 * whose dense convolutions requantise in three instructions (namespaces yfu / yf160u of yf_engine.hip); 0 = the four-instruction kernels, any rounding's constants */
int  yf_engine_create(int device, const uint8_t* table_blob, const yf_table_index* ix, int signless_dense, yf_engine** out, char* err, size_t errlen);
void yf_engine_destroy(yf_engine* e);
/* replace the table blob of a live engine (same layout, other constants: yf_network_set_requant_rounding); waits for the device first */
int  yf_engine_set_tables(yf_engine* e, const uint8_t* table_blob, const yf_table_index* ix, int signless_dense);
/* 1 if a production kernel of that shape is compiled in (frames_per_wg may carry the +200 experimental-build tag) */
int  yf_engine_variant_exists(int frames_per_wg, int waves_per_wg);
int  yf_engine_configure(yf_engine* e, int frames_per_wg, int waves_per_wg);
/* byte offsets compiled into the kernels: w_off[17], c_off[17] (dense stages), g_off[7] (depthwise), lut_off, total = 43 ints */
int  yf_engine_table_plan(int32_t* out, int cap);
const char* yf_engine_error(const yf_engine* e);
const char* yf_engine_kernel_name(const yf_engine* e);
const char* yf_engine_kernel_name_for(const yf_engine* e, long n);   /* the shape a batch of n frames runs */
const char* yf_engine_build_id(void);

/* device-resident batch; d_dump may be NULL */
int  yf_engine_run_device(yf_engine* e, const void* d_in, void* d_out, void* d_dump, long n, void* stream);
/* host batch: H2D, run, D2H through engine-owned staging buffers; synchronous */
int  yf_engine_run_host(yf_engine* e, const void* h_in, void* h_out, long n);
int  yf_engine_time_device(yf_engine* e, const void* d_in, void* d_out, long n, int iters, void* stream, float* ms_per_launch);
/* debug kernel (stage dump build) stopped after `stop_stage` fused stages: per-stage timing */
int  yf_engine_time_stages(yf_engine* e, const void* d_in, void* d_out, long n, int iters, int stop_stage, void* stream, float* ms_per_launch);
int  yf_engine_decode_device(yf_engine* e, const void* d_heads, long n, int mode, float w_scale, float h_scale,
                             void* d_dets, void* d_counts, int cap, void* stream);
/* network + box decode in ONE launch: every workgroup decodes its frames' heads while they are still in LDS */
int  yf_engine_run_decode_device(yf_engine* e, const void* d_in, void* d_out, long n, int mode, float w_scale, float h_scale,
                                 void* d_dets, void* d_counts, int cap, void* stream);
/* 28-byte yf_det records <-> 12-byte wire records (yf_network_pack_detections_device in include/yf_network.h) */
int  yf_engine_pack_detections_device(yf_engine* e, const void* d_dets, const void* d_counts, const void* d_heads, void* d_wire, long n, int cap, void* stream);
int  yf_engine_unpack_detections_device(yf_engine* e, const void* d_wire, const void* d_counts, void* d_heads, long n, int cap, void* stream);
int  yf_engine_prepare_rgb565_device(yf_engine* e, const void* d_rgb565, void* d_out, long n, void* stream);
/* camera frames -> heads (+ detections if d_dets != NULL) in one launch: frame preparation fused into the input staging */
int  yf_engine_run_camera_device(yf_engine* e, const void* d_rgb565, void* d_out, long n, int mode, float w_scale, float h_scale,
                                 void* d_dets, void* d_counts, int cap, void* stream);
/* 160x160 frames (int8 [n][160][160][3] -> [n][20][20][18]): layer-by-layer over an engine-owned HBM arena */
int  yf_engine_run_device_160(yf_engine* e, const void* d_in, void* d_out, long n, void* stream);
/* scratch regions are owned by launch streams (yf_stream_scratch.h): give a stream's regions back before destroying it; bytes held right now */
int  yf_engine_release_stream(yf_engine* e, void* stream);
size_t yf_engine_scratch_bytes(yf_engine* e);
void yf_engine_scratch_stats(yf_engine* e, unsigned long long out[6]);
long yf_engine_dump_bytes(void);
/* debug: dump build on n host frames; heads and the per-stage dump records come back to host memory (per-node observer) */
int  yf_engine_run_host_dump(yf_engine* e, const void* h_in, void* h_out, void* h_dump, long n);
/* byte offset of the tensor tflite op `op` produces inside a frame's dump record, -1 = not dumped */
long yf_engine_dump_offset(int tflite_op);

#ifdef __cplusplus
}
#endif
#endif
