/* C-ABI of libyf_network.so -- the MI355X drop-in for the reference's X-CUBE-AI network boundary.
 *
 * Part 1 re-declares, ABI-identically, the types and the eight+two entry points the reference application binds
 * (reference stm32/X-CUBE-AI/App/network.h:103-213, network_data.h:62-72, Middlewares/ST/AI/Inc/ai_platform.h).
 * The declarations are authored here (ST's headers are SLA0044-licensed and are not copied); tests/test_abi.py
 * compiles a probe against the reference's own headers and checks every sizeof/offsetof/enum value below.
 * A maintainer keeps including the reference's network.h / network_data.h and links this library in place of
 * network.c + network_data.c + NetworkRuntime700_CM7_Keil.lib.
 *
 * Part 2 declares the yf_* extension entry points (device-resident batches, detections, frame preparation, timing).
 *
 * Do not include this header together with the reference's ai_platform.h in one translation unit.
 */
#ifndef YF_NETWORK_H
#define YF_NETWORK_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define YF_API __attribute__((visibility("default")))          /* == AI_API_ENTRY, ai_platform.h:109-111 */
#else
#define YF_API
#endif

/* ------------------------------------------------------------------ Part 1: reference boundary ---------- */
typedef void*    ai_handle;                 /* ai_platform.h:431 */
typedef bool     ai_bool;
typedef int32_t  ai_i32;
typedef uint32_t ai_u32;
typedef uint16_t ai_u16;
typedef uint8_t  ai_u8;
typedef int8_t   ai_i8;
typedef uint32_t ai_signature;
typedef int32_t  ai_buffer_format;          /* ai_platform.h:211 */

#define AI_HANDLE_PTR(ptr_)   ((ai_handle)(ptr_))
#define AI_HANDLE_NULL        AI_HANDLE_PTR(NULL)
#define AI_MAGIC_MARKER       (0xA1FACADE)  /* ai_platform.h:129 */
#define AI_MAGIC_SIGNATURE    (0xa1facade)  /* ai_platform.h:85 */

/* ai_platform.h:392-413: AI_BUFFER_FMT_SET(type, sign, float, bits, fbits) evaluated (checked by tests/test_abi.py) */
#define AI_BUFFER_FORMAT_U8       ((ai_buffer_format)0x00040440)
#define AI_BUFFER_FORMAT_S8       ((ai_buffer_format)0x00840440)
#define AI_BUFFER_FMT_FLAG_CONST  (0x1U << 30)                   /* ai_platform.h:225 */

typedef struct ai_error_ {                  /* ai_platform.h:467-470 */
  ai_u32 type : 8;
  ai_u32 code : 24;
} ai_error;

typedef struct ai_buffer_meta_info_ ai_buffer_meta_info;         /* opaque here; always NULL on this path */

typedef struct ai_buffer_ {                 /* ai_platform.h:517-525; 32 bytes on LP64 */
  ai_buffer_format      format;
  ai_u16                n_batches;          /* u16: at most 65535 frames per ai_network_run call */
  ai_u16                height;
  ai_u16                width;
  ai_u32                channels;
  ai_handle             data;
  ai_buffer_meta_info*  meta_info;
} ai_buffer;

typedef struct ai_buffer_array_ {           /* ai_platform.h:532-536 */
  ai_u16      flags;
  ai_u16      size;
  ai_buffer*  buffer;
} ai_buffer_array;

typedef struct ai_platform_version_ { ai_u8 major, minor, micro, reserved; } ai_platform_version;   /* ai_platform.h:593-598 */

typedef struct ai_network_params_ {         /* ai_platform.h:348-357,606-608 */
  union {
    struct { ai_buffer params; ai_buffer activations; };
    struct { ai_signature map_signature; ai_buffer_array map_weights; ai_buffer_array map_activations; };
  };
} ai_network_params;

typedef struct ai_network_report_ {         /* ai_platform.h:626-655 */
  const char*          model_name;
  const char*          model_signature;
  const char*          model_datetime;
  const char*          compile_datetime;
  const char*          runtime_revision;
  ai_platform_version  runtime_version;
  const char*          tool_revision;
  ai_platform_version  tool_version;
  ai_platform_version  tool_api_version;
  ai_platform_version  api_version;
  ai_platform_version  interface_api_version;
  ai_u32               n_macc;
  ai_u16               n_inputs;
  ai_u16               n_outputs;
  ai_buffer*           inputs;
  ai_buffer*           outputs;
  union {
    struct { ai_buffer params; ai_buffer activations; };
    struct { ai_signature map_signature; ai_buffer_array map_weights; ai_buffer_array map_activations; };
  };
  ai_u32               n_nodes;
  ai_signature         signature;
} ai_network_report;

/* ai_platform.h:546-586 (values used on this path) */
enum {
  AI_ERROR_NONE = 0x00, AI_ERROR_TOOL_PLATFORM_API_MISMATCH = 0x01, AI_ERROR_TYPES_MISMATCH = 0x02,
  AI_ERROR_INVALID_HANDLE = 0x10, AI_ERROR_INVALID_STATE = 0x11, AI_ERROR_INVALID_INPUT = 0x12,
  AI_ERROR_INVALID_OUTPUT = 0x13, AI_ERROR_INVALID_PARAM = 0x14, AI_ERROR_INVALID_SIGNATURE = 0x15,
  AI_ERROR_INVALID_SIZE = 0x16, AI_ERROR_INVALID_VALUE = 0x17, AI_ERROR_INIT_FAILED = 0x30,
  AI_ERROR_ALLOCATION_FAILED = 0x31, AI_ERROR_DEALLOCATION_FAILED = 0x32, AI_ERROR_CREATE_FAILED = 0x33
};
enum {
  AI_ERROR_CODE_NONE = 0x0000, AI_ERROR_CODE_NETWORK = 0x0010, AI_ERROR_CODE_NETWORK_PARAMS = 0x0011,
  AI_ERROR_CODE_NETWORK_WEIGHTS = 0x0012, AI_ERROR_CODE_NETWORK_ACTIVATIONS = 0x0013, AI_ERROR_CODE_LAYER = 0x0014,
  AI_ERROR_CODE_TENSOR = 0x0015, AI_ERROR_CODE_ARRAY = 0x0016, AI_ERROR_CODE_INVALID_PTR = 0x0017,
  AI_ERROR_CODE_INVALID_SIZE = 0x0018, AI_ERROR_CODE_INVALID_FORMAT = 0x0019, AI_ERROR_CODE_OUT_OF_RANGE = 0x0020,
  AI_ERROR_CODE_INVALID_BATCH = 0x0021, AI_ERROR_CODE_MISSED_INIT = 0x0030, AI_ERROR_CODE_IN_USE = 0x0040
};

/* network.h:38-72, network_data.h:28-36 */
#define AI_NETWORK_IN_1_HEIGHT   (56)
#define AI_NETWORK_IN_1_WIDTH    (56)
#define AI_NETWORK_IN_1_CHANNEL  (3)
#define AI_NETWORK_IN_1_SIZE     (56 * 56 * 3)
#define AI_NETWORK_OUT_1_HEIGHT  (7)
#define AI_NETWORK_OUT_1_WIDTH   (7)
#define AI_NETWORK_OUT_1_CHANNEL (18)
#define AI_NETWORK_OUT_1_SIZE    (7 * 7 * 18)
#define AI_NETWORK_N_NODES       (31)
#define AI_NETWORK_DATA_ACTIVATIONS_SIZE (29784)
#define AI_NETWORK_DATA_WEIGHTS_SIZE     (11304)

/* replaces network.c:3369-3377 (ai_platform_network_create): binds *network to the static singleton context.
 * network_config must be NULL (AI_NETWORK_DATA_CONFIG, network_data.h:26). */
YF_API ai_error  ai_network_create(ai_handle* network, const ai_buffer* network_config);
/* replaces network.c:3385-3399: reads the weight blob the caller hands over (MAGIC-framed pointer map of
 * network_data.c:395-401, or the ai_buffer_array form of network_data.c:412-432), builds the device tables and
 * uploads them to HBM.  The activations arena is accepted and ignored (activations live in LDS). */
YF_API ai_bool   ai_network_init(ai_handle network, const ai_network_params* params);
/* replaces network.c:3402-3407 (ai_platform_network_process): input S8 56x56x3 x n_batches NHWC host memory,
 * output S8 7x7x18 x n_batches.  Returns the number of frames processed, <= 0 on failure. */
YF_API ai_i32    ai_network_run(ai_handle network, const ai_buffer* input, ai_buffer* output);
/* replaces network.c:3409-3413: run without returning an output */
YF_API ai_i32    ai_network_forward(ai_handle network, const ai_buffer* input);
/* replaces network.c:3363-3367: first error since the last call; reading resets it (network.h:120-132) */
YF_API ai_error  ai_network_get_error(ai_handle network);
/* replaces network.c:3379-3383 */
YF_API ai_handle ai_network_destroy(ai_handle network);
/* replace network.c:3271-3361.  Field for field what the reference's generated file answers: model_name "network", model_signature /
 * model_datetime / tool_revision = AI_NETWORK_MODEL_SIGNATURE / AI_TOOLS_DATE_TIME / AI_TOOLS_REVISION_ID of network.c:38-46, tool_version
 * 7.0.0, tool_api_version 1.4.0, api_version 1.1.0 (network_config.h:25-46), n_macc 1344320; get_report describes the buffers as
 * map_signature / map_weights / map_activations (network.c:3346-3348), the deprecated get_info as the legacy params / activations pair
 * (network.c:3301-3302).  compile_datetime, runtime_revision and runtime_version are this library's own. */
YF_API ai_bool   ai_network_get_info(ai_handle network, ai_network_report* report);
YF_API ai_bool   ai_network_get_report(ai_handle network, ai_network_report* report);
/* replace network_data.c:393-403 and :412-432 (this library ships its own copy of the weight blob) */
YF_API ai_handle ai_network_data_weights_get(void);
YF_API ai_bool   ai_network_data_params_get(ai_handle network, ai_network_params* params);
/* replaces the closed-library symbol the reference's own network_data.c:431 calls
 * (ai_platform_interface.h:869-872), so that file can also be compiled unchanged beside this library */
YF_API ai_bool   ai_platform_bind_network_params(ai_handle network, ai_network_params* params,
                                                 const ai_buffer_array* map_weights,
                                                 const ai_buffer_array* map_activations);

/* ------------------------------------------------------------------ Part 1b: runtime-level boundary ------
 * The symbols of ST's closed runtime that the reference's GENERATED network.c references (nm -u of network.c built
 * for x86-64): with these exported, network.c itself compiles and links unchanged and only the ST runtime library is
 * replaced (csrc/platform_abi.c).  `ai_network*` / `ai_context*` are opaque here (void*).
 * Replaces ai_platform_interface.h:786-966; the forward_* / nl_func / ai_sum_* kernels (layers_conv2d.h:192,
 * layers_pool.h:374, layers_generic.h:494,598, layers_nl.h:606, ai_math_helpers.h) are placeholders that are
 * referenced as function pointers by network.c's layer tables and never called. */
YF_API void*     ai_platform_context_acquire(const ai_handle handle);
YF_API ai_error  ai_platform_network_create(ai_handle* network, const ai_buffer* network_config, void* net_ctx,
                                            const ai_u8 tool_major, const ai_u8 tool_minor, const ai_u8 tool_micro);
YF_API ai_handle ai_platform_network_destroy(ai_handle network);
YF_API ai_error  ai_platform_network_get_error(ai_handle network);
YF_API void*     ai_platform_network_init(ai_handle network, const ai_network_params* params);
YF_API ai_bool   ai_platform_network_post_init(ai_handle network);
YF_API ai_i32    ai_platform_network_process(ai_handle network, const ai_buffer* input, ai_buffer* output);
YF_API ai_bool   ai_platform_get_weights_map(uint8_t** map, const uint32_t map_size, const ai_network_params* params);
YF_API ai_bool   ai_platform_get_activations_map(uint8_t** map, const uint32_t map_size, const ai_network_params* params);
YF_API ai_bool   ai_platform_api_get_network_report(ai_handle network, ai_network_report* r);
YF_API const char* ai_platform_runtime_get_revision(void);
YF_API ai_platform_version ai_platform_runtime_get_version(void);
YF_API ai_platform_version ai_platform_api_get_version(void);
YF_API ai_platform_version ai_platform_interface_api_get_version(void);
YF_API void      forward_conv2d_integer_SSSA_ch(void* layer);
YF_API void      forward_mp_integer_INT8(void* layer);
YF_API void      forward_eltwise_integer_INT8(void* layer);
YF_API void      forward_concat(void* layer);
YF_API void      nl_func_array_integer(void);
YF_API void      ai_sum_f32(void);
YF_API void      ai_sum_buffer_INT8(void);

/* ------------------------------------------------------------------ Part 2: extensions ------------------- */
typedef struct yf_det_ {      /* one detection; mirrors the fields printed at yoloface.c:148 / drawn at tflite_prediction.py:63 */
  int32_t frame;
  uint8_t anchor, row, col;
  int8_t  q_conf;             /* raw int8 confidence logit */
  float   conf;               /* sigmoid of the dequantised logit */
  int32_t x1, y1, x2, y2;
} yf_det;

enum { YF_DECODE_PY = 0,      /* yoloface/tflite/tflite_prediction.py:42-63: anchor-major, conf > 0.7, xyxy * scale, int32
                                 (float -> int32 as numpy on an x86-64 PC: truncation, out of range -> INT32_MIN) */
       YF_DECODE_FW = 1,      /* stm32/X-CUBE-AI/App/yoloface.c:98-152 on the Cortex-M7: cell-major, conf >= 0.7, axis swap,
                                 clamp, x2; float -> int is VCVT (truncation, SATURATING: a box edge beyond 2^31 becomes
                                 INT32_MAX and prints as -2 after the x2) */
       YF_DECODE_FW_HOST = 2 };/* the same loop compiled for an x86-64 host (cvttss2si: out of range -> INT32_MIN, which the
                                 clamps then turn into 0 / 55); equals YF_DECODE_FW whenever every edge fits in int32 */

/* Rounding of the requantisation step (TFLite MultiplyByQuantizedMultiplier) behind every conv / add / quantize of the int8 network.
 * "Bit-exact vs tflite" needs a definition of WHICH TFLite: the reference's script builds tf.lite.Interpreter with default arguments
 * (yoloface/tflite/tflite_prediction.py:23), i.e. the default op resolver, and TensorFlow 2.10 cannot be run where this library is
 * built, so the choice is the integrator's (DESIGN.md section 2 has the measured distance between the forms):
 *   YF_ROUND_TFLITE_REF   (default) the builtin REFERENCE kernels: SRDHM, then RoundingDivideByPOT (ties away from zero) everywhere.
 *                         SURVEY.md 8(c).3's definition of "the tflite int8 reference" (experimental_op_resolver_type=BUILTIN_REF).
 *   YF_ROUND_TIES_UP      dense CONV_2D as the default resolver's optimized kernels route it on x86 / ARM (ruy: sqrdmulh + srshl, the right
 *                         shift breaks ties UPWARD); DEPTHWISE_CONV_2D, LEAKY_RELU, ADD, QUANTIZE keep the reference form.
 *   YF_ROUND_TIES_UP_ALL  every op ties upward.
 *   YF_ROUND_SINGLE       dense CONV_2D with ONE rounding, (acc*M + 2^(30-shift)) >> (31-shift) (ruy's portable path); the rest reference.
 * The rounding lives in the per-channel constants {C64, ZR}, the LeakyReLU / QUANTIZE byte tables and the add tables that ai_network_init builds
 * (csrc/yf_host_prep.c); the four-instruction requantisation of the reference rounding's kernels serves every one of them.  The three roundings whose
 * dense convolutions have NO sign term (ties upward, single rounding) by default run a second set of the same kernels whose dense convolutions
 * requantise in THREE instructions (no carry, ZR folded into C64: 5.5 % less kernel time, profiles/r06_ties_up_epilogue_ab.txt); or-ing
 * YF_ROUND_GENERIC_KERNELS into the rounding keeps them on the reference rounding's kernels (same results; for A/B).  Call after ai_network_create, before or after
 * ai_network_init (a ready network waits for its launches, rebuilds its tables from the weights it was initialised with -- which the caller
 * still owns, as on the MCU -- and uploads them); ai_network_create resets the choice to $YF_REQUANT_ROUNDING ("ref", "ties_up",
 * "ties_up_all", "single", each optionally followed by "+generic"; unset = ref) so that an unmodified aiInit() can be steered from outside.  Affects the 56x56 and 160x160 int8
 * paths; the fp16 path has no requantisation.  Returns 0, or -1 with an error latched. */
#ifndef YF_ROUND_ENUM
#define YF_ROUND_ENUM
enum { YF_ROUND_TFLITE_REF = 0, YF_ROUND_TIES_UP = 1, YF_ROUND_TIES_UP_ALL = 2, YF_ROUND_SINGLE = 3, YF_ROUND_COUNT, YF_ROUND_GENERIC_KERNELS = 0x100 };
#endif
YF_API int  yf_network_set_requant_rounding(ai_handle network, int rounding);
YF_API int  yf_network_get_requant_rounding(ai_handle network);      /* the rounding in force, -1 for an invalid handle */
/* Select GPU (default 0 / $LOCAL_RANK is NOT read here; the caller decides).  Call before ai_network_init.  ONE DEVICE PER PROCESS: the library keeps
 * one network instance (like the reference: network.c:2929-2939) and its decode tables live in device symbols of the device the instance was
 * initialised on; a host that drives several GPUs starts one process per GPU (tools/c_host/yf_ranks.c, bench.py). */
YF_API int  yf_network_set_device(ai_handle network, int device);
/* Kernel variant: frames per workgroup (1,2,4) and waves per workgroup (4,8); 0 keeps the current value.  Without a call the choice
 * is automatic -- the throughput shape <2,8>, and batches of up to 512 frames one frame per workgroup (<1,8>: a 1-frame batch takes
 * 24 us instead of 34) -- and frames_per_wg = -1 returns to that.  A configured shape runs every batch size. */
YF_API int  yf_network_configure(ai_handle network, int frames_per_wg, int waves_per_wg);
/* Device-resident batch: d_in int8[n][56][56][3], d_out int8[n][7][7][18], both in HBM, d_in 4-byte aligned.
 * stream is a hipStream_t (NULL = default stream); asynchronous.  Returns n or <= 0 (error latched). */
YF_API long yf_network_run_device(ai_handle network, const void* d_in, void* d_out, long n, void* stream);
/* Same, and additionally dumps every fused stage's tensor (per-layer parity debugging, the counterpart of the
 * reference observer API, ai_platform_interface.h:684-731).  d_dump int8[n][yf_network_dump_bytes()]. */
YF_API long yf_network_run_device_dump(ai_handle network, const void* d_in, void* d_out, void* d_dump, long n, void* stream);
YF_API long yf_network_dump_bytes(void);
/* The network is fully convolutional; the reference ABI fixes 56x56 (network.h:48-50).  This extension also runs
 * 160x160 frames (BASELINE configs[4]): d_in int8[n][h][w][3] -> d_out int8[n][h/8][w/8][18].  56x56 takes the fused
 * LDS-resident kernel, 160x160 runs the same stage code as four banded kernels (each a group of fused stages over row bands
 * staged through LDS) over a per-frame HBM arena owned by the library, one arena per launch stream. */
YF_API long yf_network_run_device_hw(ai_handle network, int height, int width, const void* d_in, void* d_out, long n, void* stream);
/* Box decode on the GPU from device-resident heads: d_dets yf_det[n][cap], d_counts int32[n] (true count, may
 * exceed cap).  mode = YF_DECODE_PY, YF_DECODE_FW or YF_DECODE_FW_HOST. */
YF_API long yf_network_decode_device(ai_handle network, const void* d_heads, long n, int mode, float w_scale, float h_scale,
                                     void* d_dets, void* d_counts, int cap, void* stream);
/* Network and box decode in ONE launch (the firmware's ai_network_run + post-processing, yoloface.c:98-152 /
 * tflite_prediction.py:39-63, back to back): heads go to d_heads as in yf_network_run_device and every workgroup
 * decodes its frames' heads while they are still on chip.  Same records as yf_network_decode_device. */
YF_API long yf_network_run_decode_device(ai_handle network, const void* d_in, void* d_heads, long n, int mode, float w_scale, float h_scale,
                                         void* d_dets, void* d_counts, int cap, void* stream);
/* The firmware's UART text for one frame (stm32/User/main.c:46,53 and yoloface.c:148, byte for byte, CR LF line ends):
 *   === Frame N ===  /  40 dashes  /  [Face k] BBox: [x1, y1, x2, y2], Conf: c.cc  per record  /  40 dashes  /
 *   [INFO] Total faces detected: count
 * dets: the frame's firmware-mode records (host memory), count: its candidate count (lines are written for
 * min(count, cap) records, the total line carries count like the firmware's face_num).  Returns the number of
 * bytes the text needs (excluding the terminating NUL); at most buflen - 1 bytes are stored.  Host-only, no GPU. */
YF_API long yf_network_format_uart(unsigned frame_no, const yf_det* dets, int count, int cap, char* buf, size_t buflen);
/* Diagnostic: the byte offsets of the device table blob that are compiled into the kernels -- w_off[17], c_off[17] of the
 * dense stages, g_off[7] of the depthwise stages, lut_off, total (43 ints).  ai_network_init refuses to start when the
 * blob built from the caller's weights is laid out differently.  Returns the number of ints; host-only, no GPU. */
YF_API int yf_network_table_plan(int32_t* out, int cap);
/* Multi-GPU: one process per GPU, each with its own network instance on its device
 * (yf_network_set_device(local_rank) before ai_network_init).  Frames are independent: rank r of `world` runs the contiguous
 * slice [*begin, *end) of an n-frame batch (the first n % world ranks take one frame more) ... */
YF_API void yf_network_shard_range(long n, int rank, int world, long* begin, long* end);
/* ... and the only exchange is an all-gather of per-frame results (detection records + counts, optionally heads): every
 * rank contributes `bytes_per_rank` device bytes at d_send and receives world * bytes_per_rank at d_recv, in rank = frame
 * order.  nccl_comm is the host application's ncclComm_t (RCCL; the library resolves ncclAllGather from librccl.so at
 * the first call and links nothing at build time).  Asynchronous on `stream`.  Returns bytes_per_rank or <= 0. */
YF_API long yf_network_all_gather_device(ai_handle network, void* nccl_comm, const void* d_send, void* d_recv,
                                         size_t bytes_per_rank, void* stream);
/* Compact WIRE form of the detection records for that exchange (12 bytes instead of sizeof(yf_det) = 28): {u8 anchor, row, col, 0, i8 q[6], u16 0} -- the
 * firing cell and the six int8 head values of its anchor, from which frame index (= the record's position), confidence and box edges follow through the
 * decode tables: lossless whatever the edges are.  pack: d_dets yf_det[n][cap], d_counts int32[n], d_heads int8[n][7][7][18] -> d_wire uint8[n][cap][12]
 * (slots beyond min(count, cap) zeroed).  unpack: d_wire + the sender's d_counts -> d_heads int8[n][7][7][18], -128 everywhere except the transmitted
 * cells; yf_network_decode_device on it reproduces the sender's records (min(count, cap) per frame, in the sender's order).  One launch each, asynchronous
 * on `stream`; return n or <= 0. */
YF_API long yf_network_pack_detections_device(ai_handle network, const void* d_dets, const void* d_counts, const void* d_heads, void* d_wire, long n, int cap, void* stream);
YF_API long yf_network_unpack_detections_device(ai_handle network, const void* d_wire, const void* d_counts, void* d_heads, long n, int cap, void* stream);
/* Frame preparation on the GPU (yoloface.c:26-93): d_rgb565 uint8[n][112*112*2] big-endian RGB565 -> d_out int8[n][56][56][3]. */
YF_API long yf_network_prepare_rgb565_device(ai_handle network, const void* d_rgb565, void* d_out, long n, void* stream);
/* The firmware's whole per-frame pipeline in ONE launch (stm32/User/main.c:42-54): camera frames d_rgb565 uint8[n][112][112][2]
 * (big-endian RGB565, as OV_Frame.c leaves them in RGB_DATA; 16-byte aligned) -> resize_rgb565_uint8_112_to_56_direct +
 * prepare_yolo_data (yoloface.c:26-93, inside the kernel's input staging: the prepared int8 frame never exists in HBM) ->
 * ai_network_run -> heads d_heads int8[n][7][7][18] -> post_process when d_dets != NULL (records and counts as
 * yf_network_run_decode_device; pass d_dets = NULL for heads only).  Same results as yf_network_prepare_rgb565_device followed by
 * yf_network_run_decode_device.  Returns n or <= 0. */
YF_API long yf_network_run_camera_device(ai_handle network, const void* d_rgb565, void* d_heads, long n, int mode, float w_scale, float h_scale,
                                         void* d_dets, void* d_counts, int cap, void* stream);
/* `iters` back-to-back launches of the fused kernel on `stream`, bracketed by HIP events on that stream;
 * *ms_per_launch receives the average.  Synchronises the stream. */
YF_API long yf_network_time_device(ai_handle network, const void* d_in, void* d_out, long n, int iters, void* stream,
                                   float* ms_per_launch);
/* Debug: the stage-dump build of the kernel, left after `stop_stage` fused stages (1 = input staging ... 25 = conv2d_51;
 * <= 0 or > 25 = whole network): the differences between successive stops give a per-stage time profile. */
YF_API long yf_network_time_stages(ai_handle network, const void* d_in, void* d_out, long n, int iters, int stop_stage,
                                   void* stream, float* ms_per_launch);
/* fp16 side configuration (BASELINE configs[3]): the reference's fp32 ONNX export (yoloface/pytorch/yoloface-50k.onnx)
 * with fp16 weights/activations and fp32 accumulation, dense convs on v_mfma_f32_16x16x32_f16.  `yfw` is the weight
 * pack written by tools/gen_fp16_model.py (stm32h7-yolo_amd/model/yoloface_fp32.yfw).  d_in_f16: fp16 [n][56][56][3]
 * (pixel/255), d_out_f32: fp32 logits [n][7][7][18].  Independent of ai_network_init; needs only ai_network_create. */
YF_API int  yf_network_fp16_init(ai_handle network, const void* yfw, size_t bytes);
YF_API long yf_network_fp16_run_device(ai_handle network, const void* d_in_f16, void* d_out_f32, long n, void* stream);
/* Text of the last HIP/runtime failure (empty string if none). */
/* Device scratch (the fused kernels' park slots, the 160x160 arena) is owned by the launch stream and bounded: at most eight regions per kind, a region
 * whose last launch has completed is handed to the next stream that asks.  Release a stream's regions before destroying the stream; the second call
 * reports the bytes held right now.  (The reference has one context and no streams: network.c:2929-2939.) */
YF_API int yf_network_release_stream(ai_handle network, void* stream);
YF_API size_t yf_network_scratch_bytes(ai_handle network);
/* What the scratch maps did so far (all kinds of region summed; csrc/yf_stream_scratch.h): a launch whose stream has the library to itself skips the
 * event behind it (events_skipped), in company it records one (events_recorded); a NEW stream that finds every region busy waits for the event of
 * the region used longest ago (event_waits), for the whole DEVICE when only unnamed (dirty) regions are left (device_syncs: a hipDeviceSynchronize
 * inside the launch call -- a host that sees this number grow keeps more streams alive than there are regions, or drops streams without
 * yf_network_release_stream), or for another host thread's launch call to finish (acquire_waits).  Returns 0, -1 for an invalid handle. */
typedef struct yf_scratch_stats_ {
  unsigned long long events_recorded, events_skipped, event_waits, device_syncs, acquire_waits, regions;
} yf_scratch_stats;
YF_API int yf_network_scratch_stats(ai_handle network, yf_scratch_stats* out);
YF_API const char* yf_network_last_error_text(ai_handle network);
YF_API const char* yf_network_kernel_name(ai_handle network);
/* Name of the kernel shape a batch of n frames runs (see yf_network_configure). */
YF_API const char* yf_network_kernel_name_for(ai_handle network, long n);
/* Identity of the device code inside this library: the first 16 hex digits of the sha256 over the device sources and the
 * compiler flags they were built with (csrc/Makefile, BUILD_ID).  Profiles are stamped with it; bench.py reports counters of a
 * profile only when the stamp equals the id of the library it is running.  Host-only, no GPU. */
YF_API const char* yf_network_build_id(void);
/* ... and of the C host layer (sha256 over csrc/flags.mk HOST_SRCS and the C flags): binding.py checks both when it cannot run make */
YF_API const char* yf_network_host_id(void);

/* ---- per-node observer of the runtime-level interface (reference ai_platform_interface.h:684-731 datatypes, 981-1024 entry points).
 * Available to callers that link the reference's generated network.c against this library (the node list is theirs): an observed
 * ai_network_run goes through the debug build of the kernel, which dumps every node's output tensor, and the registered client is
 * called before / after every c-node, frame by frame, with the node's tensor chain; the node's output tensor then holds that node's
 * result at the address the caller's graph gives it.  AI_OBSERVER_INIT_EVT, `inner_tensors` and the scratch tensors' contents are not
 * reproduced (csrc/platform_abi.c). */
typedef struct __attribute__((packed, aligned(4))) ai_observer_node_s {
  ai_u16 c_idx;                 /* node index (position in the execution list) */
  ai_u16 type;                  /* node type */
  ai_u16 id;                    /* id the code generator gave the model layer */
  ai_u16 unused;
  const void* inner_tensors;    /* const ai_tensor_chain*: always NULL here */
  const void* tensors;          /* const ai_tensor_chain*: the caller's own chain of the node */
} ai_observer_node;
#define AI_OBSERVER_NONE_EVT    (0)
#define AI_OBSERVER_INIT_EVT    (1 << 0)
#define AI_OBSERVER_PRE_EVT     (1 << 1)
#define AI_OBSERVER_POST_EVT    (1 << 2)
#define AI_OBSERVER_FIRST_EVT   (1 << 8)
#define AI_OBSERVER_LAST_EVT    (1 << 9)
#define AI_OBSERVER_REGISTERED  (1 << 24)
#define AI_OBSERVER_MASK_EVT    (0xFF)
typedef ai_u32 (*ai_observer_node_cb)(const ai_handle cookie, const ai_u32 flags, const ai_observer_node* node);
struct ai_node_s;
typedef struct __attribute__((packed, aligned(4))) ai_observer_exec_ctx_s {
  ai_observer_node_cb on_node; ai_handle cookie; ai_u32 flags; ai_u16 c_idx; ai_u16 n_nodes; struct ai_node_s* cur;
} ai_observer_exec_ctx;
YF_API ai_bool ai_platform_observer_node_info(ai_handle network, ai_observer_node* node_info);
YF_API ai_bool ai_platform_observer_register(ai_handle network, ai_observer_node_cb cb, ai_handle cookie, ai_u32 flags);
YF_API ai_bool ai_platform_observer_register_s(ai_handle network, ai_observer_exec_ctx* ctx);
YF_API ai_bool ai_platform_observer_unregister(ai_handle network, ai_observer_node_cb cb, ai_handle cookie);
YF_API ai_bool ai_platform_observer_unregister_s(ai_handle network, ai_observer_exec_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* YF_NETWORK_H */
