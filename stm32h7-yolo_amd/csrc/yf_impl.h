/* Private names of the boundary implementation (network_abi.c), shared with the runtime-level layer (platform_abi.c). */
#ifndef YF_IMPL_H
#define YF_IMPL_H
#include "../../include/yf_network.h"
ai_error  yf_impl_create(ai_handle* network, const ai_buffer* network_config);
ai_handle yf_impl_destroy(ai_handle network);
ai_error  yf_impl_get_error(ai_handle network);
ai_bool   yf_impl_init(ai_handle network, const ai_network_params* params);
ai_i32    yf_impl_run(ai_handle network, const ai_buffer* input, ai_buffer* output);
ai_i32    yf_impl_forward(ai_handle network, const ai_buffer* input);
ai_bool   yf_impl_get_report(ai_handle network, ai_network_report* report);      /* ai_network_get_report: map_weights / map_activations arm */
ai_bool   yf_impl_get_info(ai_handle network, ai_network_report* report);        /* ai_network_get_info (deprecated): params / activations arm */
ai_bool   yf_impl_fill_report(ai_handle network, ai_network_report* report);     /* the runtime's half: what ai_platform_api_get_network_report adds to a pre-filled report */
void      yf_impl_set_tools_api_version(ai_handle network, unsigned major, unsigned minor, unsigned micro);
const char*         yf_impl_runtime_revision(void);
ai_platform_version yf_impl_runtime_version(void);
ai_platform_version yf_impl_api_version(void);
ai_platform_version yf_impl_interface_api_version(void);
const uint8_t* yf_impl_resolve_weights(const ai_network_params* p, size_t* bytes, const ai_buffer** act);
/* per-node observer support: debug build on frames [first, first + count) of input; heads + per-node dump records to host memory */
ai_i32    yf_impl_run_dump(ai_handle network, const ai_buffer* input, const ai_buffer* output, long first, long count, int8_t* heads, int8_t* dump);
long      yf_impl_dump_bytes(void);
long      yf_impl_dump_offset(int tflite_op);
void      yf_impl_fail_run(ai_handle network, unsigned code, const char* text);
/* latch an initialisation failure (first error wins, text replaces the previous one) */
void      yf_impl_fail_init(ai_handle network, unsigned code, const char* text);
#endif
