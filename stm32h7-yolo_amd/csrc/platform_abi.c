/* Runtime-level drop-in: the symbols of ST's closed NetworkRuntime700 library that the
 * reference's GENERATED network.c references, so that file can be compiled unchanged and linked against this library
 * instead of the ST runtime.  Declarations replaced: Middlewares/ST/AI/Inc/ai_platform_interface.h:786-966
 * (ai_platform_*), layers_conv2d.h:192, layers_pool.h:374, layers_generic.h:494,598, layers_nl.h:606,
 * ai_math_helpers.h (forward_* / nl_func / ai_sum_* -- only referenced as function pointers in the layer tables).
 *
 * network.c keeps ownership of its static ai_network object (`g_network`, network.c:2929-2939); this layer treats it
 * as an opaque tag: ai_platform_network_create hands it back as the handle, every other call checks the tag and
 * forwards to the fused engine.  The node list, tensors and arrays of network.c are never walked: the graph is the
 * baked yoloface graph (weights come from the caller's blob through ai_platform_get_weights_map's params).
 */
#include "yf_impl.h"
#include <stdio.h>
#include <stdlib.h>

typedef uint8_t* ai_ptr;                      /* ai_platform_interface.h: ai_ptr */
typedef uint32_t ai_size;

static void* g_tag;                           /* the caller's ai_network object (opaque) */
static ai_handle g_own;                       /* this library's context */

static ai_handle own(ai_handle h) { return (h && h == g_tag) ? g_own : AI_HANDLE_NULL; }

YF_API void* ai_platform_context_acquire(const ai_handle handle) { return own(handle) ? handle : NULL; }

YF_API ai_error ai_platform_network_create(ai_handle* network, const ai_buffer* network_config, void* net_ctx,
                                           const ai_u8 tool_major, const ai_u8 tool_minor, const ai_u8 tool_micro) {
  ai_error e; e.type = AI_ERROR_NONE; e.code = AI_ERROR_CODE_NONE;
  (void)tool_minor; (void)tool_micro;
  if (!network || !net_ctx) { e.type = AI_ERROR_CREATE_FAILED; e.code = AI_ERROR_CODE_INVALID_PTR; return e; }
  if (tool_major != 1) {                       /* tools API 1.x generated the reference model (network_config.h:34-46) */
    *network = AI_HANDLE_NULL; e.type = AI_ERROR_TOOL_PLATFORM_API_MISMATCH; e.code = AI_ERROR_CODE_NETWORK; return e;
  }
  e = yf_impl_create(&g_own, network_config);
  if (e.type != AI_ERROR_NONE) { *network = AI_HANDLE_NULL; return e; }
  g_tag = net_ctx;
  *network = (ai_handle)net_ctx;
  return e;
}

YF_API ai_handle ai_platform_network_destroy(ai_handle network) {
  if (!own(network)) return network;
  yf_impl_destroy(g_own);
  g_own = AI_HANDLE_NULL; g_tag = NULL;
  return AI_HANDLE_NULL;
}

YF_API ai_error ai_platform_network_get_error(ai_handle network) {
  if (!own(network)) { ai_error e; e.type = AI_ERROR_INVALID_HANDLE; e.code = AI_ERROR_CODE_NETWORK; return e; }
  return yf_impl_get_error(g_own);
}

/* returns the caller's context on success (network.c:3388-3389 treats NULL as failure) */
YF_API void* ai_platform_network_init(ai_handle network, const ai_network_params* params) {
  if (!own(network)) return NULL;
  return yf_impl_init(g_own, params) ? network : NULL;
}

YF_API ai_bool ai_platform_network_post_init(ai_handle network) { return own(network) != AI_HANDLE_NULL; }

YF_API ai_i32 ai_platform_network_process(ai_handle network, const ai_buffer* input, ai_buffer* output) {
  if (!own(network)) return 0;
  return output ? yf_impl_run(g_own, input, output) : yf_impl_forward(g_own, input);
}

/* network_configure_weights / _activations (network.c:3108-3267, 2943-3104) ask for the base pointers and then bind
 * their own arrays to fixed offsets; the engine does not use those arrays, but the calls must succeed. */
YF_API ai_bool ai_platform_get_weights_map(ai_ptr* map, const ai_size map_size, const ai_network_params* params) {
  if (!map || map_size < 1 || !params) return false;
  size_t bytes = 0; const ai_buffer* act = NULL;
  const uint8_t* blob = yf_impl_resolve_weights(params, &bytes, &act);
  if (!blob) return false;
  map[0] = (ai_ptr)(uintptr_t)blob;
  return true;
}

YF_API ai_bool ai_platform_get_activations_map(ai_ptr* map, const ai_size map_size, const ai_network_params* params) {
  if (!map || map_size < 1 || !params) return false;
  size_t bytes = 0; const ai_buffer* act = NULL;
  if (!yf_impl_resolve_weights(params, &bytes, &act)) return false;
  map[0] = act ? (ai_ptr)act->data : NULL;
  return true;
}

/* network.c:3317-3361 pre-fills names, dates and MACC; the runtime completes I/O descriptors and counts */
YF_API ai_bool ai_platform_api_get_network_report(ai_handle network, ai_network_report* r) {
  if (!own(network) || !r) return false;
  ai_network_report mine;
  if (!yf_impl_get_report(g_own, &mine)) return false;
  r->n_inputs = mine.n_inputs; r->n_outputs = mine.n_outputs;
  r->inputs = mine.inputs; r->outputs = mine.outputs;
  r->n_nodes = mine.n_nodes;
  r->signature = 0;
  return true;
}

YF_API const char* ai_platform_runtime_get_revision(void) { return "yf-mi355x (gfx950 fused int8 engine)"; }
static ai_platform_version ver(unsigned a, unsigned b, unsigned c) { ai_platform_version v; v.major = (ai_u8)a; v.minor = (ai_u8)b; v.micro = (ai_u8)c; v.reserved = 0; return v; }
YF_API ai_platform_version ai_platform_runtime_get_version(void) { return ver(0, 1, 0); }
YF_API ai_platform_version ai_platform_api_get_version(void) { return ver(1, 1, 0); }
YF_API ai_platform_version ai_platform_interface_api_get_version(void) { return ver(1, 3, 0); }

/* Layer kernels of the ST runtime: network.c stores their addresses in its layer objects (network.c:2204-2927) but
 * nothing on this path ever calls them -- the fused engine replaces the node walk.  Calling one is a usage error. */
static void not_a_kernel(const char* name) {
  fprintf(stderr, "libyf_network: %s() is a placeholder for the ST runtime's per-layer kernel and must not be called\n", name);
  abort();
}
YF_API void forward_conv2d_integer_SSSA_ch(void* layer) { (void)layer; not_a_kernel("forward_conv2d_integer_SSSA_ch"); }
YF_API void forward_mp_integer_INT8(void* layer) { (void)layer; not_a_kernel("forward_mp_integer_INT8"); }
YF_API void forward_eltwise_integer_INT8(void* layer) { (void)layer; not_a_kernel("forward_eltwise_integer_INT8"); }
YF_API void forward_concat(void* layer) { (void)layer; not_a_kernel("forward_concat"); }
YF_API void nl_func_array_integer(void) { not_a_kernel("nl_func_array_integer"); }
YF_API void ai_sum_f32(void) { not_a_kernel("ai_sum_f32"); }
YF_API void ai_sum_buffer_INT8(void) { not_a_kernel("ai_sum_buffer_INT8"); }
