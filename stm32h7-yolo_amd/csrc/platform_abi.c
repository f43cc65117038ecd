/* Runtime-level drop-in: the symbols of ST's closed NetworkRuntime700 library that the
 * reference's GENERATED network.c references, so that file can be compiled unchanged and linked against this library
 * instead of the ST runtime.  Declarations replaced: Middlewares/ST/AI/Inc/ai_platform_interface.h:786-966
 * (ai_platform_*), layers_conv2d.h:192, layers_pool.h:374, layers_generic.h:494,598, layers_nl.h:606,
 * ai_math_helpers.h (forward_* / nl_func / ai_sum_* -- only referenced as function pointers in the layer tables).
 *
 * network.c keeps ownership of its static ai_network object (`g_network`, network.c:2929-2939):
 * ai_platform_network_create hands it back as the handle, every other call checks it and forwards to the fused engine.
 * The engine implements ONE graph -- the 31 c-layers of the yoloface model.  ai_platform_network_init therefore walks
 * the caller's node list (from ai_network.input_node, the way the ST runtime's scheduler does, core_common.h:101-109)
 * and compares every node -- kind, kernel, stride, padding, groups, fused non-linearity, output shape, weight count, and
 * the QUANTISATION of every tensor it touches (scale and zero point of inputs, pre-activation and output tensors, the
 * per-channel filter scales; reference network.c:663-1341) -- with the graph and the tables the engine was built for; any difference latches AI_ERROR_INIT_FAILED instead of silently running a
 * different network.  Weights come from the caller's blob (ai_platform_get_weights_map's params).
 */
#include "yf_impl.h"
#include "st_graph_view.h"
#include "gen/yf_model_gen.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t* ai_ptr;                      /* ai_platform_interface.h: ai_ptr */
typedef uint32_t ai_size;

static void* g_tag;                           /* the caller's ai_network object (opaque) */
static ai_handle g_own;                       /* this library's context */

static ai_handle own(ai_handle h) { return (h && h == g_tag) ? g_own : AI_HANDLE_NULL; }

YF_API void* ai_platform_context_acquire(const ai_handle handle) { return own(handle) ? handle : NULL; }

YF_API ai_error ai_platform_network_create(ai_handle* network, const ai_buffer* network_config, void* net_ctx,
                                           const ai_u8 tool_major, const ai_u8 tool_minor, const ai_u8 tool_micro) {
  ai_error e; e.type = AI_ERROR_NONE; e.code = AI_ERROR_CODE_NONE;
  if (!network || !net_ctx) { e.type = AI_ERROR_CREATE_FAILED; e.code = AI_ERROR_CODE_INVALID_PTR; return e; }
  if (tool_major != 1) {                       /* tools API 1.x generated the reference model (network_config.h:34-46) */
    *network = AI_HANDLE_NULL; e.type = AI_ERROR_TOOL_PLATFORM_API_MISMATCH; e.code = AI_ERROR_CODE_NETWORK; return e;
  }
  e = yf_impl_create(&g_own, network_config);
  if (e.type != AI_ERROR_NONE) { *network = AI_HANDLE_NULL; return e; }
  yf_impl_set_tools_api_version(g_own, tool_major, tool_minor, tool_micro);     /* reports return it as tool_api_version */
  g_tag = net_ctx;
  *network = (ai_handle)net_ctx;
  return e;
}

static void observer_reset(void);

YF_API ai_handle ai_platform_network_destroy(ai_handle network) {
  if (!own(network)) return network;
  yf_impl_destroy(g_own);
  g_own = AI_HANDLE_NULL; g_tag = NULL;
  observer_reset();                              /* a registration does not outlive its network */
  return AI_HANDLE_NULL;
}

YF_API ai_error ai_platform_network_get_error(ai_handle network) {
  if (!own(network)) { ai_error e; e.type = AI_ERROR_INVALID_HANDLE; e.code = AI_ERROR_CODE_NETWORK; return e; }
  return yf_impl_get_error(g_own);
}

/* ---- the graph the fused engine implements, in execution order (reference network.c:2204-2927, report
 * network_generate_report.txt:290-480).  pad = {x0, y0, x1, y1} as network.c states it. */
typedef struct { uint16_t id; uint8_t kind; uint16_t groups; uint8_t k, stride, nl; uint8_t pad[4]; uint16_t oh, ow, oc; uint32_t weights; } yf_expected_node;
enum { K_CONV = 0, K_POOL = 1, K_ADD = 2, K_CONCAT = 3 };
static const yf_expected_node k_graph[31] = {
  /* id  kind      groups k  s  nl  pad            out H W C      weight elements */
  {  2, K_CONV,      1, 3, 2, 1, {1, 1, 0, 0}, 28, 28,  8,   216}, {  4, K_CONV,   8, 3, 1, 1, {1, 1, 1, 1}, 28, 28,  8,    72},
  {  5, K_CONV,      1, 1, 1, 0, {0, 0, 0, 0}, 28, 28,  4,    32}, {  7, K_CONV,   1, 1, 1, 1, {0, 0, 0, 0}, 28, 28, 18,    72},
  { 11, K_CONV,     18, 3, 2, 1, {1, 1, 0, 0}, 14, 14, 18,   162}, { 12, K_CONV,   1, 1, 1, 0, {0, 0, 0, 0}, 14, 14,  6,   108},
  { 14, K_CONV,      1, 1, 1, 1, {0, 0, 0, 0}, 14, 14, 36,   216}, { 16, K_CONV,  36, 3, 1, 1, {1, 1, 1, 1}, 14, 14, 36,   324},
  { 17, K_CONV,      1, 1, 1, 0, {0, 0, 0, 0}, 14, 14,  6,   216}, { 18, K_ADD,    0, 0, 0, 0, {0, 0, 0, 0}, 14, 14,  6,     0},
  { 20, K_CONV,      1, 1, 1, 1, {0, 0, 0, 0}, 14, 14, 18,   108}, {  8, K_POOL,   0, 8, 2, 0, {3, 3, 4, 4}, 14, 14, 18,     0},
  { 22, K_CONCAT,    0, 0, 0, 0, {0, 0, 0, 0}, 14, 14, 36,     0}, { 24, K_CONV,   1, 1, 1, 1, {0, 0, 0, 0}, 14, 14, 24,   864},
  { 28, K_CONV,     24, 3, 2, 1, {1, 1, 0, 0},  7,  7, 24,   216}, { 29, K_CONV,   1, 1, 1, 0, {0, 0, 0, 0},  7,  7,  8,   192},
  { 31, K_CONV,      1, 1, 1, 1, {0, 0, 0, 0},  7,  7, 40,   320}, { 33, K_CONV,  40, 3, 1, 1, {1, 1, 1, 1},  7,  7, 40,   360},
  { 34, K_CONV,      1, 1, 1, 0, {0, 0, 0, 0},  7,  7,  8,   320}, { 35, K_ADD,    0, 0, 0, 0, {0, 0, 0, 0},  7,  7,  8,     0},
  { 37, K_CONV,      1, 1, 1, 1, {0, 0, 0, 0},  7,  7, 40,   320}, { 39, K_CONV,  40, 3, 1, 1, {1, 1, 1, 1},  7,  7, 40,   360},
  { 40, K_CONV,      1, 1, 1, 0, {0, 0, 0, 0},  7,  7,  8,   320}, { 41, K_ADD,    0, 0, 0, 0, {0, 0, 0, 0},  7,  7,  8,     0},
  { 43, K_CONV,      1, 1, 1, 1, {0, 0, 0, 0},  7,  7, 24,   192}, { 25, K_POOL,   0, 4, 2, 0, {1, 1, 2, 2},  7,  7, 24,     0},
  { 46, K_CONCAT,    0, 0, 0, 0, {0, 0, 0, 0},  7,  7, 48,     0}, { 48, K_CONV,   1, 1, 1, 1, {0, 0, 0, 0},  7,  7, 40,  1920},
  { 50, K_CONV,     40, 3, 1, 1, {1, 1, 1, 1},  7,  7, 40,   360}, { 52, K_CONV,   1, 1, 1, 1, {0, 0, 0, 0},  7,  7, 32,  1280},
  { 53, K_CONV,      1, 1, 1, 0, {0, 0, 0, 0},  7,  7, 18,   576},
};

/* tflite tensor ids (gen/yf_model_gen.h: yf_tensor_scale_bits / yf_tensor_zero_point, equal to network.c:663-886,1225-1341) of
 * every node's inputs (as its chain lists them), of the pre-activation tensor of a conv with a fused LeakyReLU (ST keeps it
 * as `scratch1`, network.c:1225-1341) and of its output; conv = index into yf_convs (per-channel filter scales) or -1. */
typedef struct { int16_t in0, in1, pre, out, conv; } yf_expected_quant;
static const yf_expected_quant k_quant[31] = {
  { 0, -1, 51, 52,  0}, {52, -1, 53, 54,  1}, {54, -1, -1, 55,  2}, {55, -1, 56, 57,  3}, {57, -1, 60, 61,  4}, {61, -1, -1, 62,  5},
  {62, -1, 63, 64,  6}, {64, -1, 65, 66,  7}, {66, -1, -1, 67,  8}, {62, 67, -1, 68, -1}, {68, -1, 69, 70,  9}, {57, -1, -1, 58, -1},
  {58, 70, -1, 71, -1}, {71, -1, 72, 73, 10}, {73, -1, 76, 77, 11}, {77, -1, -1, 78, 12}, {78, -1, 79, 80, 13}, {80, -1, 81, 82, 14},
  {82, -1, -1, 83, 15}, {78, 83, -1, 84, -1}, {84, -1, 85, 86, 16}, {86, -1, 87, 88, 17}, {88, -1, -1, 89, 18}, {84, 89, -1, 90, -1},
  {90, -1, 91, 92, 19}, {73, -1, -1, 74, -1}, {74, 92, -1, 93, -1}, {93, -1, 94, 95, 20}, {95, -1, 96, 97, 21}, {97, -1, 98, 99, 22},
  {99, -1, -1, 100, 23},
};

/* 0 = tensor t carries exactly the per-tensor quantisation of tflite tensor `id`; otherwise a description in `what` */
static int tensor_quant_differs(const stv_tensor* t, int id, char* what, size_t n) {
  const stv_intq_info_list* q = t ? (const stv_intq_info_list*)t->klass : NULL;
  if (!q || q->size != 1 || !q->info || !q->info[0].scale || !q->info[0].zeropoint) { snprintf(what, n, "has no per-tensor quantisation record"); return 1; }
  uint32_t bits; memcpy(&bits, q->info[0].scale, 4);
  const int zp = *(const int8_t*)q->info[0].zeropoint;
  if (bits != yf_tensor_scale_bits[id]) {
    float want; memcpy(&want, &yf_tensor_scale_bits[id], 4);
    snprintf(what, n, "scale %.9g, expected %.9g", (double)*q->info[0].scale, (double)want); return 1;
  }
  if (zp != yf_tensor_zero_point[id]) { snprintf(what, n, "zero point %d, expected %d", zp, (int)yf_tensor_zero_point[id]); return 1; }
  return 0;
}

YF_API void forward_conv2d_integer_SSSA_ch(void* layer);
YF_API void forward_mp_integer_INT8(void* layer);
YF_API void forward_eltwise_integer_INT8(void* layer);
YF_API void forward_concat(void* layer);
YF_API void nl_func_array_integer(void);

static int shape4(const stv_storage* s, uint32_t out[4]) {
  if (!s->data || STV_STORAGE_SIZE(*s) != 4) return 0;
  memcpy(out, s->data, 16);
  return 1;
}

/* 0 = the caller's graph is the yoloface graph; otherwise `why` says where it differs */
static int verify_graph(const stv_network* net, char* why, size_t n) {
  static const char* const kind_name[] = {"conv2d", "pool", "eltwise add", "concat"};
  const stv_node* node = net->input_node;
  for (int i = 0; i < 31; ++i) {
    const yf_expected_node* e = &k_graph[i];
#define BAD(...)                                                                                                   \
  do {                                                                                                             \
    const int w_ = snprintf(why, n, "the caller's graph is not the yoloface graph: node %d (id %u, expected %s id %u): ", i, \
                            node ? (unsigned)node->id : 0u, kind_name[e->kind], (unsigned)e->id);               \
    if (w_ > 0 && (size_t)w_ < n) { snprintf(why + w_, n - (size_t)w_, __VA_ARGS__); }                          \
    return 1;                                                                                                      \
  } while (0)
    if (!node) BAD("the list ends after %d nodes", i);
    static const uint16_t type_of[] = {STV_LAYER_CONV2D, STV_LAYER_POOL, STV_LAYER_ELTWISE_INTEGER, STV_LAYER_CONCAT};
    void (*const fwd_of[])(void*) = {forward_conv2d_integer_SSSA_ch, forward_mp_integer_INT8, forward_eltwise_integer_INT8, forward_concat};
    if (node->type != type_of[e->kind] || node->forward != fwd_of[e->kind]) BAD("layer type 0x%x / kernel differ", (unsigned)node->type);
    if (node->id != e->id) BAD("layer id %u", (unsigned)node->id);
    if (!node->tensors || node->tensors->size < 2 || !node->tensors->chain) BAD("no tensor chain");
    const stv_tensor_list* outs = &node->tensors->chain[1];
    uint32_t sh[4];
    if (outs->size < 1 || !outs->tensor || !outs->tensor[0] || !shape4(&outs->tensor[0]->shape, sh)) BAD("no output tensor");
    if (sh[0] != 1 || sh[1] != e->oc || sh[2] != e->ow || sh[3] != e->oh)
      BAD("output shape %ux%ux%u, expected %ux%ux%u", (unsigned)sh[3], (unsigned)sh[2], (unsigned)sh[1], (unsigned)e->oh, (unsigned)e->ow, (unsigned)e->oc);
    if (e->kind == K_CONV) {
      const stv_conv2d* c = (const stv_conv2d*)node;
      uint32_t pad[4];
      if (c->groups != e->groups) BAD("groups %u, expected %u", (unsigned)c->groups, (unsigned)e->groups);
      if (c->filter_stride.data[0] != e->stride || c->filter_stride.data[1] != e->stride)
        BAD("stride %ux%u, expected %ux%u", (unsigned)c->filter_stride.data[0], (unsigned)c->filter_stride.data[1], (unsigned)e->stride, (unsigned)e->stride);
      if (c->dilation.data[0] != 1 || c->dilation.data[1] != 1) BAD("dilation %ux%u", (unsigned)c->dilation.data[0], (unsigned)c->dilation.data[1]);
      if (!shape4(&c->filter_pad, pad) || pad[0] != e->pad[0] || pad[1] != e->pad[1] || pad[2] != e->pad[2] || pad[3] != e->pad[3])
        BAD("padding {%u,%u,%u,%u}, expected {%u,%u,%u,%u}", (unsigned)pad[0], (unsigned)pad[1], (unsigned)pad[2], (unsigned)pad[3],
            (unsigned)e->pad[0], (unsigned)e->pad[1], (unsigned)e->pad[2], (unsigned)e->pad[3]);
      if ((c->nl_func != NULL) != (e->nl != 0) || (c->nl_func && c->nl_func != nl_func_array_integer)) BAD("fused non-linearity %s", c->nl_func ? "present" : "absent");
      if (e->nl && (!c->nl_params || c->nl_params->size != 256)) BAD("non-linearity table is not a 256-entry LUT");
      if (node->tensors->size < 3) BAD("no weight tensors");
      const stv_tensor_list* ws = &node->tensors->chain[2];
      if (ws->size < 2 || !ws->tensor || !ws->tensor[0] || !ws->tensor[0]->data || ws->tensor[0]->data->size != e->weights)
        BAD("weight tensor has %u elements, expected %u", ws->size >= 1 && ws->tensor && ws->tensor[0] && ws->tensor[0]->data ? (unsigned)ws->tensor[0]->data->size : 0u, (unsigned)e->weights);
      if (!ws->tensor[1] || !ws->tensor[1]->data || ws->tensor[1]->data->size != e->oc) BAD("bias tensor size");
    } else if (e->kind == K_POOL) {
      const stv_pool* p = (const stv_pool*)node;
      uint32_t pad[4];
      if (p->pool_size.data[0] != e->k || p->pool_size.data[1] != e->k) BAD("window %ux%u, expected %ux%u", (unsigned)p->pool_size.data[0], (unsigned)p->pool_size.data[1], (unsigned)e->k, (unsigned)e->k);
      if (p->pool_stride.data[0] != e->stride || p->pool_stride.data[1] != e->stride) BAD("stride %ux%u", (unsigned)p->pool_stride.data[0], (unsigned)p->pool_stride.data[1]);
      if (!shape4(&p->pool_pad, pad) || pad[0] != e->pad[0] || pad[1] != e->pad[1] || pad[2] != e->pad[2] || pad[3] != e->pad[3])
        BAD("padding {%u,%u,%u,%u}", (unsigned)pad[0], (unsigned)pad[1], (unsigned)pad[2], (unsigned)pad[3]);
    } else {
      const stv_tensor_list* ins = &node->tensors->chain[0];
      if (ins->size != 2) BAD("%u inputs, expected 2", (unsigned)ins->size);
    }
    {   /* quantisation of every tensor the node touches (network.c:663-1341) against the tables the engine was built from */
      const yf_expected_quant* qe = &k_quant[i];
      const stv_tensor_list* ins = &node->tensors->chain[0];
      char what[160];
      const int n_in = qe->in1 >= 0 ? 2 : 1;
      if (ins->size < n_in || !ins->tensor) BAD("%u inputs, expected %d", (unsigned)ins->size, n_in);
      for (int k = 0; k < n_in; ++k)
        if (tensor_quant_differs(ins->tensor[k], k ? qe->in1 : qe->in0, what, sizeof what)) BAD("input tensor %d %s", k, what);
      if (tensor_quant_differs(outs->tensor[0], qe->out, what, sizeof what)) BAD("output tensor %s", what);
      if (qe->pre >= 0) {
        const stv_tensor_list* scr = node->tensors->size >= 4 ? &node->tensors->chain[3] : NULL;
        if (!scr || scr->size < 2 || !scr->tensor) BAD("no pre-activation (scratch1) tensor");
        if (tensor_quant_differs(scr->tensor[1], qe->pre, what, sizeof what)) BAD("pre-activation tensor %s", what);
      }
      if (qe->conv >= 0) {
        const yf_conv_desc* cd = &yf_convs[qe->conv];
        const stv_tensor* wt = node->tensors->chain[2].tensor[0];
        const stv_intq_info_list* q = (const stv_intq_info_list*)wt->klass;
        if (!q || q->size != cd->cout || !q->info || !q->info[0].scale || !q->info[0].zeropoint) BAD("weight tensor has no per-channel quantisation record of %u channels", (unsigned)cd->cout);
        for (int ch = 0; ch < cd->cout; ++ch) {
          uint32_t bits; memcpy(&bits, &q->info[0].scale[ch], 4);
          if (bits != cd->wscale_bits[ch]) {
            float want; memcpy(&want, &cd->wscale_bits[ch], 4);
            BAD("weight tensor channel %d scale %.9g, expected %.9g", ch, (double)q->info[0].scale[ch], (double)want);
          }
          if (((const int8_t*)q->info[0].zeropoint)[ch] != 0) BAD("weight tensor channel %d zero point %d, expected 0", ch, (int)((const int8_t*)q->info[0].zeropoint)[ch]);
        }
      }
    }
    const stv_node* next = (node->next == node) ? NULL : node->next;     /* network.c ends the list with a self link (:2209) */
    if (i == 30 && next) BAD("more than 31 nodes");
    node = next;
#undef BAD
  }
  return 0;
}

/* returns the caller's context on success (network.c:3388-3389 treats NULL as failure) */
YF_API void* ai_platform_network_init(ai_handle network, const ai_network_params* params) {
  if (!own(network)) return NULL;
  char why[384];
  if (verify_graph((const stv_network*)network, why, sizeof why) != 0) {
    yf_impl_fail_init(g_own, AI_ERROR_CODE_NETWORK, why);
    return NULL;
  }
  return yf_impl_init(g_own, params) ? network : NULL;
}

YF_API ai_bool ai_platform_network_post_init(ai_handle network) { return own(network) != AI_HANDLE_NULL; }

/* ---- per-node observer (reference ai_platform_interface.h:684-731, 981-1024) ----------------------------------------------------
 * ST's runtime calls a registered client before and/or after every c-node with the node's tensor chain; the node's output tensor then
 * holds that node's result in the caller's activation arrays.  The fused engine has no per-node execution, so an OBSERVED run goes
 * through the debug build of the kernel instead: it dumps the output tensor of every one of the 31 nodes per frame (the six the fused
 * stages never materialise included -- raw max-pools, the convolutions in front of the residual adds, LEAKY_RELU #43 alone), and this
 * layer then walks the caller's node list in execution order, frame by frame: PRE call-back, the node's tensor copied from the dump
 * into the address the caller's graph gives it, POST call-back.  The input tensor's and the last node's data pointers are bound to
 * the caller's I/O buffers for the duration of the run, as ST's runtime does (network.c:2954-2955, 3016-3017 leave them NULL).
 * Not reproduced: AI_OBSERVER_INIT_EVT, the `inner_tensors` of a node (NULL) and the content of the scratch tensors (the
 * pre-activation values of convolutions with a fused LeakyReLU exist only as LUT indices inside the kernel).  Values follow TFLite's
 * arithmetic, like everything this library computes (ST's LUT rounding differs by 1 LSB in places). */
static struct { ai_observer_node_cb cb; ai_handle cookie; ai_u32 flags; ai_observer_exec_ctx* ctx; } g_obs;
static void observer_reset(void) { memset(&g_obs, 0, sizeof g_obs); }

/* node i of k_graph: its output tensor = C channels of P pixels; source = the dump tensor of tflite op `op` (PIX channels per pixel
 * there, this node's channels start at CH0), or the heads for the last node (op 0) */
typedef struct { int16_t op, c, p, pix, ch0; } yf_node_src;
static const yf_node_src k_node_src[31] = {
  { 2,  8, 784,  8, 0}, { 4,  8, 784,  8, 0}, { 5,  4, 784,  4, 0}, { 7, 18, 784, 18, 0}, {11, 18, 196, 18, 0}, {12,  6, 196,  6, 0},
  {14, 36, 196, 36, 0}, {16, 36, 196, 36, 0}, {17,  6, 196,  6, 0}, {18,  6, 196,  6, 0}, {22, 18, 196, 36, 18}, { 8, 18, 196, 18, 0},
  {22, 36, 196, 36, 0}, {24, 24, 196, 24, 0}, {28, 24,  49, 24, 0}, {29,  8,  49,  8, 0}, {31, 40,  49, 40, 0}, {33, 40,  49, 40, 0},
  {34,  8,  49,  8, 0}, {35,  8,  49,  8, 0}, {37, 40,  49, 40, 0}, {39, 40,  49, 40, 0}, {40,  8,  49,  8, 0}, {41,  8,  49,  8, 0},
  {43, 24,  49, 24, 0}, {25, 24,  49, 24, 0}, {46, 48,  49, 48, 0}, {48, 40,  49, 40, 0}, {50, 40,  49, 40, 0}, {52, 32,  49, 32, 0},
  { 0, 18,  49, 18, 0},
};

static const stv_node* node_at(const stv_network* net, int idx) {
  const stv_node* node = net->input_node;
  for (int i = 0; node && i < idx; ++i) node = (node->next == node) ? NULL : node->next;
  return node;
}

YF_API ai_bool ai_platform_observer_node_info(ai_handle network, ai_observer_node* node_info) {
  if (!own(network) || !node_info) return false;
  const stv_node* node = node_info->c_idx < 31 ? node_at((const stv_network*)network, node_info->c_idx) : NULL;
  if (!node) { yf_impl_fail_run(g_own, AI_ERROR_CODE_OUT_OF_RANGE, "observer: no such c-node index"); return false; }
  node_info->type = node->type; node_info->id = node->id; node_info->unused = 0;
  node_info->inner_tensors = NULL; node_info->tensors = node->tensors;
  return true;
}

YF_API ai_bool ai_platform_observer_register(ai_handle network, ai_observer_node_cb cb, ai_handle cookie, ai_u32 flags) {
  if (!own(network) || !cb) return false;
  g_obs.cb = cb; g_obs.cookie = cookie; g_obs.flags = (flags & AI_OBSERVER_MASK_EVT) | AI_OBSERVER_REGISTERED; g_obs.ctx = NULL;
  return true;
}
YF_API ai_bool ai_platform_observer_register_s(ai_handle network, ai_observer_exec_ctx* ctx) {
  if (!ctx || !ai_platform_observer_register(network, ctx->on_node, ctx->cookie, ctx->flags)) return false;
  g_obs.ctx = ctx; ctx->flags |= AI_OBSERVER_REGISTERED; ctx->c_idx = 0; ctx->n_nodes = 31; ctx->cur = NULL;
  return true;
}
YF_API ai_bool ai_platform_observer_unregister(ai_handle network, ai_observer_node_cb cb, ai_handle cookie) {
  if (!own(network) || !(g_obs.flags & AI_OBSERVER_REGISTERED) || g_obs.cb != cb || g_obs.cookie != cookie) return false;
  if (g_obs.ctx) g_obs.ctx->flags &= ~(ai_u32)AI_OBSERVER_REGISTERED;
  memset(&g_obs, 0, sizeof g_obs);
  return true;
}
YF_API ai_bool ai_platform_observer_unregister_s(ai_handle network, ai_observer_exec_ctx* ctx) {
  return ctx ? ai_platform_observer_unregister(network, ctx->on_node, ctx->cookie) : false;
}

static ai_i32 process_observed(ai_handle network, const ai_buffer* input, ai_buffer* output) {
  if (!input || !input->data || input->n_batches < 1 || (output && !output->data))      /* the plain path latches the matching error */
    return output ? yf_impl_run(g_own, input, output) : yf_impl_forward(g_own, input);
  const stv_network* net = (const stv_network*)network;
  const long ds = yf_impl_dump_bytes();
  enum { CHUNK = 32 };
  long off_of[31];
  for (int i = 0; i < 31; ++i) {
    off_of[i] = k_node_src[i].op ? yf_impl_dump_offset(k_node_src[i].op) : 0;
    if (off_of[i] < 0) { yf_impl_fail_run(g_own, AI_ERROR_CODE_NETWORK, "observer: the debug build does not dump every node"); return 0; }
  }
  /* the caller's I/O arrays: the first node's input tensor and the last node's output tensor */
  const stv_node* first = net->input_node;
  const stv_node* last = node_at(net, 30);
  if (!first || !last || !first->tensors || !last->tensors) { yf_impl_fail_run(g_own, AI_ERROR_CODE_NETWORK, "observer: no node list"); return 0; }
  stv_array* in_arr = first->tensors->chain[0].tensor[0]->data;
  stv_array* out_arr = last->tensors->chain[1].tensor[0]->data;
  const stv_array in_keep = *in_arr, out_keep = *out_arr;
  int8_t* heads = (int8_t*)malloc((size_t)CHUNK * AI_NETWORK_OUT_1_SIZE);
  int8_t* dump = (int8_t*)malloc((size_t)CHUNK * (size_t)ds);
  ai_i32 done = 0;
  const long n = input->n_batches;
  if (!heads || !dump) { yf_impl_fail_run(g_own, AI_ERROR_CODE_NETWORK, "observer: out of memory"); free(heads); free(dump); return 0; }
  for (long base = 0; base < n; base += CHUNK) {
    const long cnt = n - base < CHUNK ? n - base : CHUNK;
    if (yf_impl_run_dump(g_own, input, output, base, cnt, heads, dump) == 0) goto out;      /* validates shapes / formats and latches */
    for (long f = 0; f < cnt; ++f) {
      in_arr->data = in_arr->data_start = (uint8_t*)input->data + (base + f) * AI_NETWORK_IN_1_SIZE;
      out_arr->data = out_arr->data_start = output ? (uint8_t*)output->data + (base + f) * AI_NETWORK_OUT_1_SIZE : (uint8_t*)heads + f * AI_NETWORK_OUT_1_SIZE;
      const stv_node* node = first;
      for (int i = 0; i < 31 && node; ++i) {
        ai_observer_node on;
        on.c_idx = (ai_u16)i; on.type = node->type; on.id = node->id; on.unused = 0; on.inner_tensors = NULL; on.tensors = node->tensors;
        const ai_u32 pos = (i == 0 ? AI_OBSERVER_FIRST_EVT : 0) | (i == 30 ? AI_OBSERVER_LAST_EVT : 0);
        if (g_obs.ctx) { g_obs.ctx->c_idx = (ai_u16)i; g_obs.ctx->n_nodes = 31; g_obs.ctx->cur = (struct ai_node_s*)(uintptr_t)node; }
        if (g_obs.flags & AI_OBSERVER_PRE_EVT) g_obs.cb(g_obs.cookie, AI_OBSERVER_PRE_EVT | pos, &on);
        const yf_node_src* sv = &k_node_src[i];
        const stv_array* dst_arr = node->tensors->chain[1].tensor[0]->data;
        uint8_t* dst = dst_arr ? dst_arr->data : NULL;
        if (!dst) { yf_impl_fail_run(g_own, AI_ERROR_CODE_INVALID_PTR, "observer: a node's output tensor has no address (no activation arrays bound)"); goto out; }
        const int8_t* src = sv->op ? dump + f * ds + off_of[i] : heads + f * AI_NETWORK_OUT_1_SIZE;
        if (sv->pix == sv->c) memcpy(dst, src, (size_t)sv->c * sv->p);
        else for (int px = 0; px < sv->p; ++px) memcpy(dst + (size_t)px * sv->c, src + (size_t)px * sv->pix + sv->ch0, (size_t)sv->c);
        if (g_obs.flags & AI_OBSERVER_POST_EVT) g_obs.cb(g_obs.cookie, AI_OBSERVER_POST_EVT | pos, &on);
        node = (node->next == node) ? NULL : node->next;
      }
    }
  }
  done = (ai_i32)n;
out:
  *in_arr = in_keep; *out_arr = out_keep;
  free(heads); free(dump);
  return done;
}

YF_API ai_i32 ai_platform_network_process(ai_handle network, const ai_buffer* input, ai_buffer* output) {
  if (!own(network)) return 0;
  if ((g_obs.flags & AI_OBSERVER_REGISTERED) && (g_obs.flags & (AI_OBSERVER_PRE_EVT | AI_OBSERVER_POST_EVT)) && g_obs.cb)
    return process_observed(network, input, output);
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
  return yf_impl_fill_report(g_own, r);          /* same function as behind this library's own ai_network_get_report / get_info */
}

YF_API const char* ai_platform_runtime_get_revision(void) { return yf_impl_runtime_revision(); }
YF_API ai_platform_version ai_platform_runtime_get_version(void) { return yf_impl_runtime_version(); }
YF_API ai_platform_version ai_platform_api_get_version(void) { return yf_impl_api_version(); }
YF_API ai_platform_version ai_platform_interface_api_get_version(void) { return yf_impl_interface_api_version(); }

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
