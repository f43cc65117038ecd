/* Read-only views of the graph objects in the reference's GENERATED network.c (runtime-level drop-in).
 *
 * network.c owns a static `ai_network` object whose `input_node` heads a linked list of layer objects
 * (AI_NETWORK_OBJ_DECLARE, network.c:2929-2939; AI_LAYER_OBJ_DECLARE x31, network.c:2204-2927).  ST's closed runtime
 * walks that list (core_common.h:101-109); this library only READS it, at ai_platform_network_init time, to make sure the
 * graph it is asked to run is the one the fused engine implements.  The declarations below restate, field by field, the
 * layout those objects have when network.c is compiled for this host against ST's headers:
 *   ai_network        ai_platform_interface.h:756-777        ai_node (layer base)  core_common.h:104-112
 *   ai_layer_conv2d   layers_conv2d.h:27-34,76-78            ai_layer_pool         layers_pool.h:42-48
 *   ai_tensor_chain / ai_tensor_list / ai_tensor  ai_platform_interface.h:549-556,651-671
 *   ai_storage_klass (ai_shape)  ai_platform_interface.h:90-94,135     ai_shape_2d  :501-503     ai_array  :513-522
 *   ai_intq_info / ai_intq_info_list (a tensor's `klass`: its quantisation)  ai_platform.h:479-495
 * Own text; tests/abi/graph_probe.c prints every offset used here from BOTH sets of declarations (container test:
 * tests/test_abi.py::test_graph_views_match_reference_layout). */
#ifndef YF_ST_GRAPH_VIEW_H
#define YF_ST_GRAPH_VIEW_H
#include <stdint.h>

typedef struct { uint32_t type_size; void* data; } stv_storage;         /* type:8 | size:24, then the element array */
#define STV_STORAGE_SIZE(s) ((s).type_size >> 8)
typedef struct { uint32_t data[2]; } stv_shape2d;                        /* [0] = x / width, [1] = y / height */
typedef struct { int32_t format; uint32_t size; uint8_t* data; uint8_t* data_start; } stv_array;
typedef struct { const float* scale; const void* zeropoint; } stv_intq_info;          /* `size` scales, `size` int8 zero points */
typedef struct { uint16_t flags, size; const stv_intq_info* info; } stv_intq_info_list;   /* size > 1: per output channel */
typedef struct { uint16_t id; uint8_t flags; uint8_t data_size; } stv_tensor_info;
typedef struct { void* klass /* stv_intq_info_list* or NULL */; stv_tensor_info info; stv_storage shape; stv_storage stride; stv_array* data; } stv_tensor;
typedef struct { uint16_t size, flags; stv_tensor** tensor; void* info; } stv_tensor_list;
typedef struct { uint16_t size, flags; stv_tensor_list* chain; } stv_tensor_chain;   /* chain[0] inputs, [1] outputs, [2] weights, [3] scratch */

typedef struct stv_node_ {
  uint16_t type, id;
  uint32_t flags;
  void* klass;
  void* network;
  struct stv_node_* next;              /* NULL-terminated in execution order; the last node of network.c points to itself */
  void (*forward)(void* layer);
  const stv_tensor_chain* tensors;
} stv_node;

typedef struct {
  stv_node n;
  uint32_t groups;
  const stv_array* nl_params;
  void (*nl_func)(void);
  stv_shape2d filter_stride, dilation;
  stv_storage filter_pad;              /* 4 values */
} stv_conv2d;

typedef struct {
  stv_node n;
  stv_shape2d pool_size, pool_stride;
  stv_storage pool_pad;                /* 4 values */
  uint8_t count_include_pad;
} stv_pool;

typedef struct {
  uint32_t magic, signature;
  void* klass;
  uint32_t flags, error;
  uint16_t n_batches, batch_id;
  uint8_t buffers[40];                 /* ai_network_buffers: not read */
  stv_tensor_chain tensors;
  stv_node* input_node;
  stv_node* current_node;
  void* on_node_exec;
  void* data_exec;
  uint32_t tool_api_version;
} stv_network;

enum { STV_LAYER_CONV2D = 0x103, STV_LAYER_POOL = 0x10B, STV_LAYER_CONCAT = 0x110, STV_LAYER_ELTWISE_INTEGER = 0x114 };   /* layers_list.h:53,69,79,87 */
#endif
