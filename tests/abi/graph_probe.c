/* Prints the layout facts stm32h7-yolo_amd/csrc/st_graph_view.h relies on.  Compiled twice by tests/test_abi.py: against
 * the reference's ST headers and (-DYF_OWN_VIEWS) against the library's own views; the outputs must be identical. */
#include <stdio.h>
#include <stddef.h>
#ifdef YF_OWN_VIEWS
#include "st_graph_view.h"
typedef stv_network ai_network; typedef stv_node ai_node; typedef stv_conv2d ai_layer_conv2d; typedef stv_pool ai_layer_pool;
typedef stv_storage ai_storage_klass; typedef stv_shape2d ai_shape_2d; typedef stv_tensor ai_tensor; typedef stv_tensor_list ai_tensor_list;
typedef stv_tensor_chain ai_tensor_chain; typedef stv_array ai_array; typedef stv_intq_info ai_intq_info; typedef stv_intq_info_list ai_intq_info_list;
#define NODE(f) offsetof(stv_conv2d, n.f)
#define AI_LAYER_CONV2D_TYPE STV_LAYER_CONV2D
#define AI_LAYER_POOL_TYPE STV_LAYER_POOL
#define AI_LAYER_CONCAT_TYPE STV_LAYER_CONCAT
#define AI_LAYER_ELTWISE_INTEGER_TYPE STV_LAYER_ELTWISE_INTEGER
#else
#include "ai_platform_interface.h"
#include "core_common.h"
#include "layers_conv2d.h"
#include "layers_pool.h"
#include "layers_generic.h"
#define NODE(f) offsetof(ai_layer_conv2d, f)
#endif
#define P(T, F) printf(#T "." #F " %zu\n", offsetof(T, F))
int main(void) {
  printf("ai_network %zu\n", sizeof(ai_network)); P(ai_network, magic); P(ai_network, signature); P(ai_network, tensors); P(ai_network, input_node);
  printf("node: type %zu id %zu flags %zu next %zu forward %zu tensors %zu\n", NODE(type), NODE(id), NODE(flags), NODE(next), NODE(forward), NODE(tensors));
  printf("ai_layer_conv2d %zu\n", sizeof(ai_layer_conv2d)); P(ai_layer_conv2d, groups); P(ai_layer_conv2d, nl_params); P(ai_layer_conv2d, nl_func);
  P(ai_layer_conv2d, filter_stride); P(ai_layer_conv2d, dilation); P(ai_layer_conv2d, filter_pad);
  printf("ai_layer_pool %zu\n", sizeof(ai_layer_pool)); P(ai_layer_pool, pool_size); P(ai_layer_pool, pool_stride); P(ai_layer_pool, pool_pad);
  printf("ai_storage_klass %zu\n", sizeof(ai_storage_klass)); P(ai_storage_klass, data);
  printf("ai_shape_2d %zu\n", sizeof(ai_shape_2d));
  printf("ai_tensor %zu\n", sizeof(ai_tensor)); P(ai_tensor, info); P(ai_tensor, shape); P(ai_tensor, stride); P(ai_tensor, data);
  printf("ai_tensor_list %zu\n", sizeof(ai_tensor_list)); P(ai_tensor_list, size); P(ai_tensor_list, tensor);
  printf("ai_tensor_chain %zu\n", sizeof(ai_tensor_chain)); P(ai_tensor_chain, size); P(ai_tensor_chain, chain);
  printf("ai_array %zu\n", sizeof(ai_array)); P(ai_array, format); P(ai_array, size); P(ai_array, data);
  printf("ai_intq_info %zu\n", sizeof(ai_intq_info)); P(ai_intq_info, scale); P(ai_intq_info, zeropoint);
  printf("ai_intq_info_list %zu\n", sizeof(ai_intq_info_list)); P(ai_intq_info_list, flags); P(ai_intq_info_list, size); P(ai_intq_info_list, info); P(ai_tensor, klass);
  printf("types %d %d %d %d\n", (int)AI_LAYER_CONV2D_TYPE, (int)AI_LAYER_POOL_TYPE, (int)AI_LAYER_CONCAT_TYPE, (int)AI_LAYER_ELTWISE_INTEGER_TYPE);
  return 0;
}
