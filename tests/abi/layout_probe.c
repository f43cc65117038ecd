/* Prints the ABI facts of the boundary types.  Compiled twice by tests/test_abi.py: once against include/yf_network.h
 * (-DYF_OWN_HEADER) and, where /root/reference exists, once against the reference's ai_platform.h; the outputs
 * must be identical. */
#include <stdio.h>
#include <stddef.h>
#include <string.h>
#ifdef YF_OWN_HEADER
#include "yf_network.h"
#else
#include "ai_platform.h"
#include "ai_platform_interface.h"      /* the per-node observer's datatypes (:684-731) */
#endif
int main(void) {
  printf("S8=0x%08x U8=0x%08x CONST=0x%08x\n", (unsigned)AI_BUFFER_FORMAT_S8, (unsigned)AI_BUFFER_FORMAT_U8, (unsigned)AI_BUFFER_FMT_FLAG_CONST);
  printf("MAGIC_MARKER=0x%08x MAGIC_SIGNATURE=0x%08x\n", (unsigned)AI_MAGIC_MARKER, (unsigned)AI_MAGIC_SIGNATURE);
  printf("ai_buffer %zu: %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ai_buffer), offsetof(ai_buffer, format), offsetof(ai_buffer, n_batches),
         offsetof(ai_buffer, height), offsetof(ai_buffer, width), offsetof(ai_buffer, channels), offsetof(ai_buffer, data), offsetof(ai_buffer, meta_info));
  printf("ai_buffer_array %zu: %zu %zu %zu\n", sizeof(ai_buffer_array), offsetof(ai_buffer_array, flags), offsetof(ai_buffer_array, size), offsetof(ai_buffer_array, buffer));
  printf("ai_network_params %zu: %zu %zu %zu %zu %zu\n", sizeof(ai_network_params), offsetof(ai_network_params, params), offsetof(ai_network_params, activations),
         offsetof(ai_network_params, map_signature), offsetof(ai_network_params, map_weights), offsetof(ai_network_params, map_activations));
  printf("ai_network_report %zu: %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ai_network_report),
         offsetof(ai_network_report, model_name), offsetof(ai_network_report, runtime_revision), offsetof(ai_network_report, runtime_version),
         offsetof(ai_network_report, tool_revision), offsetof(ai_network_report, tool_version), offsetof(ai_network_report, api_version),
         offsetof(ai_network_report, n_macc), offsetof(ai_network_report, n_inputs), offsetof(ai_network_report, inputs),
         offsetof(ai_network_report, outputs), offsetof(ai_network_report, params), offsetof(ai_network_report, n_nodes), offsetof(ai_network_report, signature));
  ai_error e; memset(&e, 0, sizeof e); e.type = 0x12; e.code = 0x21; unsigned raw; memcpy(&raw, &e, 4);
  printf("ai_error %zu raw=0x%08x\n", sizeof(ai_error), raw);
  printf("errors: %d %d %d %d %d %d %d | %d %d %d %d %d %d %d\n", AI_ERROR_NONE, AI_ERROR_INVALID_HANDLE, AI_ERROR_INVALID_STATE, AI_ERROR_INVALID_INPUT,
         AI_ERROR_INVALID_OUTPUT, AI_ERROR_INIT_FAILED, AI_ERROR_CREATE_FAILED, AI_ERROR_CODE_NETWORK, AI_ERROR_CODE_NETWORK_WEIGHTS,
         AI_ERROR_CODE_NETWORK_ACTIVATIONS, AI_ERROR_CODE_INVALID_SIZE, AI_ERROR_CODE_INVALID_FORMAT, AI_ERROR_CODE_INVALID_BATCH, AI_ERROR_CODE_MISSED_INIT);
  printf("ai_observer_node %zu: %zu %zu %zu %zu %zu\n", sizeof(ai_observer_node), offsetof(ai_observer_node, c_idx), offsetof(ai_observer_node, type),
         offsetof(ai_observer_node, id), offsetof(ai_observer_node, inner_tensors), offsetof(ai_observer_node, tensors));
  printf("ai_observer_exec_ctx %zu: %zu %zu %zu %zu %zu %zu\n", sizeof(ai_observer_exec_ctx), offsetof(ai_observer_exec_ctx, on_node), offsetof(ai_observer_exec_ctx, cookie),
         offsetof(ai_observer_exec_ctx, flags), offsetof(ai_observer_exec_ctx, c_idx), offsetof(ai_observer_exec_ctx, n_nodes), offsetof(ai_observer_exec_ctx, cur));
  printf("observer events: %d %d %d %d %d %d %d\n", AI_OBSERVER_INIT_EVT, AI_OBSERVER_PRE_EVT, AI_OBSERVER_POST_EVT, AI_OBSERVER_FIRST_EVT, AI_OBSERVER_LAST_EVT,
         AI_OBSERVER_REGISTERED, AI_OBSERVER_MASK_EVT);
  return 0;
}
