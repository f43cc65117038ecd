/* Per-node observer through the REFERENCE's own headers and its unmodified generated network.c (runtime-level drop-in):
 * registers an observer (ai_platform_interface.h:981-1024), runs n frames, and writes every node's output tensor as the POST call-back
 * finds it in the caller's own tensor objects (node->tensors, chain[1] = outputs: ai_platform_interface.h:549-556, 651-671).
 *   abi_observer_probe <frames.bin> <n> <out.bin>
 * out.bin: per frame, per node in execution order: the node's output tensor bytes (AI_ARRAY data, size elements).  stdout: one line per node
 * of frame 0 ("node <c_idx> id <id> type 0x<type> bytes <n> flags 0x<flags>"), then "PRE <count> POST <count>".
 * Built only where /root/reference exists (oracle/Makefile.ref); the binary travels to the GPU box. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "network.h"
#include "network_data.h"
#include "ai_platform_interface.h"

static ai_handle network = AI_HANDLE_NULL;
AI_ALIGNED(32) static ai_u8 activations[AI_NETWORK_DATA_ACTIVATIONS_SIZE];
static FILE* g_out;
static long g_pre, g_post, g_frame_nodes;

static ai_u32 on_node(const ai_handle cookie, const ai_u32 flags, const ai_observer_node* node) {
  (void)cookie;
  if (flags & AI_OBSERVER_PRE_EVT) { ++g_pre; return 0; }
  ++g_post;
  const ai_tensor_chain* chain = node->tensors;
  const ai_tensor* t = chain->chain[1].tensor[0];                  /* AI_TENSOR_CHAIN_OUTPUT */
  const ai_array* a = t->data;
  fwrite(a->data, 1, a->size, g_out);
  if (g_frame_nodes < 31) {
    printf("node %u id %u type 0x%x bytes %u flags 0x%x\n", (unsigned)node->c_idx, (unsigned)node->id, (unsigned)node->type, (unsigned)a->size, (unsigned)flags);
    ++g_frame_nodes;
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s frames.bin n out.bin\n", argv[0]); return 2; }
  const int n = atoi(argv[2]);
  ai_i8* in_data = (ai_i8*)aligned_alloc(32, ((size_t)n * AI_NETWORK_IN_1_SIZE + 31) & ~(size_t)31);
  ai_i8* out_data = (ai_i8*)aligned_alloc(32, ((size_t)n * AI_NETWORK_OUT_1_SIZE + 31) & ~(size_t)31);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(in_data, AI_NETWORK_IN_1_SIZE, n, f) != (size_t)n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  fclose(f);
  g_out = fopen(argv[3], "wb");
  if (!g_out) return 2;
  ai_error err = ai_network_create(&network, AI_NETWORK_DATA_CONFIG);
  if (err.type != AI_ERROR_NONE) { printf("E: create type=%d code=%d\n", err.type, err.code); return 3; }
  const ai_network_params params = AI_NETWORK_PARAMS_INIT(AI_NETWORK_DATA_WEIGHTS(ai_network_data_weights_get()), AI_NETWORK_DATA_ACTIVATIONS(activations));
  if (!ai_network_init(network, &params)) { err = ai_network_get_error(network); printf("E: init type=%d code=%d\n", err.type, err.code); return 4; }

  ai_observer_node info; memset(&info, 0, sizeof info); info.c_idx = 12;
  if (!ai_platform_observer_node_info(network, &info)) return 6;
  printf("info c_idx 12: id %u type 0x%x\n", (unsigned)info.id, (unsigned)info.type);
  info.c_idx = 31;
  if (ai_platform_observer_node_info(network, &info)) return 6;              /* out of range must fail (error latched: clear it) */
  (void)ai_network_get_error(network);

  if (!ai_platform_observer_register(network, on_node, (ai_handle)&g_post, AI_OBSERVER_PRE_EVT | AI_OBSERVER_POST_EVT)) return 6;
  ai_buffer ai_input[AI_NETWORK_IN_NUM] = AI_NETWORK_IN;
  ai_buffer ai_output[AI_NETWORK_OUT_NUM] = AI_NETWORK_OUT;
  ai_input[0].n_batches = (ai_u16)n; ai_input[0].data = AI_HANDLE_PTR(in_data);
  ai_output[0].n_batches = (ai_u16)n; ai_output[0].data = AI_HANDLE_PTR(out_data);
  if (ai_network_run(network, &ai_input[0], &ai_output[0]) != n) { err = ai_network_get_error(network); printf("E: run type=%d code=%d\n", err.type, err.code); return 5; }
  fclose(g_out);
  printf("PRE %ld POST %ld\n", g_pre, g_post);
  /* unregistered: the plain (production) path must give the same heads */
  if (!ai_platform_observer_unregister(network, on_node, (ai_handle)&g_post)) return 6;
  ai_i8* out2 = (ai_i8*)aligned_alloc(32, ((size_t)n * AI_NETWORK_OUT_1_SIZE + 31) & ~(size_t)31);
  ai_output[0].data = AI_HANDLE_PTR(out2);
  if (ai_network_run(network, &ai_input[0], &ai_output[0]) != n) return 5;
  printf("heads %s\n", memcmp(out_data, out2, (size_t)n * AI_NETWORK_OUT_1_SIZE) == 0 ? "equal" : "DIFFER");
  if (ai_network_destroy(network) != AI_HANDLE_NULL) return 5;
  printf("OK %d\n", n);
  return 0;
}
