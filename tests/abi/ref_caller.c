/* Boundary check: the reference application's call sequence (stm32/X-CUBE-AI/App/yoloface.c:188-240, aiInit/aiRun)
 * written against the REFERENCE's own headers (network.h, network_data.h, ai_platform.h), compiled together with
 * the reference's unmodified network_data.c and linked to libyf_network.so in place of network.c + the ST runtime.
 * Built only where /root/reference exists (oracle/_ref/Makefile); the binary travels to the GPU box.
 *
 *   abi_ref_caller <frames.bin> <heads.bin> <n> [map]
 * "map": initialise through ai_network_data_params_get (network_data.c:412-432 -> ai_platform_bind_network_params)
 * instead of the legacy AI_NETWORK_PARAMS_INIT pair.  Exit: 0 ok, 2 usage/io, 3 create, 4 init, 5 run. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "network.h"
#include "network_data.h"

static ai_handle network = AI_HANDLE_NULL;
AI_ALIGNED(32) static ai_u8 activations[AI_NETWORK_DATA_ACTIVATIONS_SIZE];

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s frames.bin heads.bin n [map]\n", argv[0]); return 2; }
  const int n = atoi(argv[3]);
  const int use_map = argc > 4 && !strcmp(argv[4], "map");
  ai_i8* in_data = (ai_i8*)aligned_alloc(32, ((size_t)n * AI_NETWORK_IN_1_SIZE + 31) & ~(size_t)31);
  ai_i8* out_data = (ai_i8*)aligned_alloc(32, ((size_t)n * AI_NETWORK_OUT_1_SIZE + 31) & ~(size_t)31);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(in_data, AI_NETWORK_IN_1_SIZE, n, f) != (size_t)n) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  fclose(f);

  ai_error err = ai_network_create(&network, AI_NETWORK_DATA_CONFIG);
  if (err.type != AI_ERROR_NONE) { printf("E: AI ai_network_create error - type=%d code=%d\n", err.type, err.code); return 3; }

  ai_bool ok;
  if (use_map) {
    ai_network_params params;
    ok = ai_network_data_params_get(network, &params);
    params.map_activations.buffer[0].data = AI_HANDLE_PTR(activations);
    ok = ok && ai_network_init(network, &params);
  } else {
    const ai_network_params params = AI_NETWORK_PARAMS_INIT(
        AI_NETWORK_DATA_WEIGHTS(ai_network_data_weights_get()),
        AI_NETWORK_DATA_ACTIVATIONS(activations));
    ok = ai_network_init(network, &params);
  }
  if (!ok) {
    err = ai_network_get_error(network);
    printf("E: AI ai_network_init error - type=%d code=%d\n", err.type, err.code);
    return 4;
  }

  ai_buffer ai_input[AI_NETWORK_IN_NUM] = AI_NETWORK_IN;
  ai_buffer ai_output[AI_NETWORK_OUT_NUM] = AI_NETWORK_OUT;
  ai_input[0].n_batches = (ai_u16)n;
  ai_input[0].data = AI_HANDLE_PTR(in_data);
  ai_output[0].n_batches = (ai_u16)n;
  ai_output[0].data = AI_HANDLE_PTR(out_data);
  const ai_i32 n_batch = ai_network_run(network, &ai_input[0], &ai_output[0]);
  if (n_batch != n) {
    err = ai_network_get_error(network);
    printf("E: AI ai_network_run error - type=%d code=%d\n", err.type, err.code);
    return 5;
  }
  f = fopen(argv[2], "wb");
  if (!f || fwrite(out_data, AI_NETWORK_OUT_1_SIZE, n, f) != (size_t)n) return 2;
  fclose(f);
  ai_network_report rep;
  if (ai_network_get_report(network, &rep)) printf("model %s macc %u nodes %u\n", rep.model_name, (unsigned)rep.n_macc, (unsigned)rep.n_nodes);
  if (ai_network_destroy(network) != AI_HANDLE_NULL) return 5;
  printf("OK %d\n", (int)n_batch);
  return 0;
}
