/* Boundary check: the reference application's call sequence (stm32/X-CUBE-AI/App/yoloface.c:188-240, aiInit/aiRun)
 * written against the REFERENCE's own headers (network.h, network_data.h, ai_platform.h), compiled together with
 * the reference's unmodified network_data.c and linked to libyf_network.so in place of network.c + the ST runtime.
 * Built only where /root/reference exists (oracle/_ref/Makefile); the binary travels to the GPU box.
 *
 *   abi_ref_caller <frames.bin> <heads.bin> <n> [map]
 *   abi_ref_caller report            (no GPU needed: create, print ai_network_get_report / ai_network_get_info, destroy)
 * "map": initialise through ai_network_data_params_get (network_data.c:412-432 -> ai_platform_bind_network_params)
 * instead of the legacy AI_NETWORK_PARAMS_INIT pair.  Exit: 0 ok, 2 usage/io, 3 create, 4 init, 5 run. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "network.h"
#include "network_data.h"

static ai_handle network = AI_HANDLE_NULL;
AI_ALIGNED(32) static ai_u8 activations[AI_NETWORK_DATA_ACTIVATIONS_SIZE];

/* Every field of a report, one per line ("report.<field> ..."): the test compares this text between abi_ref_caller (this library's
 * ai_network_get_report / get_info) and abi_ref_runtime_caller (the reference's own network.c:3271-3361 over the runtime-level entry points).
 * Addresses are printed as set / null; compile_datetime is the one field that legitimately differs (it is the compile time of network.c). */
static void print_version(const char* tag, const char* name, ai_platform_version v) { printf("%s.%s %u.%u.%u.%u\n", tag, name, v.major, v.minor, v.micro, v.reserved); }
static void print_buffer(const char* tag, const char* name, const ai_buffer* b) {
  if (!b) { printf("%s.%s null\n", tag, name); return; }
  printf("%s.%s format=0x%08x n_batches=%u height=%u width=%u channels=%u data=%s meta_info=%s\n", tag, name, (unsigned)b->format, (unsigned)b->n_batches,
         (unsigned)b->height, (unsigned)b->width, (unsigned)b->channels, b->data ? "set" : "null", b->meta_info ? "set" : "null");
}
static void print_report(const char* tag, const ai_network_report* r, int maps) {
  printf("%s.model_name %s\n%s.model_signature %s\n%s.model_datetime %s\n", tag, r->model_name, tag, r->model_signature, tag, r->model_datetime);
  printf("%s.compile_datetime %s\n%s.runtime_revision %s\n", tag, r->compile_datetime, tag, r->runtime_revision);
  print_version(tag, "runtime_version", r->runtime_version);
  printf("%s.tool_revision [%s]\n", tag, r->tool_revision);
  print_version(tag, "tool_version", r->tool_version);
  print_version(tag, "tool_api_version", r->tool_api_version);
  print_version(tag, "api_version", r->api_version);
  print_version(tag, "interface_api_version", r->interface_api_version);
  printf("%s.n_macc %u\n%s.n_inputs %u\n%s.n_outputs %u\n", tag, (unsigned)r->n_macc, tag, (unsigned)r->n_inputs, tag, (unsigned)r->n_outputs);
  for (unsigned i = 0; i < r->n_inputs; ++i) print_buffer(tag, "inputs[]", &r->inputs[i]);
  for (unsigned i = 0; i < r->n_outputs; ++i) print_buffer(tag, "outputs[]", &r->outputs[i]);
  if (maps) {
    printf("%s.map_signature 0x%08x\n", tag, (unsigned)r->map_signature);
    printf("%s.map_weights flags=%u size=%u\n", tag, (unsigned)r->map_weights.flags, (unsigned)r->map_weights.size);
    for (unsigned i = 0; i < r->map_weights.size; ++i) print_buffer(tag, "map_weights.buffer[]", &r->map_weights.buffer[i]);
    printf("%s.map_activations flags=%u size=%u\n", tag, (unsigned)r->map_activations.flags, (unsigned)r->map_activations.size);
    for (unsigned i = 0; i < r->map_activations.size; ++i) print_buffer(tag, "map_activations.buffer[]", &r->map_activations.buffer[i]);
  } else {
    print_buffer(tag, "params", &r->params);
    print_buffer(tag, "activations", &r->activations);
  }
  printf("%s.n_nodes %u\n%s.signature 0x%08x\n", tag, (unsigned)r->n_nodes, tag, (unsigned)r->signature);
}
static int print_reports(const char* when) {
  char tag[48];
  ai_network_report rep;
  memset(&rep, 0xEE, sizeof rep);                 /* whatever the call does not write stays visible */
  if (!ai_network_get_report(network, &rep)) { printf("E: ai_network_get_report failed (%s)\n", when); return 0; }
  snprintf(tag, sizeof tag, "report[%s]", when);
  print_report(tag, &rep, 1);
  memset(&rep, 0xEE, sizeof rep);
  if (!ai_network_get_info(network, &rep)) { printf("E: ai_network_get_info failed (%s)\n", when); return 0; }
  snprintf(tag, sizeof tag, "info[%s]", when);
  print_report(tag, &rep, 0);
  return 1;
}

int main(int argc, char** argv) {
  if (argc == 2 && !strcmp(argv[1], "report")) {
    ai_error e = ai_network_create(&network, AI_NETWORK_DATA_CONFIG);
    if (e.type != AI_ERROR_NONE) { printf("E: AI ai_network_create error - type=%d code=%d\n", e.type, e.code); return 3; }
    ai_network_report none;
    if (ai_network_get_report(AI_HANDLE_NULL, &none) || ai_network_get_report(network, NULL)) { printf("E: a report without a network / a destination succeeded\n"); return 5; }
    if (!print_reports("created")) return 5;
    if (ai_network_destroy(network) != AI_HANDLE_NULL) return 5;
    return 0;
  }
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
  if (!print_reports("ready")) return 5;
  if (ai_network_destroy(network) != AI_HANDLE_NULL) return 5;
  printf("OK %d\n", (int)n_batch);
  return 0;
}
