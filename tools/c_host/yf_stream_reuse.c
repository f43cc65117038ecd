/* A host mistake the library used to turn into silent corruption: a stream destroyed with a launch STILL IN FLIGHT, and a new stream that the runtime hands the
 * same handle value (it does so at once: tools/probe/stream_id_probe.py).  Rounds 3-5 keyed the launch scratch by the handle value, so the successor shared the
 * predecessor's region -- two overlapping launches parking their conv2d_23 tensors in the same bytes.  Round 6 keys it by the runtime's stream ID where
 * hipStreamGetId exists (csrc/yf_stream_scratch.h; ROCm 7.1's runtime, which a C host links -- PyTorch 2.10's bundled one lacks the call): the successor gets a
 * region of its own.  This program does exactly that to the library and checks (1) the region count and (2) both launches' heads against synchronous reference runs.
 *
 *   yf_c_stream_reuse <repo root>     -> one JSON line; exit 0 ok, 1 a check failed, 2 error.      Built by `make chost`.  DEV / TEST TOOL. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/yf_network.h"

#define N 4096
#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)
static ai_u8 activations[AI_NETWORK_DATA_ACTIVATIONS_SIZE] __attribute__((aligned(32)));

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <repo root>\n", argv[0]); return 2; }
  (void)argv;
  ai_handle net = AI_HANDLE_NULL;
  if (ai_network_create(&net, NULL).type != AI_ERROR_NONE) return 2;
  ai_network_params params;
  memset(&params, 0, sizeof params);
  params.params.format = AI_BUFFER_FORMAT_U8; params.params.n_batches = 1; params.params.height = 1; params.params.width = 1;
  params.params.channels = AI_NETWORK_DATA_WEIGHTS_SIZE; params.params.data = ai_network_data_weights_get();
  params.activations.format = AI_BUFFER_FORMAT_U8; params.activations.n_batches = 1; params.activations.height = 1; params.activations.width = 1;
  params.activations.channels = AI_NETWORK_DATA_ACTIVATIONS_SIZE; params.activations.data = AI_HANDLE_PTR(activations);
  if (!ai_network_init(net, &params)) { fprintf(stderr, "ai_network_init: %s\n", yf_network_last_error_text(net)); return 2; }
  const int has_id = dlsym(RTLD_DEFAULT, "hipStreamGetId") != NULL;

  int8_t* h = (int8_t*)malloc((size_t)N * 9408);
  int8_t *d_in[2], *d_out[2], *d_ref[2];
  unsigned long long s = 0x1234567887654321ull;
  for (int b = 0; b < 2; ++b) {
    for (size_t i = 0; i < (size_t)N * 9408; i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(h + i, &s, 8); }
    CHECK(hipMalloc((void**)&d_in[b], (size_t)N * 9408));
    CHECK(hipMemcpy(d_in[b], h, (size_t)N * 9408, hipMemcpyHostToDevice));
    CHECK(hipMalloc((void**)&d_out[b], (size_t)N * 882));
    CHECK(hipMalloc((void**)&d_ref[b], (size_t)N * 882));
  }
  /* reference runs: one at a time on one stream, synchronised */
  hipStream_t r;
  CHECK(hipStreamCreateWithFlags(&r, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    if (yf_network_run_device(net, d_in[b], d_ref[b], N, r) != N) { fprintf(stderr, "%s\n", yf_network_last_error_text(net)); return 2; }
    CHECK(hipStreamSynchronize(r));
  }
  if (yf_network_release_stream(net, r) != 0) return 2;
  CHECK(hipStreamDestroy(r));

  /* the mistake, twenty times: launch on a stream, destroy it at once, launch on its successor */
  int same_handle = 0, mism = 0;
  yf_scratch_stats st;
  unsigned long long max_regions = 0;
  for (int k = 0; k < 20; ++k) {
    hipStream_t a, b;
    CHECK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CHECK(hipMemsetAsync(d_out[0], 0, (size_t)N * 882, a));
    if (yf_network_run_device(net, d_in[0], d_out[0], N, a) != N) { fprintf(stderr, "%s\n", yf_network_last_error_text(net)); return 2; }
    CHECK(hipStreamDestroy(a));                                    /* launch in flight, no release, no synchronise */
    CHECK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    same_handle += (a == b);
    CHECK(hipMemsetAsync(d_out[1], 0, (size_t)N * 882, b));
    if (yf_network_run_device(net, d_in[1], d_out[1], N, b) != N) { fprintf(stderr, "%s\n", yf_network_last_error_text(net)); return 2; }
    if (yf_network_scratch_stats(net, &st) != 0) return 2;
    if (st.regions > max_regions) max_regions = st.regions;
    CHECK(hipDeviceSynchronize());
    for (int q = 0; q < 2; ++q) {
      static int8_t got[(size_t)N * 882], want[(size_t)N * 882];
      CHECK(hipMemcpy(got, d_out[q], sizeof got, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(want, d_ref[q], sizeof want, hipMemcpyDeviceToHost));
      mism += memcmp(got, want, sizeof got) != 0;
    }
    if (yf_network_release_stream(net, b) != 0) return 2;
    CHECK(hipStreamDestroy(b));
  }
  /* with stream ids the successor NEVER finds the predecessor's region: at least two regions were alive at once whenever the handle value was reused */
  const int regions_ok = !has_id || !same_handle || max_regions >= 2;
  printf("{\"tool\": \"tools/c_host/yf_stream_reuse.c\", \"runtime_has_hipStreamGetId\": %s, \"cycles\": 20, \"successor_got_the_same_handle_value\": %d, "
         "\"regions_alive_at_once_max\": %llu, \"launches_that_differ_from_their_reference\": %d, \"events_recorded\": %llu, \"device_syncs\": %llu}\n",
         has_id ? "true" : "false", same_handle, max_regions, mism, st.events_recorded, st.device_syncs);
  if (ai_network_destroy(net) != AI_HANDLE_NULL) return 2;
  return (mism == 0 && regions_ok) ? 0 : 1;
}
