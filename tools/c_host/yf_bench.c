/* The bench step from a C host: no Python, no PyTorch -- the C-ABI of include/yf_network.h and the HIP runtime's C API only.
 *
 * north_star asks for "host code in C calling HIP through a thin FFI"; bench.py drives the same entry points through ctypes and uses PyTorch for device
 * memory and streams.  This program is the cross-check that nothing in the measured rate depends on that plumbing: it creates the network the way the
 * reference's aiInit does (stm32/X-CUBE-AI/App/yoloface.c:188-213: ai_network_create, AI_NETWORK_PARAMS_INIT over ai_network_data_weights_get, ai_network_init),
 * keeps eight synthetic 4096-frame batches resident in HBM (the first one carries the six golden frames), and times K steps of ONE launch each
 * (yf_network_run_decode_device: forward + box decode), consecutive steps alternating between two HIP streams as bench.py's do.  Parity: the heads of the
 * golden frames must equal tests/golden/golden_heads.bin byte for byte.
 *
 *   yf_c_bench <repo root> [steps [warmup [streams]]]         -> one JSON line on stdout; exit 0 ok, 1 parity failure, 2 usage / io / HIP / library error
 *
 * Built by `make -C stm32h7-yolo_amd/csrc chost` (gcc; links libyf_network.so and libamdhip64).  DEV / TEST TOOL: not part of the library. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../../include/yf_network.h"

#define N 4096
#define NB 8
#define CAP 4
#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static ai_u8 activations[AI_NETWORK_DATA_ACTIVATIONS_SIZE] __attribute__((aligned(32)));   /* the firmware's AI_ALIGNED(32) arena, yoloface.c:10-11: accepted, unused (activations live in LDS) */

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <repo root> [steps [warmup [streams]]]\n", argv[0]); return 2; }
  const int steps = argc > 2 ? atoi(argv[2]) : 400, warmup = argc > 3 ? atoi(argv[3]) : 100, ns = argc > 4 ? atoi(argv[4]) : 2;
  if (steps < 1 || warmup < 0 || ns < 1 || ns > 2) return 2;
  char path[1024];
  static int8_t gold_in[6 * 9408], gold_heads[6 * 882];
  snprintf(path, sizeof path, "%s/tests/golden/golden_inputs.bin", argv[1]);
  FILE* f = fopen(path, "rb");
  if (!f || fread(gold_in, 9408, 6, f) != 6) { fprintf(stderr, "cannot read %s\n", path); return 2; }
  fclose(f);
  snprintf(path, sizeof path, "%s/tests/golden/golden_heads.bin", argv[1]);
  f = fopen(path, "rb");
  if (!f || fread(gold_heads, 882, 6, f) != 6) { fprintf(stderr, "cannot read %s\n", path); return 2; }
  fclose(f);

  /* aiInit (yoloface.c:188-213) */
  ai_handle net = AI_HANDLE_NULL;
  ai_error err = ai_network_create(&net, NULL);
  if (err.type != AI_ERROR_NONE) { fprintf(stderr, "ai_network_create: type %u code %u\n", (unsigned)err.type, (unsigned)err.code); return 2; }
  ai_network_params params;
  memset(&params, 0, sizeof params);
  params.params.format = AI_BUFFER_FORMAT_U8; params.params.n_batches = 1; params.params.height = 1; params.params.width = 1;
  params.params.channels = AI_NETWORK_DATA_WEIGHTS_SIZE; params.params.data = ai_network_data_weights_get();
  params.activations.format = AI_BUFFER_FORMAT_U8; params.activations.n_batches = 1; params.activations.height = 1; params.activations.width = 1;
  params.activations.channels = AI_NETWORK_DATA_ACTIVATIONS_SIZE; params.activations.data = AI_HANDLE_PTR(activations);
  if (!ai_network_init(net, &params)) {
    err = ai_network_get_error(net);
    fprintf(stderr, "ai_network_init: type %u code %u (%s)\n", (unsigned)err.type, (unsigned)err.code, yf_network_last_error_text(net));
    return 2;
  }

  /* eight batches resident in HBM (308 MB: more than the 256 MB Infinity Cache), uniform int8 from a xorshift generator; batch 0 starts with the golden frames */
  int8_t* h_in = (int8_t*)malloc((size_t)N * 9408);
  int8_t* d_in[NB];
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (int b = 0; b < NB; ++b) {
    for (size_t i = 0; i < (size_t)N * 9408; i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(h_in + i, &s, 8); }
    if (b == 0) memcpy(h_in, gold_in, sizeof gold_in);
    CHECK(hipMalloc((void**)&d_in[b], (size_t)N * 9408));
    CHECK(hipMemcpy(d_in[b], h_in, (size_t)N * 9408, hipMemcpyHostToDevice));
  }
  hipStream_t st[2];
  int8_t* d_heads[2]; yf_det* d_dets[2]; int* d_counts[2];
  for (int k = 0; k < 2; ++k) {
    CHECK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    CHECK(hipMalloc((void**)&d_heads[k], (size_t)N * 882));
    CHECK(hipMalloc((void**)&d_dets[k], (size_t)N * CAP * sizeof(yf_det)));
    CHECK(hipMalloc((void**)&d_counts[k], (size_t)N * sizeof(int)));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  long step_no = 0;
#define STEP(K_STREAMS) do { const int k_ = (int)(step_no % (K_STREAMS)); \
    if (yf_network_run_decode_device(net, d_in[step_no % NB], d_heads[k_], N, YF_DECODE_PY, 1.f, 1.f, d_dets[k_], d_counts[k_], CAP, st[k_]) != N) { \
      fprintf(stderr, "yf_network_run_decode_device: %s\n", yf_network_last_error_text(net)); return 2; } ++step_no; } while (0)

  /* clock settle (untimed, as bench.py's): 60 ms of the same launches on one stream; then W warm-up steps; then exactly K timed steps */
  for (double t0 = now_s(); now_s() - t0 < 0.060;) { for (int i = 0; i < 8; ++i) STEP(1); CHECK(hipDeviceSynchronize()); }
  for (int i = 0; i < warmup; ++i) STEP(ns);
  CHECK(hipDeviceSynchronize());
  const double t0 = now_s();
  for (int i = 0; i < steps; ++i) STEP(ns);
  CHECK(hipDeviceSynchronize());
  const double elapsed = now_s() - t0;
  /* the kernel alone: 100 back-to-back launches on ONE stream between two events */
  step_no = 0;
  CHECK(hipEventRecord(e0, st[0]));
  for (int i = 0; i < 100; ++i) STEP(1);
  CHECK(hipEventRecord(e1, st[0]));
  CHECK(hipEventSynchronize(e1));
  float kernel_ms = 0.f;
  CHECK(hipEventElapsedTime(&kernel_ms, e0, e1));
  kernel_ms /= 100.f;

  /* parity: batch 0 once more, golden heads byte for byte; and the real frame (index 5) must fire */
  step_no = 0;
  STEP(1);
  CHECK(hipDeviceSynchronize());
  static int8_t got[6 * 882];
  static int counts[6];
  CHECK(hipMemcpy(got, d_heads[0], sizeof got, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(counts, d_counts[0], sizeof counts, hipMemcpyDeviceToHost));
  const int same = memcmp(got, gold_heads, sizeof got) == 0;
  printf("{\"tool\": \"tools/c_host/yf_bench.c\", \"host\": \"C (gcc), HIP runtime C API, no Python\", \"metric\": \"images/sec int8 YOLO-face 56x56\", \"value\": %.1f, "
         "\"unit\": \"images/s\", \"n_gpus\": 1, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"launch_streams\": %d, \"kernel_ms_alone\": %.4f, "
         "\"kernel\": \"%s\", \"build_id\": \"%s\", \"golden_heads_equal\": %s, \"detections_on_the_real_frame\": %d}\n",
         (double)N * steps / elapsed, steps, warmup, elapsed / steps * 1e3, ns, (double)kernel_ms, yf_network_kernel_name(net), yf_network_build_id(),
         same ? "true" : "false", counts[5]);
  for (int k = 0; k < 2; ++k) (void)yf_network_release_stream(net, st[k]);
  if (ai_network_destroy(net) != AI_HANDLE_NULL) return 2;
  return same && counts[5] > 0 ? 0 : 1;
}
