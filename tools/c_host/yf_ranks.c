/* The N-rank bench step from a C host: one PROCESS per GPU, RCCL all-gather of the detections, no Python, no PyTorch, no torch.distributed.
 *
 * north_star: "host code in C calling HIP through a thin FFI ... frames are batch-sharded across the 8 GPUs of one node with an RCCL all-gather of
 * detections over xGMI".  bench.py does that through torch.distributed; this program is the same flow on the C-ABI of include/yf_network.h, the HIP
 * runtime's C API and librccl's C API alone, so that nothing of the multi-GPU path depends on the Python plumbing (and so that a first 8-GPU run has a
 * second, independent host to be compared with).  The reference has no counterpart: one core, one context (stm32/X-CUBE-AI/App/network.c:51-52); per rank the
 * call order is the reference's -- aiInit once (yoloface.c:188-213), then per step what aiRun + post_process do (yoloface.c:216-240, 98-152), here ONE launch.
 *
 *   parent   forks N children BEFORE any HIP call (a process that has initialised the GPU must not fork workers), hands rank 0's ncclUniqueId round by
 *            pipe, releases the ranks into the timed region together, collects every rank's time and verdict, prints ONE JSON line (time = MAX over ranks)
 *            and exits non-zero if any child failed.
 *   child r  yf_network_set_device(r % ndev) -> aiInit -> ncclCommInitRank -> B input batches of 4096 frames in HBM (rank-seeded; rank 0's first batch starts
 *            with the golden frames) -> clock settle, W warm-up steps, K timed steps.  A step = yf_network_run_decode_device into one of FOUR alternating
 *            record buffers [records 4096 x cap x 28 B | counts 4096 x 4 B] + yf_network_all_gather_device of that buffer on the SAME stream (consecutive
 *            steps alternate between two streams; a buffer always meets the same stream, so kernel -> gather -> next kernel into the buffer is stream order).
 *   checks   rank 0: heads of the golden frames byte for byte; every rank: its own block at its own place in the gathered buffer, and the per-rank sums of
 *            the gathered COUNTS it holds -- the parent compares every rank's view with what each rank says it sent: rank = frame order on every rank.
 *
 *   yf_c_ranks <repo root> <ranks> [steps [warmup [input batches [--no-exchange]]]]
 *       --no-exchange   no communicator and no collective (RCCL refuses two ranks on one device): rehearses fork / pipes / barrier / MAX with several ranks
 *                       sharing ONE GPU; the gathered-order check is skipped.
 *   exit 0 ok, 1 a check failed, 2 usage / io / HIP / RCCL / library error.
 *
 * Built by `make -C stm32h7-yolo_amd/csrc chost` (gcc; links libyf_network.so and libamdhip64; librccl.so.1 is dlopen'ed).  DEV / TEST TOOL: not part of the library. */
#define _GNU_SOURCE
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <errno.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include "../../include/yf_network.h"

#define N 4096
#define CAP 4
#define NBUF 4
#define MAX_RANKS 64
#define ID_BYTES 128                       /* NCCL_UNIQUE_ID_BYTES (rccl.h) */
#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #call, hipGetErrorString(e_)); return 2; } } while (0)

typedef struct { char internal[ID_BYTES]; } nccl_id;
typedef int (*fn_get_id)(nccl_id*);
typedef int (*fn_init_rank)(void** comm, int nranks, nccl_id id, int rank);
typedef int (*fn_comm_destroy)(void* comm);

typedef struct {                           /* what a child reports to the parent */
  int status;                              /* 0 ok, 1 check failed, 2 error */
  int golden_ok, own_block_ok, detections_on_the_real_frame;
  double elapsed_s, kernel_ms;
  long sent_count_sum;                     /* sum of this rank's counts in the check step */
  long seen_count_sum[MAX_RANKS];          /* per rank: sum of the counts this rank holds for it after the gather */
} report;

static int g_rank = -1;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static int write_all(int fd, const void* p, size_t n) { const char* c = (const char*)p; while (n) { ssize_t w = write(fd, c, n); if (w <= 0) { if (errno == EINTR) continue; return -1; } c += w; n -= (size_t)w; } return 0; }
static int read_all(int fd, void* p, size_t n) { char* c = (char*)p; while (n) { ssize_t r = read(fd, c, n); if (r <= 0) { if (r < 0 && errno == EINTR) continue; return -1; } c += r; n -= (size_t)r; } return 0; }

static ai_u8 activations[AI_NETWORK_DATA_ACTIVATIONS_SIZE] __attribute__((aligned(32)));

/* one rank; up = pipe to the parent, down = pipe from it */
static int child(const char* root, int rank, int world, int steps, int warmup, int nb, int exchange, int up, int down, report* rep) {
  g_rank = rank;
  char path[1024];
  static int8_t gold_in[6 * 9408], gold_heads[6 * 882];
  snprintf(path, sizeof path, "%s/tests/golden/golden_inputs.bin", root);
  FILE* f = fopen(path, "rb");
  if (!f || fread(gold_in, 9408, 6, f) != 6) { fprintf(stderr, "cannot read %s\n", path); return 2; }
  fclose(f);
  snprintf(path, sizeof path, "%s/tests/golden/golden_heads.bin", root);
  f = fopen(path, "rb");
  if (!f || fread(gold_heads, 882, 6, f) != 6) { fprintf(stderr, "cannot read %s\n", path); return 2; }
  fclose(f);

  int ndev = 0;
  CHECK(hipGetDeviceCount(&ndev));
  if (ndev < 1) { fprintf(stderr, "rank %d: no HIP device\n", rank); return 2; }
  if (exchange && world > ndev) { fprintf(stderr, "rank %d: %d ranks but %d devices (RCCL refuses two ranks on one device; --no-exchange rehearses the rest)\n", rank, world, ndev); return 2; }
  const int dev = rank % ndev;
  CHECK(hipSetDevice(dev));

  /* aiInit (yoloface.c:188-213) on this rank's device */
  ai_handle net = AI_HANDLE_NULL;
  ai_error err = ai_network_create(&net, NULL);
  if (err.type != AI_ERROR_NONE) { fprintf(stderr, "rank %d: ai_network_create: type %u code %u\n", rank, (unsigned)err.type, (unsigned)err.code); return 2; }
  if (yf_network_set_device(net, dev) != 0) return 2;
  ai_network_params params;
  memset(&params, 0, sizeof params);
  params.params.format = AI_BUFFER_FORMAT_U8; params.params.n_batches = 1; params.params.height = 1; params.params.width = 1;
  params.params.channels = AI_NETWORK_DATA_WEIGHTS_SIZE; params.params.data = ai_network_data_weights_get();
  params.activations.format = AI_BUFFER_FORMAT_U8; params.activations.n_batches = 1; params.activations.height = 1; params.activations.width = 1;
  params.activations.channels = AI_NETWORK_DATA_ACTIVATIONS_SIZE; params.activations.data = AI_HANDLE_PTR(activations);
  if (!ai_network_init(net, &params)) {
    err = ai_network_get_error(net);
    fprintf(stderr, "rank %d: ai_network_init: type %u code %u (%s)\n", rank, (unsigned)err.type, (unsigned)err.code, yf_network_last_error_text(net));
    return 2;
  }

  /* communicator: rank 0's unique id goes up to the parent and comes down to everybody */
  void* comm = NULL;
  fn_comm_destroy comm_destroy = NULL;
  if (exchange) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "rank %d: librccl.so.1: %s\n", rank, dlerror()); return 2; }
    fn_get_id get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId");
    fn_init_rank init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
    if (!get_id || !init_rank || !comm_destroy) { fprintf(stderr, "rank %d: RCCL symbols missing\n", rank); return 2; }
    nccl_id id;
    memset(&id, 0, sizeof id);
    if (rank == 0) {
      const int rc = get_id(&id);
      if (rc != 0) { fprintf(stderr, "ncclGetUniqueId: ncclResult_t %d\n", rc); return 2; }
      if (write_all(up, "i", 1) != 0 || write_all(up, &id, sizeof id) != 0) return 2;      /* every message to the parent starts with a tag: 'i' id, 'r' ready, 'R' report */
    }
    if (read_all(down, &id, sizeof id) != 0) return 2;
    const int rc = init_rank(&comm, world, id, rank);
    if (rc != 0) { fprintf(stderr, "rank %d: ncclCommInitRank: ncclResult_t %d\n", rank, rc); return 2; }
  }

  /* inputs: nb batches in HBM, xorshift seeded by the rank; rank 0's batch 0 starts with the golden frames */
  int8_t* h_in = (int8_t*)malloc((size_t)N * 9408);
  int8_t** d_in = (int8_t**)calloc((size_t)nb, sizeof *d_in);
  unsigned long long s = 0x9E3779B97F4A7C15ull ^ (0xD1B54A32D192ED03ull * (unsigned long long)(rank + 1));
  for (int b = 0; b < nb; ++b) {
    for (size_t i = 0; i < (size_t)N * 9408; i += 8) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; memcpy(h_in + i, &s, 8); }
    if (b == 0 && rank == 0) memcpy(h_in, gold_in, sizeof gold_in);
    CHECK(hipMalloc((void**)&d_in[b], (size_t)N * 9408));
    CHECK(hipMemcpy(d_in[b], h_in, (size_t)N * 9408, hipMemcpyHostToDevice));
  }
  free(h_in);
  /* exchange record of one step: [records N x CAP x 28 | counts N x 4], 16-byte aligned sections (the layout of sharding.DetectionExchange) */
  const size_t off_c = ((size_t)N * CAP * sizeof(yf_det) + 15) & ~(size_t)15, rec_bytes = (off_c + (size_t)N * 4 + 15) & ~(size_t)15;
  hipStream_t st[2];
  int8_t* d_heads[NBUF]; char* d_local[NBUF]; char* d_gath[NBUF];
  for (int k = 0; k < 2; ++k) CHECK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
  for (int k = 0; k < NBUF; ++k) {
    CHECK(hipMalloc((void**)&d_heads[k], (size_t)N * 882));
    CHECK(hipMalloc((void**)&d_local[k], rec_bytes));
    CHECK(hipMemset(d_local[k], 0, rec_bytes));
    CHECK(hipMalloc((void**)&d_gath[k], rec_bytes * (size_t)world));
  }
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  long step_no = 0;
  /* a step: ONE launch (+ ONE collective on the same stream).  WITH_EXCHANGE = 0: the kernel alone (settle, kernel timing) */
#define STEP(K_STREAMS, WITH_EXCHANGE) do { const int b_ = (int)(step_no % NBUF), k_ = (int)(step_no % (K_STREAMS)); \
    if (yf_network_run_decode_device(net, d_in[step_no % nb], d_heads[b_], N, YF_DECODE_PY, 1.f, 1.f, d_local[b_], d_local[b_] + off_c, CAP, st[k_]) != N) { \
      fprintf(stderr, "rank %d: yf_network_run_decode_device: %s\n", rank, yf_network_last_error_text(net)); return 2; } \
    if ((WITH_EXCHANGE) && exchange && yf_network_all_gather_device(net, comm, d_local[b_], d_gath[b_], rec_bytes, st[k_]) != (long)rec_bytes) { \
      fprintf(stderr, "rank %d: yf_network_all_gather_device: %s\n", rank, yf_network_last_error_text(net)); return 2; } \
    ++step_no; } while (0)

  for (double t0 = now_s(); now_s() - t0 < 0.060;) { for (int i = 0; i < 8; ++i) STEP(1, 0); CHECK(hipDeviceSynchronize()); }   /* clock settle */
  for (int i = 0; i < warmup; ++i) STEP(2, 1);
  CHECK(hipDeviceSynchronize());
  /* barrier: tell the parent this rank is ready, wait for its go */
  char tok = 'r';
  if (write_all(up, &tok, 1) != 0 || read_all(down, &tok, 1) != 0) return 2;      /* (a parent that gave up has closed the pipe: read fails, this rank leaves) */
  const double t0 = now_s();
  for (int i = 0; i < steps; ++i) STEP(2, 1);
  CHECK(hipDeviceSynchronize());
  rep->elapsed_s = now_s() - t0;
  /* the kernel alone: 100 back-to-back launches on ONE stream between two events */
  step_no = 0;
  CHECK(hipEventRecord(e0, st[0]));
  for (int i = 0; i < 100; ++i) STEP(1, 0);
  CHECK(hipEventRecord(e1, st[0]));
  CHECK(hipEventSynchronize(e1));
  float kms = 0.f;
  CHECK(hipEventElapsedTime(&kms, e0, e1));
  rep->kernel_ms = kms / 100.0;

  /* checks: batch 0 once more through buffer 0 */
  step_no = 0;
  STEP(2, 1);
  CHECK(hipDeviceSynchronize());
  static int8_t got[6 * 882];
  int* counts = (int*)malloc((size_t)N * 4);
  CHECK(hipMemcpy(got, d_heads[0], sizeof got, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(counts, d_local[0] + off_c, (size_t)N * 4, hipMemcpyDeviceToHost));
  rep->golden_ok = rank != 0 || memcmp(got, gold_heads, sizeof got) == 0;
  rep->detections_on_the_real_frame = rank == 0 ? counts[5] : 0;
  rep->sent_count_sum = 0;
  for (int i = 0; i < N; ++i) rep->sent_count_sum += counts[i];
  rep->own_block_ok = 1;
  if (exchange) {
    char* h_loc = (char*)malloc(rec_bytes);
    char* h_gat = (char*)malloc(rec_bytes * (size_t)world);
    CHECK(hipMemcpy(h_loc, d_local[0], rec_bytes, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h_gat, d_gath[0], rec_bytes * (size_t)world, hipMemcpyDeviceToHost));
    rep->own_block_ok = memcmp(h_gat + rec_bytes * (size_t)rank, h_loc, rec_bytes) == 0;
    for (int r = 0; r < world; ++r) {
      const int* c = (const int*)(h_gat + rec_bytes * (size_t)r + off_c);
      long sum = 0;
      for (int i = 0; i < N; ++i) sum += c[i];
      rep->seen_count_sum[r] = sum;
    }
    free(h_loc); free(h_gat);
  }
  free(counts);
  for (int k = 0; k < 2; ++k) (void)yf_network_release_stream(net, st[k]);
  if (comm) (void)comm_destroy(comm);
  if (ai_network_destroy(net) != AI_HANDLE_NULL) return 2;
  return (rep->golden_ok && rep->own_block_ok && (rank != 0 || rep->detections_on_the_real_frame > 0)) ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <repo root> <ranks> [steps [warmup [input batches [--no-exchange]]]]\n", argv[0]); return 2; }
  const int world = atoi(argv[2]);
  const int steps = argc > 3 ? atoi(argv[3]) : 400, warmup = argc > 4 ? atoi(argv[4]) : 100, nb = argc > 5 ? atoi(argv[5]) : 8;
  const int exchange = !(argc > 6 && strcmp(argv[6], "--no-exchange") == 0);
  if (world < 1 || world > MAX_RANKS || steps < 1 || warmup < 0 || nb < 1) return 2;
  /* fork the ranks BEFORE anything touches HIP in this process (the parent never does) */
  int up[MAX_RANKS][2], down[MAX_RANKS][2];
  pid_t pid[MAX_RANKS];
  for (int r = 0; r < world; ++r) {
    if (pipe(up[r]) != 0 || pipe(down[r]) != 0) { perror("pipe"); return 2; }
    pid[r] = fork();
    if (pid[r] < 0) { perror("fork"); return 2; }
    if (pid[r] == 0) {
      /* the parent's ends of every pipe made so far are open in this child too: close them.  (The CHILD ends of earlier ranks were closed by the parent right
       * after their fork -- their descriptor numbers may have been handed out again by this rank's pipe(): a first form closed them here and with them its own pipe.) */
      for (int q = 0; q <= r; ++q) { close(up[q][0]); close(down[q][1]); }
      report rep;
      memset(&rep, 0, sizeof rep);
      rep.status = child(argv[1], r, world, steps, warmup, nb, exchange, up[r][1], down[r][0], &rep);
      (void)write_all(up[r][1], "R", 1);
      (void)write_all(up[r][1], &rep, sizeof rep);
      _exit(rep.status);
    }
    close(up[r][1]); close(down[r][0]);
  }
  signal(SIGPIPE, SIG_IGN);                /* a rank that died: writing to it fails with EPIPE instead of killing the parent */
  static report rep[MAX_RANKS];
  int reported[MAX_RANKS] = {0};
  int bad = 0;
  /* next message of rank r must carry `tag`; a rank that failed early sends its report ('R') instead: taken, and the run is marked bad */
#define EXPECT(r, tag) do { char t_ = 0; if (reported[r] || read_all(up[r][0], &t_, 1) != 0) { bad = 1; } \
    else if (t_ == 'R') { if (read_all(up[r][0], &rep[r], sizeof rep[r]) != 0) rep[r].status = 2; reported[r] = 1; if ((tag) != 'R') bad = 1; } \
    else if (t_ != (tag)) { bad = 1; } } while (0)
  if (exchange) {                          /* rank 0's unique id -> every rank */
    nccl_id id;
    memset(&id, 0, sizeof id);
    EXPECT(0, 'i');
    if (!bad && read_all(up[0][0], &id, sizeof id) != 0) bad = 1;
    for (int r = 0; r < world && !bad; ++r) if (write_all(down[r][1], &id, sizeof id) != 0) bad = 1;
  }
  for (int r = 0; r < world && !bad; ++r) EXPECT(r, 'r');                                      /* every rank is through its warm-up ... */
  for (int r = 0; r < world && !bad; ++r) if (write_all(down[r][1], "g", 1) != 0) bad = 1;     /* ... and they enter the timed region together */
  if (bad) for (int r = 0; r < world; ++r) close(down[r][1]);      /* a rank died: the others' reads end instead of waiting for ever */
  for (int r = 0; r < world; ++r) if (!reported[r]) { EXPECT(r, 'R'); if (!reported[r]) rep[r].status = 2; }
  /* Collect the ranks.  After a failure the survivors may sit in a collective whose peer is gone, for ever: they get 20 s to leave by themselves (their pipe reads
   * have ended), then SIGKILL -- the parent must not hang on the first 8-GPU run because one rank died. */
  int worst = 0, left = world, done[MAX_RANKS] = {0};
  for (double t_kill = now_s() + 20.0; left > 0;) {
    for (int r = 0; r < world; ++r) {
      if (done[r]) continue;
      int st = 0;
      const pid_t w = waitpid(pid[r], &st, bad ? WNOHANG : 0);
      if (w == 0) continue;
      done[r] = 1; --left;
      if (w < 0 || !WIFEXITED(st)) { worst = 2; bad = 1; continue; }
      if (WEXITSTATUS(st) > worst) worst = WEXITSTATUS(st);
      if (WEXITSTATUS(st) == 2) bad = 1;
    }
    if (left > 0 && bad) {
      if (now_s() > t_kill) { for (int r = 0; r < world; ++r) if (!done[r]) kill(pid[r], SIGKILL); t_kill = now_s() + 1e9; }
      else { struct timespec nap = {0, 50 * 1000 * 1000}; nanosleep(&nap, NULL); }
    }
  }
  double elapsed = 0, kernel_ms = 0;
  int order_ok = 1, golden_ok = 1, own_ok = 1;
  for (int r = 0; r < world; ++r) {
    if (rep[r].status > worst) worst = rep[r].status;
    if (rep[r].elapsed_s > elapsed) elapsed = rep[r].elapsed_s;
    if (rep[r].kernel_ms > kernel_ms) kernel_ms = rep[r].kernel_ms;
    golden_ok = golden_ok && rep[r].golden_ok; own_ok = own_ok && rep[r].own_block_ok;
    if (exchange) for (int q = 0; q < world; ++q) order_ok = order_ok && rep[r].seen_count_sum[q] == rep[q].sent_count_sum;      /* rank r holds rank q's counts at rank q's place */
  }
  if (exchange && !order_ok && worst < 1) worst = 1;
  printf("{\"tool\": \"tools/c_host/yf_ranks.c\", \"host\": \"C (gcc): one process per GPU, HIP runtime C API, librccl C API, no Python\", "
         "\"metric\": \"images/sec int8 YOLO-face 56x56\", \"value\": %.1f, \"unit\": \"images/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, "
         "\"scaling\": \"weak\", \"global_batch\": %ld, \"launch_streams\": 2, \"exchange\": \"%s\", \"exchange_buffers\": %d, \"exchange_bytes_per_rank_per_step\": %zu, "
         "\"kernel_ms_alone_max_over_ranks\": %.4f, \"golden_heads_equal\": %s, \"own_block_at_own_place\": %s, \"gathered_counts_in_rank_order_on_every_rank\": %s, "
         "\"detections_on_the_real_frame\": %d, \"build_id\": \"%s\", \"status\": %d}\n",
         elapsed > 0 ? (double)N * world * steps / elapsed : 0.0, world, steps, warmup, elapsed / steps * 1e3, (long)N * world,
         exchange ? "RCCL all-gather of detection records and counts per step (yf_network_all_gather_device)" : "none (--no-exchange rehearsal)", NBUF,
         exchange ? ((((size_t)N * CAP * sizeof(yf_det) + 15) & ~(size_t)15) + (size_t)N * 4 + 15) & ~(size_t)15 : (size_t)0,
         kernel_ms, golden_ok ? "true" : "false", own_ok ? "true" : "false", exchange ? (order_ok ? "true" : "false") : "null",
         rep[0].detections_on_the_real_frame, yf_network_build_id(), worst);
  return worst;
}
