// LDS bank behaviour on gfx950: cycles per wave-instruction of ds_read_b32 / b64 / b128 and ds_write_b32 / b64 for lane strides
// (lane l accesses byte l * STRIDE + OFFSET) -- sixteen waves of one workgroup alone on a CU, 512 back-to-back accesses each, s_memtime
// around them: the LDS pipe is saturated, so the figure is the pipe's cost per wave-instruction.  DEV TOOL.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_banks tools/probe/lds_banks.hip && /tmp/lds_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define N 512
template <int WIDTH, bool WRITE>
__global__ void probe(unsigned long long* out, int stride, int rowlanes, int rowstride, unsigned* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  // address: lanes in rows of `rowlanes`; within a row `stride` bytes apart, rows `rowstride` bytes apart
  const unsigned a = (unsigned)((lane % rowlanes) * stride + (lane / rowlanes) * rowstride);
  for (int i = threadIdx.x; i < 40960 / 4; i += 1024) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if constexpr (WRITE) {
      if constexpr (WIDTH == 4) asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(acc) : "memory");
      else { unsigned long long v = acc; asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v) : "memory"); }
    } else {
      if constexpr (WIDTH == 4) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); acc ^= v; }
      else if constexpr (WIDTH == 8) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory"); acc ^= (unsigned)v; }
      else { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); acc ^= v.x; }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  unsigned long long t2 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t2 - t0;
  if (acc == 0x1234567) sink[0] = acc;
}
// MODE 0: ds_read2_b32 (dwords at a and a + off1), 1: ds_read_u8, 2: ds_read2_b64, 3: ds_write2_b32
template <int MODE>
__global__ void probe2(unsigned long long* out, int stride, int rowlanes, int rowstride, int off1, unsigned* sink) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  const unsigned a = (unsigned)((lane % rowlanes) * stride + (lane / rowlanes) * rowstride);
  const unsigned b = a + (unsigned)off1;
  for (int i = threadIdx.x; i < 40960 / 4; i += 1024) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  unsigned acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N; ++i) {
    if constexpr (MODE == 0) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset1:61" : "=v"(v) : "v"(a) : "memory"); acc ^= (unsigned)v; }
    else if constexpr (MODE == 1) { unsigned v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(a) : "memory"); acc ^= v; }
    else if constexpr (MODE == 2) { uint4 v; asm volatile("ds_read2_b64 %0, %1 offset1:31" : "=v"(v) : "v"(a) : "memory"); acc ^= v.x; }
    else { asm volatile("ds_write2_b32 %0, %1, %2 offset1:61" :: "v"(a), "v"(acc), "v"(b) : "memory"); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned long long t2 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[0] = t2 - t0;
  if (acc == 0x1234567) sink[0] = acc;
}
template <int MODE>
double run2(int stride, int rowlanes, int rowstride, unsigned long long* d, unsigned* sink) {
  unsigned long long h = 0, best = ~0ull;
  for (int r = 0; r < 5; ++r) {
    hipLaunchKernelGGL((probe2<MODE>), dim3(1), dim3(1024), 40960, 0, d, stride, rowlanes, rowstride, 244, sink);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    if (h < best) best = h;
  }
  return (double)best / N / 16;
}
template <int WIDTH, bool WRITE>
double run(int stride, int rowlanes, int rowstride, unsigned long long* d, unsigned* sink) {
  unsigned long long h = 0, best = ~0ull;
  for (int r = 0; r < 5; ++r) {
    hipLaunchKernelGGL((probe<WIDTH, WRITE>), dim3(1), dim3(1024), 40960, 0, d, stride, rowlanes, rowstride, sink);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    if (h < best) best = h;
  }
  return (double)best / N / 16;
}
int main() {
  unsigned long long* d; unsigned* sink; hipMalloc(&d, 8); hipMalloc(&sink, 4);
  printf("LDS pipe cycles per wave-instruction (s_memtime units / 16 waves); lane l -> byte l * stride\n");
  const int strides[] = {4, 8, 12, 16, 20, 24, 32, 36, 40, 48, 64, 80, 96, 128};
  printf("%8s %10s %10s %10s %10s %10s\n", "stride", "read_b32", "read_b64", "read_b128", "write_b32", "write_b64");
  for (int s : strides) {
    printf("%8d %10.1f", s, run<4, false>(s, 64, 0, d, sink));
    if (s % 8 == 0) printf(" %10.1f", run<8, false>(s, 64, 0, d, sink)); else printf(" %10s", "-");
    if (s % 16 == 0) printf(" %10.1f", run<16, false>(s, 64, 0, d, sink)); else printf(" %10s", "-");
    printf(" %10.1f", run<4, true>(s, 64, 0, d, sink));
    if (s % 8 == 0) printf(" %10.1f", run<8, true>(s, 64, 0, d, sink)); else printf(" %10s", "-");
    printf("\n");
  }
  printf("\n%8s %12s %10s %12s %12s   (read2: second dword 244 bytes / second qword 248 bytes behind the first)\n", "stride", "read2_b32", "read_u8", "read2_b64", "write2_b32");
  for (int s2 : {1, 4, 8, 12, 20, 36, 40, 48}) {
    printf("%8d", s2);
    if (s2 % 4 == 0) printf(" %12.1f", run2<0>(s2, 64, 0, d, sink)); else printf(" %12s", "-");
    printf(" %10.1f", run2<1>(s2, 64, 0, d, sink));
    if (s2 % 8 == 0) printf(" %12.1f", run2<2>(s2, 64, 0, d, sink)); else printf(" %12s", "-");
    if (s2 % 4 == 0) printf(" %12.1f", run2<3>(s2, 64, 0, d, sink)); else printf(" %12s", "-");
    printf("\n");
  }
  printf("  read2_b32, 16 lanes x 8 B, rows 244 B (int8 T1 taps, two taps per instruction): %.1f\n", run2<0>(8, 16, 244, d, sink));
  printf("  read2_b32, 16 lanes x 40 B, rows 364 B (int8 T19 taps): %.1f\n", run2<0>(40, 16, 364, d, sink));
  printf("  read_u8, random-ish (lane * 37 mod 256): %.1f\n", run2<1>(37, 64, 0, d, sink));
  printf("\ntile rows: 16 lanes per row `stride` bytes apart, rows `rowstride` bytes apart (the depthwise tap pattern)\n");
  struct { int w, s, rs; const char* what; } cases[] = {
    {4, 8, 240, "int8 T1 8 B px, row 240"}, {4, 8, 244, "int8 T1 + 4 B skew"}, {4, 40, 1160, "int8 T4 stride 2 (40 B), rows 2 x 580"},
    {4, 40, 1164, "  + 4 B"}, {4, 36, 576, "int8 T8 36 B px, row 576"}, {4, 36, 1584, "160 T8 row 1584"}, {4, 36, 1600, "160 T8 row 1600"},
    {4, 48, 720, "int8 T15 stride 2 (48 B), rows 2 x 360"}, {4, 48, 724, "  + 4"}, {4, 48, 728, "  + 8"},
    {8, 16, 480, "fp16 T1 16 B px, row 480"}, {8, 16, 488, "  + 8 B skew"}, {8, 80, 1280, "fp16 T8 80 B px, row 1280"}, {8, 80, 1288, "  + 8 B"},
    {8, 80, 2320, "fp16 T4 stride 2 (80 B), rows 2 x 1160"}, {8, 96, 1440, "fp16 T15 stride 2 (96 B), rows 2 x 720"}, {8, 96, 1448, "  + 8"},
    {16, 16, 256, "b128 16 B px"}, {16, 48, 768, "b128 48 B px"}, {16, 80, 1280, "b128 80 B px"}, {16, 96, 1536, "b128 96 B px"}, {16, 64, 1024, "b128 64 B px"}};
  for (auto& c : cases) {
    double v = c.w == 4 ? run<4, false>(c.s, 16, c.rs, d, sink) : c.w == 8 ? run<8, false>(c.s, 16, c.rs, d, sink) : run<16, false>(c.s, 16, c.rs, d, sink);
    printf("  read_b%-3d stride %3d rowstride %5d : %6.1f   %s\n", c.w * 8, c.s, c.rs, v, c.what);
  }
  return 0;
}
