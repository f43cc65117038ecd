// What SQ_INSTS_SALU counts on gfx950: three kernels of 64 waves, each wave issues 16 000 instructions of ONE kind (s_add_u32, s_nop 0, s_waitcnt) in a
// 1000-iteration loop.  Run under `rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM --kernel-trace` and read the counter per kernel: the loop
// control is 3 scalar instructions per iteration for all three, so a kernel whose instruction kind is counted shows ~19 000 per wave and one whose kind is
// not shows ~3 000.  DEV TOOL (profiles/r04_fp16/salu_counter_probe.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
__global__ void k_s_add(int* out) {
  int s = 0;
  for (int i = 0; i < 1000; ++i) asm volatile(REP16("s_add_u32 %0, %0, 1\n\t") : "+s"(s));
  if (s == 12345) out[0] = s;
}
__global__ void k_s_nop(int* out) {
  int s = 0;
  for (int i = 0; i < 1000; ++i) asm volatile(REP16("s_nop 0\n\t") : "+s"(s));
  if (s == 12345) out[0] = s;
}
__global__ void k_s_waitcnt(int* out) {
  int s = 0;
  for (int i = 0; i < 1000; ++i) asm volatile(REP16("s_waitcnt lgkmcnt(0)\n\t") : "+s"(s));
  if (s == 12345) out[0] = s;
}
int main() {
  int* d; if (hipMalloc(&d, 64) != hipSuccess) return 1;
  hipLaunchKernelGGL(k_s_add, dim3(64), dim3(64), 0, 0, d);
  hipLaunchKernelGGL(k_s_nop, dim3(64), dim3(64), 0, 0, d);
  hipLaunchKernelGGL(k_s_waitcnt, dim3(64), dim3(64), 0, 0, d);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("ok\n");
  return 0;
}
