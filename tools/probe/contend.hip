// A kernel that does nothing for a given time on a given number of workgroups: the stand-in for a small collective's kernel (an RCCL all-gather of
// ~0.5 MB per rank runs a few workgroups for a few tens of microseconds) in tools/probe/contention_probe.py, which measures how much such a neighbour
// on another stream costs the fused kernel -- whose grid fills every CU's LDS exactly (two 79.8 KB workgroups per CU).  DEV TOOL.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probe/contend.hip -o tools/probe/contend.so
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(512) spin_kernel(long long cycles, int* sink) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();           // 100 MHz constant-rate counter
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (cycles < 0) sink[0] = lds[threadIdx.x];
}
extern "C" int spin_launch(int workgroups, int threads, int lds_bytes, int microseconds, void* stream) {
  static int* sink = nullptr;
  if (!sink && hipMalloc((void**)&sink, 64) != hipSuccess) return 1;
  hipLaunchKernelGGL(spin_kernel, dim3(workgroups), dim3(threads), lds_bytes, (hipStream_t)stream, (long long)microseconds * 100, sink);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
