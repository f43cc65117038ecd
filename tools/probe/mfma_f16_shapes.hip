// Do v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16 give the SAME bits for a lane-private dot product (8 fp16 products + an fp32 addend)?
// Lane-private: a lane's own 8 input channels against 16 output channels: four passes of the 16x16x32 form, or one 32x32x16 instruction.  DEV PROBE.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_f16_shapes.hip -o tools/probe/mfma_f16_shapes && tools/probe/mfma_f16_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void probe(const _Float16* W /*[16][8]*/, const _Float16* X /*[64][8]*/, const float* bias /*[16]*/, float* out16 /*[64][16]*/, float* out32 /*[64][16]*/) {
  const int l = threadIdx.x;
  v8h x; for (int k = 0; k < 8; ++k) x[k] = X[l * 8 + k];
  {  // 16x16x32, lane-private: lane (g = l >> 4, c = l & 15); A row c holds W[4 ps + (c & 3)] iff (c >> 2) == g
    const int g = l >> 4, c = l & 15;
    for (int ps = 0; ps < 4; ++ps) {
      v8h a; for (int k = 0; k < 8; ++k) a[k] = ((c >> 2) == g) ? W[(4 * ps + (c & 3)) * 8 + k] : (_Float16)0;
      v4f acc = {bias[4 * ps], bias[4 * ps + 1], bias[4 * ps + 2], bias[4 * ps + 3]};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, x, acc, 0, 0, 0);
      for (int j = 0; j < 4; ++j) out16[l * 16 + 4 * ps + j] = acc[j];
    }
  }
  {  // 32x32x16: lane l: row i = l % 32, k block kb = l / 32; A[i][8 kb ..] = W[4 (i >> 3) + (i & 3)] iff ((i >> 2) & 1) == kb
    const int i = l & 31, kb = l >> 5;
    v8h a; for (int k = 0; k < 8; ++k) a[k] = (((i >> 2) & 1) == kb) ? W[(4 * (i >> 3) + (i & 3)) * 8 + k] : (_Float16)0;
    v16f acc; for (int v = 0; v < 16; ++v) acc[v] = bias[v];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, x, acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out32[l * 16 + v] = acc[v];
  }
}
int main() {
  _Float16 *W, *X; float *b, *o16, *o32;
  hipMallocManaged(&W, 16 * 8 * 2); hipMallocManaged(&X, 64 * 8 * 2); hipMallocManaged(&b, 64); hipMallocManaged(&o16, 64 * 16 * 4); hipMallocManaged(&o32, 64 * 16 * 4);
  srand(7);
  long diff = 0, total = 0;
  for (int trial = 0; trial < 2000; ++trial) {
    const float sw = (trial % 3 == 0) ? 8.f : 1.f;
    for (int i = 0; i < 128; ++i) W[i] = (_Float16)(((rand() % 2001) - 1000) / 1000.f * sw);
    for (int i = 0; i < 512; ++i) X[i] = (_Float16)(((rand() % 2001) - 1000) / 500.f);
    for (int i = 0; i < 16; ++i) b[i] = ((rand() % 2001) - 1000) / 700.f;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, W, X, b, o16, o32);
    hipDeviceSynchronize();
    for (int i = 0; i < 1024; ++i) { total++; if (memcmp(&o16[i], &o32[i], 4)) { if (diff < 5) printf("trial %d elem %d: %.9g vs %.9g\n", trial, i, o16[i], o32[i]); diff++; } }
  }
  printf("%ld of %ld results differ between the 16x16x32 and the 32x32x16 form\n", diff, total);
  return 0;
}
