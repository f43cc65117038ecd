// Dev probe (not product): pins the lane<->element maps of the gfx950 int8 MFMAs with exact
// integer data, and measures issue rates of the integer ops the requantize epilogue is built from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__global__ void k_mfma64(const int8_t* A, const int8_t* B, int* D) {
  // hypothesis: lane l holds A[l&15][16*(l>>4)+j], B[16*(l>>4)+j][l&15]; D[4*(l>>4)+r][l&15]
  int l = threadIdx.x; int r = l & 15, g = l >> 4;
  v4i a, b, c = {0,0,0,0};
  int8_t ta[16], tb[16];
  for (int j = 0; j < 16; ++j) { ta[j] = A[r*64 + 16*g + j]; tb[j] = B[(16*g + j)*16 + r]; }
  memcpy(&a, ta, 16); memcpy(&b, tb, 16);
  c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4*g + i)*16 + r] = c[i];
}
__global__ void k_mfma32(const int8_t* A, const int8_t* B, int* D) {
  int l = threadIdx.x; int r = l & 15, g = l >> 4;
  long a, b; v4i c = {0,0,0,0};
  int8_t ta[8], tb[8];
  for (int j = 0; j < 8; ++j) { ta[j] = A[r*32 + 8*g + j]; tb[j] = B[(8*g + j)*16 + r]; }
  memcpy(&a, ta, 8); memcpy(&b, tb, 8);
  c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4*g + i)*16 + r] = c[i];
}
__global__ void k_mfma32x32(const int8_t* A, const int8_t* B, int* D) {
  // 32x32x32 i8: hypothesis lane l: row/col = l&31, k = 16*(l>>5)+j ; D: col=l&31,row=(reg&3)+8*(reg>>2)+4*(l>>5)
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  v4i a, b; v16i c; for (int i=0;i<16;++i) c[i]=0;
  int8_t ta[16], tb[16];
  for (int j = 0; j < 16; ++j) { ta[j] = A[r*32 + 16*h + j]; tb[j] = B[(16*h + j)*32 + r]; }
  memcpy(&a, ta, 16); memcpy(&b, tb, 16);
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[((i&3) + 8*(i>>2) + 4*h)*32 + r] = c[i];
}

// ---------------- rate kernels: each thread runs ITERS x 8 independent chains -------------
#define ITERS 2048
template<int OP> __global__ void __launch_bounds__(256) k_rate(int* out, int seed) {
  int x0 = threadIdx.x + seed, x1 = x0*3+1, x2 = x0*5+2, x3 = x0*7+3, x4=x0*11+4, x5=x0*13+5, x6=x0*17+6, x7=x0*19+7;
  int m = 0x5a3c1e77 + seed;
  long long q0=x0,q1=x1,q2=x2,q3=x3;
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (OP == 0) { // v_mad_i64_i32 + alignbit (SRDHM core)
      #define S(x) { long long p; asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(p) : "v"(x), "v"(m), "v"(q0) : "vcc"); x = (int)(p >> 31); }
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 1) { // v_mul_hi_i32
      #define S(x) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 2) { // v_mul_lo_u32
      #define S(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 3) { // v_dot4_i32_i8
      #define S(x) asm volatile("v_dot4_i32_i8 %0, %0, %1, %0" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 4) { // v_add3_u32
      #define S(x) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 5) { // v_mad_i32_i24
      #define S(x) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 6) { // v_mul_hi_i32_i24
      #define S(x) asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 7) { // v_med3_i32
      #define S(x) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(seed));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 8) { // v_perm_b32
      #define S(x) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(seed));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 9) { // v_fma_f32 (baseline full rate)
      #define S(x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 10) { // v_cvt_f32_i32 + v_cvt_i32_f32 pair
      #define S(x) asm volatile("v_cvt_f32_i32 %0, %0\n\tv_cvt_i32_f32 %0, %0" : "+v"(x));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 11) { // v_ashrrev_i32
      #define S(x) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(x) : "v"(seed));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 12) { // v_mad_u64_u32
      #define S(x) { long long p; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(p) : "v"(x), "v"(m), "v"(q0) : "vcc"); x = (int)(p >> 31); }
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 13) { // v_pk_mad_i16
      #define S(x) asm volatile("v_pk_mad_i16 %0, %0, %1, %0" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 14) { // v_pk_max_i16
      #define S(x) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x) : "v"(m));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    } else if constexpr (OP == 15) { // v_mul_f32 x2 + v_rndne
      #define S(x) asm volatile("v_rndne_f32 %0, %0" : "+v"(x));
      S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
      #undef S
    }
  }
  out[blockIdx.x*256 + threadIdx.x] = x0^x1^x2^x3^x4^x5^x6^x7^(int)q1^(int)q2^(int)q3;
}

template<int OP> __global__ void __launch_bounds__(256) k_mfma_rate(int* out, int seed) {
  v4i a = {seed, seed+1, seed+2, seed+3}, b = {seed*3, seed*5, seed*7, seed*9};
  v4i c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  v16i d0, d1; for (int i=0;i<16;++i){d0[i]=0;d1[i]=0;}
  long la = ((long)seed<<32)|seed, lb = la*3;
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (OP == 0) {
      c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0,0,0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0,0,0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0,0,0);
      c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0,0,0);
    } else if constexpr (OP == 1) {
      c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c0, 0,0,0);
      c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c1, 0,0,0);
      c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c2, 0,0,0);
      c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(la, lb, c3, 0,0,0);
    } else {
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0,0,0);
      d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0,0,0);
      d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d0, 0,0,0);
      d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d1, 0,0,0);
    }
  }
  int s = 0; for (int i=0;i<4;++i) s ^= c0[i]^c1[i]^c2[i]^c3[i]; for (int i=0;i<16;++i) s ^= d0[i]^d1[i];
  out[blockIdx.x*256 + threadIdx.x] = s;
}

// LDS byte-LUT lookup rate: each lane does ITERS*8 dependent-free ds_read_u8 at pseudo-random idx
__global__ void __launch_bounds__(256) k_lds_lut(int* out, int seed) {
  __shared__ unsigned char lut[256*4];
  for (int i = threadIdx.x; i < 1024; i += 256) lut[i] = (unsigned char)(i*37+seed);
  __syncthreads();
  unsigned x = threadIdx.x*2654435761u + seed; int acc = 0;
  for (int it = 0; it < ITERS; ++it) {
    #pragma unroll
    for (int u = 0; u < 8; ++u) { x = x*1664525u + 1013904223u; acc += lut[(x >> 24)]; }
  }
  out[blockIdx.x*256 + threadIdx.x] = acc;
}

template<class F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i=0;i<5;++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms/5;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz LDS/block %zu\n", p.gcnArchName, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  srand(1);
  { // 16x16x64
    std::vector<int8_t> A(16*64), B(64*16); for (auto& v: A) v = rand()%255-127; for (auto& v: B) v = rand()%255-127;
    std::vector<int> ref(256,0); for (int i=0;i<16;++i) for (int j=0;j<16;++j) { int s=0; for (int k=0;k<64;++k) s += A[i*64+k]*B[k*16+j]; ref[i*16+j]=s; }
    int8_t *dA,*dB; int* dD; CK(hipMalloc(&dA,A.size())); CK(hipMalloc(&dB,B.size())); CK(hipMalloc(&dD,1024));
    CK(hipMemcpy(dA,A.data(),A.size(),hipMemcpyHostToDevice)); CK(hipMemcpy(dB,B.data(),B.size(),hipMemcpyHostToDevice));
    k_mfma64<<<1,64>>>(dA,dB,dD); std::vector<int> D(256); CK(hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost));
    int bad=0; for (int i=0;i<256;++i) bad += D[i]!=ref[i]; printf("mfma_i32_16x16x64_i8 layout hypothesis: %s (%d mismatches)\n", bad?"FAIL":"PASS", bad);
  }
  { // 16x16x32
    std::vector<int8_t> A(16*32), B(32*16); for (auto& v: A) v = rand()%255-127; for (auto& v: B) v = rand()%255-127;
    std::vector<int> ref(256,0); for (int i=0;i<16;++i) for (int j=0;j<16;++j) { int s=0; for (int k=0;k<32;++k) s += A[i*32+k]*B[k*16+j]; ref[i*16+j]=s; }
    int8_t *dA,*dB; int* dD; CK(hipMalloc(&dA,A.size())); CK(hipMalloc(&dB,B.size())); CK(hipMalloc(&dD,1024));
    CK(hipMemcpy(dA,A.data(),A.size(),hipMemcpyHostToDevice)); CK(hipMemcpy(dB,B.data(),B.size(),hipMemcpyHostToDevice));
    k_mfma32<<<1,64>>>(dA,dB,dD); std::vector<int> D(256); CK(hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost));
    int bad=0; for (int i=0;i<256;++i) bad += D[i]!=ref[i]; printf("mfma_i32_16x16x32_i8 layout hypothesis: %s (%d mismatches)\n", bad?"FAIL":"PASS", bad);
  }
  { // 32x32x32
    std::vector<int8_t> A(32*32), B(32*32); for (auto& v: A) v = rand()%255-127; for (auto& v: B) v = rand()%255-127;
    std::vector<int> ref(1024,0); for (int i=0;i<32;++i) for (int j=0;j<32;++j) { int s=0; for (int k=0;k<32;++k) s += A[i*32+k]*B[k*32+j]; ref[i*32+j]=s; }
    int8_t *dA,*dB; int* dD; CK(hipMalloc(&dA,A.size())); CK(hipMalloc(&dB,B.size())); CK(hipMalloc(&dD,4096));
    CK(hipMemcpy(dA,A.data(),A.size(),hipMemcpyHostToDevice)); CK(hipMemcpy(dB,B.data(),B.size(),hipMemcpyHostToDevice));
    k_mfma32x32<<<1,64>>>(dA,dB,dD); std::vector<int> D(1024); CK(hipMemcpy(D.data(),dD,4096,hipMemcpyDeviceToHost));
    int bad=0; for (int i=0;i<1024;++i) bad += D[i]!=ref[i]; printf("mfma_i32_32x32x32_i8 layout hypothesis: %s (%d mismatches)\n", bad?"FAIL":"PASS", bad);
  }
  int nblk = p.multiProcessorCount * 8; int* out; CK(hipMalloc(&out, (size_t)nblk*256*4));
  double simds = p.multiProcessorCount * 4.0; double clk = 2.4e9;
  auto report = [&](const char* name, float ms, double instr_per_thread) {
    double wave_instr = (double)nblk * 4 * instr_per_thread; // 4 waves per block
    double per_simd_per_s = wave_instr / simds / (ms*1e-3);
    printf("%-28s %8.3f ms  -> %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, ms, clk / per_simd_per_s);
  };
  const char* names[] = {"v_mad_i64_i32(+shift)","v_mul_hi_i32","v_mul_lo_u32","v_dot4_i32_i8","v_add3_u32","v_mad_i32_i24","v_mul_hi_i32_i24","v_med3_i32","v_perm_b32","v_fma_f32","cvt_f32_i32+cvt_i32_f32(2)","v_ashrrev_i32","v_mad_u64_u32(+shift)","v_pk_mad_i16","v_pk_max_i16","v_rndne_f32"};
  float ms;
  #define R(OP) ms = timeit([&]{ k_rate<OP><<<nblk,256>>>(out, 3); }); report(names[OP], ms, (double)ITERS*8);
  R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
  ms = timeit([&]{ k_mfma_rate<0><<<nblk,256>>>(out, 3); }); report("mfma_i32_16x16x64_i8", ms, (double)ITERS*4);
  ms = timeit([&]{ k_mfma_rate<1><<<nblk,256>>>(out, 3); }); report("mfma_i32_16x16x32_i8", ms, (double)ITERS*4);
  ms = timeit([&]{ k_mfma_rate<2><<<nblk,256>>>(out, 3); }); report("mfma_i32_32x32x32_i8", ms, (double)ITERS*4);
  ms = timeit([&]{ k_lds_lut<<<nblk,256>>>(out, 3); }); report("ds_read_u8 LUT (+lcg 3 valu)", ms, (double)ITERS*8);
  return 0;
}
