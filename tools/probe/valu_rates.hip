// Issue-rate probe for the VALU ops the requantisation epilogue could be built from (gfx950).  DEV TOOL.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/valu_rates tools/probe/valu_rates.hip && ./tools/probe/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define ITERS 2048
#define OPS(X) \
  X(0,  "v_add_u32 %0, %0, %1") \
  X(1,  "v_and_b32 %0, %0, %1") \
  X(2,  "v_or_b32 %0, %0, %1") \
  X(3,  "v_xor_b32 %0, %0, %1") \
  X(4,  "v_lshlrev_b32 %0, 3, %0") \
  X(5,  "v_ashrrev_i32 %0, 3, %0") \
  X(6,  "v_max_i32 %0, %0, %1") \
  X(7,  "v_min_i32 %0, %0, %1") \
  X(8,  "v_mov_b32 %0, %1") \
  X(9,  "v_lshl_or_b32 %0, %0, 8, %1") \
  X(10, "v_or3_b32 %0, %0, %1, %1") \
  X(11, "v_lshl_add_u32 %0, %0, 2, %1") \
  X(12, "v_and_or_b32 %0, %0, %1, %1") \
  X(13, "v_med3_i32 %0, %0, %1, %1") \
  X(14, "v_alignbit_b32 %0, %0, %1, 31") \
  X(15, "v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3") \
  X(16, "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0") \
  X(17, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") \
  X(18, "v_mul_i32_i24 %0, %0, %1") \
  X(19, "v_mul_u32_u24 %0, %0, %1") \
  X(20, "v_mad_u32_u24 %0, %0, %1, %1") \
  X(21, "v_bfe_i32 %0, %0, 3, 8") \
  X(22, "v_sub_u32 %0, %0, %1") \
  X(23, "v_cvt_f32_i32 %0, %0") \
  X(24, "v_cvt_i32_f32 %0, %0") \
  X(25, "v_fma_f32 %0, %0, %1, %1") \
  X(26, "v_mul_f32 %0, %0, %1") \
  X(27, "v_add_f32 %0, %0, %1") \
  X(28, "v_pk_add_i16 %0, %0, %1") \
  X(29, "v_pk_max_i16 %0, %0, %1") \
  X(30, "v_pk_ashrrev_i16 %0, 3, %0") \
  X(31, "v_sat_pk_u8_i16 %0, %0") \
  X(32, "v_cvt_pk_u8_f32 %0, %0, 1, %1") \
  X(33, "v_fmac_f32 %0, %0, %1") \
  X(34, "v_cndmask_b32 %0, %0, %1, vcc") \
  X(35, "v_mul_hi_u32 %0, %0, %1") \
  X(36, "v_mul_lo_u32 %0, %0, %1") \
  X(37, "v_mad_i64_i32 %0, vcc, %1, %1, %0") \
  X(38, "v_sad_u32 %0, %0, %1, %1") \
  X(39, "v_add3_u32 %0, %0, %1, %1") \
  X(40, "v_xad_u32 %0, %0, %1, %1") \
  X(41, "v_add_lshl_u32 %0, %0, %1, 2") \
  X(42, "v_perm_b32 %0, %0, %1, %1") \
  X(43, "v_bfi_b32 %0, %0, %1, %1") \
  X(44, "v_pk_mul_lo_u16 %0, %0, %1") \
  X(45, "v_pk_mad_i16 %0, %0, %1, %1") \
  X(46, "v_mad_i32_i16 %0, %0, %1, %1") \
  X(47, "v_dot2_i32_i16 %0, %0, %1, %0") \
  X(48, "v_dot4_i32_i8 %0, %0, %1, %0") \
  X(49, "v_cvt_pk_i16_i32 %0, %0, %1") \
  X(50, "v_pk_min_i16 %0, %0, %1") \
  X(51, "v_max3_i32 %0, %0, %1, %1") \
  X(52, "v_cvt_f32_ubyte0 %0, %0") \
  X(53, "v_add_f64 %0, %0, %0") \
  X(54, "v_fma_f64 %0, %0, %0, %0") \
  X(55, "v_mul_hi_i32 %0, %0, %1") \
  X(56, "v_pk_lshlrev_b16 %0, 8, %0 op_sel_hi:[0,1]") \
  X(57, "v_addc_co_u32_e64 %0, vcc, %0, %1, vcc") \
  X(58, "v_pk_fma_f32 %0, %0, %0, %0") \
  X(59, "v_pk_mul_f32 %0, %0, %0") \
  X(60, "v_mad_u64_u32 %0, vcc, %1, %1, %0") \
  X(61, "v_lshrrev_b32 %0, 3, %0") \
  X(62, "v_pk_lshrrev_b16 %0, 8, %0 op_sel_hi:[0,1]") \
  X(63, "v_and_b32 %0, 0xff00ff00, %0") \
  X(64, "v_bfe_u32 %0, %0, 8, 8") \
  X(65, "v_max_i32 %0, 0, %0") \
  X(66, "v_min_u32 %0, %0, %1")

template <int OP> __global__ void __launch_bounds__(256) k_rate(int* out, int seed) {
  int x0 = threadIdx.x + seed, x1 = x0 * 3 + 1, x2 = x0 * 5 + 2, x3 = x0 * 7 + 3, x4 = x0 * 11 + 4, x5 = x0 * 13 + 5, x6 = x0 * 17 + 6, x7 = x0 * 19 + 7;
  int m = 0x5a3c1e77 + seed;
  long long q0 = x0, q1 = x1, q2 = x2, q3 = x3, q4 = x4, q5 = x5, q6 = x6, q7 = x7;
  for (int it = 0; it < ITERS; ++it) {
#define X(ID, STR) if constexpr (OP == ID) { \
    if constexpr (ID == 53 || ID == 54 || ID == 58 || ID == 59) { \
      asm volatile(STR : "+v"(q0)); asm volatile(STR : "+v"(q1)); asm volatile(STR : "+v"(q2)); asm volatile(STR : "+v"(q3)); \
      asm volatile(STR : "+v"(q4)); asm volatile(STR : "+v"(q5)); asm volatile(STR : "+v"(q6)); asm volatile(STR : "+v"(q7)); \
    } else if constexpr (ID == 37 || ID == 60) { \
      asm volatile(STR : "+v"(q0) : "v"(m) : "vcc"); asm volatile(STR : "+v"(q1) : "v"(m) : "vcc"); \
      asm volatile(STR : "+v"(q2) : "v"(m) : "vcc"); asm volatile(STR : "+v"(q3) : "v"(m) : "vcc"); \
      asm volatile(STR : "+v"(q4) : "v"(m) : "vcc"); asm volatile(STR : "+v"(q5) : "v"(m) : "vcc"); \
      asm volatile(STR : "+v"(q6) : "v"(m) : "vcc"); asm volatile(STR : "+v"(q7) : "v"(m) : "vcc"); \
    } else { \
      asm volatile(STR : "+v"(x0) : "v"(m) : "vcc"); asm volatile(STR : "+v"(x1) : "v"(m) : "vcc"); \
      asm volatile(STR : "+v"(x2) : "v"(m) : "vcc"); asm volatile(STR : "+v"(x3) : "v"(m) : "vcc"); \
      asm volatile(STR : "+v"(x4) : "v"(m) : "vcc"); asm volatile(STR : "+v"(x5) : "v"(m) : "vcc"); \
      asm volatile(STR : "+v"(x6) : "v"(m) : "vcc"); asm volatile(STR : "+v"(x7) : "v"(m) : "vcc"); \
    } }
    OPS(X)
#undef X
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ (int)q0 ^ (int)q1 ^ (int)q2 ^ (int)q3 ^ (int)q4 ^ (int)q5 ^ (int)q6 ^ (int)q7;
}

// empty loop of the same shape (loop overhead)
__global__ void __launch_bounds__(256) k_empty(int* out, int seed) {
  int x0 = threadIdx.x + seed;
  for (int it = 0; it < ITERS; ++it) asm volatile("" : "+v"(x0));
  out[blockIdx.x * 256 + threadIdx.x] = x0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int nblk = p.multiProcessorCount * 8;   // 8 blocks x 4 waves per CU = 8 waves per SIMD
  int* out; CK(hipMalloc(&out, (size_t)nblk * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double clk = p.clockRate * 1e3, simds = p.multiProcessorCount * 4.0;
  printf("device %s CUs %d clock %.0f MHz; cycles are per wave64 instruction per SIMD, 8 waves/SIMD resident\n", p.gcnArchName, p.multiProcessorCount, clk / 1e6);
#define X(ID, STR) { \
    k_rate<ID><<<nblk, 256>>>(out, 3); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) k_rate<ID><<<nblk, 256>>>(out, 3); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5; \
    const double wi = (double)nblk * 4 * ITERS * 8; \
    printf("%-100s %6.2f cyc\n", STR, clk / (wi / simds / (ms * 1e-3))); }
  OPS(X)
#undef X
  return 0;
}
