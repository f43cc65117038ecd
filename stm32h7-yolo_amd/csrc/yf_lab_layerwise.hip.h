// LABORATORY ONLY (-DYF_LAB, make lab): the layer-by-layer 160x160 form (27 launches over an HBM arena), the plain statement the banded kernels are debugged against; and the direct 4x4 form of pool_25 it uses.
// Part of yf_kernels.hip.h (included from inside namespace YF_NS; not a stand-alone header).

// pool_25: 4x4 stride 2 pad 1 on T15 (14x14x24) -> QUANTIZE#45 -> pool half of concat_46
template <int F, int NT, class T15 = B_T15, class T30 = B_T30>
YF_STAGE_FN void pool25(char* frames, int tid) {
  constexpr int PP = T30::P, OW = T30::W, LIM = T15::W - 1;
  static_assert(T15::FS == T30::FS, "one frame stride per stage");
  for (int i = tid; i < F * PP * 6; i += NT) {
    const int cg = i % 6; int t = i / 6;
    const int p = t % PP; const int f = t / PP;
    const int oy = p / OW, ox = p - oy * OW;
    char* fbase = frames + f * T15::FS;
    SplitB m;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
        m = m.mx(SplitB(lds_u32(fbase + T15::at(clampi(2 * oy - 1 + ky, 0, LIM), clampi(2 * ox - 1 + kx, 0, LIM)) + 4 * cg)));
    *reinterpret_cast<uint32_t*>(fbase + T30::at_p(p) + 4 * cg) = lut4_raw<YF_L_Q45>(m);
  }
}


// ------------------------------------------------------------------------------------------------ layer-by-layer form
// For input sizes whose activations do not fit in LDS (160x160: conv2d_6's output alone is 131 KB) the SAME stage
// functions run one kernel per fused stage over a per-frame arena in HBM (one workgroup per frame and stage, frames
// grid-strided).  Results are bit-identical to the oracle at that size; HBM traffic is no longer the algorithmic
// minimum -- fusing this variant with spatial tiles is later work.
struct GenParams {
  const int8_t* in;       // [n][G0][G0][3]
  int8_t* out;            // [n][G3][G3][18]
  long n;                 // frames in this launch (<= arena capacity)
  const uint8_t* tab;
  char* arena;            // n * FRAME_BYTES bytes of HBM scratch
};
constexpr int GEN_STAGES = 27;

template <int ST, int NW>
__global__ void __launch_bounds__(NW * 64, 2) generic_stage_kernel(const GenParams prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64, F = 1;
  uint8_t* luts = reinterpret_cast<uint8_t*>(smem);      // addressed absolutely (host-checked: no static LDS)
  char* out_all = nullptr;
  int vz = 0;
  asm volatile("" : "+v"(vz));              // a zero the compiler cannot see through: keeps the pass constants' loads vector loads
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t* __restrict__ tab = prm.tab;
  for (int i = tid; i < LUT_BYTES / 16; i += NT)
    reinterpret_cast<uint4*>(luts)[i] = reinterpret_cast<const uint4*>(tab + PLAN.lut_off)[i];
  __syncthreads();
  const AddK no_add = {};
  auto addctx = [&](int k) {
    const uint8_t* a = tab + offsetof(yf_table_index, add) + k * sizeof(yf_add);
    return AddK{uniform_u32(a + offsetof(yf_add, mo2)), uniform_u32(a + offsetof(yf_add, zro)),
                (unsigned long)uniform_u32(a + offsetof(yf_add, c64o)) | ((unsigned long)uniform_u32(a + offsetof(yf_add, c64o) + 4) << 32),
                (int)uniform_u32(a + offsetof(yf_add, rso))};
  };
  for (long fr = blockIdx.x; fr < prm.n; fr += gridDim.x) {
    char* frames = prm.arena + fr * (long)FRAME_BYTES;
    out_all = reinterpret_cast<char*>(prm.out) + fr * (long)OUT_FRAME_BYTES;
    if constexpr (ST == 0) {
      stage_input<F, NT>(frames, prm.in, fr, prm.n, (int)uniform_u32(tab + offsetof(yf_table_index, in_zp)), tid);
      fill_halo<B_T1, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW3), tid);
    } else if constexpr (ST == 1) {
      conv1_stage<F, NW>(frames, tab, load_dense(tab, YF_D_CONV1), wave, lane, vz);
    } else if constexpr (ST == 2) {
      dw_mfma_stage<F, NW, 1, B_T1, B_T2, 8, YF_L_LEAKY4>(frames, tab, load_dw(tab, YF_W_DW3), wave, lane, vz);
    } else if constexpr (ST == 3) {
      dense_stage<F, NW, 1, 1, 8, B_T2, B_T3, 0, 4, EPI_RAW, 0, B_T3>(frames, out_all, tab, load_dense(tab, YF_D_C5), no_add, wave, lane, vz);
    } else if constexpr (ST == 4) {
      fill_halo<B_T4, false, F, NT>(frames, load_halo_zp(tab, YF_W_DW10), tid);
      dense_stage<F, NW, 3, 1, 4, B_T3, B_T4, 0, 18, EPI_LUT, YF_L_LEAKY7, B_T4>(frames, out_all, tab, load_dense(tab, YF_D_C6), no_add, wave, lane, vz);
    } else if constexpr (ST == 5) {
      pool8_h<F, NT>(frames, tid);
    } else if constexpr (ST == 6) {
      pool8_v<F, NT>(frames, tid);
    } else if constexpr (ST == 7) {
      dw_mfma_stage<F, NW, 2, B_T4, B_T6, 18, YF_L_LEAKY11>(frames, tab, load_dw(tab, YF_W_DW10), wave, lane, vz);
    } else if constexpr (ST == 8) {
      dense_stage<F, NW, 1, 2, 16, B_T6, B_T7, 0, 6, EPI_RAW, 0, B_T7>(frames, out_all, tab, load_dense(tab, YF_D_C12), no_add, wave, lane, vz);
    } else if constexpr (ST == 9) {
      fill_halo<B_T8, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW15), tid);
      dense_stage<F, NW, 3, 1, 8, B_T7, B_T8, 0, 36, EPI_LUT, YF_L_LEAKY14, B_T8>(frames, out_all, tab, load_dense(tab, YF_D_C13), no_add, wave, lane, vz);
    } else if constexpr (ST == 10) {
      dw_mfma_stage<F, NW, 1, B_T8, B_T9, 36, YF_L_LEAKY16>(frames, tab, load_dw(tab, YF_W_DW15), wave, lane, vz);
    } else if constexpr (ST == 11) {
      dense_stage<F, NW, 1, 3, 16, B_T9, B_T11, 0, 6, EPI_ADD, YF_A_ADD18, B_T7>(frames, out_all, tab, load_dense(tab, YF_D_C17), addctx(YF_A_ADD18), wave, lane, vz);
    } else if constexpr (ST == 12) {
      dense_stage<F, NW, 2, 1, 8, B_T11, B_T14, YF_T14_CONV_BASE, 18, EPI_LUT, YF_L_LEAKY20, B_T14>(frames, out_all, tab, load_dense(tab, YF_D_C19), no_add, wave, lane, vz);
    } else if constexpr (ST == 13) {
      fill_halo<B_T15, false, F, NT>(frames, load_halo_zp(tab, YF_W_DW27), tid);
      dense_stage<F, NW, 2, 3, 16, B_T14, B_T15, 0, 24, EPI_LUT, YF_L_LEAKY24, B_T15>(frames, out_all, tab, load_dense(tab, YF_D_C23), no_add, wave, lane, vz);
    } else if constexpr (ST == 14) {
      pool25<F, NT>(frames, tid);
      dw_mfma_stage<F, NW, 2, B_T15, B_T17, 24, YF_L_LEAKY28>(frames, tab, load_dw(tab, YF_W_DW27), wave, lane, vz);
    } else if constexpr (ST == 15) {
      dense_stage<F, NW, 1, 2, 16, B_T17, B_T18, 0, 8, EPI_RAW, 0, B_T18>(frames, out_all, tab, load_dense(tab, YF_D_C29), no_add, wave, lane, vz);
    } else if constexpr (ST == 16) {
      fill_halo<B_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW32), tid);
      dense_stage<F, NW, 3, 1, 8, B_T18, B_T19, 0, 40, EPI_LUT, YF_L_LEAKY31, B_T19>(frames, out_all, tab, load_dense(tab, YF_D_C30), no_add, wave, lane, vz);
    } else if constexpr (ST == 17) {
      dw_mfma_stage<F, NW, 1, B_T19, B_T20, 40, YF_L_LEAKY33>(frames, tab, load_dw(tab, YF_W_DW32), wave, lane, vz);
    } else if constexpr (ST == 18) {
      dense_stage<F, NW, 1, 3, 16, B_T20, B_T22, 0, 8, EPI_ADD, YF_A_ADD35, B_T18>(frames, out_all, tab, load_dense(tab, YF_D_C34), addctx(YF_A_ADD35), wave, lane, vz);
    } else if constexpr (ST == 19) {
      fill_halo<B_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW38), tid);
      dense_stage<F, NW, 3, 1, 8, B_T22, B_T19, 0, 40, EPI_LUT, YF_L_LEAKY37, B_T19>(frames, out_all, tab, load_dense(tab, YF_D_C36), no_add, wave, lane, vz);
    } else if constexpr (ST == 20) {
      dw_mfma_stage<F, NW, 1, B_T19, B_T20, 40, YF_L_LEAKY39>(frames, tab, load_dw(tab, YF_W_DW38), wave, lane, vz);
    } else if constexpr (ST == 21) {
      dense_stage<F, NW, 1, 3, 16, B_T20, B_T26, 0, 8, EPI_ADD, YF_A_ADD41, B_T22>(frames, out_all, tab, load_dense(tab, YF_D_C40), addctx(YF_A_ADD41), wave, lane, vz);
    } else if constexpr (ST == 22) {
      dense_stage<F, NW, 2, 1, 8, B_T26, B_T30, 24, 24, EPI_LUT, YF_L_L43Q44, B_T30>(frames, out_all, tab, load_dense(tab, YF_D_C42), no_add, wave, lane, vz);
    } else if constexpr (ST == 23) {
      fill_halo<B_T19, true, F, NT>(frames, load_halo_zp(tab, YF_W_DW49), tid);
      dense_stage<F, NW, 2, 3, 16, B_T30, B_T19, 0, 40, EPI_LUT, YF_L_LEAKY48, B_T19>(frames, out_all, tab, load_dense(tab, YF_D_C47), no_add, wave, lane, vz);
    } else if constexpr (ST == 24) {
      dw_mfma_stage<F, NW, 1, B_T19, B_T20, 40, YF_L_LEAKY50>(frames, tab, load_dw(tab, YF_W_DW49), wave, lane, vz);
    } else if constexpr (ST == 25) {
      dense_stage<F, NW, 2, 3, 16, B_T20, B_T33, 0, 32, EPI_LUT, YF_L_LEAKY52, B_T33>(frames, out_all, tab, load_dense(tab, YF_D_C51), no_add, wave, lane, vz);
    } else {
      static_assert(ST == 26, "stage index");
      dense_stage<F, NW, 1, 2, 16, B_T33, B_T33, 0, 18, EPI_HEAD, 0, B_T33>(frames, out_all, tab, load_dense(tab, YF_D_C53), no_add, wave, lane, vz);
    }
  }
}

