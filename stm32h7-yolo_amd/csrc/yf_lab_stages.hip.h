// LABORATORY ONLY (-DYF_LAB, make lab): the round-2 stage forms -- constants fetched from global memory behind every stage boundary -- that the layer-by-layer 160x160 form is written in.
// Part of yf_kernels.hip.h (included from inside namespace YF_NS; not a stand-alone header).

// ==== round-2 stage forms (constants from global memory, per-job index arithmetic): what the layer-by-layer 160x160 kernels are written in
// ------------------------------------------------------------------------------------------------ epilogue store

// idx[4]: the pass's four requantised channels as unsigned bytes q + 128 (= LUT indices) of pixel p of frame f
template <int EPI, int LUT_ID, class OUT, int OUT_CH0, class ADDB>
__device__ __forceinline__ void epilogue_store(char* fbase /*frame arena*/, char* out_all, int f, int p, int chq,
                                               const int (&idx)[4], const AddK& ad) {
  if constexpr (EPI == EPI_LUT) {
    *reinterpret_cast<uint32_t*>(fbase + OUT::at_p(p) + OUT_CH0 + chq) =
        join4(lutb<LUT_ID>(idx[0]), lutb<LUT_ID>(idx[1]), lutb<LUT_ID>(idx[2]), lutb<LUT_ID>(idx[3]));
  } else if constexpr (EPI == EPI_RAW) {
    *reinterpret_cast<uint32_t*>(fbase + OUT::at_p(p) + OUT_CH0 + chq) = join4(idx[0], idx[1], idx[2], idx[3]) ^ 0x80808080u;
  } else if constexpr (EPI == EPI_ADD) {
    // tflite ADD (LUT_ID = add index): in1 = stored tensor (ADDB) -> table A, in2 = this conv's output -> table B (which
    // carries the accumulator offset), then one fused requantisation of the sum.
    typedef const __attribute__((address_space(3))) int* lds_i32_ptr;
    constexpr uint32_t LA = YF_N_LUT * 256 + LUT_ID * 2048, LB = LA + 1024;
    const uint32_t o = lds_u32(fbase + ADDB::at_p(p) + chq) ^ 0x80808080u;
    v4i sum;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      sum[j] = *(lds_i32_ptr)(uint32_t)(LA + 4 * ((o >> (8 * j)) & 255)) + *(lds_i32_ptr)(uint32_t)(LB + 4 * idx[j]);
    int r[4];
    requant4<false>(sum, v4u{ad.mo2, ad.mo2, ad.mo2, ad.mo2}, v4u{ad.zro, ad.zro, ad.zro, ad.zro},
                    v4ul{ad.c64o, ad.c64o, ad.c64o, ad.c64o}, v4i{ad.rso, ad.rso, ad.rso, ad.rso}, r);
    *reinterpret_cast<uint32_t*>(fbase + OUT::at_p(p) + OUT_CH0 + chq) = join4(r[0], r[1], r[2], r[3]) ^ 0x80808080u;
  } else {  // head: 18 channels per pixel, 2-byte aligned, staged for one coalesced copy to HBM
    static_assert(EPI == EPI_HEAD || EPI == EPI_HEAD_LDS, "epilogue kind");
    const uint32_t v = join4(idx[0], idx[1], idx[2], idx[3]) ^ 0x80808080u;
    uint16_t* dst = reinterpret_cast<uint16_t*>((EPI == EPI_HEAD ? out_all + f * OUT_FRAME_BYTES : fbase + OUT::OFF) + p * 18 + chq);
    dst[0] = (uint16_t)v;
    if (chq + 2 < 18) dst[1] = (uint16_t)(v >> 16);
  }
}

// ------------------------------------------------------------------------------------------------ dense 1x1
// Lane-private MFMA: the lane's own pixel supplies KS fragments of 16 bytes (k-steps; the last one BW = 4, 8 or 16 bytes
// wide), the A operand of k-step ks carries W[4*pass + (r&3)][16*ks ..] in row r's own slot group only, and KS MFMAs
// accumulate the 4 channels of one pass for 64 pixels.  Every lane owns ONE pixel: no lane is wasted when Cout is not a
// multiple of 16 (6, 8, 18, 24, 40), the constants of a pass are wave-uniform, and the pixel math is shared by the TPJ
// passes of a job.  MFMA count grows (KS per 4 channels) but the matrix pipe is idle anyway.
template <int F, int NW, int TPJ, int KS, int BW, class IN, class OUT, int OUT_CH0, int COUT, int EPI, int LUT_ID, class ADDB>
YF_STAGE_FN void dense_stage(char* frames, char* out_all, const uint8_t* __restrict__ tab, const yf_dense d, const AddK ad,
                             int wave, int lane, int vz) {
  constexpr int NP = (COUT + 3) / 4;                        // passes of 4 output channels
  constexpr int NCH = (NP + TPJ - 1) / TPJ;
  constexpr int P = IN::P, TOT = F * P;
  constexpr int MT = (TOT + 63) / 64;
  constexpr int JOBS = NCH * MT;
  constexpr int KROW = 16 * KS;
  static_assert(OUT::P == P || EPI == EPI_HEAD || EPI == EPI_HEAD_LDS, "1x1 conv keeps the grid");
  static_assert(IN::FS == OUT::FS && IN::FS == ADDB::FS, "one frame stride per stage");
  static_assert(IN::S >= 16 * (KS - 1) + BW && (BW == 4 || BW == 8 || BW == 16), "the pixel vector must cover all k-steps");
  const int g = lane >> 4, c = lane & 15;
  int j0, j1;
  job_range<JOBS, NW>(wave, j0, j1);
  const uint8_t* pp = tab + d.c_off;
  const bool a_on = (c >> 2) == g;
  int cur_chunk = -1;
  v4i a[TPJ][KS];
  PassV pv[TPJ];
  PassS ksr[TPJ];                           // scalar constants stay resident per chunk (one load per tile costs a wait per tile)
  for (int j = j0; j < j1; ++j) {
    const int chunk = j / MT, mt = j - chunk * MT;
    if (chunk != cur_chunk) {
      cur_chunk = chunk;
#pragma unroll
      for (int t = 0; t < TPJ; ++t) {
        const int ps = min(chunk * TPJ + t, NP - 1);
        pv[t] = load_pass_v(pp + ps * (int)sizeof(yf_pass), vz);
        ksr[t] = load_pass_s(pp + ps * (int)sizeof(yf_pass));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a[t][ks] = v4i{0, 0, 0, 0};
          if (a_on) a[t][ks] = load_wfrag(tab + d.w_off + (ps * 4 + (c & 3)) * KROW + 16 * ks, vz);
        }
      }
    }
    const int q = mt * 64 + lane;
    const int qc = min(q, TOT - 1);
    const int f = qc / P, p = qc - f * P;
    char* fbase = frames + f * IN::FS;
    v4i b[KS];
    {
      const char* src = fbase + IN::at_p(p);
      {
#pragma unroll
      for (int ks = 0; ks < KS - 1; ++ks) b[ks] = *reinterpret_cast<const v4i*>(src + 16 * ks);
      const char* last = src + 16 * (KS - 1);
      if constexpr (BW == 16) b[KS - 1] = *reinterpret_cast<const v4i*>(last);
      else if constexpr (BW == 8) { const int2 t2 = *reinterpret_cast<const int2*>(last); b[KS - 1] = v4i{t2.x, t2.y, any_value(), any_value()}; }
      else b[KS - 1] = v4i{*reinterpret_cast<const int*>(last), any_value(), any_value(), any_value()};
      }
    }
#pragma unroll
    for (int t = 0; t < TPJ; ++t) {
      const int ps = chunk * TPJ + t;
      if (ps < NP) {                                          // uniform
        const PassS k = ksr[t];
        v4i acc = {ACC0, ACC0, ACC0, ACC0};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[t][ks], b[ks], acc, 0, 0, 0);
        int idx[4];                         // no exec mask: surplus lanes redo pixel TOT-1 (same value, same address)
        requant4<true>(acc, pv[t].m2, pv[t].zr, k.c64, k.rs, idx);
        epilogue_store<EPI, LUT_ID, OUT, OUT_CH0, ADDB>(fbase, out_all, f, p, ps * 4, idx, ad);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ conv2d_1
// 3x3 stride 2, Cin 3 -> 8 on RGBX dwords, lane-private like the 1x1 stages: the lane's pixel gathers its nine taps
// (nine aligned dwords of the staged frame) into three k-steps, both 4-channel passes share them.
template <int F, int NW, class IN = B_IN, class OUT = B_T1>
YF_STAGE_FN void conv1_stage(char* frames, const uint8_t* __restrict__ tab, const yf_dense d, int wave, int lane, int vz) {
  constexpr int P = OUT::P, W1 = OUT::W, RSW = IN::RS, TOT = F * P;
  constexpr int MT = (TOT + 63) / 64;
  const int g = lane >> 4, c = lane & 15;
  const bool a_on = (c >> 2) == g;
  const uint8_t* pp = tab + d.c_off;
  v4i a[2][3];
  PassV pv[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    pv[ps] = load_pass_v(pp + ps * (int)sizeof(yf_pass), vz);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      a[ps][ks] = v4i{0, 0, 0, 0};
      if (a_on) a[ps][ks] = load_wfrag(tab + d.w_off + (ps * 4 + (c & 3)) * YF_CONV1_KROW + 16 * ks, vz);
    }
  }
  int j0, j1;
  job_range<MT, NW>(wave, j0, j1);
  const AddK ad = {};
  for (int mt = j0; mt < j1; ++mt) {
    const int q = mt * 64 + lane;
    const int qc = min(q, TOT - 1);
    const int f = qc / P, p = qc - f * P;
    const int oy = p / W1, ox = p - oy * W1;
    char* fbase = frames + f * IN::FS;
    // tap (ky,kx) of output (oy,ox) = IN[2oy-1+ky][2ox-1+kx] = halo'd dword (2oy+ky)*RSW + 2ox+kx+3
    const uint32_t* src = reinterpret_cast<const uint32_t*>(fbase + IN::OFF) + (2 * oy * RSW + 2 * ox + 3);
    const v4i b0 = {(int)src[0], (int)src[1], (int)src[2], (int)src[RSW]};
    const v4i b1 = {(int)src[RSW + 1], (int)src[RSW + 2], (int)src[2 * RSW], (int)src[2 * RSW + 1]};
    const v4i b2 = {(int)src[2 * RSW + 2], any_value(), any_value(), any_value()};
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const PassS k = load_pass_s(pp + ps * (int)sizeof(yf_pass));
      v4i acc = {ACC0, ACC0, ACC0, ACC0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][0], b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][1], b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ps][2], b2, acc, 0, 0, 0);
      int idx[4];                           // no exec mask (surplus lanes redo pixel TOT-1)
      requant4<true>(acc, pv[ps].m2, pv[ps].zr, k.c64, k.rs, idx);
      epilogue_store<EPI_LUT, YF_L_LEAKY2, OUT, 0, OUT>(fbase, nullptr, f, p, 4 * ps, idx, ad);
    }
  }
}

// ------------------------------------------------------------------------------------------------ depthwise on MFMA
// Lane-private one-hot packing.  The 64 k-slots of v_mfma_i32_16x16x64_i8 are supplied by four lane groups of 16
// slots each; rows 4g..4g+3 of the A operand are non-zero only in group g's slots.  D[4g+j][c] is then a 16-long dot
// product over data that lane (g,c) itself supplied -- 64 independent pixels per MFMA, each lane working on ITS OWN
// pixel.  One k-step carries 4 taps x 4 channels (4 aligned dwords of the pixel's halo'd neighbourhood), so the
// 9 taps of a 3x3 depthwise filter take 3 k-steps; A holds w[tap][channel j] at byte j of tap's dword in row 4g+j.
// Per 64 pixels x 4 channels: 9 ds_read_b32 off one address register, 3 MFMAs, no VALU multiply at all.
// IN has a halo holding its zero point; the zero point itself is folded into the requantisation constant.
// A job = 4 output rows x 16 columns (2 frames side by side for the 7x7 grids); border blocks are shifted inwards so
// every lane's neighbourhood address is in range.
template <int F, int NW, int STRIDE, class IN, class OUT, int C, int LUT_ID>
YF_STAGE_FN void dw_mfma_stage(char* frames, const uint8_t* __restrict__ tab, const yf_dw d, int wave, int lane, int vz) {
  constexpr int W = OUT::W, H = OUT::H;
  constexpr int FL = (W <= 8 && F % 2 == 0) ? 2 : 1;       // frames side by side in the 16 lanes of a row tile
  constexpr int NSEG = (W + 15) / 16;                       // 16-column segments, the last one shifted left (28 -> x0 in {0, 12})
  constexpr int NRB = (H + 3) / 4;                          // 4-row blocks (last one shifted up)
  constexpr int NG = (C + 3) / 4;
  constexpr int NFP = F / FL;
  constexpr int JPG = NFP * NRB * NSEG;                     // jobs per channel group
  constexpr int JOBS = NG * JPG;
  constexpr int DROW = STRIDE * IN::ROWB;                   // input bytes between consecutive output rows
  constexpr int TS = IN::S, TR = IN::ROWB;                  // tap strides: +1 column, +1 row
  static_assert(OUT::RS == W && OUT::PT == 0 && OUT::PL == 0, "depthwise outputs are plain buffers");
  static_assert(IN::FS == OUT::FS, "one frame stride per stage");
  static_assert(H >= 4 && (W >= 16 || W * FL <= 16), "tile shape");
  const int g = lane >> 4, c = lane & 15;
  const int fl = (FL == 2) ? (c >> 3) : 0;
  const int xl = (FL == 2) ? min(c & 7, W - 1) : min(c, W - 1);      // surplus lanes duplicate the last column (idempotent)
  const int lane_in = fl * IN::FS + g * DROW + xl * STRIDE * IN::S;      // this lane's pixel: row oy0+g, col x0+xl
  const int lane_out = fl * IN::FS + (g * W + xl) * OUT::S;
  const bool a_on = (c >> 2) == g;                          // A row r = c belongs to row block r>>2
  int j, j1;
  job_range<JOBS, NW>(wave, j, j1);
  while (j < j1) {
    const int cg = j / JPG;
    const int jend = min(j1, (cg + 1) * JPG);
    const uint8_t* grp = tab + d.g_off + cg * YF_DW_GROUP_BYTES;
    const uint32_t* wg = reinterpret_cast<const uint32_t*>(grp);
    v4i a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;                // k-steps: taps 0-3, 4-7, 8
    if (a_on) {
      const uint32_t* wl = wg + (c & 3);                    // masked weight dwords of channel c&3: wl[4*tap]
      a0 = v4i{(int)wl[0], (int)wl[4], (int)wl[8], (int)wl[12]};
      a1 = v4i{(int)wl[16], (int)wl[20], (int)wl[24], (int)wl[28]};
      a2[0] = (int)wl[32];
    }
    const PassV pv = load_pass_v(grp + 144, vz);
    const PassS k = load_pass_s(grp + 144);
    // one job: 9 tap dwords -> 3 MFMAs -> requantise -> LUT -> packed store
    auto taps = [&](int jj, v4i& b0, v4i& b1, v4i& b2, char*& dst) {
      int rem = jj - cg * JPG;
      const int fp = rem / (NRB * NSEG); rem -= fp * (NRB * NSEG);
      const int rb = rem / NSEG, seg = rem - rb * NSEG;
      const int oy0 = min(rb * 4, H - 4);
      const int x0 = (W >= 16) ? min(seg * 16, W - 16) : 0;
      char* fb = frames + fp * FL * IN::FS;
      const char* src = fb + IN::OFF + (oy0 * STRIDE) * IN::ROWB + x0 * STRIDE * IN::S + 4 * cg + lane_in;
      b0[0] = (int)lds_u32(src);               b0[1] = (int)lds_u32(src + TS);          b0[2] = (int)lds_u32(src + 2 * TS);
      b0[3] = (int)lds_u32(src + TR);          b1[0] = (int)lds_u32(src + TR + TS);     b1[1] = (int)lds_u32(src + TR + 2 * TS);
      b1[2] = (int)lds_u32(src + 2 * TR);      b1[3] = (int)lds_u32(src + 2 * TR + TS); b2[0] = (int)lds_u32(src + 2 * TR + 2 * TS);
      dst = fb + OUT::OFF + (oy0 * W + x0) * OUT::S + lane_out + 4 * cg;
    };
    auto conv = [&](const v4i& b0, const v4i& b1, const v4i& b2) {
      v4i acc = {ACC0, ACC0, ACC0, ACC0};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc, 0, 0, 0);
      return __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b2, acc, 0, 0, 0);
    };
    auto finish = [&](const v4i& acc, char* dst) {
      int idx[4];
      requant4<true>(acc, pv.m2, pv.zr, k.c64, k.rs, idx);
      *reinterpret_cast<uint32_t*>(dst) = join4(lutb<LUT_ID>(idx[0]), lutb<LUT_ID>(idx[1]), lutb<LUT_ID>(idx[2]), lutb<LUT_ID>(idx[3]));
    };
    // two jobs in flight per iteration: the second job's tap reads and MFMAs overlap the first one's epilogue chain
    for (; j + 1 < jend; j += 2) {
      v4i p0, p1, p2 = {0, any_value(), any_value(), any_value()}, q0, q1, q2 = {0, any_value(), any_value(), any_value()};
      char *dp, *dq;
      taps(j, p0, p1, p2, dp);
      taps(j + 1, q0, q1, q2, dq);
      const v4i ap = conv(p0, p1, p2);
      const v4i aq = conv(q0, q1, q2);
      finish(ap, dp);
      finish(aq, dq);
    }
    for (; j < jend; ++j) {
      v4i b0, b1, b2 = {0, any_value(), any_value(), any_value()};
      char* dst;
      taps(j, b0, b1, b2, dst);
      finish(conv(b0, b1, b2), dst);
    }
  }
}

