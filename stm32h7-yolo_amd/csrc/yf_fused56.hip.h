// The fused 56x56 kernel yoloface56_fused<F, NW>: one persistent workgroup walks groups of F frames through all 31 layers in LDS (stage forms: yf_kernels.hip.h, namespace v2).
// Part of yf_kernels.hip.h (included from inside namespace YF_NS; not a stand-alone header).


// ------------------------------------------------------------------------------------------------ debug dump
// Observer-style per-stage dump (reference observer API, ai_platform_interface.h:684-731): logical NHWC bytes.
template <class B, int C, int F, int NT>
__device__ __forceinline__ void dump_buf(const char* frames, int8_t* dump, long stride, long off, long first_frame,
                                         long n_frames, int tid, int ch0 = 0, int split = 1 << 30, int gap = 0) {
  if (!dump) return;
  for (int i = tid; i < F * B::P * C; i += NT) {
    const int ch = i % C; const int t = i / C;
    const int p = t % B::P; const int f = t / B::P;
    if (first_frame + f >= n_frames) continue;
    const int phys = ch0 + ch + (ch >= split ? gap : 0);
    dump[(first_frame + f) * stride + off + (long)p * C + ch] = (int8_t)frames[f * B::FS + B::at_p(p) + phys];
  }
}

// the tail sets of a batched tail (tail batching: set f belongs to frame id_of(f), -1 = none) -- laboratory dump build in the PRODUCTION stage order only
template <class B, int C, int FT, int NT, class IdOf>
__device__ __forceinline__ void dump_sets(const char* frames, int8_t* dump, long stride, long off, const IdOf& id_of, int tid, int ch0 = 0, int split = 1 << 30, int gap = 0) {
  for (int i = tid; i < FT * B::P * C; i += NT) {
    const int ch = i % C; const int t = i / C;
    const int p = t % B::P; const int f = t / B::P;
    const long id = id_of(f);
    if (id < 0) continue;
    const int phys = ch0 + ch + (ch >= split ? gap : 0);
    dump[id * stride + off + (long)p * C + ch] = (int8_t)frames[f * B::FS + B::at_p(p) + phys];
  }
}

struct DumpOffsets {   // byte offsets of each fused stage's tensor inside one frame's dump record
  enum { T1 = 0, T2 = T1 + 6272, T3 = T2 + 6272, T4 = T3 + 3136, Q21 = T4 + 14112, T6 = Q21 + 3528, T7 = T6 + 3528,
         T8 = T7 + 1176, T9 = T8 + 7056, T11 = T9 + 7056, T14 = T11 + 1176, T15 = T14 + 7056, Q45 = T15 + 4704,
         T17 = Q45 + 1176, T18 = T17 + 1176, T19 = T18 + 392, T20 = T19 + 1960, T22 = T20 + 1960, T23 = T22 + 392,
         T24 = T23 + 1960, T26 = T24 + 1960, T30 = T26 + 392, T31 = T30 + 2352, T32 = T31 + 1960, T33 = T32 + 1960,
         // tensors that the fused stages never materialise, dumped for the per-node observer (platform_abi.c): the raw max-pools (ST's
         // pool nodes carry their input's quantisation; the kernel applies QUANTIZE in the same pass) and the convolutions in front of
         // the residual adds (the add is part of their epilogue).  Debug builds park them in bytes of the concat buffers that are still
         // unwritten at that point (the conv halves) and dump them from there.
         // ... and LEAKY_RELU #43's output (production composes it with QUANTIZE #44 into one LUT)
         P8 = T33 + 1568, C17 = P8 + 3528, P25 = C17 + 1176, C34 = P25 + 1176, C40 = C34 + 392, L43 = C40 + 392,
         TOTAL = L43 + 1176 };
};

// ------------------------------------------------------------------------------------------------ the kernel
struct NetParams {
  const int8_t* in;       // [n][56][56][3] int8
  int8_t* out;            // [n][7][7][18] int8
  long n;
  const uint8_t* tab;     // device table blob (yf_host_prep.c)
  int8_t* dump;           // optional per-stage dump, [n][DumpOffsets::TOTAL]
  int stop_stage;         // debug kernel only: leave the group after this many stages (stage timing); <0 = run all
  // optional fused box decode (heads are decoded while still in LDS): dets == nullptr -> heads only
  yf_det* dets;           // [n][cap] detection records
  int* counts;            // [n] candidates per frame (may exceed cap)
  int cap, mode;          // YF_DECODE_PY / YF_DECODE_FW
  int q_thr;              // smallest quantised confidence that passes the mode's threshold (the sigmoid table is monotonic): set by the engine
  float w_scale, h_scale;
  char* scratch;          // tail batching: gridDim.x * F * TailBufs::T15_BYTES bytes (a workgroup parks one group's T15 there)
};
static_assert(sizeof(yf_table_index) <= YF_INDEX_RESERVED, "index does not fit its reserved slot");

// Issue priority per stage (s_setprio, 0..3), stage order: staging, conv2d_1, 3, 5, 6, pool_8 h, pool_8 v, conv2d_10, 12, 13, 15,
// 17, 19, 23, then the thirteen tail stages.  See the kernel: a workgroup's priority FALLS as its group advances.
#ifndef YF_POOL_MERGE
#define YF_POOL_MERGE 1          /* pool_8's two passes share stages with conv2d_10 / conv2d_13 (0: five stages of their own, the round-3 order) */
#endif
#ifndef YF_POOL8H_WAVES
#define YF_POOL8H_WAVES 3
#endif
#ifndef YF_POOL8V_WAVES
#define YF_POOL8V_WAVES 3
#endif
#ifndef YF_PRIO_LIST
#define YF_PRIO_LIST 3,3,3,3,3,3,3,3,3,3, 2,2,2,2, 1,1,1,1,1,1, 0,0,0,0,0,0,0
#endif
constexpr int STAGE_PRIO[27] = {YF_PRIO_LIST};
template <int K> __device__ __forceinline__ void stage_prio() {
  if constexpr (K == 0 || STAGE_PRIO[K] != STAGE_PRIO[K - 1]) __builtin_amdgcn_s_setprio(STAGE_PRIO[K]);
}
// Laboratory (namespace yfpd of a -DYF_LAB build, yf_engine.hip): a dump build that KEEPS the production stage order -- pools beside the branch on 3 + 5 waves,
// conv2d_10's output on concat_22's bytes, the 7x7 tail once per pair of groups on four frames through the HBM park -- so that per-stage parity is evidenced on
// the order that ships, not only on the staged order of the observer's debug build.  The six observer-only tensors are not dumped.
#if defined(YF_LAB) && defined(YF_DUMP_PROD_ORDER)
#define YF_PDUMP 1
#else
#define YF_PDUMP 0
#endif
template <bool DUMP> constexpr bool tail_batch() { return !DUMP || YF_PDUMP; }

// CAM: prm.in holds 112x112 RGB565 camera frames (25 088 B each) instead of int8 56x56x3 frames: the firmware's frame
// preparation runs inside the input staging (stage_input_cam).
template <int F, int NW, bool DUMP, bool CAM = false>
// passes per job of conv2d_13 / conv2d_23 (lab builds may override: the register-hungry settings that spill at the 128-VGPR cap)
#if !defined(YF_LAB) || !defined(YF_TPJ13)
#undef YF_TPJ13
#define YF_TPJ13 3
#endif
#if !defined(YF_LAB) || !defined(YF_TPJ23)
#undef YF_TPJ23
#define YF_TPJ23 2
#endif
#if defined(YF_LAB) && defined(YF_WHATIF_WPE)      // what-if: another register budget (waves per SIMD the kernel must fit)
#define YF_WPE(NW) YF_WHATIF_WPE
#else
#define YF_WPE(NW) ((NW) == 12 ? 6 : (NW) >= 8 ? 4 : ((NW) == 6 ? 3 : 2))
#endif
__global__ void __launch_bounds__(NW * 64, YF_WPE(NW)) yoloface56_fused(const NetParams prm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = NW * 64;
  constexpr bool BATCH = tail_batch<DUMP>();             // tail on two groups at a time (production builds)
  constexpr bool POOL_MERGE = YF_POOL_MERGE != 0 && (!DUMP || YF_PDUMP) && NW >= 8;      // (debug builds keep the staged order: their dumps and stop-stage numbers follow it)
  constexpr bool XDUMP = DUMP && !YF_PDUMP;               // the observer's extra tensors (raw pools, convolutions in front of the adds, LEAKY_RELU #43): staged-order debug build only
  typedef Buf<B_T14::OFF, 14, 14, 32, 14, 0, 0> B_T6X;                     // conv2d_10's output on concat_22's (still unwritten) bytes
  typedef typename std::conditional<POOL_MERGE, B_T6X, B_T6>::type B_T6M;
  constexpr int FT = BATCH ? 2 * F : F;                  // frames per tail run
  typedef TailBufs<BATCH ? FRAME_STRIDE / 2 : FRAME_STRIDE> U;
  constexpr int OUT_ALL_BYTES = BATCH ? 0 : (F * OUT_FRAME_BYTES + 15) & ~15;      // BATCH stages the heads inside the tail sets
  uint8_t* luts = reinterpret_cast<uint8_t*>(smem);      // LUTs are addressed absolutely: the host checks that the kernel has no static LDS
  constexpr int PRE = v2::pre_bytes<F, tail_batch<DUMP>()>();   // LUTs | depthwise job tables | zeros | two constant ring slots | halo tables
  char* out_all = smem + PRE;
  char* frames = smem + PRE + OUT_ALL_BYTES;
  long parked_first = -1;                                // first frame of the group whose T15 waits in the scratch
  const int tid0 = threadIdx.x;
  const uint8_t* __restrict__ tab = prm.tab;
  int vz = 0;
  asm volatile("" : "+v"(vz));              // a zero the compiler cannot see through: keeps the pass constants' loads vector loads
  // Issue priorities.  A static priority for the first-dispatched half of a workgroup was worth -1.9 % in round 1 and costs
  // 1.7 % with the tail on four frames.  What pays is a priority
  // LADDER over a group's stages (YF_PRIO_LIST, s_setprio before a stage whenever the level changes): 3 up to conv2d_13, 2 up
  // to conv2d_23, 1 for the first six tail stages, 0 for the rest.  The two workgroups of a CU are in different phases; the one
  // in the VALU-bound front stages then issues ahead of the one in the latency-bound tail, which only needs the slots left
  // over.  -6.8 % kernel time in-run (A/B 1.073 against no ladder); every placement of the three steps tried gave 6.0-7.3 %.
  for (int i = tid0; i < v2::LUT_B / 16; i += NT)
    reinterpret_cast<uint4*>(luts)[i] = reinterpret_cast<const uint4*>(tab + PLAN.lut_off)[i];
  for (int i = tid0; i < v2::ZERO_B / 16; i += NT) reinterpret_cast<uint4*>(smem + v2::ZERO)[i] = uint4{0, 0, 0, 0};
  constexpr int DBG_LUT = PRE + OUT_ALL_BYTES + FRAME_BYTES + (F - 1) * FRAME_STRIDE;      // debug builds: LEAKY_RELU #43 alone, behind the frame arenas
  if constexpr (DUMP) { if (tid0 < YF_DBG_LUT_BYTES / 16) reinterpret_cast<uint4*>(smem + DBG_LUT)[tid0] = reinterpret_cast<const uint4*>(tab + PLAN.lut_off + YF_N_LUT * 256 + YF_ADDLUT_BYTES)[tid0]; }
#if !(defined(YF_LAB) && defined(YF_WHATIF_NO_PROLOGUE_TABLES))   // what-if (WRONG results): the ten table builds gone -- the bound for fetching host-built tables by LDS-DMA instead
  {   // job tables of the five depthwise geometries (offsets relative to the frame arenas)
    typedef v2::JobTabs<F, tail_batch<DUMP>()> JTS;
    typedef typename JTS::U UT;
    v2::fill_jobtab<F, 1, B_T1, B_T2, JTS::JT_DW3>(smem, tid0);
    v2::fill_jobtab<F, 2, B_T4, B_T6M, JTS::JT_DW10>(smem, tid0);
    v2::fill_jobtab<F, 1, B_T8, B_T9, JTS::JT_DW15>(smem, tid0);
    v2::fill_jobtab<JTS::FT, 2, typename UT::T15, typename UT::T17, JTS::JT_DW27>(smem, tid0);
    v2::fill_jobtab<JTS::FT, 1, typename UT::T19, typename UT::T20, JTS::JT_DW32>(smem, tid0);
    typedef v2::HaloTabs<F, tail_batch<DUMP>()> HTS;       // halo pixel lists of the five depthwise inputs
    v2::build_halotab<typename HTS::G1, HTS::H_T1, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G4, HTS::H_T4, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G8, HTS::H_T8, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G15, HTS::H_T15, NT>(smem, tid0);
    v2::build_halotab<typename HTS::G19, HTS::H_T19, NT>(smem, tid0);
  }
#endif

  const long n_groups = (prm.n + F - 1) / F;
  const AddK no_add = {};
  auto addctx = [&](int k) {
    const uint8_t* a = tab + offsetof(yf_table_index, add) + k * sizeof(yf_add);
    return AddK{uniform_u32(a + offsetof(yf_add, mo2)), uniform_u32(a + offsetof(yf_add, zro)),
                (unsigned long)uniform_u32(a + offsetof(yf_add, c64o)) | ((unsigned long)uniform_u32(a + offsetof(yf_add, c64o) + 4) << 32),
                (int)uniform_u32(a + offsetof(yf_add, rso))};
  };
  constexpr long DS = DumpOffsets::TOTAL;
#ifdef YF_BARPROF
  // Stage timeline (tools/barrier_profile.py): for ONE group of every workgroup (its second: steady state) each wave stores
  // the cycle counter on arrival at and on release from every __syncthreads(): [wg][wave][40][2] in prm.dump.
  bool prof_on = false;
  int bar_no = 0;
  long long* prof_out = reinterpret_cast<long long*>(prm.dump) + ((long)blockIdx.x * NW + __builtin_amdgcn_readfirstlane(tid0 >> 6)) * 80;
#define YF_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* the LDS-DMA of the next stage's constants, as in the shipped form of YF_SYNC: the stamp follows it */ \
                       if (prof_on && (tid0 & 63) == 0 && bar_no < 40) prof_out[2 * bar_no] = __builtin_readcyclecounter(); __syncthreads(); \
                       if (prof_on && (tid0 & 63) == 0 && bar_no < 40) prof_out[2 * bar_no + 1] = __builtin_readcyclecounter(); ++bar_no; } while (0)
#else
  // the barrier behind a stage also publishes the LDS-DMA of the NEXT stage's constants, which the compiler does not see
#define YF_SYNC() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#endif
  // Stage calls: constants from an LDS ring slot, fetched one stage ahead by YF_FETCH
#define YF_HALO(B, RING, FR, G, HOFF, WI, TID) \
  v2::fill_halo_t<B, typename v2::HaloTabs<F, BATCH>::G, v2::HaloTabs<F, BATCH>::HOFF>(frames, load_halo_zp(tab, WI), TID)
#define YF_FETCH(CS, WV, LN) v2::fetch_consts<CS>(tab, WV, LN)
#define YF_CONV1(WV, LN, CS) v2::conv1_2_stage<F, NW, CS>(frames, tab, WV, LN)
#define YF_DENSE(FR, TPJ, KS, BW, IN, OUT, CH0, COUT, EPI, LUT, ADDB, DI, AD, WV, LN, CS) \
  v2::dense2_stage<FR, NW, TPJ, KS, BW, IN, OUT, CH0, COUT, EPI, LUT, ADDB, CS>(frames, out_all, tab, AD, WV, LN)
#define YF_DW(FR, STRIDE, IN, OUT, C, LUT, WI, WV, LN, CS, JTOFF) v2::dw2_stage<FR, NW, STRIDE, IN, OUT, C, LUT, CS, v2::JobTabs<F, BATCH>::JTOFF>(frames, tab, WV, LN)
#define YF_DUMP(BUF, C, OFF, ...) \
  if constexpr (DUMP) { if (prm.dump) { dump_buf<BUF, C, F, NT>(frames, prm.dump, DS, DumpOffsets::OFF, first, prm.n, tid, ##__VA_ARGS__); YF_SYNC(); } }
  int stage_no = 0;
#define YF_STAGE_END() if constexpr (DUMP) { if (++stage_no == prm.stop_stage) continue; }
#define YF_PRIO(K) stage_prio<K>()

  // Fused box decode: the staged heads of group g stay in out_all until conv2d_53 of group g+1, so they are decoded by
  // the last F waves DURING conv2d_29 of the next group (a 4-job stage: those waves are idle there), off the critical
  // path; the workgroup's last group is decoded after the loop.
  long prev_first = -1;
  auto decode_prev = [&](int w, int ln) {
    if (prm.dets != nullptr && prev_first >= 0 && w >= NW - F && prev_first + (w - (NW - F)) < prm.n) {
      int dl = ln;
      asm volatile("" : "+v"(dl));              // keep the decode's per-lane index arithmetic out of the kernel-wide hoisted set
      yfdec::decode_frame(reinterpret_cast<const int8_t*>(out_all) + (w - (NW - F)) * OUT_FRAME_BYTES, prev_first + (w - (NW - F)), dl,
                          prm.mode, prm.w_scale, prm.h_scale, prm.dets, prm.counts, prm.cap);
    }
  };
  for (long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const long first = grp * F;
#ifdef YF_BARPROF
    bar_no = 0;
    prof_on = !DUMP && prm.dump != nullptr && grp == (long)blockIdx.x + gridDim.x;
#endif
    // Loop-invariant code motion hoists the per-lane index arithmetic of every stage out of this loop and parks the
    // results in VGPRs for the whole kernel.  YF_LAUNDER selects stage groups (1 front 28x28, 2 middle 14x14, 4 tail
    // 7x7) whose thread index is laundered once per group, i.e. recomputed instead of parked.
    int tid = tid0, tid_f = tid0, tid_m = tid0, tid_t = tid0;
    if constexpr ((YF_LAUNDER & 1) != 0) asm volatile("" : "+v"(tid_f));
    if constexpr ((YF_LAUNDER & 2) != 0) asm volatile("" : "+v"(tid_m));
    if constexpr ((YF_LAUNDER & 4) != 0) asm volatile("" : "+v"(tid_t));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L_f = tid_f & 63, L_m = tid_m & 63, L_t = tid_t & 63;
    const int W_f = __builtin_amdgcn_readfirstlane(tid_f >> 6), W_m = __builtin_amdgcn_readfirstlane(tid_m >> 6), W_t = __builtin_amdgcn_readfirstlane(tid_t >> 6);
    (void)lane; (void)wave;
#if !defined(YF_BARPROF)
    // previous group's arena is dead.  LDS-only barrier: __syncthreads() would also wait for the acknowledgements of the
    // previous group's head / detection / parking stores (vmcnt), which nothing in this group depends on
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    YF_SYNC();
#endif
    stage_no = 0;
    YF_PRIO(0);
    int tid_s = tid0;
    asm volatile("" : "+v"(tid_s));         // the staging offsets are cheap: recomputed per group instead of parked in VGPRs for the whole kernel
    if constexpr (CAM) stage_input_cam<F, NT>(frames, reinterpret_cast<const uint8_t*>(prm.in), first, prm.n, (int)uniform_u32(tab + offsetof(yf_table_index, in_zp)), tid_s);
    else stage_input<F, NT>(frames, prm.in, first, prm.n, (int)uniform_u32(tab + offsetof(yf_table_index, in_zp)), tid_s);
    YF_FETCH(0, W_f, L_f);                                                                            // conv2d_1's constants -> ring slot 0
    YF_HALO(B_T1, true, F, G1, H_T1, YF_W_DW3, tid_f);
    YF_SYNC();
    YF_STAGE_END()
    YF_PRIO(1);
    YF_FETCH(1, W_f, L_f);
    YF_CONV1(W_f, L_f, 0);                                                                            // conv2d_1
    YF_SYNC(); YF_DUMP(B_T1, 8, T1)
    YF_STAGE_END()
    YF_PRIO(2);
    YF_FETCH(2, W_f, L_f);
    YF_DW(F, 1, B_T1, B_T2, 8, YF_L_LEAKY4, YF_W_DW3, W_f, L_f, 1, JT_DW3);                              // conv2d_3
    YF_SYNC(); YF_DUMP(B_T2, 8, T2)
    YF_STAGE_END()
    YF_PRIO(3);
    YF_FETCH(3, W_f, L_f);
    YF_DENSE(F, 1, 1, 8, B_T2, B_T3, 0, 4, EPI_RAW, 0, B_T3, YF_D_C5, no_add, W_f, L_f, 2);              // conv2d_5
    YF_SYNC(); YF_DUMP(B_T3, 4, T3)
    YF_STAGE_END()
    YF_PRIO(4);
    YF_HALO(B_T4, false, F, G4, H_T4, YF_W_DW10, tid_f);
    YF_FETCH(4, W_f, L_f);
    YF_DENSE(F, 5, 1, 4, B_T3, B_T4, 0, 18, EPI_LUT, YF_L_LEAKY7, B_T4, YF_D_C6, no_add, W_f, L_f, 3);   // conv2d_6: all five passes per job (25 jobs per two frames instead of 50)
    YF_SYNC(); YF_DUMP(B_T4, 18, T4)
    YF_STAGE_END()
    if constexpr (POOL_MERGE) {
      // pool_8 shares two stages with the branch beside it (the form the fp16 kernel took in round 4): {horizontal pass || conv2d_10} both only read T4,
      // {vertical pass || conv2d_13} touch disjoint buffers (HB -> concat_22's pool half; T7 -> T8), conv2d_12 in between.  conv2d_10's output T6 cannot
      // sit on HB's bytes then (its slot in the staged order): it goes to concat_22's region, which nothing writes before the vertical pass.  Two barrier
      // intervals less per group than {h}, {v}, {conv2d_10}, {conv2d_12}, {conv2d_13}.
      constexpr int PH = YF_POOL8H_WAVES, PV = YF_POOL8V_WAVES;
      YF_PRIO(5);
      YF_FETCH(5, W_m, L_m);
      if (W_m < PH) pool8_h<F, PH * 64>(frames, tid_m);                                               // pool_8 (h): T4 -> HB ...
      else v2::dw2_stage<F, NW - PH, 2, B_T4, B_T6M, 18, YF_L_LEAKY11, 4, v2::JobTabs<F, BATCH>::JT_DW10>(frames, tab, W_m - PH, L_m);   // ... beside conv2d_10: T4 -> T6
      YF_SYNC(); YF_DUMP(B_T6M, 18, T6)
      YF_PRIO(8);
      YF_FETCH(6, W_m, L_m);
      YF_DENSE(F, 1, 2, 16, B_T6M, B_T7, 0, 6, EPI_RAW, 0, B_T7, YF_D_C12, no_add, W_m, L_m, 5);          // conv2d_12
      YF_SYNC(); YF_DUMP(B_T7, 6, T7)
      YF_PRIO(9);
      YF_HALO(B_T8, true, F, G8, H_T8, YF_W_DW15, tid_m);
      YF_FETCH(7, W_m, L_m);
      if (W_m < PV) pool8_v<F, PV * 64, false>(frames, tid_m);                                        // pool_8 (v) + QUANTIZE#21: HB -> concat_22 ...
      else v2::dense2_stage<F, NW - PV, YF_TPJ13, 1, 8, B_T7, B_T8, 0, 36, EPI_LUT, YF_L_LEAKY14, B_T8, 6>(frames, out_all, tab, no_add, W_m - PV, L_m);   // ... beside conv2d_13: T7 -> T8
      YF_SYNC(); YF_DUMP(B_T14, 18, Q21) YF_DUMP(B_T8, 36, T8)
    } else {
  YF_PRIO(5);
#if !(defined(YF_LAB) && defined(YF_WHATIF_NO_POOL8H))   // what-if (WRONG results): the horizontal pass and its barrier gone -- the bound for folding it into conv2d_6's epilogue
      pool8_h<F, NT>(frames, tid_f);                                                                   // pool_8 (h)
      YF_SYNC();
#endif
      YF_STAGE_END()
      YF_PRIO(6);
      pool8_v<F, NT, XDUMP>(frames, tid_m);                                                      // pool_8 (v) + QUANTIZE#21
      YF_SYNC();                                    // T6 (written next) aliases HB (read by pool_8 v)
      YF_PRIO(7);
      YF_FETCH(5, W_m, L_m);
      YF_DW(F, 2, B_T4, B_T6, 18, YF_L_LEAKY11, YF_W_DW10, W_m, L_m, 4, JT_DW10);                          // conv2d_10
      YF_SYNC(); YF_DUMP(B_T14, 18, Q21) YF_DUMP(B_T14, 18, P8, YF_T14_CONV_BASE) YF_DUMP(B_T6, 18, T6)
      YF_STAGE_END()
      YF_PRIO(8);
      YF_FETCH(6, W_m, L_m);
      YF_DENSE(F, 1, 2, 16, B_T6, B_T7, 0, 6, EPI_RAW, 0, B_T7, YF_D_C12, no_add, W_m, L_m, 5);            // conv2d_12
      YF_SYNC(); YF_DUMP(B_T7, 6, T7)
      YF_STAGE_END()
      YF_PRIO(9);
      YF_HALO(B_T8, true, F, G8, H_T8, YF_W_DW15, tid_m);
      YF_FETCH(7, W_m, L_m);
      YF_DENSE(F, YF_TPJ13, 1, 8, B_T7, B_T8, 0, 36, EPI_LUT, YF_L_LEAKY14, B_T8, YF_D_C13, no_add, W_m, L_m, 6); // conv2d_13
      YF_SYNC(); YF_DUMP(B_T8, 36, T8)
      YF_STAGE_END()
    }
    YF_PRIO(10);
    YF_FETCH(8, W_m, L_m);
    YF_DW(F, 1, B_T8, B_T9, 36, YF_L_LEAKY16, YF_W_DW15, W_m, L_m, 7, JT_DW15);                          // conv2d_15
    YF_SYNC(); YF_DUMP(B_T9, 36, T9)
    YF_STAGE_END()
    YF_PRIO(11);
    YF_FETCH(9, W_m, L_m);
    if constexpr (XDUMP)   // debug builds: conv2d_17's own output is parked in the (still unwritten) conv half of concat_22 for the dump
      v2::dense2_stage<F, NW, 1, 3, 16, B_T9, B_T11, 0, 6, EPI_ADD, YF_A_ADD18, B_T7, 8, B_T14::OFF + YF_T14_CONV_BASE, B_T14::S>(frames, out_all, tab, addctx(YF_A_ADD18), W_m, L_m);
    else
    YF_DENSE(F, 1, 3, 16, B_T9, B_T11, 0, 6, EPI_ADD, YF_A_ADD18, B_T7, YF_D_C17, addctx(YF_A_ADD18), W_m, L_m, 8);   // conv2d_17 + eltwise_18
    YF_SYNC(); YF_DUMP(B_T11, 6, T11) if constexpr (XDUMP) { YF_DUMP(B_T14, 6, C17, YF_T14_CONV_BASE) }
    YF_STAGE_END()
    YF_PRIO(12);
    YF_FETCH(10, W_m, L_m);
    YF_DENSE(F, 3, 1, 8, B_T11, B_T14, YF_T14_CONV_BASE, 18, EPI_LUT, YF_L_LEAKY20, B_T14, YF_D_C19, no_add, W_m, L_m, 9);  // conv2d_19 -> concat_22
    YF_SYNC(); YF_DUMP(B_T14, 36, T14, 0, 18, 2)
    YF_STAGE_END()
    YF_PRIO(13);
    // BATCH: the parked group's T15 goes from the scratch straight into the odd tail sets by LDS-DMA (no registers), issued
    // here -- their bytes (T9/T11's old slots) are dead once conv2d_19 is through.  The wait that guards it is the explicit
    // s_waitcnt vmcnt(0) in front of the barrier behind conv2d_23 (this toolchain also waits at conv2d_23's first LDS access: its
    // alias analysis cannot tell the DMA's destination from the stage's buffers, so the transfer overlaps less than it could).
    // One wave-instruction moves 64 x 16 contiguous bytes.
    if constexpr (BATCH) {
      if (parked_first >= 0) {
        constexpr int PV = TailBufs<FRAME_BYTES>::T15_BYTES / 16, WI = (PV + 63) / 64;      // vectors / wave-instructions per frame
        const char* park = prm.scratch + (long)blockIdx.x * (F * PV * 16);
        for (int j = W_m; j < F * WI; j += NW) {
          const int f = j / WI, k0 = (j - f * WI) * 64;
          if (k0 + L_m < PV)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uintptr_t)(park + (f * PV + k0 + L_m) * 16),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(uint32_t)(PRE + (2 * f + 1) * U::T15::FS + 16 * k0),
                                             16, 0, 0);
        }
      }
    }
    YF_HALO(B_T15, false, F, G15, H_T15, YF_W_DW27, tid_m);
    YF_FETCH(11, W_m, L_m);
    YF_DENSE(F, YF_TPJ23, 3, 16, B_T14, B_T15, 0, 24, EPI_LUT, YF_L_LEAKY24, B_T15, YF_D_C23, no_add, W_m, L_m, 10);   // conv2d_23
    if constexpr (BATCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the parked T15's LDS-DMA must have landed before the barrier that publishes the odd sets
    YF_SYNC(); YF_DUMP(B_T15, 24, T15)
    YF_STAGE_END()
    // ---- the 7x7 tail.  BATCH: it runs once per PAIR of groups on FT = 2F frames.  Its thirteen stages are latency chains
    // (98 pixels per group: one or two jobs per wave), so twice the jobs per stage cost far less than twice the time.  The
    // first group of a pair parks its T15 (5.4 KB per frame) in a per-workgroup HBM scratch and skips the tail; the second
    // group fetches it back into the odd tail sets -- set f of the tail sits at f * FRAME_BYTES / 2, so the even sets ARE the
    // arenas' own T15 -- and runs the tail for both.  A workgroup's last group runs the tail alone when it has no partner.
    long odd_first = -1;                          // first frame of the odd sets (the parked group), -1: none
    if constexpr (BATCH) {
      constexpr int V = U::T15_BYTES / 16;
      uint4* park = reinterpret_cast<uint4*>(prm.scratch) + (long)blockIdx.x * (F * V);
      if (parked_first < 0 && grp + gridDim.x < n_groups) {
        for (int i = tid_t; i < F * V; i += NT) {
          const int f = i / V, k = i - f * V;
          park[i] = *reinterpret_cast<const uint4*>(frames + f * FRAME_STRIDE + 16 * k);
        }
        parked_first = first;
        continue;
      }
      if (parked_first >= 0) {                  // its T15 is already in the odd sets (LDS-DMA issued before conv2d_23)
        odd_first = parked_first;
        parked_first = -1;
      }
    }
    // frame number of tail set f (BATCH: even sets = this group, odd sets = the parked one), -1 = nothing to write
    auto frame_of = [&](int f) -> long {
      long id = first + f;
      if constexpr (BATCH) id = (f & 1) ? (odd_first >= 0 ? odd_first + (f >> 1) : -1) : first + (f >> 1);
      return id < prm.n ? id : -1;
    };
#define YF_DUMP_T(BUF, C, OFF, ...) \
  if constexpr (DUMP) { if (prm.dump) { if constexpr (BATCH) dump_sets<BUF, C, FT, NT>(frames, prm.dump, DS, DumpOffsets::OFF, frame_of, tid, ##__VA_ARGS__); \
                                        else dump_buf<BUF, C, FT, NT>(frames, prm.dump, DS, DumpOffsets::OFF, first, prm.n, tid, ##__VA_ARGS__); \
                                        YF_SYNC(); } }
    YF_PRIO(14);
    YF_FETCH(12, W_t, L_t);
    {   // pool_25 + QUANTIZE#45 on the first waves (by columns), conv2d_27 on the others: both only read T15
      constexpr int PW = v2::pool25_waves<FT>();
      static_assert(PW < NW, "waves left for conv2d_27");
      if (W_t < PW) v2::pool25_cols<FT, typename U::T15, typename U::T30, XDUMP>(frames, W_t * 64 + L_t);
      else v2::dw2_stage<FT, NW - PW, 2, typename U::T15, typename U::T17, 24, YF_L_LEAKY28, 11, v2::JobTabs<F, BATCH>::JT_DW27>(frames, tab, W_t - PW, L_t);
    }
    YF_SYNC(); YF_DUMP_T(typename U::T30, 24, Q45) if constexpr (XDUMP) { YF_DUMP_T(typename U::T30, 24, P25, 24) } YF_DUMP_T(typename U::T17, 24, T17)
    YF_STAGE_END()
    YF_PRIO(15);
    YF_FETCH(13, W_t, L_t);
    YF_DENSE(FT, 1, 2, 16, typename U::T17, typename U::T18, 0, 8, EPI_RAW, 0, typename U::T18, YF_D_C29, no_add, W_t, L_t, 12);   // conv2d_29
    if constexpr (!BATCH) decode_prev(W_t, L_t);                                                    // previous group's boxes
    YF_SYNC(); YF_DUMP_T(typename U::T18, 8, T18)
    YF_STAGE_END()
    YF_PRIO(16);
    YF_HALO(typename U::T19, true, FT, G19, H_T19, YF_W_DW32, tid_t);
    YF_FETCH(14, W_t, L_t);
    YF_DENSE(FT, 5, 1, 8, typename U::T18, typename U::T19, 0, 40, EPI_LUT, YF_L_LEAKY31, typename U::T19, YF_D_C30, no_add, W_t, L_t, 13);  // conv2d_30
    YF_SYNC(); YF_DUMP_T(typename U::T19, 40, T19)
    YF_STAGE_END()
    YF_PRIO(17);
    YF_FETCH(15, W_t, L_t);
    YF_DW(FT, 1, typename U::T19, typename U::T20, 40, YF_L_LEAKY33, YF_W_DW32, W_t, L_t, 14, JT_DW32);    // conv2d_32
    YF_SYNC(); YF_DUMP_T(typename U::T20, 40, T20)
    YF_STAGE_END()
    YF_PRIO(18);
    YF_FETCH(16, W_t, L_t);
    if constexpr (XDUMP)   // debug builds: conv2d_34's own output -> the conv half of concat_46 (written by conv2d_42 only)
      v2::dense2_stage<FT, NW, 1, 3, 16, typename U::T20, typename U::T22, 0, 8, EPI_ADD, YF_A_ADD35, typename U::T18, 15, U::T30::OFF + 24, U::T30::S>(frames, out_all, tab, addctx(YF_A_ADD35), W_t, L_t);
    else
    YF_DENSE(FT, 1, 3, 16, typename U::T20, typename U::T22, 0, 8, EPI_ADD, YF_A_ADD35, typename U::T18, YF_D_C34, addctx(YF_A_ADD35), W_t, L_t, 15);   // conv2d_34 + eltwise_35
    YF_SYNC(); YF_DUMP_T(typename U::T22, 8, T22) if constexpr (XDUMP) { YF_DUMP_T(typename U::T30, 8, C34, 24) }
    YF_STAGE_END()
    YF_PRIO(19);
    YF_HALO(typename U::T19, true, FT, G19, H_T19, YF_W_DW38, tid_t);
    YF_FETCH(17, W_t, L_t);
    YF_DENSE(FT, 5, 1, 8, typename U::T22, typename U::T19, 0, 40, EPI_LUT, YF_L_LEAKY37, typename U::T19, YF_D_C36, no_add, W_t, L_t, 16);  // conv2d_36
    YF_SYNC(); YF_DUMP_T(typename U::T19, 40, T23)
    YF_STAGE_END()
    YF_PRIO(20);
    YF_FETCH(18, W_t, L_t);
    YF_DW(FT, 1, typename U::T19, typename U::T20, 40, YF_L_LEAKY39, YF_W_DW38, W_t, L_t, 17, JT_DW32);    // conv2d_38
    YF_SYNC(); YF_DUMP_T(typename U::T20, 40, T24)
    YF_STAGE_END()
    YF_PRIO(21);
    YF_FETCH(19, W_t, L_t);
    if constexpr (XDUMP)
      v2::dense2_stage<FT, NW, 1, 3, 16, typename U::T20, typename U::T26, 0, 8, EPI_ADD, YF_A_ADD41, typename U::T22, 18, U::T30::OFF + 24, U::T30::S>(frames, out_all, tab, addctx(YF_A_ADD41), W_t, L_t);
    else
    YF_DENSE(FT, 1, 3, 16, typename U::T20, typename U::T26, 0, 8, EPI_ADD, YF_A_ADD41, typename U::T22, YF_D_C40, addctx(YF_A_ADD41), W_t, L_t, 18);   // conv2d_40 + eltwise_41
    YF_SYNC(); YF_DUMP_T(typename U::T26, 8, T26) if constexpr (XDUMP) { YF_DUMP_T(typename U::T30, 8, C40, 24) }
    YF_STAGE_END()
    YF_PRIO(22);
    YF_FETCH(20, W_t, L_t);
    if constexpr (XDUMP)   // debug builds: LEAKY_RELU #43's output (through the debug LUT) -> T20's slot, dead since conv2d_40
      v2::dense2_stage<FT, NW, 2, 1, 8, typename U::T26, typename U::T30, 24, 24, EPI_LUT, YF_L_L43Q44, typename U::T30, 19, U::T20::OFF, U::T20::S, DBG_LUT>(frames, out_all, tab, no_add, W_t, L_t);
    else
    YF_DENSE(FT, 3, 1, 8, typename U::T26, typename U::T30, 24, 24, EPI_LUT, YF_L_L43Q44, typename U::T30, YF_D_C42, no_add, W_t, L_t, 19);  // conv2d_42 -> concat_46
    YF_SYNC(); YF_DUMP_T(typename U::T30, 48, T30) if constexpr (XDUMP) { YF_DUMP_T(typename U::T20, 24, L43) }
    YF_STAGE_END()
    YF_PRIO(23);
    YF_HALO(typename U::T19, true, FT, G19, H_T19, YF_W_DW49, tid_t);
    YF_FETCH(21, W_t, L_t);
    YF_DENSE(FT, 2, 3, 16, typename U::T30, typename U::T19, 0, 40, EPI_LUT, YF_L_LEAKY48, typename U::T19, YF_D_C47, no_add, W_t, L_t, 20);   // conv2d_47
    YF_SYNC(); YF_DUMP_T(typename U::T19, 40, T31)
    YF_STAGE_END()
    YF_PRIO(24);
    YF_FETCH(22, W_t, L_t);
    YF_DW(FT, 1, typename U::T19, typename U::T20, 40, YF_L_LEAKY50, YF_W_DW49, W_t, L_t, 21, JT_DW32);    // conv2d_49
    YF_SYNC(); YF_DUMP_T(typename U::T20, 40, T32)
    YF_STAGE_END()
    YF_PRIO(25);
    YF_FETCH(23, W_t, L_t);
    // the decode's two look-up tables (2 KB) -> the first bytes of the frame arenas, dead since conv2d_47 (T15 / T30 of set 0), two stages ahead of
    // the decode: the barrier behind this stage waits for the transfer, the one behind conv2d_53 would do so on the critical path
    if (BATCH && prm.dets != nullptr && W_t < 2) {
      int dl = L_t;
      asm volatile("" : "+v"(dl));
      const uint8_t* src = reinterpret_cast<const uint8_t*>(W_t == 0 ? yfdec::d_sig_bits : yfdec::d_exp_bits) + 16 * dl;
      const uint32_t dst = (uint32_t)(PRE + OUT_ALL_BYTES + 1024 * W_t);
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    YF_DENSE(FT, 2, 3, 16, typename U::T20, typename U::T33, 0, 32, EPI_LUT, YF_L_LEAKY52, typename U::T33, YF_D_C51, no_add, W_t, L_t, 22);   // conv2d_51
    YF_SYNC(); YF_DUMP_T(typename U::T33, 32, T33)
    YF_STAGE_END()
    YF_PRIO(26);
    if constexpr (!BATCH) {
      YF_DENSE(FT, 1, 2, 16, typename U::T33, typename U::T33, 0, 18, EPI_HEAD, 0, typename U::T33, YF_D_C53, no_add, W_t, L_t, 23);   // conv2d_53
      YF_SYNC();
      // head: F*882 contiguous bytes -> HBM, 2-byte granules (882 is not a multiple of 4)
      const long valid = min((long)F, prm.n - first);
      const int n16 = (int)(valid * (OUT_FRAME_BYTES / 2));
      uint16_t* dst = reinterpret_cast<uint16_t*>(prm.out + first * OUT_FRAME_BYTES);
      const uint16_t* srcp = reinterpret_cast<const uint16_t*>(out_all);
      for (int i = tid; i < n16; i += NT) dst[i] = srcp[i];
      prev_first = first;
    } else {
      YF_DENSE(FT, 1, 2, 16, typename U::T33, typename U::HEAD, 0, 18, EPI_HEAD_LDS, 0, typename U::T33, YF_D_C53, no_add, W_t, L_t, 23);   // conv2d_53
      YF_SYNC();
      // heads: 882 bytes per frame from its set -> HBM, 2-byte granules; the boxes of set w are decoded by wave w meanwhile
      constexpr int H16 = OUT_FRAME_BYTES / 2;
      // with a decode the first FT waves decode (one frame each) while the other waves copy the heads; without one every wave copies
      const bool split = prm.dets != nullptr && FT < NW;
      const int c0 = split ? tid_t - FT * 64 : tid_t, cstep = split ? NT - FT * 64 : NT;
      if (!split || W_t >= FT) {
        for (int i = c0; i < FT * H16; i += cstep) {
          const int f = i / H16, k = i - f * H16;
          const long id = frame_of(f);
          if (id >= 0) reinterpret_cast<uint16_t*>(prm.out + id * OUT_FRAME_BYTES)[k] = *reinterpret_cast<const uint16_t*>(frames + f * U::HEAD::FS + U::HEAD::OFF + 2 * k);
        }
      }
      for (int f = W_t; f < FT; f += NW) {
        const long id = frame_of(f);
        if (prm.dets != nullptr && id >= 0) {
          int dl = L_t;
          asm volatile("" : "+v"(dl));
          yfdec::decode_frame_lds(reinterpret_cast<const int8_t*>(frames + f * U::HEAD::FS + U::HEAD::OFF), id, dl, prm.mode, prm.w_scale, prm.h_scale, prm.dets, prm.counts, prm.cap,
                                  (uint32_t)(PRE + OUT_ALL_BYTES), prm.q_thr);
        }
      }
    }
#undef YF_DUMP_T
  }
  if constexpr (!BATCH) {   // boxes of this workgroup's last group
    const int tid = tid0, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    decode_prev(wave, lane);
  }
#undef YF_HALO
#undef YF_FETCH
#undef YF_CONV1
#undef YF_DENSE
#undef YF_DW
#undef YF_DUMP
#undef YF_STAGE_END
#undef YF_PRIO
#undef YF_SYNC
}

template <int F, int NW, bool DUMP>
constexpr size_t lds_bytes() { return (size_t)v2::pre_bytes<F, tail_batch<DUMP>()>() + (tail_batch<DUMP>() ? 0 : (F * OUT_FRAME_BYTES + 15) & ~15) + (size_t)FRAME_BYTES + (size_t)(F - 1) * FRAME_STRIDE + (DUMP ? YF_DBG_LUT_BYTES : 0); }
template <bool DUMP>
constexpr size_t scratch_bytes_per_frame_slot() { return tail_batch<DUMP>() ? (size_t)TailBufs<FRAME_BYTES>::T15_BYTES : 0; }
#undef YF_PDUMP
