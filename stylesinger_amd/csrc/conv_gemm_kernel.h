// Generic fp32-MFMA implicit-GEMM 1-D convolution for gfx950 (CDNA4).
//
//   out[b][t][n] = epi( sum_j sum_ci A'[b][t + tap_off[j]][ci] * W[n][j][ci] )
//
// Design (MI355X-first):
//   * v_mfma_f32_32x32x2_f32: exact-fp32 matrix FMA at the 157 TF/s fp32 rate. One wave owns a
//     (32*TM)x(32*TN) output sub-tile in AGPR/VGPRs (16 regs per 32x32 block).
//   * activations are channels-last so a conv tap is a row shift of the same [rows][C] panel:
//     no im2col is ever materialised; out-of-range rows are zero-filled in the global->LDS stage.
//   * both operands are staged K-contiguous ([row][32+4] floats, 144-B row stride) so every lane
//     fetches its 4 consecutive K values with ONE conflict-free ds_read_b128; the K order inside a
//     32-chunk is permuted (k = 8q + 4h + s) identically for A and B, which the sum does not care about.
//   * double-buffered LDS, one barrier per K-chunk, register-staged prefetch of chunk c+1 issued
//     before the MFMAs of chunk c.
//   * block -> tile map keeps all N-tiles of one M-tile on one XCD (ids congruent mod 8) so the
//     activation panel is fetched once per XCD L2.
//   * epilogues are fused: bias / activation / residual / gate / residual+skip / DDPM posterior step.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/stylesinger_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = BK;  // floats; no padding: the 16-B slots of a row are XOR-swizzled instead (see lds_slot)

// LDS image of a staged operand tile: row r holds 32 consecutive K values = 8 slots of 16 B. Slot s of row r lives
// at physical slot s ^ ((r >> 1) & 7). With 128-B rows the 64 banks (256 B) hold two rows, so a ds_read_b128 lane
// group (rows {0-3,12-15,20-27} of one 16-B column, MI355X_MICROARCH.md §LDS) conflicts iff two rows agree in
// parity and in (r>>1)&7, i.e. are congruent mod 16 - none are. ds_write_b128 (8 contiguous lanes = the 8 slots of
// one row) is conflict-free too. Dropping the +4 padding cuts a 64x128 tile to 48 KiB -> 3 blocks per CU.
__device__ __forceinline__ int lds_slot(int row, int slot) { return row * LDS_LD + ((slot ^ ((row >> 1) & 7)) << 2); }

// BF16 = true: same staging and epilogues, but both operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way
// from LDS to the matrix core and the products run on v_mfma_f32_32x32x16_bf16 (fp32 accumulate) - BASELINE config 4.
// The matrix pipe is then ~16x faster than the fp32 form, so the loop is latency bound and prefetches two chunks ahead.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool BF16 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_gemm_kernel(const ss_conv_gemm_args a, int m_tiles_per_item, int m_tiles,
                                                        int n_tiles, int dbg) {
  constexpr int WTM = BM / WAVES_M;  // rows per wave
  constexpr int WTN = BN / WAVES_N;  // cols per wave
  constexpr int TM = WTM / 32;
  constexpr int TN = WTN / 32;
  constexpr int NT = 64 * WAVES_M * WAVES_N;  // threads per block
  constexpr int RP = NT / 8;                   // tile rows staged per pass (8 threads x float4 cover the 32-wide chunk)
  static_assert(TM >= 1 && TN >= 1 && WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(BM % RP == 0 && BN % RP == 0, "tile rows must be a multiple of the staging pass");
  constexpr int A_F4 = BM / RP;  // float4 per thread per chunk
  constexpr int B_F4 = BN / RP;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BM][LDS_LD]
  float* Bs = smem + 2 * BM * LDS_LD;      // [2][BN][LDS_LD]

  // ---- block -> (m tile, n tile): all n tiles of an m tile share (blockIdx % 8) -> same XCD ----
  const int id = blockIdx.x;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int mt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (mt >= m_tiles) return;
  const int b = mt / m_tiles_per_item;
  const int t0 = (mt % m_tiles_per_item) * BM;
  const int n0 = nt * BN;

#ifdef SS_KERNEL_TIMESTAMPS  // per-wave phase stamps for tools/phase_times.py; never compiled into the shipped library
  const unsigned long long ts0 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
#endif
  ss_apply_wave_prio(dbg & 3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;

  const int len = a.lens ? a.lens[b] : a.T;
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;  // weight set of this batch item (grouped launch)
  const float* Wg = a.W + (int64_t)grp_w * a.w_group_stride;
  const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
  const float* abiasg = a.a_bias ? a.a_bias + (int64_t)grp_w * a.a_bias_group_stride : nullptr;

  const int kchunks_per_tap = a.Kp / BK;
  const int nchunks = a.ntaps * kchunks_per_tap;
  const int ldw = a.ntaps * a.Kp;

  // ---- operand fetch through buffer resources: the hardware range check IS the zero padding ----
  // A: records = len*lda*4 bytes of this item -> rows >= len read 0; a negative row gives a negative byte offset,
  //    i.e. >= 2^31 as unsigned, also out of range -> 0. No clamps, no validity masks, 32-bit address math.
  // W: records = Np*ldw*4 -> packed rows beyond Np read 0.
  // (descriptor inputs go through readfirstlane so that hipcc can PROVE them wave-uniform; otherwise every
  //  buffer op is wrapped in a waterfall loop - cdna_hip_programming.md T20)
  auto uniform_ptr = [](const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Wg), 0, __builtin_amdgcn_readfirstlane(a.Np * ldw * 4), 0x00020000);
  // A-prologue bias through a descriptor as well (a plain pointer select would become a FLAT load, which also
  // counts on lgkmcnt and would stall the LDS fragment reads); no bias -> 0 records -> reads 0.
  const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(abiasg ? abiasg : Wg), 0, __builtin_amdgcn_readfirstlane(abiasg ? a.Cin * 4 : 0), 0x00020000);

  // staging coordinates: 8 threads x float4 cover one 32-wide K chunk of a row; RP rows per pass
  const int st_c4 = tid & 7;
  const int st_row = tid >> 3;
  const int a_off0 = ((t0 + st_row) * a.lda + st_c4 * 4) * 4;  // byte offset of (row, col) of pass 0, tap offset 0
  const int a_pass = RP * a.lda * 4;
  const int w_off0 = ((n0 + st_row) * ldw + st_c4 * 4) * 4;
  const int w_pass = RP * ldw * 4;
  const float pro_scale = a.a_scale, pro_slope = a.a_lrelu;
  // A-prologue form (wave-uniform). VALU instructions are NOT free beside v_mfma_f32_32x32x2_f32 (tools/ubench/mfma_valu.hip:
  // ~2.8 matrix-pipe cycles per VALU op beyond the first two per MFMA), so layers without a prologue must not pay for one:
  //   0 = none (no bias, scale 1, slope 1): registers go to LDS untouched - the range-checked fetch already returned 0 on padding
  //   1 = leaky-relu only, slope in (0,1]: max(v, slope*v) (== max(v,0) + slope*min(v,0) bit for bit; keeps 0 at 0)
  //   2 = general: lrelu((v + bias) * scale), explicit zero on padding
  const int pro_mode = (abiasg || pro_scale != 1.0f) ? 2 : (pro_slope == 1.0f ? 0 : (pro_slope > 0.0f && pro_slope < 1.0f ? 1 : 2));

  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 ra[A_F4], rb[B_F4];
  float4 rpb;
  u32x4 ra2[BF16 ? A_F4 : 1], rb2[BF16 ? B_F4 : 1];  // second register stage of the bf16 loop
  float4 rpb2;

  // K-chunk cursor (uniform): chunk c = (tap, ci0); advanced incrementally, no division in the loop
  struct Cursor { int tap, ci0; };
  auto advance = [&](Cursor& k) {
    k.ci0 += BK;
    if (k.ci0 >= a.Kp) { k.ci0 = 0; ++k.tap; }
  };

  // Issue the fetches of a chunk. Nothing here depends on loaded data, so no s_waitcnt is placed before the
  // MFMAs that follow in program order: the L2/HBM latency hides under them.
  // `dead` = 0x80000000 turns the whole fetch into out-of-range reads (return 0, no memory traffic): the bf16 loop issues
  // its prefetch unconditionally so that no control-flow join hides the in-flight count from the s_waitcnt insertion.
  auto load_a_pm = [&](const Cursor& k, u32x4* ra, float4& rpb, int dead, auto pm_tag) {
    constexpr int PM = decltype(pm_tag)::value;
    const int ci = k.ci0 + st_c4 * 4;
    const bool ci_ok = ci < a.Cin;  // only false in the zero-padded tail of a Cin that is not a multiple of 32
    const int chunk_off = (a.tap_off[k.tap] * a.lda + k.ci0) * 4;
    const int oob = (ci_ok ? 0 : (int)0x80000000) | dead;  // OR-ed into the offset: one branch-free load either way
    if constexpr (PM == 2)  // only the general prologue has a bias to fetch
      rpb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_bias, (ci * 4) | dead, 0, 0));
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
      ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (a_off0 + i * a_pass + chunk_off) | oob, 0, 0);
  };
  auto load_b_to = [&](int c, u32x4* rb, int dead = 0) {
#pragma unroll
    for (int i = 0; i < B_F4; ++i)
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (w_off0 + i * w_pass + c * (BK * 4)) | dead, 0, 0);
  };
  auto load_a_to = [&](const Cursor& k, u32x4* ra, float4& rpb, int dead = 0) { load_a_pm(k, ra, rpb, dead, std::integral_constant<int, 2>{}); };
  auto load_a = [&](const Cursor& k) { load_a_to(k, ra, rpb); };
  auto load_b = [&](int c) { load_b_to(c, rb); };
  // A prologue: lrelu((x + bias) * scale) on real elements, exact 0 on padding; branch-free
  // (lrelu(x,s) = max(x,0) + s*min(x,0), identity for s = 1), then the swizzled LDS write.
  auto store_a_from = [&](int buf, const Cursor& k, const u32x4* ra, const float4& rpb, auto pm_tag) {
    constexpr int PM = decltype(pm_tag)::value;
    float* Ad = As + buf * BM * LDS_LD;
    if constexpr (PM == 0) {
#pragma unroll
      for (int i = 0; i < A_F4; ++i) *reinterpret_cast<float4*>(Ad + lds_slot(st_row + i * RP, st_c4)) = __builtin_bit_cast(float4, ra[i]);
      return;
    }
    if constexpr (PM == 1) {
#pragma unroll
      for (int i = 0; i < A_F4; ++i) {
        float4 v = __builtin_bit_cast(float4, ra[i]);
        v.x = ss_lrelu_max(v.x, pro_slope * v.x); v.y = ss_lrelu_max(v.y, pro_slope * v.y);
        v.z = ss_lrelu_max(v.z, pro_slope * v.z); v.w = ss_lrelu_max(v.w, pro_slope * v.w);
        *reinterpret_cast<float4*>(Ad + lds_slot(st_row + i * RP, st_c4)) = v;
      }
      return;
    }
    const int r0 = t0 + st_row + a.tap_off[k.tap];
    const bool c_ok = k.ci0 + st_c4 * 4 < a.Cin;
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      float4 v = __builtin_bit_cast(float4, ra[i]);
      const bool ok = c_ok && (unsigned)(r0 + i * RP) < (unsigned)len;
      v.x = (v.x + rpb.x) * pro_scale; v.y = (v.y + rpb.y) * pro_scale;
      v.z = (v.z + rpb.z) * pro_scale; v.w = (v.w + rpb.w) * pro_scale;
      v.x = fmaxf(v.x, 0.f) + pro_slope * fminf(v.x, 0.f); v.y = fmaxf(v.y, 0.f) + pro_slope * fminf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f) + pro_slope * fminf(v.z, 0.f); v.w = fmaxf(v.w, 0.f) + pro_slope * fminf(v.w, 0.f);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(Ad + lds_slot(st_row + i * RP, st_c4)) = v;
    }
  };
  auto store_b_from = [&](int buf, const u32x4* rb) {
    float* Bd = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int i = 0; i < B_F4; ++i)
      *reinterpret_cast<float4*>(Bd + lds_slot(st_row + i * RP, st_c4)) = __builtin_bit_cast(float4, rb[i]);
  };
  auto store_a = [&](int buf, const Cursor& k) { store_a_from(buf, k, ra, rpb, std::integral_constant<int, 2>{}); };
  auto store_b = [&](int buf) { store_b_from(buf, rb); };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  Cursor kc{0, 0};
  load_a(kc);
  load_b(0);
  store_a(0, kc);
  store_b(0);
  __syncthreads();

  const int l31 = lane & 31;
  const int lh = lane >> 5;
  // fragment addressing: lane (l31, lh) reads row (tile_row0 + l31), slot 2q + lh. Rows m*32 + l31 of a wave
  // tile keep (row>>1)&7 == (l31>>1)&7 because wave tile origins are multiples of 32, so one swizzle per lane.
  const int swz = (l31 >> 1) & 7;
  const int a_row = (wm * WTM + l31) * LDS_LD;
  const int b_row = (wn * WTN + l31) * LDS_LD;
  auto read_frags = [&](const float* Ac, const float* Bc, int q, float4 (&af)[TM], float4 (&bf)[TN]) {
    const int so = ((2 * q + lh) ^ swz) << 2;
#pragma unroll
    for (int m = 0; m < TM; ++m) af[m] = *reinterpret_cast<const float4*>(Ac + a_row + m * 32 * LDS_LD + so);
#pragma unroll
    for (int n = 0; n < TN; ++n) bf[n] = *reinterpret_cast<const float4*>(Bc + b_row + n * 32 * LDS_LD + so);
  };
  auto mfma_group = [&](const float4 (&af)[TM], const float4 (&bf)[TN]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const float av = (s == 0) ? af[m].x : (s == 1) ? af[m].y : (s == 2) ? af[m].z : af[m].w;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const float bv = (s == 0) ? bf[n].x : (s == 1) ? bf[n].y : (s == 2) ? bf[n].z : bf[n].w;
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m][n], 0, 0, 0);
        }
      }
    }
  };
  // bf16 form: two fragment quads (k = 8q+4h+s for q, q+1) give the 8 K values a lane feeds to one 32x32x16 op; A and
  // B use the same K assignment, which is all the sum needs.
  auto pack8 = [](const float4& u, const float4& v) {
    bf16x8 r;
    r[0] = (__bf16)u.x; r[1] = (__bf16)u.y; r[2] = (__bf16)u.z; r[3] = (__bf16)u.w;
    r[4] = (__bf16)v.x; r[5] = (__bf16)v.y; r[6] = (__bf16)v.z; r[7] = (__bf16)v.w;
    return r;
  };
  auto mfma_pair_bf16 = [&](const float4 (&a0)[TM], const float4 (&a1)[TM], const float4 (&b0)[TN], const float4 (&b1)[TN]) {
    bf16x8 av[TM], bv[TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) av[m] = pack8(a0[m], a1[m]);
#pragma unroll
    for (int n = 0; n < TN; ++n) bv[n] = pack8(b0[n], b1[n]);
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
  };
  auto compute_chunk_bf16 = [&](int cur) {
    const float* Ac = As + cur * BM * LDS_LD;
    const float* Bc = Bs + cur * BN * LDS_LD;
    float4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    mfma_pair_bf16(af0, af1, bf0, bf1);
    read_frags(Ac, Bc, 2, af0, bf0);
    read_frags(Ac, Bc, 3, af1, bf1);
    mfma_pair_bf16(af0, af1, bf0, bf1);
  };
  // Main loop. The MFMA stream of a chunk (4 groups of 4*TM*TN matrix ops, 64 cycles each) is the clock; every
  // other instruction of the chunk is placed BETWEEN groups so that it issues in the shadow of in-flight MFMAs
  // (one wave per SIMD cannot rely on other waves to fill the pipe):
  //   [frags q0,q1] [A fetch c+1] G0 [frags q2] [W fetch c+1] G1 [frags q3] G2 [prologue + LDS write c+1] G3 | barrier
  // The last chunk is peeled so the steady-state body is one straight-line block.
#ifdef SS_KERNEL_TIMESTAMPS
  const unsigned long long ts1 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
#endif
  if constexpr (BF16) {
    // chunk j's registers live in stage j&1 ((ra,rb) = stage 0, (ra2,rb2) = stage 1); in iteration c the fetch of chunk
    // c+2 is issued, chunk c is multiplied from LDS[c&1], then chunk c+1 (fetched one iteration ago) moves to LDS.
    Cursor k1 = kc, k2;
    advance(k1);
    k2 = k1;
    if (nchunks > 1) { load_a_to(k1, ra2, rpb2); load_b_to(1, rb2); }
    int c = 0;
    auto body = [&](u32x4* raN, float4& rpbN, u32x4* rbN, u32x4* raNN, float4& rpbNN, u32x4* rbNN) {
      const bool more = c + 2 < nchunks;
      const int dead = more ? 0 : (int)0x80000000;
      if (more) advance(k2);  // scalar bookkeeping only
      load_a_to(k2, raNN, rpbNN, dead);
      load_b_to(more ? c + 2 : 0, rbNN, dead);
      __builtin_amdgcn_sched_barrier(0);
      compute_chunk_bf16(c & 1);
      __builtin_amdgcn_sched_barrier(0);
      store_a_from((c + 1) & 1, k1, raN, rpbN, std::integral_constant<int, 2>{});
      store_b_from((c + 1) & 1, rbN);
      k1 = k2;
      __syncthreads();
      ++c;
    };
    // nchunks-1 bodies in all; pairs run as one straight-line loop body (no branch inside: the accumulators stay in
    // AGPRs and the s_waitcnt insertion sees exact in-flight counts), an odd one is peeled after the loop.
    const int nbodies = nchunks - 1;
    for (int i = 0; i + 2 <= nbodies; i += 2) {
      body(ra2, rpb2, rb2, ra, rpb, rb);
      body(ra, rpb, rb, ra2, rpb2, rb2);
    }
    if (nbodies & 1) body(ra2, rpb2, rb2, ra, rpb, rb);
  } else {
    // two chunks per loop iteration so that the LDS buffer index is a compile-time constant: every fragment / staging address
    // is then one per-thread base register + an immediate offset (no per-chunk address arithmetic on the VALU)
    // (the prologue form is dispatched OUTSIDE the loop: one branch-free loop copy per form - a branch inside the body makes
    //  the compiler bounce the accumulators through VGPRs and wait for all outstanding loads at the join)
    auto run_loop = [&](auto pm_tag) {
      auto body = [&](auto cur_tag, int c) {
        constexpr int cur = decltype(cur_tag)::value;
        const float* Ac = As + cur * BM * LDS_LD;
        const float* Bc = Bs + cur * BN * LDS_LD;
        float4 af0[TM], bf0[TN], af1[TM], bf1[TN];
        read_frags(Ac, Bc, 0, af0, bf0);
        read_frags(Ac, Bc, 1, af1, bf1);
        advance(kc);
        load_a_pm(kc, ra, rpb, 0, pm_tag);
        __builtin_amdgcn_sched_barrier(0);  // pin: fetches are ISSUED here, two MFMA groups before their first use
        mfma_group(af0, bf0);
        read_frags(Ac, Bc, 2, af0, bf0);
        load_b(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(af1, bf1);
        read_frags(Ac, Bc, 3, af1, bf1);
        mfma_group(af0, bf0);
        __builtin_amdgcn_sched_barrier(0);  // pin: the LDS writes of chunk c+1 go in the shadow of the last group
        store_a_from(cur ^ 1, kc, ra, rpb, pm_tag);
        store_b(cur ^ 1);
        mfma_group(af1, bf1);
        __syncthreads();
      };
      int c = 0;
      for (; c + 2 < nchunks; c += 2) {
        body(std::integral_constant<int, 0>{}, c);
        body(std::integral_constant<int, 1>{}, c + 1);
      }
      if (c + 1 < nchunks) body(std::integral_constant<int, 0>{}, c);
    };
    if (pro_mode == 0) run_loop(std::integral_constant<int, 0>{});
    else if (pro_mode == 1) run_loop(std::integral_constant<int, 1>{});
    else run_loop(std::integral_constant<int, 2>{});
  }
  // Epilogue operands that live in HBM (GATE: the hoisted conditioner slab E, 40 KB row stride; RESSKIP: x and the
  // skip accumulator) are fetched BEFORE the last chunk's MFMAs when they fit in registers, so their miss latency
  // (~2 us) hides under >= 2048 MFMA cycles instead of stalling the epilogue.
  const int row_base = t0 + wm * WTM;
  const int col_base = n0 + wn * WTN;
  auto row_of = [&](int m, int r) { return row_base + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh; };
  constexpr bool PREFETCH = (EPI == SS_EPI_GATE || EPI == SS_EPI_RESSKIP) && (TM * TN <= 2);
  float pre[PREFETCH ? TM : 1][PREFETCH ? TN : 1][16];
  if constexpr (PREFETCH && EPI == SS_EPI_GATE) {
    const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int pc = col_base + n * 32 + l31;
      const bool col_ok = ((pc >> 6) * 32 + l31) < a.N;
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_of(m, r);
          pre[m][n][r] = (Eb && col_ok && row < a.T) ? Eb[(int64_t)row * a.lde + pc] : 0.f;
        }
    }
  }
  if constexpr (PREFETCH && EPI == SS_EPI_RESSKIP) {
    const float* Rb = a.R + (int64_t)b * a.r_batch_stride;
    const float* C2b = a.C2 + (int64_t)b * a.c2_batch_stride;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = col_base + n * 32 + l31;
      const bool col_ok = col < a.N;
      const bool first = col < a.Nh;
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_of(m, r);
          const bool ok = col_ok && row < a.T;
          if (first) pre[m][n][r] = ok ? Rb[(int64_t)row * a.ldr + col] : 0.f;
          else pre[m][n][r] = (ok && a.accumulate) ? C2b[(int64_t)row * a.ldc2 + (col - a.Nh)] : 0.f;
        }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (BF16) {
    compute_chunk_bf16((nchunks - 1) & 1);
  } else {
    const int cur = (nchunks - 1) & 1;
    const float* Ac = As + cur * BM * LDS_LD;
    const float* Bc = Bs + cur * BN * LDS_LD;
    float4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    mfma_group(af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    mfma_group(af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    mfma_group(af0, bf0);
    mfma_group(af1, bf1);
  }

  // ------------------------------------------------------------------------------------------
  // Epilogue. C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // ------------------------------------------------------------------------------------------
#ifdef SS_KERNEL_TIMESTAMPS
  const unsigned long long ts2 = (dbg & 16) ? __builtin_readcyclecounter() : 0ull;
#endif
  // Every epilogue is two-phase: (1) issue ALL the global reads it needs into registers, (2) compute + store.
  // The output may alias the inputs (in-place residual updates), so the compiler cannot hoist a load above a
  // store by itself; without the split every element pays a full load round trip.
  if constexpr (EPI == SS_EPI_STORE) {
    float* Cb = a.C + (int64_t)b * a.c_batch_stride;
    const float* Rb = a.R ? a.R + (int64_t)b * a.r_batch_stride : nullptr;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = col_base + n * 32 + l31;
      const bool col_ok = col < a.N;
      const float bs = (biasg && col_ok) ? biasg[col] : 0.f;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        float rv[16], pv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_of(m, r);
          const bool ok = col_ok && row < a.T;
          rv[r] = (Rb && ok) ? Rb[(int64_t)row * a.ldr + col] : 0.f;
          pv[r] = (a.accumulate && ok) ? Cb[(int64_t)row * a.ldc + col] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_of(m, r);
          if (!col_ok || row >= a.T) continue;
          float v = (acc[m][n][r] + bs) * a.pre_scale;
          v = ss_apply_act(v, a.act, a.act_slope);
          v = (v + rv[r]) * a.post_scale + pv[r];
          if (a.mask_rows && row >= len) v = 0.f;
          Cb[(int64_t)row * a.ldc + col] = v;
        }
      }
    }
  } else if constexpr (EPI == SS_EPI_GATE) {
    if constexpr (TN % 2 == 0) {
      float* Cb = a.C + (int64_t)b * a.c_batch_stride;
      const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
      // interior wave tiles (all rows < min(T, len), all channels < N) skip every per-element predicate
      const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
      const bool interior = (row_base + WTM <= row_lim) && (((col_base + WTN) >> 1) <= a.N);
      auto gate_tile = [&](auto checked_tag) {
        constexpr bool CHECK = decltype(checked_tag)::value;
#pragma unroll
        for (int n = 0; n < TN; n += 2) {
          const int pc0 = col_base + n * 32 + l31;  // packed column of the first member of the pair
          const int pc1 = pc0 + 32;
          const int oc = (pc0 >> 6) * 32 + l31;  // output channel
          const bool col_ok = !CHECK || oc < a.N;
          const float b0 = (biasg && col_ok) ? biasg[pc0] : 0.f;
          const float b1 = (biasg && col_ok) ? biasg[pc1] : 0.f;
#pragma unroll
          for (int m = 0; m < TM; ++m) {
            float e0[16], e1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if constexpr (PREFETCH) {
                e0[r] = pre[m][n][r];
                e1[r] = pre[m][n + 1][r];
              } else {
                const int row = row_of(m, r);
                const bool ok = Eb && col_ok && (!CHECK || row < a.T);
                e0[r] = ok ? Eb[(int64_t)row * a.lde + pc0] : 0.f;
                e1[r] = ok ? Eb[(int64_t)row * a.lde + pc1] : 0.f;
              }
            }
            float* cp = Cb + (int64_t)(row_base + m * 32 + 4 * lh) * a.ldc + oc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rr = (r & 3) + 8 * (r >> 2);
              const float v0 = acc[m][n][r] + b0 + e0[r];
              const float v1 = acc[m][n + 1][r] + b1 + e1[r];
              float g = (a.gate_mode == 0) ? ss_sigmoid_fast(v0) * ss_tanh_fast(v1) : ss_tanh_fast(v0) * ss_sigmoid_fast(v1);
              if constexpr (CHECK) {
                const int row = row_base + m * 32 + 4 * lh + rr;
                if (!col_ok || row >= a.T) continue;
                if (a.mask_rows && row >= len) g = 0.f;
              }
              cp[(int64_t)rr * a.ldc] = g;
            }
          }
        }
      };
      if (interior) gate_tile(std::false_type{});
      else gate_tile(std::true_type{});
    }
  } else if constexpr (EPI == SS_EPI_RESSKIP) {
    float* Cb = a.C + (int64_t)b * a.c_batch_stride;
    float* C2b = a.C2 + (int64_t)b * a.c2_batch_stride;
    const float* Rb = a.R + (int64_t)b * a.r_batch_stride;
    const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
    const bool interior = (row_base + WTM <= row_lim) && (col_base + WTN <= a.N);
    auto rs_tile = [&](auto checked_tag) {
      constexpr bool CHECK = decltype(checked_tag)::value;
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        const int col = col_base + n * 32 + l31;
        const bool col_ok = !CHECK || col < a.N;
        const float bs = (biasg && col_ok) ? biasg[col] : 0.f;
        const bool first = col < a.Nh;  // uniform per 32-column block
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          float pv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if constexpr (PREFETCH) {
              pv[r] = pre[m][n][r];
            } else {
              const int row = row_of(m, r);
              const bool ok = col_ok && (!CHECK || row < a.T);
              if (first) pv[r] = ok ? Rb[(int64_t)row * a.ldr + col] : 0.f;
              else pv[r] = (ok && a.accumulate) ? C2b[(int64_t)row * a.ldc2 + (col - a.Nh)] : 0.f;
            }
          }
          float* cp = first ? Cb + (int64_t)(row_base + m * 32 + 4 * lh) * a.ldc + col
                            : C2b + (int64_t)(row_base + m * 32 + 4 * lh) * a.ldc2 + (col - a.Nh);
          const int64_t ld = first ? a.ldc : a.ldc2;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2);
            const float v = acc[m][n][r] + bs;
            float o = first ? (pv[r] + v) * a.post_scale : v + pv[r];
            if constexpr (CHECK) {
              const int row = row_base + m * 32 + 4 * lh + rr;
              if (!col_ok || row >= a.T) continue;
              if (a.mask_rows && row >= len) o = 0.f;
            }
            cp[rr * ld] = o;
          }
        }
      }
    };
    if (interior) rs_tile(std::false_type{});
    else rs_tile(std::true_type{});
  } else if constexpr (EPI == SS_EPI_DDPM) {
    // v = eps_theta. x0 = clamp(recip*x - recipm1*eps, -1, 1); mean = c1*x0 + c2*x; x <- mean + sigma*z
    float* Cb = a.C + (int64_t)b * a.c_batch_stride;
    const SsPhilox rng(a.seed + (a.seed_dev ? a.seed_dev[0] : 0ull));
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = col_base + n * 32 + l31;
      const bool col_ok = col < a.N;
      const float bs = (biasg && col_ok) ? biasg[col] : 0.f;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        float xv[16], zv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_of(m, r);
          const bool ok = col_ok && row < a.T;
          xv[r] = ok ? Cb[(int64_t)row * a.ldc + col] : 0.f;
          zv[r] = (ok && a.noise && a.ddpm_sigma != 0.f) ? a.noise[((int64_t)b * a.T + row) * a.N + col] : 0.f;
        }
        if (a.ddpm_sigma != 0.f && !a.noise && col_ok) {   // in-kernel noise: one Philox block per four consecutive frames of this lane (ss_mel_draw4)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row4 = row_of(m, 4 * j);     // a multiple of 4: tile origin, 32 m, 8 j, 4 lh
            if (row4 < a.T) {
              float z4[4];
              ss_mel_draw4(rng, (uint32_t)(row4 >> 2), (uint32_t)a.N, (uint32_t)col, (uint32_t)b, a.step, z4);
#pragma unroll
              for (int i = 0; i < 4; ++i) zv[4 * j + i] = z4[i];
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_of(m, r);
          if (!col_ok || row >= a.T) continue;
          const float eps = acc[m][n][r] + bs;
          const float x = xv[r];
          float x0 = a.ddpm_recip * x - a.ddpm_recipm1 * eps;
          x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
          if (a.ddpm_x0_pred) x0 = eps;  // the network output IS x0 (ProDiffusion.p_sample, prodiff.py:150-153), no clamp
          const float mean = a.ddpm_c1 * x0 + a.ddpm_c2 * x;
          float xn = mean + a.ddpm_sigma * zv[r];
          if (a.mask_rows && row >= len) xn = 0.f;
          Cb[(int64_t)row * a.ldc + col] = xn;
        }
      }
    }
  }
#ifdef SS_KERNEL_TIMESTAMPS
  if ((dbg & 16) && EPI == SS_EPI_GATE && a.C2 && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.C2) + ((size_t)blockIdx.x * (NT / 64) + wave) * 8;
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = __builtin_readcyclecounter();
    o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
    o[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    o[6] = blockIdx.x;
  }
#endif
}

inline int dbg_flags() {  // bits 0-1: wave priority mode (ss_set_tuning); bit 4: timestamps (debug builds only)
#ifdef SS_KERNEL_TIMESTAMPS
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SS_DBG");
    v = e ? (atoi(e) & ~3) : 0;
  }
  return v | (g_ss_tuning.wave_prio & 3);
#else
  return g_ss_tuning.wave_prio & 3;
#endif
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, bool BF16 = false>
int launch(const ss_conv_gemm_args& a, hipStream_t stream) {
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const int n_cols = (EPI == SS_EPI_GATE) ? a.Np : a.N;
  const int n_tiles = ss_cdiv(n_cols, BN);
  const int m_tiles_pad = ss_cdiv(m_tiles, 8) * 8;
  const int grid = m_tiles_pad * n_tiles;
  const size_t lds = (size_t)2 * (BM + BN) * LDS_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BM, BN, WAVES_M, WAVES_N, EPI, BF16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WAVES_M, WAVES_N, EPI, BF16>), dim3(grid), dim3(64 * WAVES_M * WAVES_N), lds, stream, a,
                     m_tiles_per_item, m_tiles, n_tiles, dbg_flags());
  SS_CHECK_LAUNCH("ss_conv_gemm");
  return SS_OK;
}


template <int EPI>
int launch_tile(int tile, const ss_conv_gemm_args& a, hipStream_t stream) {
  constexpr bool G = EPI == SS_EPI_GATE;  // GATE needs an even number of 32-col blocks per wave
  if (a.mfma_bf16) {  // bf16-operand form: the production tiles only
    switch (tile) {
      case SS_TILE_128x128: return launch<128, 128, 2, 2, EPI, true>(a, stream);
      case SS_TILE_64x128: return launch<64, 128, 2, 2, EPI, true>(a, stream);
      case SS_TILE_128x64: return launch<128, 64, 4, 1, EPI, true>(a, stream);
      case SS_TILE_64x64:
        if constexpr (!G) return launch<64, 64, 2, 2, EPI, true>(a, stream);
        break;
      case SS_TILE_128x32:
        if constexpr (!G) return launch<128, 32, 4, 1, EPI, true>(a, stream);
        break;
      default: break;
    }
    ss_set_error("ss_conv_gemm: tile %d has no bf16 form (epilogue %d)", tile, EPI);
    return SS_ERR_ARG;
  }
  switch (tile) {
    case SS_TILE_128x128: return launch<128, 128, 2, 2, EPI>(a, stream);
    case SS_TILE_64x128: return launch<64, 128, 2, 2, EPI>(a, stream);
    case SS_TILE_128x64: return launch<128, 64, 4, 1, EPI>(a, stream);
    case SS_TILE_64x64:
      if constexpr (!G) return launch<64, 64, 2, 2, EPI>(a, stream);
      break;
    case SS_TILE_128x32:
      if constexpr (!G) return launch<128, 32, 4, 1, EPI>(a, stream);
      break;
    default: break;
  }
  ss_set_error("ss_conv_gemm: bad tile %d for epilogue %d", tile, EPI);
  return SS_ERR_ARG;
}
}  // namespace
