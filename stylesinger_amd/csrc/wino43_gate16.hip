// Winograd F(4,3) dilated conv + conditioner addend + gate (modules/diff/net.py:66-73) on 16x16x4 fp32 MFMA tiles: the round-3 form of
// the dominant kernel for SINGLE-ROUND launches (BASELINE config 2: 12 000 - 24 000 frames per launch on 1024 SIMDs).
//
// Why a second tiling of the same arithmetic (wino43_gate.hip has the algebra and the 32x32x2 form): at the C2 shape the 32x32 kernel
// launches 1504 wave tiles of 32 quads x 32 columns on 1024 SIMDs - a SIMD holds two of them or one, so the launch lasts as long as
// two tiles while a quarter of the matrix pipes idle (executed MFMA fraction 0.44, profiles/r02_pmc_gate.json). Here a wave tile is
// 16*MT quads x 16 columns (MT = 3: 48 x 16, 25 % smaller than 32 x 32), a workgroup 16*MT quads x 64 packed columns:
//   mel  (8 x 1500 frames, 512 columns): MT = 3 -> 8 x 8 x 8 = 512 workgroups = exactly 2 per CU, 2 equal waves per SIMD
//   f0   (16 x 1500 frames, 384 columns): MT = 3 -> 16 x 8 x 6 = 768 workgroups = exactly 3 per CU
// Differences from the 32x32 kernel that follow from the 16-column wave tile:
//   * a wave's 16 MFMA columns are 8 first-operand and 8 second-operand columns of the SAME 8 channels (packed columns
//     n0 + 8w + c and n0 + 32 + 8w + c), so the gate product is an in-wave DPP exchange (row_ror:8) - no LDS, no barrier in the epilogue;
//   * every weight element is used by exactly one wave: the B operand goes global -> registers directly (two chunks ahead, 3 register
//     stages), only the transformed A tile is staged through LDS (2 x 16*MT x 32 floats);
//   * v_mfma_f32_16x16x4_f32 has a 40-cycle dependent latency at a 32-cycle issue rate: the MT row tiles of a component interleave.
// Arithmetic (transforms, exact-fp32 products, fp32 accumulation, gate) is the 32x32 kernel's; only the order in which the K products
// enter an accumulator differs (a 16x16x4 MFMA consumes K = {4h+e, 8+4h+e, 16+4h+e, 24+4h+e} of a chunk), so the two forms agree to
// fp32 rounding, not bit for bit (tests/test_gpu_round3.py).
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// SS_G16_ABL (debug builds only, tools/ablate_g16.sh; results are wrong by design): 1 = no global fetches inside the loop, 2 = no LDS
// stores, 3 = no MFMAs, 4 = no barriers in the loop, 5 = no conditioner-addend loads, 6 = no activations / exchange in the epilogue,
// 7 = no weight fetches inside the loop, 8 = no raw-row fetches inside the loop
#ifndef SS_G16_ABL
#define SS_G16_ABL 0
#endif

namespace {

constexpr int BK = 32;
constexpr int LD = BK;
constexpr int BN = 64;
constexpr int NC = 6;

// 16-byte slot swizzle of a [rows][32] fp32 tile. Reads: lane (r = l & 15, kg = l >> 4) takes slots 2kg, 2kg+1 of row r; with the
// ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) this map is conflict-free (simulated; SQ_LDS_BANK_CONFLICT = 0).
__device__ __forceinline__ int swz16(int row) { return ((row >> 1) & 7) ^ ((((row >> 2) ^ (row >> 3)) & 1) << 1); }
__device__ __forceinline__ int lds_slot16(int row, int slot) { return row * LD + ((slot ^ swz16(row)) << 2); }


// elementwise helpers on float4 / float2 (the two staging slot widths)
__device__ __forceinline__ float4 vfma(float c, const float4& r, const float4& v) {
  return make_float4(fmaf(c, r.x, v.x), fmaf(c, r.y, v.y), fmaf(c, r.z, v.z), fmaf(c, r.w, v.w));
}
__device__ __forceinline__ float2 vfma(float c, const float2& r, const float2& v) { return make_float2(fmaf(c, r.x, v.x), fmaf(c, r.y, v.y)); }
__device__ __forceinline__ float4 vadd(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float2 vadd(const float2& a, const float2& b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float4 vsub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float2 vsub(const float2& a, const float2& b) { return make_float2(a.x - b.x, a.y - b.y); }

// WL = weight layout: false = the packed rows of ss_pack_conv_weight ([Np][6][Kp], shared with the 32x32x2 kernel); true = the
// lane-contiguous repack of ss_pack_gate16_weights ([n tile][wave][K chunk][component][half][lane][4 floats]): one fetch instruction of
// a wave is 1 KB contiguous (8 cache lines) instead of 16 columns x 64 B (16 lines).
// The kernel body as a device function of (workgroup id, LDS base): the __global__ wrapper below passes blockIdx.x and its dynamic LDS; the
// dataflow experiment of fused_gate_res.hip (round 5) calls the same body from a launch that also holds the residual projection's workgroups.
// Returns false for the padding workgroups of the XCD-aligned grid (no tile).
// ST_AUX: cache-policy bits of the output stores (0 in the product; 16 = sc1, write-through to memory: the publish form of the dataflow experiment)
template <int MT, bool KS, bool WL, int ST_AUX = 0>
__device__ __forceinline__ bool wino43_gate16_body(const ss_conv_gemm_args& a, const float* __restrict__ W16, int q_tiles_per_item, int q_tiles, int n_tiles,
                                                   int log2d, unsigned long long* clock_probe, const int block_id, float* __restrict__ smem_) {
  constexpr int BQ = 16 * MT;
  constexpr int NFULL = BQ / 32;             // staging passes of 32 rows x 8 sixteen-byte slots
  constexpr bool HALF = (BQ % 32) != 0;      // + one pass of 16 rows x 16 eight-byte half slots
  const bool probing = clock_probe != nullptr && block_id == 0;
  unsigned long long probe_c0 = 0, probe_r0 = 0;
  if (probing) {
    probe_c0 = __builtin_readcyclecounter();
    probe_r0 = __builtin_amdgcn_s_memrealtime();
  }
  float* smem = static_cast<float*>(__builtin_assume_aligned(smem_, 16));
  float* As = smem;  // [2][BQ][LD]; KS: [2][6][BQ][LD] (all six components of a K chunk staged at once)
  constexpr int SZC = BQ * LD;  // floats of one staged component

  const int id = block_id;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int qt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (qt >= q_tiles) return false;
  // block-uniform by construction; telling the compiler so makes lens[b] a scalar load instead of a vector load + vmcnt(0) in front of
  // the first row fetch
  const int b = __builtin_amdgcn_readfirstlane(qt / q_tiles_per_item);
  const int q0 = __builtin_amdgcn_readfirstlane((qt % q_tiles_per_item) * BQ);
  const int n0 = nt * BN;
  const int d = 1 << log2d;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kg = lane >> 4;

  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const float* Wg = (WL ? W16 : a.W) + (int64_t)grp_w * a.w_group_stride;
  const float* abiasg = a.a_bias ? a.a_bias + (int64_t)grp_w * a.a_bias_group_stride : nullptr;
  const int kchunks = a.Kp / BK;
  const int ldw = NC * a.Kp;

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
  const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(abiasg ? abiasg : Wg), 0, __builtin_amdgcn_readfirstlane(abiasg ? a.Cin * 4 : 0), 0x00020000);

  // ---- staging roles: thread -> (quad row, K slot) of the raw rows; roff = byte offset of raw row r (frame t + (r-1)d) or out of
  // range (-> the fetch returns 0) when that frame is outside [0, len); mc = what dstep enters each component with (sum of the
  // component's coefficients over the VALID rows)
  const int st_c4 = tid & 7, st_row = tid >> 3;     // full passes: 32 rows x 8 slots of 16 B
  const int sh_c2 = tid & 15, sh_row = tid >> 4;    // half pass: 16 rows x 16 half slots of 8 B
  int roff4[NFULL > 0 ? NFULL : 1][6];
  float mc4[NC][NFULL > 0 ? NFULL : 1];
  int roffh[6];
  float mch[NC];
  auto row_setup = [&](int q, int col_floats, int (&ro)[6], auto&& set_mc) {
    const int t = q + 3 * (q & ~(d - 1));
    float v[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int tr = t + (r - 1) * d;
      const bool ok = (unsigned)tr < (unsigned)len;
      v[r] = ok ? 1.0f : 0.0f;
      ro[r] = ok ? (tr * a.lda + col_floats) * 4 : (int)0x80000000;
    }
    set_mc(0, 4.f * v[0] - 5.f * v[2] + v[4]);
    set_mc(5, 4.f * v[1] - 5.f * v[3] + v[5]);
    set_mc(1, v[4] - 4.f * v[2]);  // coefficient sums of the shared terms A, B, C, D (see `build`)
    set_mc(2, v[3] - 4.f * v[1]);
    set_mc(3, v[4] - v[2]);
    set_mc(4, v[3] - v[1]);
  };
#pragma unroll
  for (int i = 0; i < NFULL; ++i) row_setup(q0 + st_row + i * 32, st_c4 * 4, roff4[i], [&](int j, float x) { mc4[j][i] = x; });
  if constexpr (HALF) row_setup(q0 + NFULL * 32 + sh_row, sh_c2 * 2, roffh, [&](int j, float x) { mch[j] = x; });

  // ---- B operand: lane (lc, kg) of wave w owns packed column pc and K floats [8kg, 8kg+8) of every chunk
  const int c7 = lc & 7, chi = lc >> 3;
  const int pc = n0 + 8 * wave + c7 + 32 * chi;
  const int w_voff = WL ? (nt * ldw * BN + wave * (ldw * BN / 4)) * 4 + lane * 16 : (pc * ldw + kg * 8) * 4;

  u32x4 rr4[NFULL > 0 ? NFULL : 1][6];
  u32x2 rr2[6];
  float4 rpb4;
  float2 rpb2;
  auto load_rows = [&](int ci0b) {  // ci0b = byte offset of the K chunk inside a row (wave-uniform -> SGPR soffset)
    ci0b = __builtin_amdgcn_readfirstlane(ci0b);
    if constexpr (NFULL > 0) rpb4 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_bias, st_c4 * 16, ci0b, 0));
    if constexpr (HALF) rpb2 = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_bias, sh_c2 * 8, ci0b, 0));
#pragma unroll
    for (int i = 0; i < NFULL; ++i)
#pragma unroll
      for (int r = 0; r < 6; ++r) rr4[i][r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, roff4[i][r], ci0b, 0);
    if constexpr (HALF) {
#pragma unroll
      for (int r = 0; r < 6; ++r) rr2[r] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_a, roffh[r], ci0b, 0);
    }
  };
  float4 bst[3][2];
  auto load_b = [&](auto stag, int comp, int k) {  // weights of component `comp`, K chunk k (wave-uniform offsets)
    constexpr int S = decltype(stag)::value;
    if constexpr (WL) {
      const int cb = __builtin_amdgcn_readfirstlane(((k * NC + comp) * 2) * 1024);
      bst[S][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, cb, 0));
      bst[S][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, cb + 1024, 0));
    } else {
      const int cb = __builtin_amdgcn_readfirstlane((comp * a.Kp + k * BK) * 4);
      bst[S][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, cb, 0));
      bst[S][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff + 16, cb, 0));
    }
  };
  int a_wr4[NFULL > 0 ? NFULL : 1];
#pragma unroll
  for (int i = 0; i < NFULL; ++i) a_wr4[i] = lds_slot16(st_row + i * 32, st_c4);
  int a_wrh = lds_slot16(NFULL * 32 + sh_row, sh_c2 >> 1) + (sh_c2 & 1) * 2;
  // Input transform with shared sub-expressions (18 instead of 28 VALU ops per element and K chunk - VALU instructions take
  // matrix-pipe time on gfx950):
  //   A = r4 - 4 r2, B = r3 - 4 r1, C = r4 - r2, D = r3 - r1   ->   c1 = A + B, c2 = A - B, c3 = C + 2 D, c4 = C - 2 D
  //   c0 = 4 r0 - 5 r2 + r4, c5 = 4 r1 - 5 r3 + r5; dstep enters every term through the sum of its coefficients over the valid rows.
  // The six components of a K chunk are consumed in the order c1, c0, c5, c2, c3, c4 ("positions" 0..5) so that the raw rows are dead
  // after the third build and the rows of the NEXT K chunk can be fetched three positions (not one) before their first use:
  //   position 5 of k-1 : build c1(k) = A + B   (rows(k) must have landed; A, B kept)
  //   position 0        : build c0(k), C        position 1 : build c5(k), D        -> rows(k) dead: fetch rows(k+1) at position 2
  //   position 2, 3, 4  : build c2 = A - B, c3 = C + 2 D, c4 = C - 2 D from the kept terms
  float4 tA4[NFULL > 0 ? NFULL : 1], tB4[NFULL > 0 ? NFULL : 1], tC4[NFULL > 0 ? NFULL : 1], tD4[NFULL > 0 ? NFULL : 1];
  float2 tAh, tBh, tCh, tDh;
  // BUILD = index in the consumption order of the component being built
  auto build = [&](auto ptag, const auto& pb, float m0, float mA, float mB, float mC, float mD, float m5, const auto& r0, const auto& r1,
                   const auto& r2, const auto& r3, const auto& r4, const auto& r5, auto& tA, auto& tB, auto& tC, auto& tD) {
    constexpr int P = decltype(ptag)::value;
    if constexpr (P == 0) {  // c1
      tA = vfma(mA, pb, vfma(-4.f, r2, r4));
      tB = vfma(mB, pb, vfma(-4.f, r1, r3));
      return vadd(tA, tB);
    } else if constexpr (P == 1) {  // c0 (+ C)
      tC = vfma(mC, pb, vsub(r4, r2));
      return vfma(m0, pb, vfma(-5.f, r2, vfma(4.f, r0, r4)));
    } else if constexpr (P == 2) {  // c5 (+ D)
      tD = vfma(mD, pb, vsub(r3, r1));
      return vfma(m5, pb, vfma(-5.f, r3, vfma(4.f, r1, r5)));
    } else if constexpr (P == 3) {
      return vsub(tA, tB);          // c2
    } else if constexpr (P == 4) {
      return vfma(2.f, tD, tC);     // c3
    } else {
      return vfma(-2.f, tD, tC);    // c4
    }
  };
  auto store_a = [&](float* Ad, auto ptag) {
#pragma unroll
    for (int i = 0; i < NFULL; ++i) {
      auto R = [&](int q) { return __builtin_bit_cast(float4, rr4[i][q]); };
      const float4 v = build(ptag, rpb4, mc4[0][i], mc4[1][i], mc4[2][i], mc4[3][i], mc4[4][i], mc4[5][i], R(0), R(1), R(2), R(3), R(4),
                             R(5), tA4[i], tB4[i], tC4[i], tD4[i]);
      *reinterpret_cast<float4*>(Ad + a_wr4[i]) = v;
    }
    if constexpr (HALF) {
      auto R = [&](int q) { return __builtin_bit_cast(float2, rr2[q]); };
      const float2 v = build(ptag, rpb2, mch[0], mch[1], mch[2], mch[3], mch[4], mch[5], R(0), R(1), R(2), R(3), R(4), R(5), tAh, tBh, tCh, tDh);
      *reinterpret_cast<float2*>(Ad + a_wrh) = v;
    }
  };

  f32x4 acc[NC][MT];
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][m][r] = 0.f;

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  using P5 = std::integral_constant<int, 5>;
  // ORD[p] = component consumed at position p
  constexpr int ORD[6] = {1, 0, 5, 2, 3, 4};
  const int cs = BK * 4;    // bytes of one K chunk
  // chunk (k, p): weights of component ORD[p], K chunk k (load_b); register stage p % 3; LDS buffer p & 1
  load_rows(0);
  load_b(P0{}, ORD[0], 0);
  load_b(P1{}, ORD[1], 0);
  store_a(As, P0{});
  if constexpr (KS) {
    // K-staged form: the six components of chunk k+1 are built during chunk k (one per position, into the OTHER half of the LDS), so a
    // K chunk needs ONE barrier instead of six; chunk 0 is built here, then the raw rows of chunk 1 are fetched.
    store_a(As + 1 * SZC, P1{});
    store_a(As + 2 * SZC, P2{});
    store_a(As + 3 * SZC, P3{});
    store_a(As + 4 * SZC, P4{});
    store_a(As + 5 * SZC, P5{});
    __builtin_amdgcn_sched_barrier(0);
    load_rows(cs);
#pragma unroll
    for (int i = 0; i < NFULL; ++i) a_wr4[i] += 6 * SZC;
    a_wrh += 6 * SZC;
  }
  __syncthreads();

  // fragment addresses: row tile m, lane row lc, slots 2kg + h
  int a_rd[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) a_rd[m][h] = lds_slot16(16 * m + lc, 2 * kg + h);

  int ks_delta = 6 * SZC;
  auto mfma_half = [&](auto jtag, const float4 (&af)[MT], const float4& bf) {
    constexpr int J = decltype(jtag)::value;
    if constexpr (SS_G16_ABL == 3) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[J][m][0] += af[m].x * bf.x + af[m].w * bf.w;
      return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[J][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, bf.x, acc[J][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[J][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, bf.y, acc[J][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[J][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, bf.z, acc[J][m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[J][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, bf.w, acc[J][m], 0, 0, 0);
  };
  // position P of K chunk k: MFMAs of component ORD[P] from LDS buffer P&1 and weight stage P%3. In their shadow the component of
  // position P+1 is built (registers -> other LDS buffer), the weights of position P+2 are fetched into stage (P+2)%3, and at position 2
  // (the raw rows are dead by then) the raw rows of K chunk k+1.
  auto chunk = [&](auto ptag, auto stage_tag, auto fetch_b_tag, auto fetch_rows_tag, int k) {
    constexpr int P = decltype(ptag)::value;
    constexpr int CUR = P & 1;
    constexpr int S = P % 3;
    constexpr int PN = (P + 1) % 6, P2N = (P + 2) % 6;
    const float* Ac = As + (KS ? P : CUR) * SZC;   // KS: a_rd / a_wr carry the half of the LDS this K chunk reads / writes
    float4 af0[MT], af1[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) af0[m] = *reinterpret_cast<const float4*>(Ac + a_rd[m][0]);
#pragma unroll
    for (int m = 0; m < MT; ++m) af1[m] = *reinterpret_cast<const float4*>(Ac + a_rd[m][1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(std::integral_constant<int, ORD[P]>{}, af0, bst[S][0]);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(stage_tag)::value && SS_G16_ABL != 2) {
      if constexpr (KS) store_a(As + P * SZC, ptag);   // component of the same position, NEXT K chunk
      else store_a(As + (CUR ^ 1) * SZC, std::integral_constant<int, PN>{});
    }
    __builtin_amdgcn_sched_barrier(0);  // stores first, then the fetches into the SAME registers
    if constexpr (decltype(fetch_b_tag)::value && SS_G16_ABL != 1 && SS_G16_ABL != 7)
      load_b(std::integral_constant<int, (P + 2) % 3>{}, ORD[P2N], k + (P + 2) / 6);
    if constexpr (decltype(fetch_rows_tag)::value && SS_G16_ABL != 1 && SS_G16_ABL != 8) load_rows((k + (KS ? 2 : 1)) * cs);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(std::integral_constant<int, ORD[P]>{}, af1, bst[S][1]);
    if constexpr (!KS) {
      if constexpr (SS_G16_ABL != 4) __syncthreads();
    } else if constexpr (P == 5 && decltype(stage_tag)::value) {
      if constexpr (SS_G16_ABL != 4) __syncthreads();
      // swap the halves: what was written becomes what is read
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        a_rd[m][0] += ks_delta;
        a_rd[m][1] += ks_delta;
      }
#pragma unroll
      for (int i = 0; i < NFULL; ++i) a_wr4[i] -= ks_delta;
      a_wrh -= ks_delta;
      ks_delta = -ks_delta;
    }
  };
  using Yes = std::true_type;
  using No = std::false_type;
  if constexpr (KS) {
    // chunk k builds chunk k+1 from rows(k+1); rows(k+2) are fetched at position 2, after the last build that reads the raw rows
    for (int k = 0; k + 2 < kchunks; ++k) {
      chunk(P0{}, Yes{}, Yes{}, No{}, k);
      chunk(P1{}, Yes{}, Yes{}, No{}, k);
      chunk(P2{}, Yes{}, Yes{}, Yes{}, k);
      chunk(P3{}, Yes{}, Yes{}, No{}, k);
      chunk(P4{}, Yes{}, Yes{}, No{}, k);
      chunk(P5{}, Yes{}, Yes{}, No{}, k);
    }
    const int k = kchunks - 2;   // builds the last chunk, fetches no rows
    chunk(P0{}, Yes{}, Yes{}, No{}, k);
    chunk(P1{}, Yes{}, Yes{}, No{}, k);
    chunk(P2{}, Yes{}, Yes{}, No{}, k);
    chunk(P3{}, Yes{}, Yes{}, No{}, k);
    chunk(P4{}, Yes{}, Yes{}, No{}, k);
    chunk(P5{}, Yes{}, Yes{}, No{}, k);
  } else {
    for (int k = 0; k + 1 < kchunks; ++k) {
      chunk(P0{}, Yes{}, Yes{}, No{}, k);
      chunk(P1{}, Yes{}, Yes{}, No{}, k);
      chunk(P2{}, Yes{}, Yes{}, Yes{}, k);   // rows(k) are dead: fetch rows(k+1), three positions before position 5 builds c1(k+1)
      chunk(P3{}, Yes{}, Yes{}, No{}, k);
      chunk(P4{}, Yes{}, Yes{}, No{}, k);    // fetches the weights of (k+1, position 0)
      chunk(P5{}, Yes{}, Yes{}, No{}, k);    // builds c1(k+1); fetches the weights of (k+1, position 1)
    }
  }
  {
  // The conditioner addend of the epilogue (48 values per lane at MT = 3; a 40 KB row stride, i.e. one HBM / L2 miss per element) is
  // fetched under the last three chunks, into the registers the raw rows and the shared terms no longer need: in a single-round launch
  // every workgroup reaches its epilogue at the same time, so loads issued there are fully exposed (ablation: 4.4 of 59 us).
  // From here on the lane coordinates are derived afresh (two mbcnt instructions): nothing lane-dependent of the prologue has to stay
  // in a register across the K loop for the epilogue's sake (the K-staged MT = 2 form otherwise spills 7 registers to scratch).
  const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int lc = lane_e & 15, kg = lane_e >> 4;
  const int c7 = lc & 7, chi = lc >> 3;
  const int pc = n0 + 8 * wave + c7 + 32 * chi;
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Eb ? Eb : Wg), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
  const int oc = (n0 >> 1) + 8 * wave + c7;  // output channel: both its operands live in this wave (lanes lc and lc ^ 8)
  const bool col_ok = oc < a.N;
  const int oob = col_ok ? 0 : (int)0x80000000;
  const int lde4 = a.lde * 4, ldc4 = a.ldc * 4;
  float pe[MT][4][4];
  auto fetch_addend = [&]() {
    if (a.e_tiled) {   // the addend in this kernel's fetch order (ss_gate16_tile_addend): [tile][wave][m][r][lane][o], 16 bytes per lane
      const __amdgpu_buffer_rsrc_t rsrc_t = __builtin_amdgcn_make_buffer_rsrc(
          uniform_ptr(a.E), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)q_tiles * n_tiles * (MT * 4096) * 4)), 0x00020000);
      const int tile_b = __builtin_amdgcn_readfirstlane(((qt * n_tiles + nt) * (MT * 4096) + wave * (MT * 1024)) * 4);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 v = SS_G16_ABL == 5 ? make_float4(0.f, 0.f, 0.f, 0.f)
                                           : __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_t, lane_e * 16, tile_b + (m * 4 + r) * 1024, 0));
          pe[m][r][0] = v.x;
          pe[m][r][1] = v.y;
          pe[m][r][2] = v.z;
          pe[m][r][3] = v.w;
        }
      return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int qm = q0 + 16 * m + 4 * kg;          // multiple of 4
      const int tm = qm + 3 * (qm & ~(d - 1));      // frame of quad qm
      const int e_base = tm * lde4 + (pc * 4 + oob);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int dr = r + 3 * (r & ~(d - 1));      // wave-uniform: frame of quad qm + r = tm + dr
#pragma unroll
        for (int o = 0; o < 4; ++o)
          pe[m][r][o] = SS_G16_ABL == 5 ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, e_base, (dr + o * d) * lde4, 0));
      }
    }
  };
  {
    const int k = kchunks - 1;
    using St = std::integral_constant<bool, !KS>;   // KS: everything this chunk reads was built during the previous one
    chunk(P0{}, St{}, Yes{}, No{}, k);
    chunk(P1{}, St{}, Yes{}, No{}, k);
    chunk(P2{}, St{}, Yes{}, No{}, k);   // the last build that reads the raw rows happened in P1; P2 built c2 from (A, B)
    __builtin_amdgcn_sched_barrier(0);
    fetch_addend();
    __builtin_amdgcn_sched_barrier(0);
    chunk(P3{}, St{}, Yes{}, No{}, k);
    chunk(P4{}, St{}, No{}, No{}, k);
    chunk(P5{}, No{}, No{}, No{}, k);
  }

  // ---- epilogue: output transform, conditioner addend, gate; accumulator (m, r) of this lane = quad 16m + 4kg + r, column lc ----
  // VALU instructions take matrix-pipe time and this is the densest VALU block of the kernel: every address is one
  // per-lane base per row tile + a wave-uniform SGPR offset (frame of quad q+r = frame of q + r + 3 (r & ~(d-1)) for q = 0 mod 4), stores
  // are buffer stores (out-of-range rows / columns are dropped by the range check, no exec masking), the activation is one
  // multiply-exp2-add-rcp-fma chain, and the common case (no bias pointer, tile entirely inside [0, len)) skips the bias adds and the
  // padding-row zeroing.
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);
  const bool use_sig = (chi == 0) == (a.gate_mode == 0);
  // sigmoid(x) = rcp(1 + exp2(-x log2 e)); tanh(x) = 2 sigmoid(2x) - 1: one exp2 + one rcp either way, selected per lane by (mul, scale, shift)
  const float am = (use_sig ? -1.0f : -2.0f) * 1.44269504088896340736f, as = use_sig ? 1.0f : 2.0f, ah = use_sig ? 0.0f : -1.0f;
  auto act = [&](float x) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * am)), as, ah); };
  auto partner = [](float x) {  // the value of lane lc ^ 8 of the same 16-lane row (DPP row_ror:8)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xf, 0xf, true));
  };
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  // lanes holding the first operand write frames t, t+d; their partners t+2d, t+3d (both lanes compute the same products)
  const int my_first = chi ? 2 * d : 0;
  auto epilogue = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    float bs = 0.f;
    if constexpr (!FAST) bs = (a.bias && col_ok) ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int qm = q0 + 16 * m + 4 * kg;          // multiple of 4
      const int tm = qm + 3 * (qm & ~(d - 1));      // frame of quad qm
      const int c_base = (tm + my_first) * ldc4 + (oc * 4 + oob);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int dr = r + 3 * (r & ~(d - 1));
        const float a0 = acc[0][m][r], a5 = acc[5][m][r];
        float s12 = acc[1][m][r] + acc[2][m][r], d12 = acc[1][m][r] - acc[2][m][r];
        const float s34 = acc[3][m][r] + acc[4][m][r], d34 = acc[3][m][r] - acc[4][m][r];
        if constexpr (!FAST) {
          s12 += bs;  // enters z0, z2
          d12 += bs;  // enters z1, z3
        }
        const float z0 = a0 + s12 + s34;
        const float z1 = fmaf(2.0f, d34, d12);
        const float z2 = fmaf(4.0f, s34, s12);
        const float z3 = fmaf(8.0f, d34, d12) + a5;
        const float u0 = act(z0 + pe[m][r][0]);
        const float u1 = act(z1 + pe[m][r][1]);
        const float u2 = act(z2 + pe[m][r][2]);
        const float u3 = act(z3 + pe[m][r][3]);
        float g0, g1, g2, g3;
        if constexpr (SS_G16_ABL == 6) {
          g0 = z0 + pe[m][r][0]; g1 = z1 + pe[m][r][1]; g2 = z2 + pe[m][r][2]; g3 = z3 + pe[m][r][3];
        } else {
          g0 = u0 * partner(u0); g1 = u1 * partner(u1); g2 = u2 * partner(u2); g3 = u3 * partner(u3);
        }
        float ga = chi ? g2 : g0, gb = chi ? g3 : g1;
        if constexpr (!FAST) {
          const int ta = tm + dr + my_first;
          if (ta >= row_lim) ga = 0.f;
          if (ta + d >= row_lim) gb = 0.f;
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ga), rsrc_c, c_base, dr * ldc4, ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gb), rsrc_c, c_base, (dr + d) * ldc4, ST_AUX);
      }
    }
  };
  // last frame this tile can touch: quad q0 + BQ - 1, frame + 3d
  const int q_last = q0 + BQ - 1;
  const bool fast = a.bias == nullptr && (q_last + 3 * (q_last & ~(d - 1)) + 3 * d) < row_lim;  // block-uniform
  if (fast) epilogue(std::true_type{});
  else epilogue(std::false_type{});
  }
  if (probing && wave == 0 && __builtin_amdgcn_mbcnt_lo(~0u, 0u) == 0 && __builtin_amdgcn_mbcnt_hi(~0u, 0u) == 0) {   // thread 0, without keeping threadIdx alive
    atomicAdd(clock_probe, (unsigned long long)__builtin_readcyclecounter() - probe_c0);
    atomicAdd(clock_probe + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - probe_r0);
  }
  return true;
}

template <int MT, bool KS, bool WL>
__global__ __launch_bounds__(256, (MT <= 2 ? 3 : 2)) void wino43_gate16_kernel(const ss_conv_gemm_args a, const float* __restrict__ W16, int q_tiles_per_item,
                                                               int q_tiles, int n_tiles, int log2d, unsigned long long* clock_probe) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  wino43_gate16_body<MT, KS, WL>(a, W16, q_tiles_per_item, q_tiles, n_tiles, log2d, clock_probe, (int)blockIdx.x, smem);
}

template <int MT, bool KS, bool WL>
int launch16(const ss_conv_gemm_args& a, const float* W16, int dilation, int log2d, hipStream_t stream) {
  constexpr int BQ = 16 * MT;
  const int quads_per_item = ss_cdiv(a.T, 4 * dilation) * dilation;
  const int q_tiles_per_item = ss_cdiv(quads_per_item, BQ);
  const int q_tiles = q_tiles_per_item * a.B;
  const int n_tiles = a.Np / BN;
  const int grid = ss_cdiv(q_tiles, 8) * 8 * n_tiles;
  const size_t lds = (size_t)(KS ? 12 : 2) * BQ * LD * sizeof(float);
  if constexpr (KS && MT == 3) {   // 72 KB of dynamic LDS: above the 64 KB a kernel gets without asking. The attribute is per device and
    // cheap, so it is set on every launch (a process may drive several GPUs) and a failure is reported, never cached
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino43_gate16_kernel<MT, KS, WL>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
  }
  hipLaunchKernelGGL((wino43_gate16_kernel<MT, KS, WL>), dim3(grid), dim3(256), lds, stream, a, W16, q_tiles_per_item, q_tiles, n_tiles, log2d,
                     g_ss_tuning.clock_probe);
  return 0;
}

}  // namespace

#ifndef SS_FUSED_TU   // fused_gate_res.hip includes this file for the kernel bodies above only

// Workgroups a tiling of `quads` rows launches, and the model used to pick one: a launch costs (work per wave tile) x (workgroup
// layers per CU). MT = 0 in the return value means "the 32x32x2 kernel (wino43_gate.hip) is the better fit".
extern "C" int ss_wino43_gate16_pick(int B, int T, int Np, int dilation) {
  const int quads_per_item = ss_cdiv(T, 4 * dilation) * dilation;
  const int n_tiles = Np / BN;
  const int n_cu = ss_n_cu();
  auto layers = [&](int bq) { return ss_cdiv((long)ss_cdiv(quads_per_item, bq) * B * n_tiles, n_cu); };
  // many rounds per launch: tile granularity no longer matters and the 32x32 tile moves half the LDS bytes per flop
  if (layers(64) >= 6) return 0;
  int best = 0;  // the 64-quad tile of the 32x32x2 kernel
  long best_cost = 4L * layers(64);
  // MT = 1 (16 quads = 64 frames per workgroup) only for launches whose MT = 2 grid leaves CUs without a workgroup (one short utterance:
  // the B = 1 latency shape of inference/StyleSinger.py:175-186): half the work per workgroup, twice the workgroups
  const bool tiny = (long)ss_cdiv(quads_per_item, 32) * B * n_tiles <= n_cu;
  for (int mt = 3; mt >= (tiny ? 1 : 2); --mt) {
    const long cost = (long)mt * layers(16 * mt);
    // ties go to the smaller tile: MT = 2 fits three workgroups per CU (168 registers; MT = 3: 252 -> two) and measured 1 % faster where
    // both fill the chip evenly (BASELINE config 2: mel 768 workgroups of MT = 2 vs 512 of MT = 3)
    if (cost < best_cost || (cost == best_cost && best != 0)) {
      best_cost = cost;
      best = mt;
    }
  }
  return best;
}

namespace {

// [Np][6][Kp] fp32 -> [n tile (64 columns)][wave][K chunk][component][half][lane][4 floats]: lane = kg * 16 + chi * 8 + c7 holds column
// 64 nt + 8 w + c7 + 32 chi, K elements 32 k + 8 kg + 4 half + (0..3)
__global__ void pack_gate16_kernel(const float* __restrict__ src, float* __restrict__ dst, int Np, int Kp) {
  const int64_t n = (int64_t)Np * NC * Kp;
  const int kchunks = Kp / BK;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % Kp);
    const int j = (int)((i / Kp) % NC);
    const int col = (int)(i / ((int64_t)Kp * NC));
    const int nt = col / BN, cl = col % BN;
    const int chi = cl / 32, w = (cl % 32) / 8, c7 = cl % 8;
    const int k = kk / BK, kg = (kk % BK) / 8, h = (kk % 8) / 4, e = kk % 4;
    const int lane = kg * 16 + chi * 8 + c7;
    dst[((((int64_t)(nt * 4 + w) * kchunks + k) * NC + j) * 2 + h) * 256 + lane * 4 + e] = src[i];
  }
}

// E [B][T][lde] (Np packed columns of one layer) -> the fetch order of wino43_gate16_kernel<MT>: one float4 (frames o = 0..3 of a quad) per
// (tile, wave, row tile m, quad r, lane); the index math is the kernel's own (fetch_addend / epilogue)
__global__ void gate16_tile_addend_kernel(const float* __restrict__ E, int lde, int64_t e_bs, float4* __restrict__ out, int B, int T, int Np,
                                          int d, int MT, int q_tiles_per_item) {
  const int n_tiles = Np / BN;
  const int64_t n = (int64_t)B * q_tiles_per_item * n_tiles * MT * 1024;   // float4 count
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int mr = (int)((i >> 6) % (MT * 4));
    const int wave = (int)((i / (64 * MT * 4)) & 3);
    const int64_t tile = i / (MT * 1024);
    const int nt = (int)(tile % n_tiles);
    const int qt = (int)(tile / n_tiles);
    const int b = qt / q_tiles_per_item, q0 = (qt % q_tiles_per_item) * 16 * MT;
    const int m = mr >> 2, r = mr & 3;
    const int lc = lane & 15, kg = lane >> 4, c7 = lc & 7, chi = lc >> 3;
    const int pc = nt * BN + 8 * wave + c7 + 32 * chi;
    const int qm = q0 + 16 * m + 4 * kg;
    const int tm = qm + 3 * (qm & ~(d - 1));
    const int dr = r + 3 * (r & ~(d - 1));
    const float* Eb = E + (int64_t)b * e_bs + pc;
    float v[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int t = tm + dr + o * d;
      v[o] = t < T ? Eb[(int64_t)t * lde] : 0.f;
    }
    out[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

int gate16_impl(const ss_conv_gemm_args* args, const float* W16, int dilation, int mt, void* stream, const char* who) {
  SS_CHECK_ARG(args != nullptr, "%s: null args", who);
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "%s: null A/W/C", who);
  SS_CHECK_ARG(dilation >= 1 && (dilation & (dilation - 1)) == 0 && dilation <= 64, "%s: dilation %d must be a power of two <= 64", who, dilation);
  SS_CHECK_ARG((a.Cin % BK) == 0 && a.Kp == a.Cin && (a.lda & 3) == 0, "%s: Cin=%d must be a multiple of 32 and Kp == Cin", who, a.Cin);
  SS_CHECK_ARG((a.Np % 64) == 0 && 2 * a.N <= a.Np, "%s: Np=%d must be a multiple of 64 and >= 2*N", who, a.Np);
  SS_CHECK_ARG((int64_t)a.T * a.lda * 4 < (1ll << 31) && (!a.E || (int64_t)a.T * a.lde * 4 < (1ll << 31)) &&
                   (int64_t)a.Np * NC * a.Kp * 4 < (1ll << 31),
               "%s: item too large for 32-bit offsets", who);
  SS_CHECK_ARG((int64_t)a.T * a.ldc * 4 < (1ll << 31), "%s: output item too large for 32-bit offsets", who);
  SS_CHECK_ARG(mt >= 0 && mt <= 3, "%s: mt=%d must be 0 (auto), 1, 2 or 3", who, mt);
  SS_CHECK_ARG(!a.e_tiled || (mt != 0 && a.E), "%s: a tiled addend (e_tiled) is laid out for ONE tiling: give mt explicitly", who);
  int log2d = 0;
  while ((1 << log2d) < dilation) ++log2d;
  // K-staged form (one barrier per K chunk, six components staged at once): needs at least two K chunks
  const bool ks = g_ss_tuning.gate16_ks != 0 && a.Kp >= 2 * BK;
  if (mt == 0) {
    mt = ss_wino43_gate16_pick(a.B, a.T, a.Np, dilation);
    if (mt == 0) return ss_wino43_gate(args, dilation, stream);   // the 32x32x2 kernel reads the packed rows in args->W
    if (mt == 1 && !ks) mt = 2;   // the 16-quad tiling only exists in the K-staged form
  }
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  if (mt == 1) {   // small launches only: the K-staged form with the weights in whichever layout the caller has
    SS_CHECK_ARG(ks, "%s: mt = 1 needs the K-staged form (Kp >= 64, gate16_ks on)", who);
    rc = W16 ? launch16<1, true, true>(a, W16, dilation, log2d, st) : launch16<1, true, false>(a, nullptr, dilation, log2d, st);
  } else if (W16) {
    if (mt == 2) rc = ks ? launch16<2, true, true>(a, W16, dilation, log2d, st) : launch16<2, false, true>(a, W16, dilation, log2d, st);
    else rc = ks ? launch16<3, true, true>(a, W16, dilation, log2d, st) : launch16<3, false, true>(a, W16, dilation, log2d, st);
  } else {
    if (mt == 2) rc = ks ? launch16<2, true, false>(a, nullptr, dilation, log2d, st) : launch16<2, false, false>(a, nullptr, dilation, log2d, st);
    else rc = ks ? launch16<3, true, false>(a, nullptr, dilation, log2d, st) : launch16<3, false, false>(a, nullptr, dilation, log2d, st);
  }
  SS_CHECK_ARG(rc == 0, "%s: hipFuncSetAttribute failed (%d)", who, rc);
  SS_CHECK_LAUNCH(who);
  return SS_OK;
}

}  // namespace

extern "C" int ss_wino43_gate16(const ss_conv_gemm_args* args, int dilation, int mt, void* stream) {
  return gate16_impl(args, nullptr, dilation, mt, stream, "ss_wino43_gate16");
}

extern "C" int ss_wino43_gate16w(const ss_conv_gemm_args* args, const float* W16, int dilation, int mt, void* stream) {
  SS_CHECK_ARG(W16 != nullptr, "ss_wino43_gate16w: null W16");
  return gate16_impl(args, W16, dilation, mt, stream, "ss_wino43_gate16w");
}

extern "C" int ss_pack_gate16_weights(const float* src, float* dst, int Np, int Kp, void* stream) {
  SS_CHECK_ARG(src && dst && src != dst && Np > 0 && (Np % BN) == 0 && Kp > 0 && (Kp % BK) == 0, "ss_pack_gate16_weights: Np %% 64, Kp %% 32, out of place");
  const int64_t n = (int64_t)Np * NC * Kp;
  const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(pack_gate16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, Np, Kp);
  SS_CHECK_LAUNCH("ss_pack_gate16_weights");
  return SS_OK;
}

extern "C" int64_t ss_gate16_tiled_floats(int B, int T, int Np, int dilation, int mt) {
  if (B <= 0 || T <= 0 || Np <= 0 || (Np % BN) != 0 || dilation < 1 || mt < 1 || mt > 3) return -1;
  const int quads_per_item = ss_cdiv(T, 4 * dilation) * dilation;
  return (int64_t)ss_cdiv(quads_per_item, 16 * mt) * B * (Np / BN) * mt * 4096;
}

extern "C" int ss_gate16_tile_addend(const float* E, int lde, int64_t e_batch_stride, float* E16, int B, int T, int Np, int dilation, int mt,
                                     void* stream) {
  SS_CHECK_ARG(E && E16 && B > 0 && T > 0 && Np > 0 && (Np % BN) == 0 && lde >= Np && mt >= 1 && mt <= 3, "ss_gate16_tile_addend: bad args");
  SS_CHECK_ARG(dilation >= 1 && (dilation & (dilation - 1)) == 0 && dilation <= 64, "ss_gate16_tile_addend: dilation %d must be a power of two <= 64", dilation);
  SS_CHECK_ARG((((uintptr_t)E16) & 15) == 0, "ss_gate16_tile_addend: E16 must be 16-byte aligned");
  const int quads_per_item = ss_cdiv(T, 4 * dilation) * dilation;
  const int q_tiles_per_item = ss_cdiv(quads_per_item, 16 * mt);
  const int64_t n = (int64_t)B * q_tiles_per_item * (Np / BN) * mt * 1024;
  SS_CHECK_ARG(n * 16 < (1ll << 31), "ss_gate16_tile_addend: launch too large for the kernel's 32-bit offsets");
  const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipLaunchKernelGGL(gate16_tile_addend_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, E, lde, e_batch_stride, reinterpret_cast<float4*>(E16), B, T, Np,
                     dilation, mt, q_tiles_per_item);
  SS_CHECK_LAUNCH("ss_gate16_tile_addend");
  return SS_OK;
}
#endif  // SS_FUSED_TU
