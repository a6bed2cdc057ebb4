// bf16-operand dilated conv + conditioner addend + gate (modules/diff/net.py:66-73) for MANY-ROUND launches (BASELINE config 4:
// 180 000 rows per launch): 256 rows x 256 packed columns per workgroup, 8 waves, both operands by LDS-DMA.
//
// Why a second bf16 kernel (gemm_bf16.hip keeps the generic one, 128x128 tiles / 4 waves / register staging, 25 % of the bf16 roof at
// this shape, "no single bound: the phases of a workgroup run one after the other"):
//   * 256 x 256 tile, wave tile 128 x 64 (4 x 2 accumulators of 32x32): 6 fragment reads feed 8 MFMAs (4 feed 4 before), and the tile
//     moves 3x fewer bytes from L2 into LDS per flop;
//   * the A operand (y = x + dstep, bf16) is staged ONCE per 64-channel chunk WITH its dilation halo (rows t0-8 .. t0+264) and the three
//     taps read the same LDS image at row offsets 8-d, 8, 8+d: a third of the A traffic of three shifted tiles;
//   * global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane): no staging registers, no ds_write, no VALU in the loop; the XOR slot
//     swizzle is applied to the per-lane SOURCE address because the DMA image is lane-linear. The weight tile of step s+1 and the A
//     chunk of the next channel chunk are in flight under the MFMAs of step s; one raw s_barrier + counted s_waitcnt per step.
// Arithmetic contract unchanged: bf16 operands (rounded once where they are produced), exact products, fp32 accumulation, fp32 addend,
// hardware exp/rcp activations, bf16 gate output - the same values as gemm_bf16_kernel<GATE> up to the K summation order.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include "pair16.h"
#include <type_traits>
#include <utility>

typedef ss_f32x16 f32x16;
typedef ss_bf16x8 bf16x8;

// SS_G256_ABL (debug builds only; results wrong by design): 1 = no DMA inside the loop, 2 = no MFMAs, 3 = no barriers, 4 = no epilogue
// loads/stores except one store per lane
#ifndef SS_G256_ABL
#define SS_G256_ABL 0
#endif

namespace {

constexpr int BM = 256, BN = 256, BKH = 64;
constexpr int HALO = 8;                       // rows staged before the tile (dilations up to 8)
constexpr int AROWS = 320;                    // 5 DMA instructions per wave (8 rows each); rows >= BM + 2 HALO are zero fill
constexpr int ROWB = BKH * 2;                 // bytes per LDS row (64 bf16)

__device__ __forceinline__ uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
// LDS-DMA of 64 x 16 bytes (lane i lands at lds_dst + 16 i); a __device__ helper so that the host pass keeps the kernel's launch stub
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

// SPLIT ("bf16x2" precision, ss_gemm_bf16_args.split): operands are (hi, mid) bf16 pairs interleaved by 32 channels, so a 128-byte LDS row is
// 32 channels of BOTH planes (slots 0-3 hi, 4-7 mid) and every line of the DMA / LDS plan below is unchanged; a step then covers 32 channels
// (CCS = K / 32 chunks per tap: 24 steps for K = 256) and runs 2 k-steps x 3 products (mid*hi, hi*mid, hi*hi) = 48 MFMAs per wave from 12
// fragment reads per k-step - 1.5x the matrix work per byte staged. Outputs leave as (hi, mid) pairs in the same interleaved layout.
// SPLIT = 2 ("fp16x2"): the same image with fp16 terms; the A operand's second plane is staged but never read, a step runs 2 k-steps x 2
// products (hi*hi, hi*lo) = 32 MFMAs from 8 + 8 fragment reads (step_w2), the accumulators are scaled by args.out_scale in the epilogue and the
// output is fp16(g) in the hi slots only (the second plane of the output rows is left untouched).
template <class F, int... I>
__device__ __forceinline__ void unrolled_steps(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

template <int CCS, int SPLIT>
__global__ __launch_bounds__(512, 2) void gate256_kernel(const ss_gemm_bf16_args a, int m_tiles_per_item, int m_tiles, int n_tiles, int d) {
  extern __shared__ __attribute__((aligned(16))) char smem_g256[];   // 144 KB (split: 160 KB, the epilogue's staging tile is twice as wide): one workgroup per CU
  // [A0 40 K][B0 32 K][A1 40 K][B1 32 K]: the operands of the LAST step live in A1 / B1, so the first 72 KB are free while it runs
  char* const A0 = smem_g256;
  char* const B0 = A0 + AROWS * ROWB;
  char* const A1 = B0 + BN * ROWB;
  char* const B1 = A1 + AROWS * ROWB;
  // epilogue view of the same memory: two 64-KB addend quarters and a 16-KB output staging tile
  char* const EQ0 = smem_g256;
  char* const EQ1 = smem_g256 + 64 * 1024;
  char* const OUT = smem_g256 + 128 * 1024;

  // consecutive workgroups walk row tiles of the SAME column tile (ids = mod 8 -> one XCD): the 3 x 256-column weight slice stays in
  // that XCD's L2 while the activations stream through
  const int id = blockIdx.x;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int mt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (mt >= m_tiles) return;
  const int b = mt / m_tiles_per_item;
  const int t0 = (mt % m_tiles_per_item) * BM;
  const int n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  constexpr bool W2 = SPLIT == 2;
  const int ldw = 3 * a.K * (SPLIT ? 2 : 1);            // bf16 per packed weight row (3 taps; both planes when split)

  auto uniform_ptr = [](const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.W + (int64_t)grp_w * a.w_group_stride), 0, __builtin_amdgcn_readfirstlane(a.Np * ldw * 2), 0x00020000);

  // ---- DMA roles. A: 40 pieces of 8 rows x 128 B (rows t0 - 8 + r, r < 320; r >= 272 forced out of range); wave w issues pieces
  // w, w + 8, ..., w + 32. B: 32 pieces; wave w issues w, w + 8, w + 16, w + 24. Lane i of a piece lands at (row i >> 3, physical slot
  // i & 7) and therefore fetches logical slot (i & 7) ^ ((row >> 1) & 7).
  // (piece w + 8 j starts 64 j rows after piece w and ((r >> 1) & 7) does not depend on j: ONE per-lane offset per operand)
  const int r0 = 8 * wave + (lane >> 3);
  const int slot0 = (lane & 7) ^ ((r0 >> 1) & 7);
  const int a_voff = ((t0 - HALO + r0) * a.lda + slot0 * 8) * 2;   // may be negative: >= 2^31 as unsigned -> out of range -> zeros
  const int b_voff = ((n0 + r0) * ldw + slot0 * 8) * 2;
  const int a_tail_dead = wave < 2 ? 0 : (int)0x80000000;          // piece w + 32 = rows 256 + 8 w ..: only rows < 272 exist
  // SPLIT = 2: the A operand's second plane (logical slots 4-7) is never read by the matrix cores - its lanes fetch nothing (out of range: the
  // DMA writes zeros), which halves the A traffic from L2 / HBM
  const int a_lo_dead = (W2 && slot0 >= 4) ? (int)0x80000000 : 0;
  auto dma_a = [&](char* buf, int cc, int dead) {
#pragma unroll
    for (int j = 0; j < 5; ++j)
      // the row offset of piece j goes into the VGPR offset (one add), NOT the SGPR offset: for the rows before the item (t0 - 8 + r < 0)
      // the per-lane offset is negative, i.e. >= 2^31 as unsigned, and the hardware adds the SGPR offset without wrapping - a positive
      // SGPR part would leave valid rows of later pieces out of range
      glds16(rsrc_a, buf + (wave + 8 * j) * 8 * ROWB, (a_voff + 64 * j * a.lda * 2) | dead | (j == 4 ? a_tail_dead : 0) | a_lo_dead, cc * (BKH * 2));
  };
  auto piece_a = [&](char* buf, int cc, int j) {
    glds16(rsrc_a, buf + (wave + 8 * j) * 8 * ROWB, (a_voff + 64 * j * a.lda * 2) | (j == 4 ? a_tail_dead : 0) | a_lo_dead, cc * (BKH * 2));
  };
  auto piece_b = [&](char* buf, int cc, int tap, int j) {
    glds16(rsrc_w, buf + (wave + 8 * j) * 8 * ROWB, b_voff, (tap * CCS + cc) * (BKH * 2) + 64 * j * ldw * 2);
  };
  auto dma_b = [&](char* buf, int cc, int tap, int dead) {
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(rsrc_w, buf + (wave + 8 * j) * 8 * ROWB, b_voff | dead, (tap * CCS + cc) * (BKH * 2) + 64 * j * ldw * 2);
  };

  // ---- fragment addresses. A, tap j: row = HALO + (j - 1) d + 128 wm + 32 m + l31; k-step ks reads slot (2 ks + lh) ^ swz(row).
  // 32 m more rows leave the swizzle unchanged ((16 m) & 7 == 0): m is an immediate offset. Kept as (row base, swizzle) per tap and
  // combined with one XOR per read: the 128 accumulator registers leave no room for a table of all 16 addresses.
  int a_base[3], a_swz[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = HALO + (j - 1) * d + 128 * wm + l31;
    a_base[j] = row * ROWB;
    a_swz[j] = ((row >> 1) & 7) ^ lh;
  }
  const int b_base = (64 * wn + l31) * ROWB;
  const int b_swz = (((64 * wn + l31) >> 1) & 7) ^ lh;

  // conditioner addend: fp32 [rows][lde], the tile's 256 packed columns are 1 KB contiguous per row -> one row per DMA instruction.
  // Quarter q = tile rows 128 h + 32 q + (0..31), h = 0, 1 -> 64 pieces; wave w issues pieces w, w + 8, ..., w + 56.
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Eb ? (const void*)Eb : (const void*)a.W), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
  const int e_voff = ((t0 + wave) * a.lde + n0) * 4 + lane * 16;   // piece w + 8 j: 8 j rows further (j < 4), 128 + 8 (j - 4) for j >= 4
  auto piece_e = [&](char* buf, int q, int j) {
    glds16(rsrc_e, buf + (wave + 8 * j) * 1024, e_voff, (q * 32 + (j < 4 ? 8 * j : 128 + 8 * (j - 4))) * a.lde * 4);
  };
  auto dma_e = [&](char* buf, int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      glds16(rsrc_e, buf + (wave + 8 * j) * 1024, e_voff, (q * 32 + (j < 4 ? 8 * j : 128 + 8 * (j - 4))) * a.lde * 4);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // step S = 3 cc + tap, fully unrolled (CCS is a template parameter): MFMAs of (A chunk cc at the tap's row offset, weight tile
  // (cc, tap)). At its top: my pieces of this step's operands have landed - counted vmcnt: the A chunk of cc+1 is issued in the tap-1
  // step AFTER that step's weight pieces, so at the top of the tap-2 step its 5 pieces may still fly; barrier (everyone's pieces landed,
  // everyone finished reading step S-1); then the weight pieces of step S+1 (and, at tap 1, the A chunk cc+1) are issued and fly under
  // this step's MFMAs. Weight tiles alternate B0 / B1 every step, A chunks A0 / A1 every channel chunk.
  // ---- SPLIT: 2 k-steps x 3 product groups of 8 MFMAs per step: G0 = mid x hi, G1 = hi x mid, G2 = hi x hi (G2 reuses G1's A and G0's B
  // fragments). Software pipeline ACROSS the step barrier: the last two groups of a step (G1, G2 of its second k-step: 16 MFMAs whose
  // fragments already sit in registers) are issued AFTER the next step's barrier, with that step's DMA pieces placed between them and its
  // first fragment reads in flight - so the matrix pipe has work while every wave of the workgroup is issuing DMA / waiting on LDS
  // (before: ~2 k of a step's 5.2 k cycles with the pipe idle, 410 us per C4 launch).
  [[maybe_unused]] bf16x8 p_ah[4], p_bh[2], p_bm[2];
  auto mfma8 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[2]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
  };
  auto step_split = [&](auto stag) {
    constexpr int S = decltype(stag)::value;
    constexpr int CC = S / 3, TAP = S % 3;
    constexpr bool LAST = S + 1 >= 3 * CCS;
    const char* Ac = (CC & 1) ? A1 : A0;
    const char* Bc = (S & 1) ? B1 : B0;
    char* Bn = (S & 1) ? B0 : B1;
    char* An = (CC & 1) ? A0 : A1;
    if constexpr (TAP == 2 && CC + 1 < CCS) wait_vmcnt<5>();   // the 5 pieces of A chunk cc+1, issued last step after its weight pieces
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    auto rd_a = [&](int slot, bf16x8 (&f)[4]) {   // a_swz / b_swz carry lh: slot (2 ks) ^ swz = hi, (4 + 2 ks) ^ swz = mid of k-step ks
      const int ao = a_base[TAP] + ((slot ^ a_swz[TAP]) << 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) f[m] = *reinterpret_cast<const bf16x8*>(Ac + ao + m * 32 * ROWB);
    };
    auto rd_b = [&](int slot, bf16x8 (&f)[2]) {
      const int bo = b_base + ((slot ^ b_swz) << 4);
#pragma unroll
      for (int n = 0; n < 2; ++n) f[n] = *reinterpret_cast<const bf16x8*>(Bc + bo + n * 32 * ROWB);
    };
    bf16x8 am0[4], bh0[2];
    rd_a(4, am0);
    rd_b(0, bh0);
    __builtin_amdgcn_sched_barrier(0);
    // DMA pieces this step issues (same order as dma_b, dma_a, dma_e: the vmcnt counts above rely on it): the weight tile of step S+1, at tap
    // 1 the A chunk cc+1, in the last step the first addend quarter (into A0 / B0, which that step does not read)
    constexpr int NB = LAST ? 0 : 4, NA = (TAP == 1 && CC + 1 < CCS) ? 5 : 0, NE = LAST ? 8 : 0, NP = NB + NA + NE;
    auto piece = [&](int i) {
      if (i < NB) piece_b(Bn, (S + 1) / 3, (S + 1) % 3, i);
      else if (i < NB + NA) piece_a(An, CC + 1, i - NB);
      else piece_e(EQ0, 0, i - NB - NA);
    };
    if constexpr (S > 0) {   // the 16 MFMAs deferred by step S-1, one DMA piece after every MFMA until the pieces are out
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i >> 1) & 3, n = i & 1;
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p_ah[m], i < 8 ? p_bm[n] : p_bh[n], acc[m][n], 0, 0, 0);
        if (i < NP) {
          __builtin_amdgcn_sched_barrier(0);
          piece(i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      static_assert(NP <= 16, "more DMA pieces than deferred MFMAs");
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) piece(i);
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 ah0[4], bm0[2], am1[4], bh1[2];
    rd_a(0, ah0);
    rd_b(4, bm0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(am0, bh0);          // G0(0)
    __builtin_amdgcn_sched_barrier(0);
    rd_a(6, am1);
    rd_b(2, bh1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(ah0, bm0);          // G1(0)
    __builtin_amdgcn_sched_barrier(0);
    mfma8(ah0, bh0);          // G2(0)
    __builtin_amdgcn_sched_barrier(0);
    rd_a(2, p_ah);
    rd_b(6, p_bm);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(am1, bh1);          // G0(1); G1(1) = p_ah x p_bm and G2(1) = p_ah x p_bh run after the next barrier
#pragma unroll
    for (int n = 0; n < 2; ++n) p_bh[n] = bh1[n];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LAST) {
      mfma8(p_ah, p_bm);
      mfma8(p_ah, p_bh);
    }
  };
  // ---- SPLIT = 2: products H0 = hi x hi, L0 = hi x lo of k-step 0 run inside the step, H1 / L1 of k-step 1 (fragments p_ah, p_bh, p_bm) are
  // deferred past the next barrier exactly like G1(1) / G2(1) above: the same pipeline with one product group less per k-step.
  auto step_w2 = [&](auto stag) {
    constexpr int S = decltype(stag)::value;
    constexpr int CC = S / 3, TAP = S % 3;
    constexpr bool LAST = S + 1 >= 3 * CCS;
    const char* Ac = (CC & 1) ? A1 : A0;
    const char* Bc = (S & 1) ? B1 : B0;
    char* Bn = (S & 1) ? B0 : B1;
    char* An = (CC & 1) ? A0 : A1;
    if constexpr (TAP == 2 && CC + 1 < CCS) wait_vmcnt<5>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    auto rd_a = [&](int slot, bf16x8 (&f)[4]) {
      const int ao = a_base[TAP] + ((slot ^ a_swz[TAP]) << 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) f[m] = *reinterpret_cast<const bf16x8*>(Ac + ao + m * 32 * ROWB);
    };
    auto rd_b = [&](int slot, bf16x8 (&f)[2]) {
      const int bo = b_base + ((slot ^ b_swz) << 4);
#pragma unroll
      for (int n = 0; n < 2; ++n) f[n] = *reinterpret_cast<const bf16x8*>(Bc + bo + n * 32 * ROWB);
    };
    auto mm8 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[2]) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = ss_mfma_32x32x16<true>(fa[m], fb[n], acc[m][n]);
    };
    bf16x8 ah0[4], bh0[2];
    rd_a(0, ah0);
    rd_b(0, bh0);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NB = LAST ? 0 : 4, NA = (TAP == 1 && CC + 1 < CCS) ? 5 : 0, NE = LAST ? 8 : 0, NP = NB + NA + NE;
    auto piece = [&](int i) {
      if (i < NB) piece_b(Bn, (S + 1) / 3, (S + 1) % 3, i);
      else if (i < NB + NA) piece_a(An, CC + 1, i - NB);
      else piece_e(EQ0, 0, i - NB - NA);
    };
    if constexpr (S > 0) {   // the 16 MFMAs deferred by step S-1 (L1 then H1), one DMA piece after every MFMA until the pieces are out
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i >> 1) & 3, n = i & 1;
        acc[m][n] = ss_mfma_32x32x16<true>(p_ah[m], i < 8 ? p_bm[n] : p_bh[n], acc[m][n]);
        if (i < NP) {
          __builtin_amdgcn_sched_barrier(0);
          piece(i);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      static_assert(NP <= 16, "more DMA pieces than deferred MFMAs");
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) piece(i);
    }
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 bm0[2];
    rd_b(4, bm0);
    __builtin_amdgcn_sched_barrier(0);
    mm8(ah0, bh0);            // H0
    __builtin_amdgcn_sched_barrier(0);
    rd_a(2, p_ah);
    rd_b(2, p_bh);
    __builtin_amdgcn_sched_barrier(0);
    mm8(ah0, bm0);            // L0
    __builtin_amdgcn_sched_barrier(0);
    rd_b(6, p_bm);            // H1 = p_ah x p_bh and L1 = p_ah x p_bm run after the next barrier
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LAST) {
      mm8(p_ah, p_bm);
      mm8(p_ah, p_bh);
    }
  };
  auto step = [&](auto stag) {
    constexpr int S = decltype(stag)::value;
    constexpr int CC = S / 3, TAP = S % 3;
    constexpr bool LAST = S + 1 >= 3 * CCS;
    const char* Ac = (CC & 1) ? A1 : A0;
    const char* Bc = (S & 1) ? B1 : B0;
    char* Bn = (S & 1) ? B0 : B1;
    char* An = (CC & 1) ? A0 : A1;
    // (no DMA is ever issued past the last step: the epilogue's addend quarters reuse this memory and must not race with zero fills)
    if constexpr (TAP == 2 && CC + 1 < CCS) wait_vmcnt<5>();   // the 5 pieces of A chunk cc+1, issued last step after its weight pieces
    else wait_vmcnt<0>();
    if constexpr (SS_G256_ABL != 3) __builtin_amdgcn_s_barrier();
    if constexpr (SS_G256_ABL != 1 && !LAST) dma_b(Bn, (S + 1) / 3, (S + 1) % 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAP == 1 && SS_G256_ABL != 1 && CC + 1 < CCS) dma_a(An, CC + 1, 0);
    if constexpr (LAST) {   // the first addend quarter flies under the last step (into A0 / B0, which that step does not read)
      __builtin_amdgcn_sched_barrier(0);
      dma_e(EQ0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SPLIT) {
      static_assert(!SPLIT, "the split form has its own step (step_split)");
    } else {
    // fragments of k-step ks+1 are read before the MFMAs of k-step ks issue (two register sets of 6 x 16 B): the two waves of a SIMD
    // belong to the same workgroup and leave every barrier in lockstep, so a partner's MFMAs do NOT cover this wave's LDS latency
    bf16x8 af[2][4], bf[2][2];
    auto read_frags = [&](int ks, bf16x8 (&fa)[4], bf16x8 (&fb)[2]) {
      const int ao = a_base[TAP] + (((2 * ks) ^ a_swz[TAP]) << 4);
      const int bo = b_base + (((2 * ks) ^ b_swz) << 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) fa[m] = *reinterpret_cast<const bf16x8*>(Ac + ao + m * 32 * ROWB);
#pragma unroll
      for (int n = 0; n < 2; ++n) fb[n] = *reinterpret_cast<const bf16x8*>(Bc + bo + n * 32 * ROWB);
    };
    read_frags(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) read_frags(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          if constexpr (SS_G256_ABL == 2) acc[m][n][0] += (float)af[ks & 1][m][0] * (float)bf[ks & 1][n][0];
          else acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][m], bf[ks & 1][n], acc[m][n], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };
  dma_a(A0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  dma_b(B0, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  static_assert((CCS & 1) == 0, "the epilogue's LDS plan assumes the last step reads A1 / B1");
  if constexpr (W2) unrolled_steps(step_w2, std::make_integer_sequence<int, 3 * CCS>{});
  else if constexpr (SPLIT) unrolled_steps(step_split, std::make_integer_sequence<int, 3 * CCS>{});
  else unrolled_steps(step, std::make_integer_sequence<int, 3 * CCS>{});

  // ---- epilogue. A single workgroup per CU has nobody to overlap its epilogue with, so nothing here may wait on HBM or issue narrow
  // stores (ablation: with plain per-lane addend loads and 2-byte stores the epilogue took 2/3 of the launch):
  //   * the fp32 addend tile (256 rows x 1 KB) arrives by LDS-DMA in four quarters of 64 rows (one row per DMA instruction), double
  //     buffered: quarter 0 was issued inside the last MFMA step into the 72 KB that step no longer uses, quarter q+1 flies while q is used;
  //   * the bf16 gate outputs of a quarter are staged in LDS and leave as 16-byte stores of row-contiguous runs (256 B per row).
  // Pass q handles accumulator block m = q of every wave: tile rows 128 wm + 32 q + (0..31), LDS row k = 32 wm + (0..31).
  const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
  const int pcl = 64 * wn + l31;                 // packed column inside the tile (first operand; the second sits 32 further)
  const int ocl = 32 * wn + l31;                 // output channel inside the tile
  const bool ch_ok = (n0 >> 1) + ocl < a.N;
  const float b0 = (biasg && ch_ok) ? biasg[n0 + pcl] : 0.f, b1 = (biasg && ch_ok) ? biasg[n0 + pcl + 32] : 0.f;
  const bool sig_first = a.gate_mode == 0;
  const float L2E = 1.44269504088896340736f;
  const float m0 = (sig_first ? -1.0f : -2.0f) * L2E, s0 = sig_first ? 1.0f : 2.0f, h0 = sig_first ? 0.0f : -1.0f;
  const float m1 = (sig_first ? -2.0f : -1.0f) * L2E, s1 = sig_first ? 2.0f : 1.0f, h1 = sig_first ? -1.0f : 0.0f;
  auto act = [](float x, float mul, float sc, float sh) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * mul)), sc, sh); };
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr((uint16_t*)a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 2)), 0x00020000);
  __builtin_amdgcn_s_barrier();   // everyone is done with A1 / B1: the second quarter and the staging tile may overwrite them
  dma_e(EQ1, 1);
  const int e_rd = (32 * wm + 4 * lh) * 1024 + pcl * 4;        // + rr * 1024 (+ 128 for the second operand)
  constexpr int OROW = SPLIT ? 512 : 256;                       // bytes per staged output row: 128 channels (split: hi and mid, interleaved by 32)
  const int o_wr = (32 * wm + 4 * lh) * OROW + (SPLIT ? wn * 128 + l31 * 2 : ocl * 2);   // + rr * OROW
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const char* Eq = (q & 1) ? EQ1 : EQ0;
    // my pieces of quarter q have landed; younger operations that may still fly, in issue order
    //   E0 | E1 | pass 0: E2, NST stores | pass 1: E3, NST stores | pass 2: NST stores | pass 3: NST stores
    constexpr int NST = SPLIT ? 4 : 2;   // stores per thread and pass
    if (q == 0) wait_vmcnt<8>();
    else if (q == 1) wait_vmcnt<8 + NST>();
    else if (q == 2) wait_vmcnt<8 + 2 * NST>();
    else wait_vmcnt<2 * NST>();
    __builtin_amdgcn_s_barrier();  // everyone's pieces landed; everyone finished reading the staging tile of quarter q-1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      const float e0 = *reinterpret_cast<const float*>(Eq + e_rd + rr * 1024);
      const float e1 = *reinterpret_cast<const float*>(Eq + e_rd + rr * 1024 + 128);
      float g;
      if constexpr (W2) g = act(fmaf(acc[q][0][r], a.out_scale, b0 + e0), m0, s0, h0) * act(fmaf(acc[q][1][r], a.out_scale, b1 + e1), m1, s1, h1);
      else g = act(acc[q][0][r] + b0 + e0, m0, s0, h0) * act(acc[q][1][r] + b1 + e1, m1, s1, h1);
      if (t0 + 128 * wm + 32 * q + 4 * lh + rr >= row_lim) g = 0.f;
      const uint16_t gh = ss_f2t<W2>(g);
      *reinterpret_cast<uint16_t*>(OUT + o_wr + rr * OROW) = gh;
      // second term; fp16x2: the gate output only ever feeds the matrix cores' A operand (hi term) - its second term is not written at all
      if constexpr (SPLIT == 1) *reinterpret_cast<uint16_t*>(OUT + o_wr + rr * OROW + 64) = f2bf(g - __builtin_bit_cast(float, (uint32_t)gh << 16));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): my staging writes are done
    __builtin_amdgcn_s_barrier();         // the staging tile is complete; everyone finished reading addend quarter q
    if (q + 2 < 4) dma_e((q & 1) ? EQ1 : EQ0, q + 2);
    if constexpr (SPLIT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // 64 rows x 512 B = 2048 pieces of 16 B, four per thread: piece p = (row p >> 5, 16 bytes p & 31 of the row's 512)
        const int p = tid + 512 * j;
        const int k = p >> 5, c16 = p & 31;
        const int grow = t0 + 128 * (k >> 5) + 32 * q + (k & 31);
        const uint4 v = *reinterpret_cast<const uint4*>(OUT + p * 16);
        const bool ok = (n0 >> 1) + 32 * (c16 >> 3) < a.N && !(W2 && (c16 & 4));   // N is a multiple of 32 (checked by the launcher); W2: the hi halves only
        const int off = ok ? grow * a.ldc * 2 + n0 * 2 + c16 * 16 : (int)0x80000000;   // logical channel n0/2 sits at physical element n0
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc_c, off, 0, 0);
      }
    } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // 64 rows x 256 B = 1024 pieces of 16 B, two per thread: piece p = (row p >> 4, 8 channels p & 15)
      const int p = tid + 512 * j;
      const int k = p >> 4, c8 = p & 15;
      const int grow = t0 + 128 * (k >> 5) + 32 * q + (k & 31);
      const uint4 v = *reinterpret_cast<const uint4*>(OUT + p * 16);
      const bool ok = (n0 >> 1) + 8 * c8 < a.N;   // N is a multiple of 8 (checked by the launcher)
      const int off = ok ? (grow * a.ldc + (n0 >> 1) + 8 * c8) * 2 : (int)0x80000000;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc_c, off, 0, 0);   // rows >= T dropped
    }
    }
  }
}

}  // namespace

// 1 if ss_gemm_bf16 should hand this GATE launch to the 256x256 kernel: three symmetric taps, dilation <= 8, K a multiple of 64,
// Np a multiple of 256, and enough rows that 256-row tiles fill the chip several times over
extern "C" int ss_gemm_bf16_gate256_ok(const ss_gemm_bf16_args* a) {
  if (!a || a->epi != SS_HEPI_GATE || a->ntaps != 3) return 0;
  const int d = a->tap_off[2];
  if (d < 1 || d > HALO || a->tap_off[0] != -d || a->tap_off[1] != 0) return 0;
  if (a->K != 256 || (a->Np % BN) != 0 || (a->lda % 8) != 0 || (a->N % 8) != 0 || (a->ldc % 8) != 0) return 0;
  if (a->split && ((a->N % 32) != 0 || a->lda < 2 * a->K || a->ldc < 2 * a->N)) return 0;
  // 32-bit offsets inside an item (the launcher asserts the same): longer items take the generic kernel
  if ((int64_t)a->T * a->lda * 2 >= (1ll << 31) || (int64_t)a->T * a->lde * 4 >= (1ll << 31) || (int64_t)a->T * a->ldc * 2 >= (1ll << 31)) return 0;
  const long tiles = (long)ss_cdiv(a->T, BM) * a->B * (a->Np / BN);
  return tiles >= 4l * ss_n_cu() ? 1 : 0;   // four rounds of one workgroup per CU (1024 tiles on the 256 CUs of an MI355X)
}

extern "C" int ss_gemm_bf16_gate256(const ss_gemm_bf16_args* args, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_gemm_bf16_gate256: null args");
  const ss_gemm_bf16_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "ss_gemm_bf16_gate256: null A/W/C");
  SS_CHECK_ARG(a.epi == SS_HEPI_GATE && a.ntaps == 3 && a.tap_off[1] == 0 && a.tap_off[0] == -a.tap_off[2] && a.tap_off[2] >= 1 &&
                   a.tap_off[2] <= HALO, "ss_gemm_bf16_gate256: GATE with taps (-d, 0, d), 1 <= d <= 8 only");
  SS_CHECK_ARG(a.K == 256 && (a.Np % BN) == 0 && 2 * a.N <= a.Np && (a.lda % 8) == 0 && (a.N % 8) == 0 && (a.ldc % 8) == 0,
               "ss_gemm_bf16_gate256: K = 256, Np %% 256, lda %% 8, N %% 8, ldc %% 8");
  SS_CHECK_ARG((int64_t)a.T * a.lda * 2 < (1ll << 31) && (int64_t)a.T * a.lde * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 2 < (1ll << 31) &&
                   (int64_t)a.Np * 3 * a.K * 4 < (1ll << 31), "ss_gemm_bf16_gate256: item too large for 32-bit offsets");
  SS_CHECK_ARG(a.split == 0 || ((a.split == 1 || a.split == 2) && (a.N % 32) == 0 && a.lda >= 2 * a.K && a.ldc >= 2 * a.N), "ss_gemm_bf16_gate256: split operands need N %% 32 == 0, lda >= 2 K, ldc >= 2 N");
  SS_CHECK_ARG(a.split != 2 || (a.out_scale > 0.f && a.out_scale <= 1.f), "ss_gemm_bf16_gate256: split = 2 needs 0 < out_scale <= 1");
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const int n_tiles = a.Np / BN;
  const int grid = ss_cdiv(m_tiles, 8) * 8 * n_tiles;
  // operands 144 KB; the split epilogue's view is 2 x 64 KB addend quarters + a 32 KB staging tile = all 160 KB of the CU
  const size_t lds = a.split ? (size_t)160 * 1024 : (size_t)2 * AROWS * ROWB + (size_t)2 * BN * ROWB;
  auto go = [&](auto kern) {
    // the attribute is per device and cheap: set on every launch (a process may drive several GPUs), failures are reported, not cached
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      ss_set_error("ss_gemm_bf16_gate256: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(e));
      return SS_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, a, m_tiles_per_item, m_tiles, n_tiles, a.tap_off[2]);
    return SS_OK;
  };
  // plain: 4 channel chunks of 64 x 3 taps = 12 steps; split: 8 chunks of 32 (both planes) x 3 taps = 24 steps
  SS_PROPAGATE(a.split == 2 ? go(&gate256_kernel<8, 2>) : a.split ? go(&gate256_kernel<8, 1>) : go(&gate256_kernel<4, 0>));
  SS_CHECK_LAUNCH("ss_gemm_bf16_gate256");
  return SS_OK;
}
