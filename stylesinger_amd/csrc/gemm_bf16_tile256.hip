// Split-operand ("bf16x2") 1-tap GEMMs of the denoiser loops for MANY-ROUND launches (BASELINE config 4: 180 000 - 360 000 rows per launch):
// the residual projection on the pair-only stream (SS_HEPI_RESX with X == NULL: K = N = C) and the K = L*C skip GEMM (SS_HEPI_STORE), on the
// 256 x 256 / 8-wave / LDS-DMA skeleton of gate256_kernel<split> (gemm_bf16_gate256.hip) instead of the generic 128 x 128 register-staged kernel.
//
// Why (tools/kbench_h.py --which res --split --pair-only, 180 000 rows): the generic kernel runs the split residual projection in 168 us for
// 553 MB of algorithmic traffic (3.3 TB/s) and 28 us of matrix time - neither bound: with 32-channel chunks its loop is 8 latency-bound
// iterations per tile (fetch -> registers -> ds_write -> barrier) and N = 256 is two column tiles, so the A operand crosses L2 -> LDS twice. Here
//   * one workgroup owns all N <= 256 columns of its 256 rows: A is staged once; both operands arrive by LDS-DMA (no staging registers, no
//     ds_write, no VALU in the loop), one raw s_barrier + s_waitcnt per 32-channel step, the next step's 8 DMA pieces per wave are issued between
//     the 16 MFMAs the previous step deferred past the barrier (same software pipeline as gate256_kernel<split>);
//   * epilogues go through LDS in four passes of 64 rows so that every thread owns CONSECUTIVE channels of a row: the (hi, mid) pairs of the
//     stream / the fp32 outputs move as 16-byte vectors, full lines per wave instruction.
// Arithmetic contract = gemm_bf16_kernel<..., SPLIT>: hi*hi + hi*mid + mid*hi of (hi, mid) bf16 pairs interleaved by 32 channels, fp32
// accumulation (the order of the K products inside an accumulator differs: results agree to fp32 rounding).
// W2 (ss_gemm_bf16_args.split = 2, "fp16x2"): fp16 terms, the A operand's second plane is neither fetched (dead DMA lanes write zeros) nor read, two products hi*hi + hi*lo per
// k-step (32 MFMAs per step, the 16 of the second k-step deferred), accumulators scaled by args.out_scale where the epilogue first touches them.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include "pair16.h"
#include <type_traits>

typedef ss_f32x16 f32x16;
typedef ss_bf16x8 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 256, BN = 256;
constexpr int ROWB = 128;                     // bytes per LDS row: 32 channels x (hi | mid)

__device__ __forceinline__ uint16_t f2bf(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

// ONE (with W2; "fp16sd", ss_gemm_bf16_args.one_product): a single fp16 weight term - the lo plane of the weights is neither fetched (dead DMA lanes, as the
// A operand's second plane) nor read, 16 MFMAs per step instead of 32 (the 8 of the second k-step deferred past the barrier with the DMA pieces between them)
// DENSE (with ONE; both operands compact: a_compact and one_product = 2, K %% 128 == 0): nothing but single fp16 terms on either side, so a 128-byte LDS
// row holds 64 CHANNELS (two chunks side by side) instead of 32 channels and a dead plane: every DMA lane fetches, a step is four k-steps of one product
// (32 MFMAs, the 16 of its second half deferred past the barrier), and the K loop has half the steps, barriers and DMA instructions.
template <int EPI, bool W2, bool ONE = false, bool DENSE = false>
__global__ __launch_bounds__(512, 2) void tile256s_kernel(const ss_gemm_bf16_args a, int m_tiles_per_item, int m_tiles, int kchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_t256[];   // 128 KB: [A0 32 K][B0 32 K][A1 32 K][B1 32 K]; epilogue: 2 x 64 KB staging
  char* const A0 = smem_t256;
  char* const B0 = A0 + BM * ROWB;
  char* const A1 = B0 + BN * ROWB;
  char* const B1 = A1 + BM * ROWB;

  const int mt = blockIdx.x;
  if (mt >= m_tiles) return;
  const int b = mt / m_tiles_per_item;
  const int t0 = (mt % m_tiles_per_item) * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, lh = lane >> 5;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const bool w_compact = ONE && a.one_product == 2;       // the one-term pack without its zero second plane: rows of K fp16
  const int ldw = w_compact ? a.K : 2 * a.K;              // 16-bit elements per packed weight row (both planes, one tap)
  const int b_chunk = (w_compact && !DENSE) ? ROWB / 2 : ROWB;

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

  // DMA roles: 32 pieces of 8 rows x 128 B per operand and step; wave w issues pieces w, w + 8, w + 16, w + 24 of both. Lane i of a piece
  // lands at (row i >> 3, physical slot i & 7) and fetches logical slot (i & 7) ^ ((row >> 1) & 7) (64 j more rows leave the swizzle unchanged).
  const int r0 = 8 * wave + (lane >> 3);
  const int slot0 = (lane & 7) ^ ((r0 >> 1) & 7);
  const int a_voff = ((t0 + r0) * a.lda + slot0 * 8) * 2;   // rows >= len are out of range: the DMA writes zeros
  const int b_voff = (r0 * ldw + slot0 * 8) * 2;            // packed weight rows >= Np read zeros
  const int a_lo_dead = (W2 && !DENSE && slot0 >= 4) ? (int)0x80000000 : 0;   // W2: the A operand's second plane is never read - its lanes fetch nothing (zeros)
  const int b_lo_dead = (ONE && !DENSE && slot0 >= 4) ? (int)0x80000000 : 0;  // ONE: nor is the weights'
  // a_compact (W2, STORE): a row of A holds its hi terms only - chunk c at 64 c bytes of the row instead of 128 c (the LDS image keeps its 128-byte
  // rows, the dead lanes of the second plane write zeros as before): half the bytes the launch pulls from HBM
  const int a_chunk = (W2 && a.a_compact && !DENSE) ? ROWB / 2 : ROWB;   // (DENSE: a step takes 128 contiguous bytes = 64 channels of a compact row)
  auto piece = [&](char* Ab, char* Bb, int c, int i) {     // i = 0..3: A pieces, 4..7: B pieces of chunk c
    const int j = i & 3;
    if (i < 4) glds16(rsrc_a, Ab + (wave + 8 * j) * 8 * ROWB, (a_voff + 64 * j * a.lda * 2) | a_lo_dead, c * a_chunk);
    else glds16(rsrc_w, Bb + (wave + 8 * j) * 8 * ROWB, b_voff | b_lo_dead, c * b_chunk + 64 * j * ldw * 2);
  };

  // fragment addresses (see gate256_kernel): row = 128 wm + 32 m + l31 for A, 64 wn + 32 n + l31 for B; slot (2 ks) ^ swz = hi, (4 + 2 ks) ^ swz = mid
  const int a_base = (128 * wm + l31) * ROWB, a_swz = (((128 * wm + l31) >> 1) & 7) ^ lh;
  const int b_base = (64 * wn + l31) * ROWB, b_swz = (((64 * wn + l31) >> 1) & 7) ^ lh;

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // the two product groups (hi x mid, hi x hi of the second k-step) a step defers past the next barrier; zero fragments before the first step
  bf16x8 p_ah[4], p_bh[2], p_bm[2];
  [[maybe_unused]] bf16x8 p_a6[4];   // DENSE: the A fragments of channels 48..63 (p_ah: 32..47, p_bh / p_bm: the weights' of the same two k-steps)
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p_ah[m][e] = (__bf16)0.f;
      if constexpr (DENSE) p_a6[m][e] = (__bf16)0.f;
    }
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      p_bh[n][e] = (__bf16)0.f;
      p_bm[n][e] = (__bf16)0.f;
    }
  auto mfma8 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[2]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = ss_mfma_32x32x16<W2>(fa[m], fb[n], acc[m][n]);
  };
  auto step = [&](const char* Ac, const char* Bc, char* An, char* Bn, int c, bool more) {
    wait_vmcnt<0>();                  // my pieces of chunk c have landed (nothing younger is in flight)
    __builtin_amdgcn_s_barrier();     // everyone's have; everyone finished reading chunk c-1's buffers
    auto rd_a = [&](int slot, bf16x8 (&f)[4]) {
      const int ao = a_base + ((slot ^ a_swz) << 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) f[m] = *reinterpret_cast<const bf16x8*>(Ac + ao + m * 32 * ROWB);
    };
    auto rd_b = [&](int slot, bf16x8 (&f)[2]) {
      const int bo = b_base + ((slot ^ b_swz) << 4);
#pragma unroll
      for (int n = 0; n < 2; ++n) f[n] = *reinterpret_cast<const bf16x8*>(Bc + bo + n * 32 * ROWB);
    };
    [[maybe_unused]] bf16x8 am0[4], ah0[4];
    bf16x8 bh0[2];
    if constexpr (W2) rd_a(0, ah0);
    else rd_a(4, am0);
    rd_b(0, bh0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DENSE) {            // 64 channels per step: the previous step's channels 32..63 (two deferred groups) first, a DMA piece after each of the first 8
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i >> 1) & 3, n = i & 1;
        acc[m][n] = ss_mfma_32x32x16<true>(i < 8 ? p_ah[m] : p_a6[m], i < 8 ? p_bh[n] : p_bm[n], acc[m][n]);
        if (i < 8) {
          __builtin_amdgcn_sched_barrier(0);
          if (more) piece(An, Bn, c + 1, i);   // wave-uniform branch
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 a2[4], b2[2];
      rd_a(2, a2);
      rd_b(2, b2);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(ah0, bh0);                // channels 0..15
      __builtin_amdgcn_sched_barrier(0);
      rd_a(4, p_ah);
      rd_b(4, p_bh);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(a2, b2);                  // channels 16..31; 32..47 (p_ah x p_bh) and 48..63 (p_a6 x p_bm) after the next barrier
      __builtin_amdgcn_sched_barrier(0);
      rd_a(6, p_a6);
      rd_b(6, p_bm);
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    if constexpr (ONE) {              // one weight term: the 8 MFMAs of the second k-step deferred by the previous step, a DMA piece after each
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = (i >> 1) & 3, n = i & 1;
        acc[m][n] = ss_mfma_32x32x16<true>(p_ah[m], p_bh[n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) piece(An, Bn, c + 1, i);   // wave-uniform branch
        __builtin_amdgcn_sched_barrier(0);
      }
      mfma8(ah0, bh0);                // k-step 0 now; the second k-step's group (p_ah x p_bh) after the next barrier
      __builtin_amdgcn_sched_barrier(0);
      rd_a(2, p_ah);
      rd_b(2, p_bh);
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {    // the 16 MFMAs deferred by the previous step, one DMA piece of the next chunk after each of the first 8
      const int m = (i >> 1) & 3, n = i & 1;
      acc[m][n] = ss_mfma_32x32x16<W2>(p_ah[m], i < 8 ? p_bm[n] : p_bh[n], acc[m][n]);
      if (i < 8) {
        __builtin_amdgcn_sched_barrier(0);
        if (more) piece(An, Bn, c + 1, i);   // wave-uniform branch
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (W2) {   // hi x hi, hi x lo of k-step 0 now; the second k-step's two groups (p_ah x p_bm, p_ah x p_bh) after the next barrier
      bf16x8 bm0[2];
      rd_b(4, bm0);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(ah0, bh0);
      __builtin_amdgcn_sched_barrier(0);
      rd_a(2, p_ah);
      rd_b(2, p_bh);
      __builtin_amdgcn_sched_barrier(0);
      mfma8(ah0, bm0);
      __builtin_amdgcn_sched_barrier(0);
      rd_b(6, p_bm);
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    bf16x8 bm0[2], am1[4], bh1[2];
    rd_a(0, ah0);
    rd_b(4, bm0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(am0, bh0);
    __builtin_amdgcn_sched_barrier(0);
    rd_a(6, am1);
    rd_b(2, bh1);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(ah0, bm0);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(ah0, bh0);
    __builtin_amdgcn_sched_barrier(0);
    rd_a(2, p_ah);
    rd_b(6, p_bm);
    __builtin_amdgcn_sched_barrier(0);
    mfma8(am1, bh1);
#pragma unroll
    for (int n = 0; n < 2; ++n) p_bh[n] = bh1[n];
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(A0, B0, 0, i);
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < kchunks; c += 2) {   // kchunks is even (checked by the launcher)
    step(A0, B0, A1, B1, c, true);
    step(A1, B1, A0, B0, c + 1, c + 2 < kchunks);
  }
  if constexpr (!ONE) mfma8(p_ah, p_bm);
  mfma8(p_ah, p_bh);
  if constexpr (DENSE) mfma8(p_a6, p_bm);   // channels 48..63 of the last step, after its 32..47: every accumulator takes its products in channel order

  // ---- epilogue: four passes of 64 rows (accumulator block m = q of every wave: tile rows 128 wm + 32 q + (0..31) -> staging row 32 wm + ..),
  // staged as fp32 [64][256] in alternating 64-KB halves of the operand memory, then processed row-contiguously
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
  __builtin_amdgcn_s_barrier();   // everyone is done reading the operand buffers
  const int st_wr = (32 * wm + 4 * lh) * (BN * 4) + (64 * wn + l31) * 4;   // + rr * BN * 4 (+ 128 for n = 1)
  if constexpr (EPI == SS_HEPI_STORE) {
    float* Cb = (float*)a.C + (int64_t)b * a.c_batch_stride;
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Cb), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);
    const int c4 = (tid & 63) * 4;                 // this thread's 4 columns in every row it handles
    const int dead = c4 < a.N ? 0 : (int)0x80000000;   // N is a multiple of 4
    float bs[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bs[e] = (biasg && !dead) ? biasg[c4 + e] : 0.f;
    const bool relu = a.act == SS_ACT_RELU;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      char* St = smem_t256 + (q & 1) * 64 * 1024;
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(St + st_wr + ((r & 3) + 8 * (r >> 2)) * (BN * 4) + n * 128) = acc[q][n][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): my staging writes are done
      __builtin_amdgcn_s_barrier();         // the staging tile of pass q is complete (pass q-1's tile, the other half, is being read at most)
#pragma unroll
      for (int j = 0; j < 8; ++j) {         // 64 rows x 64 float4 = 4096 pieces, eight per thread: piece p = (row p >> 6, columns 4 (p & 63))
        const int k = (tid >> 6) + 8 * j;   // staging row of piece tid + 512 j
        const int grow = t0 + 128 * (k >> 5) + 32 * q + (k & 31);
        float4 v = *reinterpret_cast<const float4*>(St + k * (BN * 4) + c4 * 4);
        if constexpr (W2) v = make_float4(fmaf(v.x, a.out_scale, bs[0]), fmaf(v.y, a.out_scale, bs[1]), fmaf(v.z, a.out_scale, bs[2]), fmaf(v.w, a.out_scale, bs[3]));
        else { v.x += bs[0]; v.y += bs[1]; v.z += bs[2]; v.w += bs[3]; }
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (grow >= row_lim) v = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_c, (grow * a.ldc + c4) * 4 | dead, 0, 0);   // rows >= T dropped
      }
      // (no barrier here: pass q+1 writes the OTHER half, and pass q+2's writes to this half come after pass q+1's barrier, which every
      // thread reaches only after its pass-q reads)
    }
  } else {   // SS_HEPI_RESX on the pair-only stream: Y = pair(x + cur_bias) is read, x updated, Y = pair(x_new + next_bias) rewritten in place
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.Y + (int64_t)b * a.y_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldy * 2)), 0x00020000);
    const float* nbg = a.next_bias ? a.next_bias + (int64_t)grp_w * a.next_bias_group_stride : nullptr;
    const float* cbg = a.cur_bias + (int64_t)grp_w * a.cur_bias_group_stride;
    const int g8 = tid & 31, col0 = g8 * 8;         // this thread's 8 channels (N is a multiple of 32: valid or dead as a whole)
    const int dead = col0 < a.N ? 0 : (int)0x80000000;
    float bs[8], nb[8], cb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bs[e] = (biasg && !dead) ? biasg[col0 + e] : 0.f;
      nb[e] = (nbg && !dead) ? nbg[col0 + e] : 0.f;
      cb[e] = !dead ? cbg[col0 + e] : 0.f;
    }
    const int phys = (col0 >> 5) * 64 + (col0 & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      char* St = smem_t256 + (q & 1) * 64 * 1024;
      // the stream's pairs of this pass are fetched before the staging barrier: their latency hides under it
      u32x4 hv[4], mv[4];
      int yo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {   // 64 rows x 32 groups = 2048 items, four per thread: item (row p >> 5, group p & 31), p = tid + 512 j
        const int k = (tid >> 5) + 16 * j;
        const int grow = t0 + 128 * (k >> 5) + 32 * q + (k & 31);
        yo[j] = (grow * a.ldy + phys) * 2 | dead;
        hv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, yo[j], 0, 0);
        mv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, yo[j], 64, 0);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(St + st_wr + ((r & 3) + 8 * (r >> 2)) * (BN * 4) + n * 128) = acc[q][n][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = (tid >> 5) + 16 * j;
        const int grow = t0 + 128 * (k >> 5) + 32 * q + (k & 31);
        const float4 a0 = *reinterpret_cast<const float4*>(St + k * (BN * 4) + col0 * 4), a1 = *reinterpret_cast<const float4*>(St + k * (BN * 4) + col0 * 4 + 16);
        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const bool pad = grow >= row_lim;
        u32x4 ho, mo;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          uint32_t hp = 0, mp = 0;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int e = 2 * e2 + kk;
            float hf, mf, xn;
            if constexpr (W2) {
              hf = ss_t2f_packed<true>(hv[j][e2], kk);
              mf = ss_t2f_packed<true>(mv[j][e2], kk);
              xn = (((hf + mf) - cb[e]) + fmaf(av[e], a.out_scale, bs[e])) * a.post_scale;
            } else {
              hf = __builtin_bit_cast(float, kk ? (hv[j][e2] & 0xffff0000u) : (hv[j][e2] << 16));
              mf = __builtin_bit_cast(float, kk ? (mv[j][e2] & 0xffff0000u) : (mv[j][e2] << 16));
              xn = (((hf + mf) - cb[e]) + (av[e] + bs[e])) * a.post_scale;
            }
            const float yv = pad ? 0.f : xn + nb[e];
            uint16_t yh, ym;
            if constexpr (W2) {
              yh = ss_f2t<true>(yv);
              ym = ss_f2t<true>(yv - ss_t2f<true>(yh));
            } else {
              yh = f2bf(yv);
              ym = f2bf(yv - bf2f(yh));
            }
            hp |= (uint32_t)yh << (16 * kk);
            mp |= (uint32_t)ym << (16 * kk);
          }
          ho[e2] = hp;
          mo[e2] = mp;
        }
        __builtin_amdgcn_raw_buffer_store_b128(ho, rsrc_y, yo[j], 0, 0);    // rows >= T: out of range, dropped
        __builtin_amdgcn_raw_buffer_store_b128(mo, rsrc_y, yo[j], 64, 0);
      }
    }
  }
}

// (round 4 measured a variant of the long-K STORE GEMM with its A operand prefetched two chunks ahead - three A buffers, 160 KB of LDS: 1070.9 ->
// 1053.1 us back to back, profiles/r04_kbench_skip_deep.log; not latency-bound. Removed in round 5 together with its "skip_deep" knob.)

}  // namespace

// 1 if ss_gemm_bf16 should hand this launch to the 256-row kernel: split operands, one tap, STORE or RESX on the pair-only stream, N <= 256,
// an even number of 32-channel chunks, and at least two rounds of 256-row tiles
extern "C" int ss_gemm_bf16_tile256_ok(const ss_gemm_bf16_args* a) {
  if (!a || (a->split != 1 && a->split != 2) || a->ntaps != 1 || a->tap_off[0] != 0) return 0;
  if (a->epi == SS_HEPI_RESX ? !(a->X == nullptr && a->Y && a->cur_bias && (a->N % 32) == 0 && a->ldy >= 2 * a->N) : a->epi != SS_HEPI_STORE) return 0;
  if (a->epi == SS_HEPI_STORE && ((a->N % 4) != 0 || (a->ldc % 4) != 0 || (a->act != SS_ACT_NONE_ && a->act != SS_ACT_RELU_))) return 0;
  if ((a->a_compact || a->one_product == 2) && !(a->split == 2 && a->epi == SS_HEPI_STORE)) return 0;
  if (a->N > BN || (a->K % 64) != 0 || a->lda < (a->a_compact ? 1 : 2) * a->K || (a->lda % 8) != 0) return 0;
  if ((int64_t)a->T * a->lda * 2 >= (1ll << 31) || (int64_t)a->T * a->ldc * 4 >= (1ll << 31) || (int64_t)a->T * a->ldy * 2 >= (1ll << 31) ||
      (int64_t)a->Np * a->K * 4 >= (1ll << 31)) return 0;
  return (long)ss_cdiv(a->T, BM) * a->B >= 2L * ss_n_cu() ? 1 : 0;
}

extern "C" int ss_gemm_bf16_tile256(const ss_gemm_bf16_args* args, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_gemm_bf16_tile256: null args");
  const ss_gemm_bf16_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && (a.split == 1 || a.split == 2) && a.ntaps == 1 && a.tap_off[0] == 0, "ss_gemm_bf16_tile256: split operands, one tap at offset 0");
  SS_CHECK_ARG(a.split != 2 || (a.out_scale > 0.f && a.out_scale <= 1.f), "ss_gemm_bf16_tile256: split = 2 needs 0 < out_scale <= 1");
  SS_CHECK_ARG(!(a.a_compact || a.one_product == 2) || (a.split == 2 && a.epi == SS_HEPI_STORE),
               "ss_gemm_bf16_tile256: a_compact / one_product = 2 are the fp16 (split = 2) STORE form only");
  SS_CHECK_ARG(a.one_product >= 0 && a.one_product <= 2, "ss_gemm_bf16_tile256: one_product = 0 | 1 | 2");
  SS_CHECK_ARG(a.N > 0 && a.N <= BN && a.Np >= a.N && (a.K % 64) == 0 && a.lda >= (a.a_compact ? 1 : 2) * a.K && (a.lda % 8) == 0,
               "ss_gemm_bf16_tile256: N <= 256, K %% 64 == 0, lda >= 2 K (K with a_compact)");
  SS_CHECK_ARG((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0 && (a.a_batch_stride & 7) == 0, "ss_gemm_bf16_tile256: A/W must be 16-byte aligned");
  SS_CHECK_ARG((int64_t)a.T * a.lda * 2 < (1ll << 31) && (int64_t)a.Np * a.K * 4 < (1ll << 31), "ss_gemm_bf16_tile256: item too large for 32-bit offsets");
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const size_t lds = (size_t)128 * 1024;
  // both operands compact and K a multiple of 128: 64 channels per step (tile256s_kernel<.., DENSE>); knob "skip_dense" = 0 keeps 32-channel steps (A/B)
  const bool dense = a.epi == SS_HEPI_STORE && a.split == 2 && a.a_compact && a.one_product == 2 && (a.K % 128) == 0 && g_ss_tuning.skip_dense != 0;
  auto go = [&](auto kern) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      ss_set_error("ss_gemm_bf16_tile256: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(e));
      return SS_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(m_tiles), dim3(512), lds, (hipStream_t)stream, a, m_tiles_per_item, m_tiles, a.K / (dense ? 64 : 32));
    return SS_OK;
  };
  if (a.epi == SS_HEPI_STORE) {
    SS_CHECK_ARG(a.C && (a.N % 4) == 0 && (a.ldc % 4) == 0 && (int64_t)a.T * a.ldc * 4 < (1ll << 31) && (a.act == SS_ACT_NONE_ || a.act == SS_ACT_RELU_),
                 "ss_gemm_bf16_tile256: STORE needs C, N %% 4 == 0, ldc %% 4 == 0, act none | relu");
    SS_PROPAGATE(a.split == 2 ? (dense ? go(&tile256s_kernel<SS_HEPI_STORE, true, true, true>) : a.one_product ? go(&tile256s_kernel<SS_HEPI_STORE, true, true>) : go(&tile256s_kernel<SS_HEPI_STORE, true>))
                              : go(&tile256s_kernel<SS_HEPI_STORE, false>));
  } else {
    SS_CHECK_ARG(a.epi == SS_HEPI_RESX && a.X == nullptr && a.Y && a.cur_bias && (a.N % 32) == 0 && a.ldy >= 2 * a.N && (a.ldy % 8) == 0 &&
                     (int64_t)a.T * a.ldy * 2 < (1ll << 31), "ss_gemm_bf16_tile256: RESX on the pair-only stream (X = NULL, Y, cur_bias), N %% 32 == 0");
    SS_PROPAGATE(a.split == 2 ? go(&tile256s_kernel<SS_HEPI_RESX, true>) : go(&tile256s_kernel<SS_HEPI_RESX, false>));
  }
  SS_CHECK_LAUNCH("ss_gemm_bf16_tile256");
  return SS_OK;
}
