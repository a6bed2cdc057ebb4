// bf16-operand implicit-GEMM conv for the denoisers' hidden layers (BASELINE config 4: "bf16 MFMA").
//
//   out[b][t][n] = epi( sum_j sum_k A[b][t + tap_off[j]][k] * W[n][j][k] )        A, W: bf16 in HBM; fp32 accumulate
//
// Why a second GEMM kernel (the fp32 kernel has a BF16 template mode, conv_gemm_kernel.h): in that mode both operands travel
// as fp32 (HBM -> LDS) and every wave rounds its own fragments on the way into the matrix core. v_mfma_f32_32x32x16_bf16 is
// 16x faster than the fp32 form while a VALU instruction still costs ~2.8 SIMD cycles (tools/ubench/mfma_valu.hip), so those
// conversions + the fp32 prologue (12 VALU per MFMA) bound the loop: 310 TF/s = 12 % of the bf16 roof at the C4 shape.
// Here every operand is rounded ONCE where it is produced (weights at pack time: the "bf16 weight copies" of SURVEY.md §8f-3;
// activations in the epilogue of the kernel that writes them) and stored as bf16 in HBM:
//   * K chunks of 64 bf16 = the same 128-byte LDS rows / 16-byte-slot XOR swizzle as the fp32 kernel; staging is a plain
//     16-byte copy (no VALU at all), fragments are one ds_read_b128 = 8 bf16 = one MFMA operand;
//   * fetch addresses = per-thread VGPR offset + wave-uniform SGPR offset; two K chunks in flight in registers (the MFMA
//     phase of a chunk is only ~512 cycles per wave); chunk pairs with compile-time LDS buffer index;
//   * epilogues: GATE (conditioner addend, sigmoid*tanh, bf16 out), RESX (x <- (x + y)/sqrt(2) in fp32 AND the next layer's
//     operand bf16(x + dstep_next)), STORE (bias, act, fp32 out).
// Arithmetic contract = oracle/restatement.py with set_matmul_rounding("bf16"): RNE rounding of both matmul operands, exact
// products, fp32 accumulation; everything outside the matmuls fp32.
#include "common.h"
#include <stdlib.h>
#include "../../include/stylesinger_hip.h"
#include "pair16.h"
#include <type_traits>

typedef ss_f32x16 f32x16;
typedef ss_bf16x8 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BKH = 64;  // bf16 per K chunk
constexpr int LDH = 64;  // bf16 per LDS row (128 bytes)

// byte offset of 16-byte slot `slot` of row `row` (slot ^ ((row >> 1) & 7): conflict-free ds_read_b128 / ds_write_b128, see
// conv_gemm_kernel.h)
__device__ __forceinline__ int lds_off(int row, int slot) { return row * (LDH * 2) + ((slot ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ uint16_t f2bf(float x) {  // RNE, like torch's .bfloat16()
  return __builtin_bit_cast(uint16_t, (__bf16)x);
}

__device__ __forceinline__ float bf2f(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }

// SS_HABL (debug builds only, tools/ablate_h.sh): timing ablations. 1 = no global fetches in the loop, 2 = no MFMAs,
// 3 = no addend loads in the epilogue, 4 = no epilogue stores. Results are wrong by design.
#ifndef SS_HABL
#define SS_HABL 0
#endif

// SPLIT = 1 ("bf16x2"): operands are (hi, mid) bf16 pairs interleaved by 32 channels - a 128-byte K chunk holds 32 channels of BOTH planes
// (slots 0-3 hi, 4-7 mid), the fetch / staging code is the same, and a chunk feeds 2 k-steps x 3 products (mid*hi, hi*mid, hi*hi) instead of 4 x 1.
// SPLIT = 2 ("fp16x2"): the same layouts with fp16 terms; the A operand's second plane is neither fetched nor read (2 products: hi*lo, hi*hi) and the
// accumulator is scaled by args.out_scale (the weights carry a power-of-two shift, pair16.h) before anything is added to it.
template <int BM, int BN, int EPI, int SPLIT>
__global__ __launch_bounds__(256, (BM >= 128 ? 2 : 3)) void gemm_bf16_kernel(const ss_gemm_bf16_args a, int m_tiles_per_item, int m_tiles, int n_tiles) {
  constexpr int WM = 2, WN = 2;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int AP = BM / 32, BP = BN / 32;  // 16-byte loads per thread per chunk (256 threads x 16 B = 32 rows x 128 B per pass)
  static_assert(EPI != SS_HEPI_GATE || TN % 2 == 0, "GATE pairs 32-column blocks");
  extern __shared__ __attribute__((aligned(16))) char smem_h[];
  char* As = smem_h;                        // [2][BM][128 B]
  char* Bs = smem_h + 2 * BM * (LDH * 2);   // [2][BN][128 B]

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
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  constexpr bool W2 = SPLIT == 2;        // fp16 terms, weights-only split
  [[maybe_unused]] const float osc = a.out_scale;
  constexpr int KCH = SPLIT ? 32 : 64;   // channels per 128-byte K chunk
  const int nchunks_tap = a.K / KCH;
  const int nchunks = a.ntaps * nchunks_tap;
  const int ldw = a.ntaps * a.K * (SPLIT ? 2 : 1);  // bf16 per packed weight row

  auto uniform_ptr = [](const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  };
  // range-checked fetches: rows outside [0, len) of the item read 0 (= the conv's zero padding), packed rows beyond Np read 0
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.W + (int64_t)grp_w * a.w_group_stride), 0, __builtin_amdgcn_readfirstlane(a.Np * ldw * 2), 0x00020000);

  const int st_slot = tid & 7, st_row = tid >> 3;  // 8 threads x 16 B cover one 128-byte chunk row; 32 rows per pass
  int a_voff[AP], w_voff[BP], a_lds[AP], b_lds[BP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    a_voff[i] = ((t0 + st_row + i * 32) * a.lda + st_slot * 8) * 2;
    a_lds[i] = lds_off(st_row + i * 32, st_slot);
  }
#pragma unroll
  for (int i = 0; i < BP; ++i) {
    w_voff[i] = ((n0 + st_row + i * 32) * ldw + st_slot * 8) * 2;
    b_lds[i] = lds_off(st_row + i * 32, st_slot);
  }
  const int lda2 = a.lda * 2;
  [[maybe_unused]] const int a_lo_dead = (W2 && st_slot >= 4) ? (int)0x80000000 : 0;   // SPLIT = 2: the A operand's second plane is never read - not fetched either (zeros)

  // chunk c = (tap, k0): SGPR byte offsets of the A fetch (tap row shift + channel offset) and of the W fetch
  auto a_soff = [&](int c) {
    const int tap = c / nchunks_tap, cc = c - tap * nchunks_tap;
    return a.tap_off[tap] * lda2 + cc * (BKH * 2);
  };
  u32x4 ra[2][AP], rb[2][BP];  // two register stages
  auto fetch = [&](auto st_tag, int c, int dead) {
    constexpr int ST = decltype(st_tag)::value;
    // a negative tap shift must stay in the VGPR offset (the range check has to see the row): one v_add per load, only for taps
    const int so = a_soff(c);
#if SS_HABL == 1
    if (c > 1) return;
#endif
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      if constexpr (W2) ra[ST][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (a_voff[i] + so) | dead | a_lo_dead, 0, 0);
      else ra[ST][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (a_voff[i] + so) | dead, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BP; ++i) rb[ST][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[i] | dead, c * (BKH * 2), 0);
  };
  auto stage = [&](auto st_tag, auto buf_tag) {
    constexpr int ST = decltype(st_tag)::value;
    constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<u32x4*>(As + BUF * BM * (LDH * 2) + a_lds[i]) = ra[ST][i];
#pragma unroll
    for (int i = 0; i < BP; ++i) *reinterpret_cast<u32x4*>(Bs + BUF * BN * (LDH * 2) + b_lds[i]) = rb[ST][i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const int swz = (l31 >> 1) & 7;
  const int a_row = (wm * 32 * TM + l31) * (LDH * 2);
  const int b_row = (wn * 32 * TN + l31) * (LDH * 2);
  auto compute = [&](auto buf_tag) {
    constexpr int BUF = decltype(buf_tag)::value;
    const char* Ac = As + BUF * BM * (LDH * 2) + a_row;
    const char* Bc = Bs + BUF * BN * (LDH * 2) + b_row;
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {  // 2 k-steps of 16 channels: slots 2 ks + lh (hi) and 4 + 2 ks + lh (mid) of the row
        const int sh = ((2 * ks + lh) ^ swz) << 4, sm = ((4 + 2 * ks + lh) ^ swz) << 4;
        [[maybe_unused]] bf16x8 am[TM];
        bf16x8 ah[TM], bh[TN], bm[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          ah[m] = *reinterpret_cast<const bf16x8*>(Ac + m * 32 * (LDH * 2) + sh);
          if constexpr (!W2) am[m] = *reinterpret_cast<const bf16x8*>(Ac + m * 32 * (LDH * 2) + sm);
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          bh[n] = *reinterpret_cast<const bf16x8*>(Bc + n * 32 * (LDH * 2) + sh);
          bm[n] = *reinterpret_cast<const bf16x8*>(Bc + n * 32 * (LDH * 2) + sm);
        }
        // product outermost: consecutive MFMAs write different accumulators
        if constexpr (!W2) {
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[m], bh[n], acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n) acc[m][n] = ss_mfma_32x32x16<W2>(ah[m], bm[n], acc[m][n]);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n) acc[m][n] = ss_mfma_32x32x16<W2>(ah[m], bh[n], acc[m][n]);
      }
    } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {  // 4 k-steps of 16: lane (l31, lh) feeds k = 16*ks + 8*lh .. +8 of its row (same for A and B)
      const int so = ((2 * ks + lh) ^ swz) << 4;
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) af[m] = *reinterpret_cast<const bf16x8*>(Ac + m * 32 * (LDH * 2) + so);
#pragma unroll
      for (int n = 0; n < TN; ++n) bf[n] = *reinterpret_cast<const bf16x8*>(Bc + n * 32 * (LDH * 2) + so);
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
#if SS_HABL == 2
          asm volatile("" ::"v"(af[m]), "v"(bf[n]));
#else
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
#endif
        }
    }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // chunk j lives in register stage j & 1 and LDS buffer j & 1. Iteration c: fetch chunk c+2 into stage c&1 (its old content,
  // chunk c, went to LDS one iteration ago), multiply chunk c from LDS, then move chunk c+1 (stage (c+1)&1) to LDS.
  fetch(I0{}, 0, 0);
  stage(I0{}, I0{});
  if (nchunks > 1) fetch(I1{}, 1, 0);
  __syncthreads();
  auto body = [&](auto par_tag, int c) {
    constexpr int P = decltype(par_tag)::value;
    const bool more = c + 2 < nchunks;
    fetch(std::integral_constant<int, P>{}, more ? c + 2 : 0, more ? 0 : (int)0x80000000);  // unconditional: branch-free body
    __builtin_amdgcn_sched_barrier(0);
    compute(std::integral_constant<int, P>{});
    __builtin_amdgcn_sched_barrier(0);
    stage(std::integral_constant<int, P ^ 1>{}, std::integral_constant<int, P ^ 1>{});
    __syncthreads();
  };
  // the fetch above overwrites stage P while ... chunk c (stage P) is already in LDS: safe. But stage(P^1) must read chunk c+1,
  // fetched one iteration earlier into stage P^1: also safe.
  int c = 0;
  for (; c + 2 < nchunks; c += 2) {
    body(I0{}, c);
    body(I1{}, c + 1);
  }
  // GATE: the conditioner addend (fp32, one HBM miss per element) is fetched under the last two chunks' MFMAs, into the registers the
  // fetch stages no longer need; with the loads in the epilogue every workgroup of a round stalled on them (tools/ablate_h.sh: 56 of 276 us).
  const int row_base = t0 + wm * 32 * TM + 4 * lh;
  const int col_base = n0 + wn * 32 * TN;
  [[maybe_unused]] float pe0[TM][TN / 2 > 0 ? TN / 2 : 1][16], pe1[TM][TN / 2 > 0 ? TN / 2 : 1][16];
  [[maybe_unused]] auto prefetch_e = [&](int m) {
    if constexpr (EPI == SS_HEPI_GATE) {
      const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
      const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
          uniform_ptr(Eb ? (const void*)Eb : (const void*)a.W), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
      const int lde4 = a.lde * 4;
#pragma unroll
      for (int n = 0; n < TN; n += 2) {
        const int pc0 = col_base + n * 32 + l31;
        const int oc = (pc0 >> 6) * 32 + l31;
        const int dead = oc < a.N ? 0 : (int)0x80000000;
        const int eoff = ((row_base + m * 32) * a.lde + pc0) * 4 | dead;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = ((r & 3) + 8 * (r >> 2)) * lde4;
#if SS_HABL == 3
          pe0[m][n / 2][r] = (float)ro;
          pe1[m][n / 2][r] = (float)eoff;
#else
          pe0[m][n / 2][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, eoff + ro, 0, 0));
          pe1[m][n / 2][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, eoff + ro, 128, 0));
#endif
        }
      }
    }
  };
  if (c + 1 < nchunks) {  // one body left (nchunks even): chunk c is in buffer 0, chunk c+1 in stage 1
    prefetch_e(0);
    __builtin_amdgcn_sched_barrier(0);
    compute(I0{});
    stage(I1{}, I1{});
    __syncthreads();
    if constexpr (TM > 1) prefetch_e(1);
    __builtin_amdgcn_sched_barrier(0);
    compute(I1{});
  } else {
    prefetch_e(0);
    if constexpr (TM > 1) prefetch_e(1);
    __builtin_amdgcn_sched_barrier(0);
    compute(I0{});  // nchunks odd: the last chunk sits in buffer 0
  }

  // ---------------------------------------------------------------- epilogue (C/D layout: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*lh)
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  if constexpr (EPI == SS_HEPI_STORE) {
    float* Cb = (float*)a.C + (int64_t)b * a.c_batch_stride;
    const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = col_base + n * 32 + l31;
      if (col >= a.N) continue;
      const float bs = biasg ? biasg[col] : 0.f;
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_base + m * 32 + (r & 3) + 8 * (r >> 2);
          if (row >= a.T) continue;
          float v = ss_apply_act(W2 ? fmaf(acc[m][n][r], osc, bs) : acc[m][n][r] + bs, a.act, 0.f);
          if (row >= row_lim) v = 0.f;
          Cb[(int64_t)row * a.ldc + col] = v;
        }
    }
  } else if constexpr (EPI == SS_HEPI_GATE) {
    // addend already in registers (prefetch_e above); 32-bit buffer addressing for the bf16 stores
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr((uint16_t*)a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 2)), 0x00020000);
    const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
    const bool sig_first = a.gate_mode == 0;
    const float m0 = sig_first ? -1.0f : -2.0f, s0 = sig_first ? 1.0f : 2.0f, h0 = sig_first ? 0.0f : -1.0f;
    const float m1 = sig_first ? -2.0f : -1.0f, s1 = sig_first ? 2.0f : 1.0f, h1 = sig_first ? -1.0f : 0.0f;
    auto act = [](float x, float mul, float sc, float sh) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __expf(x * mul)), sc, sh); };
    const int ldc2 = a.ldc * 2;
#pragma unroll
    for (int n = 0; n < TN; n += 2) {
      const int pc0 = col_base + n * 32 + l31;  // packed column of the first operand; the second sits 32 further
      const int oc = (pc0 >> 6) * 32 + l31;     // output channel
      const int dead = oc < a.N ? 0 : (int)0x80000000;
      const float b0 = (biasg && !dead) ? biasg[pc0] : 0.f, b1 = (biasg && !dead) ? biasg[pc0 + 32] : 0.f;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int row0 = row_base + m * 32;
        const int coff = (row0 * a.ldc + (SPLIT ? (oc >> 5) * 64 + (oc & 31) : oc)) * 2 | dead;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          float g;
          if constexpr (W2) g = act(fmaf(acc[m][n][r], osc, b0 + pe0[m][n / 2][r]), m0, s0, h0) * act(fmaf(acc[m][n + 1][r], osc, b1 + pe1[m][n / 2][r]), m1, s1, h1);
          else g = act(acc[m][n][r] + b0 + pe0[m][n / 2][r], m0, s0, h0) * act(acc[m][n + 1][r] + b1 + pe1[m][n / 2][r], m1, s1, h1);
          if (row0 + rr >= row_lim) g = 0.f;
          const uint16_t gh = ss_f2t<W2>(g);
#if SS_HABL == 4
          if (g == 12345.678f)   // ablation build (tools/ablate_h.sh): no output stores
#endif
          {
            __builtin_amdgcn_raw_buffer_store_b16(gh, rsrc_c, coff + rr * ldc2, 0, 0);   // rows >= T: out of range, dropped
            // second term 32 elements further; fp16x2: the gate output is only ever a matrix-core A operand (hi term), its second term is not written
            if constexpr (SPLIT == 1) __builtin_amdgcn_raw_buffer_store_b16(f2bf(g - bf2f(gh)), rsrc_c, coff + rr * ldc2, 64, 0);
          }
        }
      }
    }
  } else {  // SS_HEPI_RESX: x <- (x + acc + bias) * post_scale (fp32, in place); y = bf16(x_new + next_bias) for the next layer's conv
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.X ? (void*)(a.X + (int64_t)b * a.x_batch_stride) : (void*)a.W), 0, __builtin_amdgcn_readfirstlane(a.X ? (int)((int64_t)a.T * a.ldx * 4) : 0), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.Y ? (void*)(a.Y + (int64_t)b * a.y_batch_stride) : (void*)a.X), 0,
        __builtin_amdgcn_readfirstlane(a.Y ? (int)((int64_t)a.T * a.ldy * 2) : 0), 0x00020000);
    const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
    const float* nbg = a.next_bias ? a.next_bias + (int64_t)grp_w * a.next_bias_group_stride : nullptr;
    const int ldx4 = a.ldx * 4, ldy2 = a.ldy * 2;
    if constexpr (SPLIT) {
      if (a.X == nullptr) {
        // Pair-only residual stream (the form the bf16x2 loops launch): the accumulator tile goes through LDS so that every thread owns 8
        // CONSECUTIVE channels of a row - the (hi, mid) pairs of the stream are then read and written as 16-byte vectors (full lines per wave
        // instruction) instead of 2-byte accesses per lane (the per-lane form was VMEM-instruction bound: 64 narrow accesses per thread).
        static_assert(BM * BN * 4 <= 2 * (BM + BN) * (LDH * 2), "the staged accumulator tile must fit the operand buffers");
        __syncthreads();   // every wave is done reading the last chunk's fragments
        float* St = reinterpret_cast<float*>(smem_h);   // [BM][BN] fp32
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              St[(wm * 32 * TM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * 32 * TN + n * 32 + l31] = acc[m][n][r];
        __syncthreads();
        const int g8 = tid & 15, col0 = n0 + g8 * 8;    // this thread's 8 channels (N is a multiple of 32: the group is valid or not as a whole)
        const int dead = col0 < a.N ? 0 : (int)0x80000000;
        const float* cbg = a.cur_bias + (int64_t)grp_w * a.cur_bias_group_stride;
        float bs[8], nb[8], cb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          bs[e] = (biasg && !dead) ? biasg[col0 + e] : 0.f;
          nb[e] = (nbg && !dead) ? nbg[col0 + e] : 0.f;
          cb[e] = !dead ? cbg[col0 + e] : 0.f;
        }
        const int phys = (col0 >> 5) * 64 + (col0 & 31);   // element of the hi terms inside the row; the mid terms sit 32 elements (64 B) further
#pragma unroll
        for (int i = 0; i < BM / 16; ++i) {
          const int row_l = (tid >> 4) + 16 * i, row = t0 + row_l;
          const float4 a0 = *reinterpret_cast<const float4*>(St + row_l * BN + g8 * 8), a1 = *reinterpret_cast<const float4*>(St + row_l * BN + g8 * 8 + 4);
          const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
          const int yo = (row * a.ldy + phys) * 2 | dead;
          const u32x4 hv = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, yo, 0, 0), mv = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, yo, 64, 0);
          const bool pad = row >= row_lim;
          u32x4 ho, mo;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            uint32_t hp = 0, mp = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const int e = 2 * e2 + k;
              float hf, mf, xn;
              if constexpr (W2) {
                hf = ss_t2f_packed<true>(hv[e2], k);
                mf = ss_t2f_packed<true>(mv[e2], k);
                xn = (((hf + mf) - cb[e]) + fmaf(av[e], osc, bs[e])) * a.post_scale;
              } else {
                hf = __builtin_bit_cast(float, k ? (hv[e2] & 0xffff0000u) : (hv[e2] << 16));
                mf = __builtin_bit_cast(float, k ? (mv[e2] & 0xffff0000u) : (mv[e2] << 16));
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
              hp |= (uint32_t)yh << (16 * k);
              mp |= (uint32_t)ym << (16 * k);
            }
            ho[e2] = hp;
            mo[e2] = mp;
          }
          __builtin_amdgcn_raw_buffer_store_b128(ho, rsrc_y, yo, 0, 0);    // rows >= T are out of range: dropped
          __builtin_amdgcn_raw_buffer_store_b128(mo, rsrc_y, yo, 64, 0);
        }
        return;
      }
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int col = col_base + n * 32 + l31;
      const int dead = col < a.N ? 0 : (int)0x80000000;
      const float bs = (biasg && !dead) ? biasg[col] : 0.f;
      const float nb = (nbg && !dead) ? nbg[col] : 0.f;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const int row0 = row_base + m * 32;
        const int xoff = (row0 * a.ldx + col) * 4 | dead;
        const int yoff = (row0 * a.ldy + (SPLIT ? (col >> 5) * 64 + (col & 31) : col)) * 2 | dead;
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          xv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, xoff + ((r & 3) + 8 * (r >> 2)) * ldx4, 0, 0));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          float xn = (xv[r] + (W2 ? fmaf(acc[m][n][r], osc, bs) : acc[m][n][r] + bs)) * a.post_scale;
          const bool pad = row0 + rr >= row_lim;
          if (pad) xn = 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, xn), rsrc_x, xoff + rr * ldx4, 0, 0);
          const float yv = pad ? 0.f : xn + nb;
          const uint16_t yh = ss_f2t<W2>(yv);
          __builtin_amdgcn_raw_buffer_store_b16(yh, rsrc_y, yoff + rr * ldy2, 0, 0);
          if constexpr (SPLIT) __builtin_amdgcn_raw_buffer_store_b16(ss_f2t<W2>(yv - ss_t2f<W2>(yh)), rsrc_y, yoff + rr * ldy2, 64, 0);
        }
      }
    }
  }
}

template <int BM, int BN, int EPI, int SPLIT>
int launch_h(const ss_gemm_bf16_args& a, hipStream_t stream) {
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const int n_cols = (EPI == SS_HEPI_GATE) ? a.Np : a.N;
  const int n_tiles = ss_cdiv(n_cols, BN);
  const int grid = ss_cdiv(m_tiles, 8) * 8 * n_tiles;
  const size_t lds = (size_t)2 * (BM + BN) * (LDH * 2);
  // per device and cheap: set on every launch (a process may drive several GPUs), a failure is reported, never cached
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, EPI, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) {
    ss_set_error("ss_gemm_bf16: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(attr));
    return SS_ERR_HIP;
  }
  hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, EPI, SPLIT>), dim3(grid), dim3(256), lds, stream, a, m_tiles_per_item, m_tiles, n_tiles);
  SS_CHECK_LAUNCH("ss_gemm_bf16");
  return SS_OK;
}

template <int EPI, int SPLIT>
int launch_tiles_s(const ss_gemm_bf16_args& a, hipStream_t stream) {
  const int n_cols = (EPI == SS_HEPI_GATE) ? a.Np : a.N;
  const long big = (long)ss_cdiv(a.T, 128) * a.B * ss_cdiv(n_cols, 128);
  const int env_tile = g_ss_tuning.htile;  // experiments: 64 / 128 force the row tile
  if (env_tile == 64) return launch_h<64, 128, EPI, SPLIT>(a, stream);
  if (env_tile == 128) return launch_h<128, 128, EPI, SPLIT>(a, stream);
  if (big >= 512) return launch_h<128, 128, EPI, SPLIT>(a, stream);
  return launch_h<64, 128, EPI, SPLIT>(a, stream);
}
template <int EPI>
int launch_tiles(const ss_gemm_bf16_args& a, hipStream_t stream) {
  return a.split == 2 ? launch_tiles_s<EPI, 2>(a, stream) : a.split ? launch_tiles_s<EPI, 1>(a, stream) : launch_tiles_s<EPI, 0>(a, stream);
}

// x[r][c] (+ bias[c]) -> bf16 ; rows >= lens[b] -> 0
__global__ void to_bf16_kernel(const float* __restrict__ x, const float* __restrict__ bias, uint16_t* __restrict__ y, int B, int T, int C,
                               int ldx, int ldy, const int32_t* __restrict__ lens, int group_size, int64_t bias_gs) {
  const int64_t n = (int64_t)B * T * (C / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4)) * 4;
    const int64_t r = i / (C / 4);
    const int b = (int)(r / T), t = (int)(r % T);
    float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c4);
    if (bias) {
      const float* bp = bias + (group_size > 0 ? (int64_t)(b / group_size) * bias_gs : 0) + c4;
      v.x += bp[0]; v.y += bp[1]; v.z += bp[2]; v.w += bp[3];
    }
    if (lens && t >= lens[b]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    ushort4 o;
    o.x = f2bf(v.x); o.y = f2bf(v.y); o.z = f2bf(v.z); o.w = f2bf(v.w);
    *reinterpret_cast<ushort4*>(y + r * ldy + c4) = o;
  }
}

// the split form, pairs interleaved by 32: hi = RNE(v) at (c >> 5) * 64 + (c & 31), mid = RNE(v - hi) 32 elements further; F16: fp16 terms of
// v * scale (a power of two: the weights' shift of the "fp16x2" mode, 1 for activations)
template <bool F16>
__global__ void split_bf16_kernel(const float* __restrict__ x, const float* __restrict__ bias, uint16_t* __restrict__ y, int B, int T, int C,
                                  int ldx, int ldy, const int32_t* __restrict__ lens, int group_size, int64_t bias_gs, float scale) {
  const int64_t n = (int64_t)B * T * (C / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4)) * 4;
    const int64_t r = i / (C / 4);
    const int b = (int)(r / T), t = (int)(r % T);
    float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c4);
    if (bias) {
      const float* bp = bias + (group_size > 0 ? (int64_t)(b / group_size) * bias_gs : 0) + c4;
      v.x += bp[0]; v.y += bp[1]; v.z += bp[2]; v.w += bp[3];
    }
    if (lens && t >= lens[b]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (F16) v = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    ushort4 h, m;
    h.x = ss_f2t<F16>(v.x); h.y = ss_f2t<F16>(v.y); h.z = ss_f2t<F16>(v.z); h.w = ss_f2t<F16>(v.w);
    m.x = ss_f2t<F16>(v.x - ss_t2f<F16>(h.x)); m.y = ss_f2t<F16>(v.y - ss_t2f<F16>(h.y)); m.z = ss_f2t<F16>(v.z - ss_t2f<F16>(h.z));
    m.w = ss_f2t<F16>(v.w - ss_t2f<F16>(h.w));
    uint16_t* yp = y + r * ldy + (c4 >> 5) * 64 + (c4 & 31);
    *reinterpret_cast<ushort4*>(yp) = h;
    *reinterpret_cast<ushort4*>(yp + 32) = m;
  }
}

}  // namespace

extern "C" int ss_gemm_bf16(const ss_gemm_bf16_args* args, void* stream_) {
  SS_CHECK_ARG(args != nullptr, "ss_gemm_bf16: null args");
  const ss_gemm_bf16_args& a = *args;
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(a.A && a.W, "ss_gemm_bf16: null A/W");
  SS_CHECK_ARG(a.B > 0 && a.T > 0 && a.N > 0, "ss_gemm_bf16: bad dims B=%d T=%d N=%d", a.B, a.T, a.N);
  SS_CHECK_ARG(a.ntaps >= 1 && a.ntaps <= 4, "ss_gemm_bf16: ntaps=%d out of range", a.ntaps);
  SS_CHECK_ARG(a.K > 0 && (a.K % (a.split ? 32 : BKH)) == 0 && (a.lda % 8) == 0, "ss_gemm_bf16: K=%d must be a multiple of 64 (32 when split), lda=%d of 8", a.K, a.lda);
  SS_CHECK_ARG((a.Np & 31) == 0 && (a.epi == SS_HEPI_GATE ? (a.Np & 63) == 0 && 2 * a.N <= a.Np : a.Np >= a.N), "ss_gemm_bf16: bad Np=%d for N=%d", a.Np, a.N);
  SS_CHECK_ARG((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0 && (a.a_batch_stride & 7) == 0, "ss_gemm_bf16: A/W must be 16-byte aligned");
  // 32-bit buffer offsets per item: the output element is 2 bytes for the GATE epilogue (16-bit terms), 4 for STORE; RESX writes X (fp32) and Y (terms)
  SS_CHECK_ARG((int64_t)a.T * a.lda * 2 < (1ll << 31) && (int64_t)a.T * a.lde * 4 < (1ll << 31) && (int64_t)a.T * a.ldx * 4 < (1ll << 31) &&
                   (int64_t)a.T * a.ldc * (a.epi == SS_HEPI_GATE ? 2 : 4) < (1ll << 31) && (int64_t)a.T * a.ldy * 2 < (1ll << 31),
               "ss_gemm_bf16: item too large for 32-bit offsets");
  SS_CHECK_ARG(a.split >= 0 && a.split <= 2, "ss_gemm_bf16: split=%d", a.split);
  SS_CHECK_ARG(a.split != 2 || (a.out_scale > 0.f && a.out_scale <= 1.f), "ss_gemm_bf16: split = 2 needs 0 < out_scale <= 1 (got %g)", (double)a.out_scale);
  if (a.a_compact || a.one_product == 2) {   // the compact operands exist in the many-round STORE kernel only
    SS_CHECK_ARG(a.epi == SS_HEPI_STORE && a.C && ss_gemm_bf16_tile256_ok(&a),
                 "ss_gemm_bf16: a_compact / one_product = 2 need a launch ss_gemm_bf16_tile256 takes (split = 2, STORE, >= 2 rounds of 256-row tiles)");
    return ss_gemm_bf16_tile256(&a, stream_);
  }
  if (a.split) {   // pairs interleaved by 32: physical rows hold 2 x the logical channels
    SS_CHECK_ARG((a.K % 32) == 0 && a.lda >= 2 * a.K, "ss_gemm_bf16: split operands need K %% 32 == 0 and lda >= 2 K (K=%d lda=%d)", a.K, a.lda);
    SS_CHECK_ARG(a.epi != SS_HEPI_GATE || (a.ldc >= 2 * a.N && (a.N % 32) == 0), "ss_gemm_bf16: split GATE writes 2 N bf16 per row (ldc=%d N=%d)", a.ldc, a.N);
    SS_CHECK_ARG(a.epi != SS_HEPI_RESX || !a.Y || (a.ldy >= 2 * a.N && (a.N % 32) == 0), "ss_gemm_bf16: split RESX writes 2 N bf16 per row of Y");
  }
  switch (a.epi) {
    case SS_HEPI_STORE:
      SS_CHECK_ARG(a.C != nullptr, "ss_gemm_bf16: STORE needs C");
      if (g_ss_tuning.gate256 && ss_gemm_bf16_tile256_ok(&a)) return ss_gemm_bf16_tile256(&a, stream_);   // many-round split launches
      return launch_tiles<SS_HEPI_STORE>(a, stream);
    case SS_HEPI_GATE:
      SS_CHECK_ARG(a.C != nullptr, "ss_gemm_bf16: GATE needs C");
      // many-round launches of the 3-tap dilated conv (BASELINE config 4) go to the 256x256 / 8-wave / LDS-DMA kernel
      // fp16x2, very many tiles: 256 x 128 tiles with a compact A image, two workgroups per CU ("gate128" knob; C4 batch 11.9 -> 11.2 s)
      if (g_ss_tuning.gate128 && ss_gemm_bf16_gate128_ok(&a)) return ss_gemm_bf16_gate128(&a, stream_);
      if (g_ss_tuning.gate256 && ss_gemm_bf16_gate256_ok(&a)) return ss_gemm_bf16_gate256(&a, stream_);
      return launch_tiles<SS_HEPI_GATE>(a, stream);
    case SS_HEPI_RESX:
      SS_CHECK_ARG(a.X != nullptr || (a.split && a.Y && a.cur_bias), "ss_gemm_bf16: RESX needs X (or, with split operands, Y + cur_bias: the pair-only stream)");
      // (a 128-row / two-workgroups-per-CU form of this launch measured 131.9 vs 128.9 us in round 4 - HBM-bound either way - and was removed in round 5)
      if (g_ss_tuning.gate256 && ss_gemm_bf16_tile256_ok(&a)) return ss_gemm_bf16_tile256(&a, stream_);
      return launch_tiles<SS_HEPI_RESX>(a, stream);
    default: break;
  }
  ss_set_error("ss_gemm_bf16: bad epilogue %d", a.epi);
  return SS_ERR_ARG;
}

extern "C" int ss_to_bf16(const float* x, const float* bias, uint16_t* y, int B, int T, int C, int ldx, int ldy, const int32_t* lens,
                          int group_size, int64_t bias_group_stride, void* stream) {
  SS_CHECK_ARG(x && y && B > 0 && T > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0, "ss_to_bf16: bad args");
  const int64_t n = (int64_t)B * T * (C / 4);
  const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipLaunchKernelGGL(to_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, y, B, T, C, ldx, ldy, lens, group_size,
                     bias_group_stride);
  SS_CHECK_LAUNCH("ss_to_bf16");
  return SS_OK;
}

extern "C" int ss_split_bf16(const float* x, const float* bias, uint16_t* y, int B, int T, int C, int ldx, int ldy, const int32_t* lens,
                             int group_size, int64_t bias_group_stride, void* stream) {
  SS_CHECK_ARG(x && y && B > 0 && T > 0 && C > 0 && (C & 31) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ldy >= 2 * C, "ss_split_bf16: bad args");
  const int64_t n = (int64_t)B * T * (C / 4);
  const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipLaunchKernelGGL(split_bf16_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, y, B, T, C, ldx, ldy, lens, group_size,
                     bias_group_stride, 1.0f);
  SS_CHECK_LAUNCH("ss_split_bf16");
  return SS_OK;
}

extern "C" int ss_split_f16(const float* x, const float* bias, float scale, uint16_t* y, int B, int T, int C, int ldx, int ldy, const int32_t* lens,
                            int group_size, int64_t bias_group_stride, void* stream) {
  SS_CHECK_ARG(x && y && B > 0 && T > 0 && C > 0 && (C & 31) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ldy >= 2 * C, "ss_split_f16: bad args");
  SS_CHECK_ARG(scale > 0.f && scale < 65536.f, "ss_split_f16: scale=%g", (double)scale);
  const int64_t n = (int64_t)B * T * (C / 4);
  const int grid = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  hipLaunchKernelGGL(split_bf16_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, bias, y, B, T, C, ldx, ldy, lens, group_size,
                     bias_group_stride, scale);
  SS_CHECK_LAUNCH("ss_split_f16");
  return SS_OK;
}
