// The "fp16x2" residual projection on the pair-only stream (ss_gemm_bf16_args.split = 2, SS_HEPI_RESX with X == NULL: K = N = C) for many-round
// launches with TWO workgroups per CU: 128 rows x all N <= 256 columns per workgroup, 4 waves, 80 KB of LDS.
//
// Why (tile256s_kernel<RESX, true>, profiles/r04_bench_c4_fp16x2_20steps_kernel_stats.csv): 125.5 us for 461 MB of algorithmic traffic and 19 us
// of matrix time per BASELINE config 4 launch - HBM-bound, but in PHASES: a 128 KB workgroup owns its CU, streams the A operand (the loop), then
// reads and rewrites the stream's pairs (the epilogue); the memory system sees one kind of traffic per CU at a time and nothing while the
// accumulators cross LDS. With the compact A image of the fp16x2 mode (only the hi plane of the A operand is staged: gate128_kernel) a 128-row
// tile needs 2 x (8 + 32) KB = 80 KB, so two independent workgroups share a CU and one's epilogue traffic runs under the other's loop.
// Everything else is tile256s_kernel<RESX, true>: the 128 x 64 wave tile, 32-channel steps of 2 k-steps x 2 products with the second k-step's
// 16 MFMAs deferred past the next barrier and the next step's DMA pieces (2 A + 8 B per wave) issued between them, the epilogue in four
// LDS-staged passes (here of 32 rows) with the stream's pairs as 16-byte vectors. Index math: gate128_layout.h (namespace t128), checked on the
// host by tools/layout_check_gate128.cpp. Arithmetic and summation order = tile256s_kernel<RESX, true>: bit-identical results.
// MEASURED (profiles/r04_kbench_tile128.log, BASELINE config 4 shape, back to back): 131.9 us against the 256-row kernel's 128.9 us. The
// hypothesis above is wrong for this launch: it is HBM-bound as a whole (461 MB at 3.6 TB/s with reads and writes mixed), not phase-bound, and
// the second workgroup only re-streams the 256 KB of weights once more per 128 rows. Kept behind the "tile128" knob (default 0) as the
// two-workgroup skeleton for 1-tap GEMMs that DO have matrix or epilogue time to hide.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include "pair16.h"
#include "gate128_layout.h"
#include <type_traits>

typedef ss_f32x16 f32x16;
typedef ss_bf16x8 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

using namespace t128;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

__global__ __launch_bounds__(256, 2) void tile128_resx_kernel(const ss_gemm_bf16_args a, int m_tiles_per_item, int m_tiles, int kchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_t128[];   // 80 KB: [A0 8 K][B0 32 K][A1 8 K][B1 32 K]; epilogue: 2 x 32 KB staging
  char* const A0 = smem_t128;
  char* const B0 = A0 + BM * A_ROWB;
  char* const A1 = B0 + BN * B_ROWB;
  char* const B1 = A1 + BM * A_ROWB;

  const int mt = blockIdx.x;
  if (mt >= m_tiles) return;
  const int b = mt / m_tiles_per_item;
  const int t0 = (mt % m_tiles_per_item) * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave;
  const int l31 = lane & 31, lh = lane >> 5;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const int ldw = 2 * a.K;            // 16-bit terms per packed weight row (both planes, one tap)

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

  // DMA roles (gate128_layout.h, t128): per 32-channel chunk 8 A pieces of 16 rows x 64 B (hi plane only) - wave w issues pieces w, w + 4 (64
  // rows further, same swizzle) - and 32 B pieces of 8 rows x 128 B - wave w issues pieces w + 4 j, j < 8 (32 j rows further, same swizzle)
  const int a_voff = ((t0 + a_dma_row(wave, lane)) * a.lda + a_dma_slot(wave, lane) * 8) * 2;   // rows >= len are out of range: the DMA writes zeros
  const int b_voff = (b_dma_row(wave, lane) * ldw + b_dma_slot(wave, lane) * 8) * 2;            // packed weight rows >= Np read zeros
  auto piece = [&](char* Ab, char* Bb, int c, int i) {     // i = 0, 1: A pieces, 2..9: B pieces of chunk c
    if (i < 2) glds16(rsrc_a, Ab + (wave + 4 * i) * 1024, a_voff + 64 * i * a.lda * 2, c * 128);
    else glds16(rsrc_w, Bb + (wave + 4 * (i - 2)) * 1024, b_voff, c * 128 + 32 * (i - 2) * ldw * 2);
  };

  // fragment addresses: A row 32 m + l31, B row 64 wn + 32 n + l31; (base, swizzle ^ lh), one XOR per read
  const int a_base = a_frag_row(0, l31) * A_ROWB, a_sw = a_swz(a_frag_row(0, l31)) ^ lh;
  const int b_base = b_frag_row(wn, 0, l31) * B_ROWB, b_sw = b_swz(b_frag_row(wn, 0, l31)) ^ lh;

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // the two product groups (hi x lo, hi x hi of the second k-step) a step defers past the next barrier; zero fragments before the first step
  bf16x8 p_ah[4], p_bh[2], p_bm[2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int e = 0; e < 8; ++e) p_ah[m][e] = (__bf16)0.f;
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
      for (int n = 0; n < 2; ++n) acc[m][n] = ss_mfma_32x32x16<true>(fa[m], fb[n], acc[m][n]);
  };
  auto step = [&](const char* Ac, const char* Bc, char* An, char* Bn, int c, bool more) {
    wait_vmcnt<0>();                  // my pieces of chunk c have landed (nothing younger is in flight)
    __builtin_amdgcn_s_barrier();     // everyone's have; everyone finished reading chunk c-1's buffers
    auto rd_a = [&](int ks2, bf16x8 (&f)[4]) {
      const int ao = a_base + ((ks2 ^ a_sw) << 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) f[m] = *reinterpret_cast<const bf16x8*>(Ac + ao + m * 32 * A_ROWB);
    };
    auto rd_b = [&](int slot, bf16x8 (&f)[2]) {
      const int bo = b_base + ((slot ^ b_sw) << 4);
#pragma unroll
      for (int n = 0; n < 2; ++n) f[n] = *reinterpret_cast<const bf16x8*>(Bc + bo + n * 32 * B_ROWB);
    };
    bf16x8 ah0[4], bh0[2];
    rd_a(0, ah0);
    rd_b(0, bh0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {    // the 16 MFMAs deferred by the previous step, one DMA piece of the next chunk after each of the first 10
      const int m = (i >> 1) & 3, n = i & 1;
      acc[m][n] = ss_mfma_32x32x16<true>(p_ah[m], i < 8 ? p_bm[n] : p_bh[n], acc[m][n]);
      if (i < 10) {
        __builtin_amdgcn_sched_barrier(0);
        if (more) piece(An, Bn, c + 1, i);   // wave-uniform branch
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
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
  };
#pragma unroll
  for (int i = 0; i < 10; ++i) piece(A0, B0, 0, i);
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < kchunks; c += 2) {   // kchunks is even (checked by the launcher)
    step(A0, B0, A1, B1, c, true);
    step(A1, B1, A0, B0, c + 1, c + 2 < kchunks);
  }
  mfma8(p_ah, p_bm);
  mfma8(p_ah, p_bh);

  // ---- epilogue: four passes of 32 rows (accumulator block m = q of every wave = tile rows 32 q + (0..31)), staged as fp32 [32][256] in
  // alternating 32-KB halves of the operand memory, then processed row-contiguously: Y = pair(x + cur_bias) is read, x updated,
  // Y = pair(x_new + next_bias) rewritten in place
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
  __builtin_amdgcn_s_barrier();   // everyone is done reading the operand buffers
  const int st_wr = st_write(wn, 0, l31, lh, 0);   // + rr * ST_ROWB (+ 128 for n = 1)
  const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.Y + (int64_t)b * a.y_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldy * 2)), 0x00020000);
  const float* nbg = a.next_bias ? a.next_bias + (int64_t)grp_w * a.next_bias_group_stride : nullptr;
  const float* cbg = a.cur_bias + (int64_t)grp_w * a.cur_bias_group_stride;
  const int col0 = item_col0(tid);                // this thread's 8 channels in every row it handles (N is a multiple of 32: valid or dead as a whole)
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
    char* St = smem_t128 + (q & 1) * 32 * 1024;
    // the stream's pairs of this pass are fetched before the staging barrier: their latency hides under it
    u32x4 hv[4], mv[4];
    int yo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // 32 rows x 32 groups = 1024 items, four per thread: item p = tid + 256 j -> row (tid >> 5) + 8 j, group tid & 31
      const int grow = t0 + 32 * q + item_row(tid + 256 * j);
      yo[j] = (grow * a.ldy + phys) * 2 | dead;
      hv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, yo[j], 0, 0);
      mv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, yo[j], 64, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(St + st_wr + g128::acc_rr(r) * ST_ROWB + n * 128) = acc[q][n][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): my staging writes are done
    __builtin_amdgcn_s_barrier();         // the staging tile of pass q is complete (pass q-1's tile, the other half, is being read at most)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = item_row(tid + 256 * j);
      const int grow = t0 + 32 * q + k;
      const float4 a0 = *reinterpret_cast<const float4*>(St + k * ST_ROWB + col0 * 4), a1 = *reinterpret_cast<const float4*>(St + k * ST_ROWB + col0 * 4 + 16);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const bool pad = grow >= row_lim;
      u32x4 ho, mo;
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        uint32_t hp = 0, mp = 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const int e = 2 * e2 + kk;
          const float hf = ss_t2f_packed<true>(hv[j][e2], kk);
          const float mf = ss_t2f_packed<true>(mv[j][e2], kk);
          const float xn = (((hf + mf) - cb[e]) + fmaf(av[e], a.out_scale, bs[e])) * a.post_scale;
          const float yv = pad ? 0.f : xn + nb[e];
          const uint16_t yh = ss_f2t<true>(yv), ym = ss_f2t<true>(yv - ss_t2f<true>(yh));
          hp |= (uint32_t)yh << (16 * kk);
          mp |= (uint32_t)ym << (16 * kk);
        }
        ho[e2] = hp;
        mo[e2] = mp;
      }
      __builtin_amdgcn_raw_buffer_store_b128(ho, rsrc_y, yo[j], 0, 0);    // rows >= T: out of range, dropped
      __builtin_amdgcn_raw_buffer_store_b128(mo, rsrc_y, yo[j], 64, 0);
    }
    // (no barrier here: pass q+1 writes the OTHER half, and pass q+2's writes to this half come after pass q+1's barrier, which every
    // thread reaches only after its pass-q reads)
  }
}

}  // namespace

// 1 if ss_gemm_bf16 should hand this launch to the 128-row kernel: fp16x2 operands, one tap, RESX on the pair-only stream, N <= 256, an even
// number of 32-channel chunks, the weights small enough to stay in L2 while every 128-row tile re-streams them (K <= 512), at least two rounds of tiles (two per CU and round)
extern "C" int ss_gemm_bf16_tile128_ok(const ss_gemm_bf16_args* a) {
  if (!a || a->split != 2 || a->ntaps != 1 || a->tap_off[0] != 0 || a->epi != SS_HEPI_RESX) return 0;
  if (!(a->X == nullptr && a->Y && a->cur_bias && (a->N % 32) == 0 && a->ldy >= 2 * a->N && (a->ldy % 8) == 0)) return 0;
  if (a->N > BN || (a->K % 64) != 0 || a->K > 512 || a->lda < 2 * a->K || (a->lda % 8) != 0 || !(a->out_scale > 0.f && a->out_scale <= 1.f)) return 0;
  if ((int64_t)a->T * a->lda * 2 >= (1ll << 31) || (int64_t)a->T * a->ldy * 2 >= (1ll << 31) || (int64_t)a->Np * a->K * 4 >= (1ll << 31)) return 0;
  return (long)ss_cdiv(a->T, BM) * a->B >= 4L * ss_n_cu() ? 1 : 0;
}

extern "C" int ss_gemm_bf16_tile128(const ss_gemm_bf16_args* args, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_gemm_bf16_tile128: null args");
  const ss_gemm_bf16_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.split == 2 && a.ntaps == 1 && a.tap_off[0] == 0 && a.out_scale > 0.f && a.out_scale <= 1.f,
               "ss_gemm_bf16_tile128: fp16x2 operands (split = 2, 0 < out_scale <= 1), one tap at offset 0");
  SS_CHECK_ARG(a.N > 0 && a.N <= BN && a.Np >= a.N && (a.K % 64) == 0 && a.lda >= 2 * a.K && (a.lda % 8) == 0, "ss_gemm_bf16_tile128: N <= 256, K %% 64 == 0, lda >= 2 K");
  SS_CHECK_ARG((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0 && (a.a_batch_stride & 7) == 0, "ss_gemm_bf16_tile128: A/W must be 16-byte aligned");
  SS_CHECK_ARG((int64_t)a.T * a.lda * 2 < (1ll << 31) && (int64_t)a.Np * a.K * 4 < (1ll << 31), "ss_gemm_bf16_tile128: item too large for 32-bit offsets");
  SS_CHECK_ARG(a.epi == SS_HEPI_RESX && a.X == nullptr && a.Y && a.cur_bias && (a.N % 32) == 0 && a.ldy >= 2 * a.N && (a.ldy % 8) == 0 &&
                   (((uintptr_t)a.Y) & 15) == 0 && (a.y_batch_stride & 7) == 0 && (int64_t)a.T * a.ldy * 2 < (1ll << 31),
               "ss_gemm_bf16_tile128: RESX on the pair-only stream (X = NULL, Y, cur_bias), N %% 32 == 0");
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const size_t lds = (size_t)80 * 1024;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile128_resx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    ss_set_error("ss_gemm_bf16_tile128: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(e));
    return SS_ERR_HIP;
  }
  hipLaunchKernelGGL(tile128_resx_kernel, dim3(m_tiles), dim3(256), lds, (hipStream_t)stream, a, m_tiles_per_item, m_tiles, a.K / 32);
  SS_CHECK_LAUNCH("ss_gemm_bf16_tile128");
  return SS_OK;
}
