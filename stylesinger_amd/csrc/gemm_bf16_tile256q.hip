// The "fp16q4" long-K STORE GEMM (ss_gemm_bf16_args.split = 3, SS_HEPI_STORE: the K = L*C skip GEMM of the denoiser loops) for many-round
// launches: tile256s_kernel<STORE, true> (gemm_bf16_tile256.hip: 256 rows x all N <= 256 columns per workgroup, 8 waves, both operands by LDS-DMA)
// with the SECOND product of every 32-channel step - activation x weight-lo - on the block-scaled fp4 matrix instruction, exactly as
// gate128q_kernel does it (gemm_bf16_gate128q.hip): consecutive chunks (2 p, 2 p + 1) form a pair; a lane converts its own four fp16 A fragments
// of the pair to fp4 in registers (fixed power-of-two scale args.q_scale: the A operand here is the gate output z in (-1, 1), scale 2^-2 in the
// numerics study) and meets the pair's weight-lo terms, packed once in that lane order in the second half of the ODD chunk's weight line
// (stylesinger_amd.lib.pack_skip_q4, element order g128q-style: t128q::q_kindex). Per pair 32 fp16 MFMAs + 8 block-scaled ones instead of 64.
// The launch sits at the chip's power-limited matrix rate, so fewer matrix instructions are what makes it faster.
// NOT YET RUN ON HARDWARE (written after the round's GPU budget was spent): reached only through ss_gemm_bf16_tile256q, not dispatched to.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include "pair16.h"
#include "gate128_layout.h"
#include <type_traits>

typedef ss_f32x16 f32x16;
typedef ss_bf16x8 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BM = 256, BN = 256;
constexpr int ROWB = 128;   // bytes per LDS row: A = 32 channels x (hi | unused plane), B = 32 channels hi | the pair's fp4 lo terms + scales

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

__global__ __launch_bounds__(512, 2) void tile256q_store_kernel(const ss_gemm_bf16_args a, int m_tiles_per_item, int m_tiles, int kchunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_t256q[];   // 128 KB: [A0 32 K][B0 32 K][A1 32 K][B1 32 K]; epilogue: 2 x 64 KB staging
  char* const A0 = smem_t256q;
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
  const int ldw = 2 * a.K;            // 16-bit terms per packed weight row (one 64-element line per 32-channel chunk)

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

  // DMA roles: tile256s_kernel's. 32 pieces of 8 rows x 128 B per operand and step; wave w issues pieces w, w + 8, w + 16, w + 24 of both. Lane i
  // of a piece lands at (row i >> 3, physical slot i & 7) and fetches logical slot (i & 7) ^ ((row >> 1) & 7). Dead lanes (nothing fetched, zeros
  // written): the A operand's second plane (slots 4-7) always; the weight line's slots 4-7 in EVEN chunks, slot 7 in odd ones
  const int r0 = 8 * wave + (lane >> 3);
  const int slot0 = (lane & 7) ^ ((r0 >> 1) & 7);
  const int a_voff = (((t0 + r0) * a.lda + slot0 * 8) * 2) | (slot0 >= 4 ? (int)0x80000000 : 0);   // rows >= len are out of range anyway
  const int b_voff = (r0 * ldw + slot0 * 8) * 2;                                                  // packed weight rows >= Np read zeros
  const int b_dead_even = slot0 >= 4 ? (int)0x80000000 : 0, b_dead_odd = slot0 == 7 ? (int)0x80000000 : 0;
  auto piece = [&](char* Ab, char* Bb, int c, int i) {     // i = 0..3: A pieces, 4..7: B pieces of chunk c
    const int j = i & 3;
    if (i < 4) glds16(rsrc_a, Ab + (wave + 8 * j) * 8 * ROWB, a_voff + 64 * j * a.lda * 2, c * ROWB);
    else glds16(rsrc_w, Bb + (wave + 8 * j) * 8 * ROWB, b_voff | ((c & 1) ? b_dead_odd : b_dead_even), c * ROWB + 64 * j * ldw * 2);
  };

  // fragment addresses: row = 128 wm + 32 m + l31 for A, 64 wn + 32 n + l31 for B; slot (2 ks) ^ swz = hi plane of k-step ks; B slot 4 ^ swz = this
  // lane half's 32 fp4 lo terms of the pair (odd chunks), byte lh of logical slot 6 = their E8M0 scale
  const int a_base = (128 * wm + l31) * ROWB, a_swz = (((128 * wm + l31) >> 1) & 7) ^ lh;
  const int b_row = 64 * wn + l31;
  const int b_base = b_row * ROWB, b_swz = ((b_row >> 1) & 7) ^ lh;
  const int b_scale_off = b_base + ((6 ^ ((b_row >> 1) & 7)) << 4) + lh;

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  const float qs = a.q_scale;
  const int sa = (int)((__builtin_bit_cast(unsigned, qs) >> 23) & 0xffu);   // E8M0 byte of the power of two qs
  // deferred past the next barrier: k-step 1 of a step (fragments p_ah, p_bh) and, after odd steps, the pair's block-scaled group (aq, bq, sb);
  // all zero before the first step, so that the first step's deferred instructions add nothing
  bf16x8 p_ah[4], p_bh[2];
  unsigned aq[4][4];
  v8i bq[2];
  int sb[2] = {127, 127};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int e = 0; e < 8; ++e) p_ah[m][e] = (__bf16)0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) aq[m][i] = 0u;
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int e = 0; e < 8; ++e) p_bh[n][e] = (__bf16)0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) bq[n][i] = 0;
  }
  auto cvt8 = [&](const bf16x8& f) {   // 8 fp16 -> 8 fp4: element t in nibble t & 1 of byte t >> 1 (tools/ubench/cvt_fp4_probe.hip)
    const ss_f16x8 v = __builtin_bit_cast(ss_f16x8, f);
    unsigned r = 0;
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{v[0], v[1]}, qs, 0);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{v[2], v[3]}, qs, 1);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{v[4], v[5]}, qs, 2);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{v[6], v[7]}, qs, 3);
    return r;
  };
  auto mfma_h = [&](int m, int n, const bf16x8 (&fa)[4], const bf16x8 (&fb)[2]) { acc[m][n] = ss_mfma_32x32x16<true>(fa[m], fb[n], acc[m][n]); };
  auto mfma_q = [&](int m, int n) {
    v8i av;
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = (int)aq[m][i];
#pragma unroll
    for (int i = 4; i < 8; ++i) av[i] = 0;
    acc[m][n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bq[n], acc[m][n], 4, 4, 0, sa, 0, sb[n]);
  };
  auto step = [&](auto par_tag, const char* Ac, const char* Bc, char* An, char* Bn, int c, bool more) {
    constexpr int PAR = decltype(par_tag)::value;   // chunk parity inside its pair
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
    bf16x8 ah0[4], bh0[2];
    rd_a(0, ah0);
    rd_b(0, bh0);
    __builtin_amdgcn_sched_barrier(0);
    // deferred by the previous step: its k-step 1 (8) and, when that step closed a pair (this one opens the next: PAR == 0), the pair's
    // block-scaled group (8); one DMA piece of the next chunk after each of the first 8
    constexpr int ND = PAR == 0 ? 16 : 8;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      if (i < 8) mfma_h((i >> 1) & 3, i & 1, p_ah, p_bh);
      else mfma_q(((i - 8) >> 1) & 3, (i - 8) & 1);
      if (i < 8) {
        __builtin_amdgcn_sched_barrier(0);
        if (more) piece(An, Bn, c + 1, i);   // wave-uniform branch
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    rd_a(2, p_ah);
    rd_b(2, p_bh);
    [[maybe_unused]] bf16x8 bqr[2];
    if constexpr (PAR == 1) rd_b(4, bqr);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) mfma_h((k >> 1) & 3, k & 1, ah0, bh0);   // k-step 0 of this step
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PAR == 1) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const u32x4 w = __builtin_bit_cast(u32x4, bqr[n]);
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[n][i] = (int)w[i];
        sb[n] = *reinterpret_cast<const uint8_t*>(Bc + b_scale_off + n * 32 * ROWB);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {   // my own A values of this step as fp4: registers 2 * PAR + ks of the pair's operand
      aq[m][2 * PAR] = cvt8(ah0[m]);
      aq[m][2 * PAR + 1] = cvt8(p_ah[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(A0, B0, 0, i);
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < kchunks; c += 2) {   // kchunks is even (checked by the launcher)
    step(P0{}, A0, B0, A1, B1, c, true);
    step(P1{}, A1, B1, A0, B0, c + 1, c + 2 < kchunks);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) mfma_h((i >> 1) & 3, i & 1, p_ah, p_bh);   // the last (odd) step's deferred work
#pragma unroll
  for (int i = 0; i < 8; ++i) mfma_q((i >> 1) & 3, i & 1);

  // ---- epilogue: tile256s_kernel<STORE, true>'s - four passes of 64 rows staged as fp32 [64][256] in alternating 64-KB halves
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
  __builtin_amdgcn_s_barrier();   // everyone is done reading the operand buffers
  const int st_wr = (32 * wm + 4 * lh) * (BN * 4) + (64 * wn + l31) * 4;   // + rr * BN * 4 (+ 128 for n = 1)
  float* Cb = (float*)a.C + (int64_t)b * a.c_batch_stride;
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Cb), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);
  const int c4 = (tid & 63) * 4;
  const int dead = c4 < a.N ? 0 : (int)0x80000000;
  float bs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bs[e] = (biasg && !dead) ? biasg[c4 + e] : 0.f;
  const bool relu = a.act == SS_ACT_RELU;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    char* St = smem_t256q + (q & 1) * 64 * 1024;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) *reinterpret_cast<float*>(St + st_wr + ((r & 3) + 8 * (r >> 2)) * (BN * 4) + n * 128) = acc[q][n][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = (tid >> 6) + 8 * j;
      const int grow = t0 + 128 * (k >> 5) + 32 * q + (k & 31);
      float4 v = *reinterpret_cast<const float4*>(St + k * (BN * 4) + c4 * 4);
      v = make_float4(fmaf(v.x, a.out_scale, bs[0]), fmaf(v.y, a.out_scale, bs[1]), fmaf(v.z, a.out_scale, bs[2]), fmaf(v.w, a.out_scale, bs[3]));
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      if (grow >= row_lim) v = make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_c, (grow * a.ldc + c4) * 4 | dead, 0, 0);
    }
  }
}

}  // namespace

// 1 if this STORE launch can run on the fp16q4 256-row kernel: split = 3 operands, one tap, N <= 256, K a multiple of 64, at least two rounds of tiles
extern "C" int ss_gemm_bf16_tile256q_ok(const ss_gemm_bf16_args* a) {
  if (!a || a->split != 3 || a->ntaps != 1 || a->tap_off[0] != 0 || a->epi != SS_HEPI_STORE) return 0;
  if ((a->N % 4) != 0 || (a->ldc % 4) != 0 || (a->act != SS_ACT_NONE_ && a->act != SS_ACT_RELU_)) return 0;
  if (a->N > BN || (a->K % 64) != 0 || a->lda < 2 * a->K || (a->lda % 8) != 0 || !(a->out_scale > 0.f && a->out_scale <= 1.f) || !(a->q_scale > 0.f)) return 0;
  if ((int64_t)a->T * a->lda * 2 >= (1ll << 31) || (int64_t)a->T * a->ldc * 4 >= (1ll << 31) || (int64_t)a->Np * a->K * 4 >= (1ll << 31)) return 0;
  return (g_ss_tuning.q4_force || (long)ss_cdiv(a->T, BM) * a->B >= 2L * ss_n_cu()) ? 1 : 0;
}

extern "C" int ss_gemm_bf16_tile256q(const ss_gemm_bf16_args* args, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_gemm_bf16_tile256q: null args");
  const ss_gemm_bf16_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C && a.split == 3 && a.ntaps == 1 && a.tap_off[0] == 0 && a.epi == SS_HEPI_STORE && a.out_scale > 0.f && a.out_scale <= 1.f,
               "ss_gemm_bf16_tile256q: fp16q4 operands (split = 3, 0 < out_scale <= 1), one tap at offset 0, STORE");
  {
    int ex = 0;
    const float mant = frexpf(a.q_scale, &ex);
    SS_CHECK_ARG(a.q_scale > 0.f && mant == 0.5f && ex >= -20 && ex <= 20, "ss_gemm_bf16_tile256q: q_scale must be a power of two (got %g)", (double)a.q_scale);
  }
  SS_CHECK_ARG(a.N > 0 && a.N <= BN && a.Np >= a.N && (a.K % 64) == 0 && a.lda >= 2 * a.K && (a.lda % 8) == 0, "ss_gemm_bf16_tile256q: N <= 256, K %% 64 == 0, lda >= 2 K");
  SS_CHECK_ARG((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0 && (a.a_batch_stride & 7) == 0, "ss_gemm_bf16_tile256q: A/W must be 16-byte aligned");
  SS_CHECK_ARG((int64_t)a.T * a.lda * 2 < (1ll << 31) && (int64_t)a.Np * a.K * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 4 < (1ll << 31),
               "ss_gemm_bf16_tile256q: item too large for 32-bit offsets");
  SS_CHECK_ARG((a.N % 4) == 0 && (a.ldc % 4) == 0 && (a.act == SS_ACT_NONE_ || a.act == SS_ACT_RELU_), "ss_gemm_bf16_tile256q: N %% 4 == 0, ldc %% 4 == 0, act none | relu");
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const size_t lds = (size_t)128 * 1024;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile256q_store_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    ss_set_error("ss_gemm_bf16_tile256q: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(e));
    return SS_ERR_HIP;
  }
  SS_PROPAGATE(ss_q4_guard_launch(&a, 1, stream));
  hipLaunchKernelGGL(tile256q_store_kernel, dim3(m_tiles), dim3(512), lds, (hipStream_t)stream, a, m_tiles_per_item, m_tiles, a.K / 32);
  SS_CHECK_LAUNCH("ss_gemm_bf16_tile256q");
  return SS_OK;
}

// K index of element e of lane half h in chunk pair p of a 1-tap GEMM (the order lib.pack_skip_q4 packs the fp4 lo terms in)
extern "C" int ss_tile256q_kindex(int32_t* out, int n_pairs) {
  SS_CHECK_ARG(out != nullptr && n_pairs > 0, "ss_tile256q_kindex: bad args");
  for (int p = 0; p < n_pairs; ++p)
    for (int h = 0; h < 2; ++h)
      for (int e = 0; e < 32; ++e) out[(p * 2 + h) * 32 + e] = t128q_kindex(p, h, e);
  return n_pairs * 64;
}
