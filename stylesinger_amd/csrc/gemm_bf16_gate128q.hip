// The "fp16q4" gate (ss_gemm_bf16_args.split = 3): gate128_kernel (gemm_bf16_gate128.hip: 256 x 128 tiles, compact A image, two workgroups per
// CU) with the SECOND product of every 32-channel step - activation x weight-lo, a correction of relative size 2^-12 that needs two or three
// significant bits (oracle/second_product_numerics.py: 3.4e-5 / 4.2e-5 on the reference's goldens, bar 1e-4) - moved from the fp16 matrix
// instruction to the block-scaled fp4 one (v_mfma_scale_f32_32x32x64_f8f6f4: 3.7x the fp16 issue rate, tools/ubench/mfma_mx_layout.hip).
//   * Steps are paired in issue order (2 p, 2 p + 1). The instruction pairs lane (i, h) of A with lane (j, h) of B element by element, so its
//     K = 64 can be ANY 64 K indices: here the 2 x 2 k-steps of the pair. Lane (row, h) already holds exactly those 32 values of its row in the
//     four fp16 A fragments of the pair - it converts them in registers (v_cvt_scalef32_pk_fp4_f16, fixed power-of-two scale args.q_scale,
//     semantics measured by tools/ubench/cvt_fp4_probe.hip). NO fp4 copy of any activation exists in HBM or LDS.
//   * The weights' lo plane is packed once in that element order (gate128_layout.h, g128q; stylesinger_amd.lib.pack_gate_q4): 16 bytes per
//     lane half + one E8M0 scale byte, inside the second half of the weight line of the pair's odd step. Even steps fetch only their hi halves.
//   * Per pair 32 fp16 MFMAs + 8 block-scaled ones instead of 64: 0.63 of the matrix time at the measured issue rates.
// NOT YET RUN ON HARDWARE (written after the round's GPU budget was spent): reached only through ss_gemm_bf16_gate128q, not dispatched to.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include "pair16.h"
#include "gate128_layout.h"
#include <type_traits>
#include <utility>

typedef ss_f32x16 f32x16;
typedef ss_bf16x8 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

namespace {

using namespace g128;
using g128q::CCS;   // K = 256 channels per tap = 8 chunks of 32

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}
template <class F, int... I>
__device__ __forceinline__ void unrolled_steps(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}

__global__ __launch_bounds__(256, 2) void gate128q_kernel(const ss_gemm_bf16_args a, int m_tiles_per_item, int m_tiles, int n_tiles, int d,
                                                          unsigned long long* clock_probe) {
  // ss_set_clock_probe: workgroup 0 reports the shader cycles and 100 MHz ticks its first wave lived (-> the clock the launch sustained)
  const bool probing = clock_probe != nullptr && blockIdx.x == 0;
  unsigned long long probe_c0 = 0, probe_r0 = 0;
  if (probing) {
    probe_c0 = __builtin_readcyclecounter();
    probe_r0 = __builtin_amdgcn_s_memrealtime();
  }
  extern __shared__ __attribute__((aligned(16))) char smem_g128q[];   // 80 KB: two workgroups per CU
  // [A0 20 K][B0 16 K][A1 20 K][B1 16 K]: the operands of the LAST step live in A1 / B1, so the first 36 KB are free while it runs
  char* const A0 = smem_g128q;
  char* const B0 = A0 + AROWS * A_ROWB;
  char* const A1 = B0 + BN * B_ROWB;
  char* const B1 = A1 + AROWS * A_ROWB;
  // epilogue view: two 32-KB addend quarters and a 16-KB output staging tile
  char* const EQ0 = smem_g128q;
  char* const EQ1 = smem_g128q + 32 * 1024;
  char* const OUT = smem_g128q + 64 * 1024;

  // consecutive workgroups walk row tiles of the SAME column tile (ids = mod 8 -> one XCD): the weight slice stays in that XCD's L2
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
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const int ldw = 3 * a.K * 2;   // 16-bit terms per packed weight row: 3 taps, both planes

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

  // ---- DMA roles (gate128_layout.h). A: wave w issues pieces w + 4 j (j < 5): piece w + 4 j starts 64 j rows after piece w and the swizzle
  // (row >> 2) & 3 does not depend on j: ONE per-lane offset. Only the hi plane (first 64 bytes of a chunk's 128-byte line) is fetched.
  // Rows before the item (negative offset = >= 2^31 unsigned) and rows >= len are out of range: the DMA writes zeros (the conv's padding).
  const int a_voff = ((t0 - HALO + a_dma_row(wave, lane)) * a.lda + a_dma_slot(wave, lane) * 8) * 2;
  const int a_tail_dead = wave < 1 ? 0 : (int)0x80000000;   // piece w + 16 = LDS rows 256 + 16 w ..: only rows < BM + 2 HALO = 272 are ever read
  auto piece_a = [&](char* buf, int cc, int j) {
    // the row offset of piece j goes into the VGPR offset (one add): for rows before the item the per-lane offset is negative and the
    // hardware adds the SGPR offset without wrapping - a positive SGPR part would leave valid rows of later pieces out of range
    glds16(rsrc_a, buf + (wave + 4 * j) * 1024, (a_voff + 64 * j * a.lda * 2) | (j == 4 ? a_tail_dead : 0), cc * 128);
  };
  // weight lines of EVEN steps carry nothing in their second half (slots 4-7): those lanes fetch nothing; odd steps carry the pair's fp4 lo
  // terms in slots 4, 5 and their scales in slot 6 (slot 7 unused)
  const int b_slot = b_dma_slot(wave, lane);
  const int b_voff = ((n0 + b_dma_row(wave, lane)) * ldw + b_slot * 8) * 2;   // piece w + 4 j: 32 j rows further, same swizzle
  const int b_dead_even = b_slot >= 4 ? (int)0x80000000 : 0, b_dead_odd = b_slot == 7 ? (int)0x80000000 : 0;
  auto piece_b = [&](char* buf, int S, int j) {   // the weight tile of step S
    glds16(rsrc_w, buf + (wave + 4 * j) * 1024, b_voff | ((S & 1) ? b_dead_odd : b_dead_even), g128q::step_line(S) * 128 + 32 * j * ldw * 2);
  };

  // ---- fragment addresses: (row base, swizzle ^ lh) per tap, one XOR per read; 32 m more rows leave the swizzle unchanged
  int a_base[3], a_sw[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = a_frag_row(wm, 0, l31, (j - 1) * d);
    a_base[j] = row * A_ROWB;
    a_sw[j] = a_swz(row) ^ lh;
  }
  const int b_row = b_frag_row(wn, 0, l31);
  const int b_base = b_row * B_ROWB;
  const int b_sw = b_swz(b_row) ^ lh;
  const int b_scale_off = b_base + ((6 ^ b_swz(b_row)) << 4) + lh;   // byte lh of logical slot 6: this lane half's block scale (+ n * 32 * B_ROWB)

  // conditioner addend: fp32 [rows][lde]; the tile's 128 packed columns are 512 B contiguous per row -> two rows per DMA instruction
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Eb ? (const void*)Eb : (const void*)a.W), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
  // piece w + 4 j of quarter q: tile rows e_dma_tile_row(w + 4 j, lane, q) = [2 w + (lane >> 5)] + 32 q + (j < 4 ? 8 j : 128 + 8 (j - 4))
  const int e_voff = ((t0 + 2 * wave + (lane >> 5)) * a.lde + n0) * 4 + e_dma_col_byte(lane);
  auto piece_e = [&](char* buf, int q, int j) {
    glds16(rsrc_e, buf + (wave + 4 * j) * 1024, e_voff, (q * 32 + (j < 4 ? 8 * j : 128 + 8 * (j - 4))) * a.lde * 4);
  };
  auto dma_e = [&](char* buf, int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) piece_e(buf, q, j);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // step S = 3 cc + tap. Per step the fp16 product a x hi only: k-step 0 (8 MFMAs) inside the step, k-step 1 (8) deferred past the next
  // barrier as in gate128_kernel. Per PAIR of steps one block-scaled group (8 MFMAs) a_q x lo_q, issued after the barrier that follows the odd
  // step (deferred with that step's k-step 1: 16 deferred MFMAs after odd steps, 8 after even ones). The DMA pieces of a step are issued one
  // after each of its first MFMAs (deferred ones first, then k-step 0's).
  const float qs = a.q_scale;                                    // the activations' fixed fp4 scale: q = fp4(v / qs)
  const int sa = (int)((__builtin_bit_cast(unsigned, qs) >> 23) & 0xffu);   // ... as the instruction's E8M0 scale byte: qs is a power of two, its biased exponent IS 127 + log2(qs)
  bf16x8 p_ah[4], p_bh[2];
  unsigned aq[4][4];          // [m][register r]: the pair's 32 A values of this lane as fp4, r = 2 * (step parity) + k-step
  v8i bq[2];                  // [n]: the pair's 32 weight-lo values of this lane as fp4 (registers 0-3)
  int sb[2] = {127, 127};     // [n]: their block scale
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 8; ++i) bq[n][i] = 0;
  auto cvt8 = [&](const bf16x8& f) {   // 8 fp16 -> 8 fp4: element t in nibble t & 1 of byte t >> 1
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
  auto step = [&](auto stag) {
    constexpr int S = decltype(stag)::value;
    constexpr int CC = S / 3, TAP = S % 3;
    constexpr bool LAST = S + 1 >= 3 * CCS, ODD = (S & 1) != 0;
    const char* Ac = (CC & 1) ? A1 : A0;
    const char* Bc = (S & 1) ? B1 : B0;
    char* Bn = (S & 1) ? B0 : B1;
    char* An = (CC & 1) ? A0 : A1;
    if constexpr (TAP == 2 && CC + 1 < CCS) wait_vmcnt<5>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    auto rd_a = [&](int ks2, bf16x8 (&f)[4]) {   // ks2 = 2 ks; a_sw carries lh
      const int ao = a_base[TAP] + ((ks2 ^ a_sw[TAP]) << 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) f[m] = *reinterpret_cast<const bf16x8*>(Ac + ao + m * 32 * A_ROWB);
    };
    auto rd_b = [&](int slot, bf16x8 (&f)[2]) {   // slot = 2 ks (hi plane) or 4 (this lane half's fp4 lo terms of the pair); b_sw carries lh
      const int bo = b_base + ((slot ^ b_sw) << 4);
#pragma unroll
      for (int n = 0; n < 2; ++n) f[n] = *reinterpret_cast<const bf16x8*>(Bc + bo + n * 32 * B_ROWB);
    };
    bf16x8 ah0[4], bh0[2];
    rd_a(0, ah0);
    rd_b(0, bh0);
    __builtin_amdgcn_sched_barrier(0);
    // DMA pieces this step issues, in this order (the vmcnt counts rely on it): the weight tile of step S+1, at tap 1 the A chunk cc+1, in the
    // last step the first addend quarter (into A0 / B0, which that step does not read)
    constexpr int NB = LAST ? 0 : 4, NA = (TAP == 1 && CC + 1 < CCS) ? 5 : 0, NE = LAST ? 8 : 0, NP = NB + NA + NE;
    auto piece = [&](int i) {
      if (i < NB) piece_b(Bn, S + 1, i);
      else if (i < NB + NA) piece_a(An, CC + 1, i - NB);
      else piece_e(EQ0, 0, i - NB - NA);
    };
    // MFMA slots of this step before its second k-step's fragments are needed: the deferred ones of step S-1 (its k-step 1; after an odd step
    // also the pair's block-scaled group), then this step's k-step 0
    constexpr int ND = S == 0 ? 0 : (ODD ? 8 : 16);   // S odd -> step S-1 was even: 8 deferred; S even -> S-1 odd: 16
    static_assert(NP <= ND + 8, "more DMA pieces than MFMA slots to spread them over");
    if constexpr (S == 0) {
#pragma unroll
      for (int i = 0; i < NP; ++i) piece(i);
    }
    auto spread = [&](int i) {   // one DMA piece after each of the step's first NP MFMAs
      if (S > 0 && i < NP) {
        __builtin_amdgcn_sched_barrier(0);
        piece(i);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      if (i < 8) mfma_h((i >> 1) & 3, i & 1, p_ah, p_bh);   // k-step 1 of step S-1
      else mfma_q(((i - 8) >> 1) & 3, (i - 8) & 1);          // the block-scaled group of the pair that ended with step S-1
      spread(i);
    }
    __builtin_amdgcn_sched_barrier(0);
    // this step's k-step 1 fragments (used after the next barrier) and, in odd steps, the pair's weight-lo terms + scales: in flight under k-step 0
    rd_a(2, p_ah);
    rd_b(2, p_bh);
    [[maybe_unused]] bf16x8 bqr[2];
    if constexpr (ODD) rd_b(4, bqr);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mfma_h((k >> 1) & 3, k & 1, ah0, bh0);                 // k-step 0 of this step
      spread(ND + k);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ODD) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const u32x4 w = __builtin_bit_cast(u32x4, bqr[n]);
#pragma unroll
        for (int i = 0; i < 4; ++i) bq[n][i] = (int)w[i];
        sb[n] = *reinterpret_cast<const uint8_t*>(Bc + b_scale_off + n * 32 * B_ROWB);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // my own A values of this step as fp4: registers 2 * parity + ks of the pair's operand
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      aq[m][2 * (S & 1)] = cvt8(ah0[m]);
      aq[m][2 * (S & 1) + 1] = cvt8(p_ah[m]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LAST) {   // nothing follows: the deferred work of the last (odd) step now
#pragma unroll
      for (int i = 0; i < 8; ++i) mfma_h((i >> 1) & 3, i & 1, p_ah, p_bh);
#pragma unroll
      for (int i = 0; i < 8; ++i) mfma_q((i >> 1) & 3, i & 1);
    }
  };
#pragma unroll
  for (int j = 0; j < 5; ++j) piece_a(A0, 0, j);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) piece_b(B0, 0, j);
  __builtin_amdgcn_sched_barrier(0);
  static_assert((CCS & 1) == 0, "the epilogue's LDS plan assumes the last step reads A1 / B1");
  unrolled_steps(step, std::make_integer_sequence<int, 3 * CCS>{});

  // ---- epilogue (gate256_kernel's, with this tile's constants): the fp32 addend tile (256 rows x 512 B) arrives by LDS-DMA in four quarters of
  // 64 rows, double buffered (quarter 0 was issued inside the last MFMA step into the 36 KB that step no longer uses); the fp16 gate outputs of
  // a quarter are staged in LDS and leave as 16-byte stores (the hi halves of the pair layout's 128-byte groups only).
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
  const int e_rd = e_read_lds(wm, wn, l31, lh, 0, 0);        // + rr * E_ROWB (+ 128 for the second operand)
  const int o_wr = out_write_lds(wm, wn, l31, lh, 0);        // + rr * OUT_ROWB
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const char* Eq = (q & 1) ? EQ1 : EQ0;
    // my pieces of quarter q have landed; younger operations that may still fly, in issue order
    //   E0 | E1 | pass 0: E2, NST stores | pass 1: E3, NST stores | pass 2: NST stores | pass 3: NST stores
    constexpr int NST = 4;   // stores per thread and pass
    if (q == 0) wait_vmcnt<8>();
    else if (q == 1) wait_vmcnt<8 + NST>();
    else if (q == 2) wait_vmcnt<8 + 2 * NST>();
    else wait_vmcnt<2 * NST>();
    __builtin_amdgcn_s_barrier();  // everyone's pieces landed; everyone finished reading the staging tile of quarter q-1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = acc_rr(r);
      const float e0 = *reinterpret_cast<const float*>(Eq + e_rd + rr * E_ROWB);
      const float e1 = *reinterpret_cast<const float*>(Eq + e_rd + rr * E_ROWB + 128);
      float g = act(fmaf(acc[q][0][r], a.out_scale, b0 + e0), m0, s0, h0) * act(fmaf(acc[q][1][r], a.out_scale, b1 + e1), m1, s1, h1);
      if (t0 + 128 * wm + 32 * q + 4 * lh + rr >= row_lim) g = 0.f;
      *reinterpret_cast<uint16_t*>(OUT + o_wr + rr * OUT_ROWB) = ss_f2t<true>(g);   // the second plane of the output rows is not written
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): my staging writes are done
    __builtin_amdgcn_s_barrier();         // the staging tile is complete; everyone finished reading addend quarter q
    if (q + 2 < 4) dma_e((q & 1) ? EQ1 : EQ0, q + 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // 64 rows x 256 B = 1024 pieces of 16 B, four per thread
      const int p = tid + 256 * j;
      const int c16 = out_store_c16(p);
      const int grow = t0 + out_store_tile_row(p, q);
      const uint4 v = *reinterpret_cast<const uint4*>(OUT + p * 16);
      const bool ok = (n0 >> 1) + 32 * (c16 >> 3) < a.N && !(c16 & 4);   // N is a multiple of 32 (checked by the launcher); hi halves only
      const int off = ok ? grow * a.ldc * 2 + n0 * 2 + c16 * 16 : (int)0x80000000;   // logical channel n0/2 sits at physical element n0
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), rsrc_c, off, 0, 0);   // rows >= T dropped
    }
  }
  if (probing && tid == 0) {
    atomicAdd(clock_probe, (unsigned long long)__builtin_readcyclecounter() - probe_c0);
    atomicAdd(clock_probe + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - probe_r0);
  }
}

}  // namespace

// 1 if ss_gemm_bf16 should hand this GATE launch to the two-workgroups-per-CU kernel: fp16x2 operands, three symmetric taps, dilation <= 8,
// K = 256, Np a multiple of 128, and enough rows that 256-row tiles fill the chip several times over
extern "C" int ss_gemm_bf16_gate128q_ok(const ss_gemm_bf16_args* a) {
  if (!a || a->split != 3 || a->epi != SS_HEPI_GATE || a->ntaps != 3) return 0;
  const int d = a->tap_off[2];
  if (d < 1 || d > HALO || a->tap_off[0] != -d || a->tap_off[1] != 0) return 0;
  if (a->K != 256 || (a->Np % BN) != 0 || (a->lda % 8) != 0 || (a->N % 32) != 0 || (a->ldc % 8) != 0 || (a->lde % 4) != 0) return 0;
  if (a->lda < 2 * a->K || a->ldc < 2 * a->N || 2 * a->N > a->Np || !(a->out_scale > 0.f && a->out_scale <= 1.f) || !(a->q_scale > 0.f)) return 0;
  if ((int64_t)a->T * a->lda * 2 >= (1ll << 31) || (int64_t)a->T * a->lde * 4 >= (1ll << 31) || (int64_t)a->T * a->ldc * 2 >= (1ll << 31)) return 0;
  // everything the launcher insists on: a launch that misses one of these falls back to the fp16x2 kernels instead of failing
  if ((((uintptr_t)a->A) & 15) != 0 || (((uintptr_t)a->W) & 15) != 0 || (a->a_batch_stride & 7) != 0 || !a->C) return 0;
  if (a->E && ((((uintptr_t)a->E) & 15) != 0 || (a->e_batch_stride & 3) != 0)) return 0;
  const long tiles = (long)ss_cdiv(a->T, BM) * a->B * (a->Np / BN);
  return (g_ss_tuning.q4_force || tiles >= 8l * ss_n_cu()) ? 1 : 0;   // four rounds of two workgroups per CU
}

// ---- fp16q4 range guard: max |a| / (6 q_scale) over the hi plane of the A operand (rows < lens[b]) -> atomicMax on the float's bits
namespace {
__global__ void q4_guard_kernel(const uint16_t* __restrict__ A, int64_t a_batch_stride, int lda, int K, int T, const int32_t* __restrict__ lens, float inv_limit,
                                unsigned int* __restrict__ out, int64_t n) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {   // one 16-byte group of 8 channels each
    const int g = (int)(i % (K / 8));
    const int64_t r = i / (K / 8);
    const int b = (int)(r / T), t = (int)(r % T);
    if (lens && t >= lens[b]) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(A + (int64_t)b * a_batch_stride + (int64_t)t * lda + (g >> 2) * 64 + (g & 3) * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      m = fmaxf(m, fabsf(ss_t2f_packed<true>(w[k], 0)));
      m = fmaxf(m, fabsf(ss_t2f_packed<true>(w[k], 1)));
    }
  }
  m *= inv_limit;
  if (!(m == m)) m = INFINITY;   // a NaN operand counts as out of range
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
}  // namespace

int ss_q4_guard_launch(const ss_gemm_bf16_args* a, int which, void* stream) {
  if (!g_ss_q4_guard) return SS_OK;
  const int64_t n = (int64_t)a->B * a->T * (a->K / 8);
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(q4_guard_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a->A, a->a_batch_stride, a->lda, a->K, a->T, a->lens, 1.0f / (6.0f * a->q_scale),
                     g_ss_q4_guard + which, n);
  SS_CHECK_LAUNCH("ss_q4_guard");
  return SS_OK;
}

extern "C" int ss_gemm_bf16_gate128q(const ss_gemm_bf16_args* args, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_gemm_bf16_gate128q: null args");
  const ss_gemm_bf16_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "ss_gemm_bf16_gate128q: null A/W/C");
  SS_CHECK_ARG(a.split == 3 && a.out_scale > 0.f && a.out_scale <= 1.f, "ss_gemm_bf16_gate128q: fp16q4 operands only (split = 3, 0 < out_scale <= 1)");
  {
    int ex = 0;
    const float mant = frexpf(a.q_scale, &ex);
    SS_CHECK_ARG(a.q_scale > 0.f && mant == 0.5f && ex >= -20 && ex <= 20, "ss_gemm_bf16_gate128q: q_scale must be a power of two (got %g)", (double)a.q_scale);
  }
  SS_CHECK_ARG(a.epi == SS_HEPI_GATE && a.ntaps == 3 && a.tap_off[1] == 0 && a.tap_off[0] == -a.tap_off[2] && a.tap_off[2] >= 1 &&
                   a.tap_off[2] <= HALO, "ss_gemm_bf16_gate128q: GATE with taps (-d, 0, d), 1 <= d <= 8 only");
  SS_CHECK_ARG(a.K == 256 && (a.Np % BN) == 0 && 2 * a.N <= a.Np && (a.lda % 8) == 0 && (a.N % 32) == 0 && (a.ldc % 8) == 0 && (a.lde % 4) == 0 &&
                   a.lda >= 2 * a.K && a.ldc >= 2 * a.N, "ss_gemm_bf16_gate128q: K = 256, Np %% 128, N %% 32, lda %% 8 and >= 2 K, ldc %% 8 and >= 2 N, lde %% 4");
  SS_CHECK_ARG((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0 && (a.a_batch_stride & 7) == 0 && (!a.E || ((((uintptr_t)a.E) & 15) == 0 && (a.e_batch_stride & 3) == 0)),
               "ss_gemm_bf16_gate128q: A / W / E must be 16-byte aligned");
  SS_CHECK_ARG((int64_t)a.T * a.lda * 2 < (1ll << 31) && (int64_t)a.T * a.lde * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 2 < (1ll << 31) &&
                   (int64_t)a.Np * 3 * a.K * 4 < (1ll << 31), "ss_gemm_bf16_gate128q: item too large for 32-bit offsets");
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const int n_tiles = a.Np / BN;
  const int grid = ss_cdiv(m_tiles, 8) * 8 * n_tiles;
  const size_t lds = (size_t)80 * 1024;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gate128q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    ss_set_error("ss_gemm_bf16_gate128q: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(e));
    return SS_ERR_HIP;
  }
  SS_PROPAGATE(ss_q4_guard_launch(&a, 0, stream));
  hipLaunchKernelGGL(gate128q_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, m_tiles_per_item, m_tiles, n_tiles, a.tap_off[2], g_ss_tuning.clock_probe);
  SS_CHECK_LAUNCH("ss_gemm_bf16_gate128q");
  return SS_OK;
}

extern "C" int ss_gate128q_kindex(int32_t* out, int n) {
  SS_CHECK_ARG(out != nullptr && n >= g128q::PAIRS * 2 * 32, "ss_gate128q_kindex: need room for %d entries", g128q::PAIRS * 2 * 32);
  for (int p = 0; p < g128q::PAIRS; ++p)
    for (int h = 0; h < 2; ++h)
      for (int e = 0; e < 32; ++e) out[(p * 2 + h) * 32 + e] = g128q::q_kindex(p, h, e, 256);
  return g128q::PAIRS * 2 * 32;
}
