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
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

template <int CCS>
__global__ __launch_bounds__(512, 2) void gate256_kernel(const ss_gemm_bf16_args a, int m_tiles_per_item, int m_tiles, int n_tiles, int d) {
  extern __shared__ __attribute__((aligned(16))) char smem_g256[];   // 144 KB: one workgroup per CU
  char* const A0 = smem_g256;
  char* const A1 = A0 + AROWS * ROWB;
  char* const B0 = A1 + AROWS * ROWB;
  char* const B1 = B0 + BN * ROWB;

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
  const int len = a.lens ? a.lens[b] : a.T;
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const int ldw = 3 * a.K;            // bf16 per packed weight row (3 taps)

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
  int a_voff[5], b_voff[4];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int r = 8 * (wave + 8 * j) + (lane >> 3);
    const int slot = (lane & 7) ^ ((r >> 1) & 7);
    const int grow = t0 - HALO + r;   // may be negative: the byte offset is then >= 2^31 as unsigned -> out of range -> zeros
    a_voff[j] = (r < BM + 2 * HALO) ? (grow * a.lda + slot * 8) * 2 : (int)0x80000000;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = 8 * (wave + 8 * j) + (lane >> 3);
    const int slot = (lane & 7) ^ ((r >> 1) & 7);
    b_voff[j] = ((n0 + r) * ldw + slot * 8) * 2;
  }
  // `dead` = 0x80000000 turns a DMA into out-of-range reads (zeros written, no memory traffic): the pieces past the last step are
  // still issued - into buffers nobody reads any more - so that every step has the same instruction stream and the same counted waits
  auto dma_a = [&](char* buf, int cc, int dead) {
#pragma unroll
    for (int j = 0; j < 5; ++j) glds16(rsrc_a, buf + (wave + 8 * j) * 8 * ROWB, a_voff[j] | dead, cc * (BKH * 2));
  };
  auto dma_b = [&](char* buf, int cc, int tap, int dead) {
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(rsrc_w, buf + (wave + 8 * j) * 8 * ROWB, b_voff[j] | dead, (tap * a.K + cc * BKH) * 2);
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
  auto step = [&](auto stag) {
    constexpr int S = decltype(stag)::value;
    constexpr int CC = S / 3, TAP = S % 3;
    constexpr bool LAST = S + 1 >= 3 * CCS;
    const char* Ac = (CC & 1) ? A1 : A0;
    const char* Bc = (S & 1) ? B1 : B0;
    char* Bn = (S & 1) ? B0 : B1;
    char* An = (CC & 1) ? A0 : A1;
    if constexpr (TAP == 2) wait_vmcnt<5>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    dma_b(Bn, (S + 1) / 3, (S + 1) % 3, LAST ? (int)0x80000000 : 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAP == 1) dma_a(An, CC + 1, CC + 1 >= CCS ? (int)0x80000000 : 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {   // one k-step at a time: 6 fragment reads (24 registers) feed 8 MFMAs; the partner wave of the SIMD
      bf16x8 af[4], bf[2];             // covers the LDS latency with its own MFMAs
      const int ao = a_base[TAP] + (((2 * ks) ^ a_swz[TAP]) << 4);
      const int bo = b_base + (((2 * ks) ^ b_swz) << 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const bf16x8*>(Ac + ao + m * 32 * ROWB);
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[n] = *reinterpret_cast<const bf16x8*>(Bc + bo + n * 32 * ROWB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bf[n], acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  dma_a(A0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  dma_b(B0, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  auto run = [&](auto... ss) { (step(ss), ...); };
  using std::integral_constant;
  run(integral_constant<int, 0>{}, integral_constant<int, 1>{}, integral_constant<int, 2>{}, integral_constant<int, 3>{},
      integral_constant<int, 4>{}, integral_constant<int, 5>{}, integral_constant<int, 6>{}, integral_constant<int, 7>{},
      integral_constant<int, 8>{});
  if constexpr (CCS == 4) run(integral_constant<int, 9>{}, integral_constant<int, 10>{}, integral_constant<int, 11>{});

  // ---- epilogue (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 lh): addend, gate, bf16 store; per row block of 32
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Eb ? (const void*)Eb : (const void*)a.W), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr((uint16_t*)a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 2)), 0x00020000);
  const float* biasg = a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : nullptr;
  const int pc0 = n0 + 64 * wn + l31;            // packed column of the first operand; the second sits 32 further
  const int oc = (pc0 >> 6) * 32 + l31;          // output channel
  const int dead = oc < a.N ? 0 : (int)0x80000000;
  const float b0 = (biasg && !dead) ? biasg[pc0] : 0.f, b1 = (biasg && !dead) ? biasg[pc0 + 32] : 0.f;
  const bool sig_first = a.gate_mode == 0;
  const float L2E = 1.44269504088896340736f;
  const float m0 = (sig_first ? -1.0f : -2.0f) * L2E, s0 = sig_first ? 1.0f : 2.0f, h0 = sig_first ? 0.0f : -1.0f;
  const float m1 = (sig_first ? -2.0f : -1.0f) * L2E, s1 = sig_first ? 2.0f : 1.0f, h1 = sig_first ? -1.0f : 0.0f;
  auto act = [](float x, float mul, float sc, float sh) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * mul)), sc, sh); };
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const int lde4 = a.lde * 4, ldc2 = a.ldc * 2;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row0 = t0 + 128 * wm + 32 * m + 4 * lh;
    const int eoff = (row0 * a.lde + pc0) * 4 | dead;
    const int coff = (row0 * a.ldc + oc) * 2 | dead;
    float e0[16], e1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      e0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, eoff, rr * lde4, 0));
      e1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, eoff, rr * lde4 + 128, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      float g = act(acc[m][0][r] + b0 + e0[r], m0, s0, h0) * act(acc[m][1][r] + b1 + e1[r], m1, s1, h1);
      if (row0 + rr >= row_lim) g = 0.f;
      __builtin_amdgcn_raw_buffer_store_b16(f2bf(g), rsrc_c, coff, rr * ldc2, 0);   // rows >= T: out of range, dropped
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
  if ((a->K != 256 && a->K != 192) || (a->Np % BN) != 0 || (a->lda % 8) != 0) return 0;
  const long tiles = (long)ss_cdiv(a->T, BM) * a->B * (a->Np / BN);
  return tiles >= 1024 ? 1 : 0;
}

extern "C" int ss_gemm_bf16_gate256(const ss_gemm_bf16_args* args, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_gemm_bf16_gate256: null args");
  const ss_gemm_bf16_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "ss_gemm_bf16_gate256: null A/W/C");
  SS_CHECK_ARG(a.epi == SS_HEPI_GATE && a.ntaps == 3 && a.tap_off[1] == 0 && a.tap_off[0] == -a.tap_off[2] && a.tap_off[2] >= 1 &&
                   a.tap_off[2] <= HALO, "ss_gemm_bf16_gate256: GATE with taps (-d, 0, d), 1 <= d <= 8 only");
  SS_CHECK_ARG((a.K == 256 || a.K == 192) && (a.Np % BN) == 0 && 2 * a.N <= a.Np && (a.lda % 8) == 0, "ss_gemm_bf16_gate256: K 192 | 256, Np %% 256, lda %% 8");
  SS_CHECK_ARG((int64_t)a.T * a.lda * 2 < (1ll << 31) && (int64_t)a.T * a.lde * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 2 < (1ll << 31) &&
                   (int64_t)a.Np * 3 * a.K * 2 < (1ll << 31), "ss_gemm_bf16_gate256: item too large for 32-bit offsets");
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const int n_tiles = a.Np / BN;
  const int grid = ss_cdiv(m_tiles, 8) * 8 * n_tiles;
  const size_t lds = (size_t)2 * AROWS * ROWB + (size_t)2 * BN * ROWB;
  auto go = [&](auto kern) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, a, m_tiles_per_item, m_tiles, n_tiles, a.tap_off[2]);
  };
  if (a.K == 256) go(&gate256_kernel<4>);
  else go(&gate256_kernel<3>);
  SS_CHECK_LAUNCH("ss_gemm_bf16_gate256");
  return SS_OK;
}
