// Winograd F(4,3) dilated conv + conditioner addend + gate (modules/diff/net.py:66-73) with the fp32 products computed on the BF16 matrix
// cores from split operands ("bf16x3" precision mode, opt-in: hparams['mfma_precision'] = "bf16x3"; the default stays exact-fp32 MFMA).
//
// The fp32 matrix pipe of gfx950 is 16x slower than the bf16 one (157 vs 2500 TFLOP/s). Each operand is written as the sum of three bf16
// terms, a = a_hi + a_mid + a_lo (round-to-nearest each: together all 24 significand bits), and a product keeps the six partial products
// whose weight is >= 2^-16 of the leading one, accumulated in fp32 smallest first:
//     a.b ~= (hi.lo + lo.hi + mid.mid) + (hi.mid + mid.hi) + hi.hi
// i.e. 6 bf16 MFMAs (v_mfma_f32_16x16x32_bf16, 16 cycles, K = 32) where the exact form issues 8 fp32 MFMAs of 32 cycles: 0.375 of the matrix
// time. Numerics BEFORE this kernel (oracle/bf16x3_numerics.py: the arithmetic swapped into the oracle's denoiser GEMMs, against the REAL
// reference's goldens): a 4096x256x512 GEMM is 2.0e-6 from float64 (plain fp32: 3.5e-6); mel L1 vs the reference 3.07e-7 after the 100-step
// chain and 3.44e-7 after the 1000-step chain (plain fp32: 3.14e-7 / 3.48e-7) - indistinguishable from fp32.
//
// Structure: the 16x16-tile kernel of wino43_gate16.hip (wave tile 16 MT quads x 16 columns, in-wave DPP gate exchange, weights global ->
// registers, same prologue / epilogue / tile pick), with
//   * the weights pre-split at pack time into three bf16 planes per Winograd component ([Np][6][3][Kp] bf16);
//   * the transformed A tile split into three bf16 planes when it is staged (v_cvt_pk_bf16_f32: round-to-nearest-even, the same rounding
//     torch's .bfloat16() applies to the weights): LDS image [6 components][3 planes][16 MT rows][32 bf16] in two halves of three components:
//     while the matrix cores work on one half the other is built (54 KB at MT = 3 -> 2 workgroups per CU; MT = 3 makes the C2 mel launch
//     exactly 512 workgroups = 2 per CU). The first form of this kernel staged all six components in one VALU-only phase followed by a
//     matrix-only phase: 67 us per C2 mel launch - the two workgroups of a CU stayed in lockstep and nothing overlapped;
//   * 64-byte LDS rows: 16-byte slot s of row r lives at r * 64 + ((s ^ ((r & 8) ? 3 : 0)) << 4), conflict-free for the ds_read_b128 lane
//     groups of gfx950 ({0-3, 12-15, 20-27}, ...: per residue of r mod 4 the four lanes of a group hit slots 0, 1, 2, 3).
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 32;
constexpr int BN = 64;
constexpr int NC = 6;
constexpr int ROWB = BK * 2;   // bytes of one LDS row (32 bf16)

__device__ __forceinline__ int swz64(int row) { return (row & 8) ? 3 : 0; }

// two fp32 values -> their three bf16 terms, packed pairwise (low half = first value)
__device__ __forceinline__ void split3(float x, float y, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  auto pk = [](float p, float q) {
    const f32x2 v = {p, q};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  };
  hi = pk(x, y);
  const float rx = x - __builtin_bit_cast(float, hi << 16), ry = y - __builtin_bit_cast(float, hi & 0xffff0000u);
  mid = pk(rx, ry);
  lo = pk(rx - __builtin_bit_cast(float, mid << 16), ry - __builtin_bit_cast(float, mid & 0xffff0000u));
}

__device__ __forceinline__ float4 vfma(float c, const float4& r, const float4& v) {
  return make_float4(fmaf(c, r.x, v.x), fmaf(c, r.y, v.y), fmaf(c, r.z, v.z), fmaf(c, r.w, v.w));
}
__device__ __forceinline__ float2 vfma(float c, const float2& r, const float2& v) { return make_float2(fmaf(c, r.x, v.x), fmaf(c, r.y, v.y)); }
__device__ __forceinline__ float4 vadd(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float2 vadd(const float2& a, const float2& b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float4 vsub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float2 vsub(const float2& a, const float2& b) { return make_float2(a.x - b.x, a.y - b.y); }

template <int MT>
__global__ __launch_bounds__(256, 2) void wino43_gate16x_kernel(const ss_conv_gemm_args a, const uint16_t* __restrict__ Wx,
                                                                                 int q_tiles_per_item, int q_tiles, int n_tiles, int log2d) {
  constexpr int BQ = 16 * MT;
  constexpr int NFULL = BQ / 32;             // staging passes of 32 rows x 8 four-float slots
  constexpr bool HALF = (BQ % 32) != 0;      // + one pass of 16 rows x 16 two-float half slots
  constexpr int PLANE = BQ * ROWB;           // bytes of one bf16 plane of one component
  extern __shared__ __attribute__((aligned(16))) char smem_x[];   // [6][3][BQ][64 B]

  const int id = blockIdx.x;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int qt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (qt >= q_tiles) return;
  const int b = __builtin_amdgcn_readfirstlane(qt / q_tiles_per_item);
  const int q0 = __builtin_amdgcn_readfirstlane((qt % q_tiles_per_item) * BQ);
  const int n0 = nt * BN;
  const int d = 1 << log2d;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const float* abiasg = a.a_bias ? a.a_bias + (int64_t)grp_w * a.a_bias_group_stride : nullptr;
  const int kchunks = a.Kp / BK;
  // weights: [n tile][wave][K chunk][component][plane][lane][8 bf16] (ss_split3_weights): one fetch instruction of a wave = 1 KB contiguous
  const int wtile = NC * 3 * a.Kp * BN;   // bf16 of one 64-column tile

  auto uniform_ptr = [](const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Wx + (int64_t)grp_w * a.w_group_stride), 0, __builtin_amdgcn_readfirstlane((a.Np / BN) * wtile * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(abiasg ? (const void*)abiasg : (const void*)Wx), 0, __builtin_amdgcn_readfirstlane(abiasg ? a.Cin * 4 : 0), 0x00020000);

  // ---- staging roles (as wino43_gate16.hip): thread -> (quad row, K slot) of the raw rows; roff = byte offset of raw row r (frame
  // t + (r-1)d) or out of range (-> 0) outside [0, len); mc = what dstep enters each term with (coefficient sums over the VALID rows)
  const int st_c4 = tid & 7, st_row = tid >> 3;
  const int sh_c2 = tid & 15, sh_row = tid >> 4;
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
    set_mc(1, v[4] - 4.f * v[2]);
    set_mc(2, v[3] - 4.f * v[1]);
    set_mc(3, v[4] - v[2]);
    set_mc(4, v[3] - v[1]);
  };
#pragma unroll
  for (int i = 0; i < NFULL; ++i) row_setup(q0 + st_row + i * 32, st_c4 * 4, roff4[i], [&](int j, float x) { mc4[j][i] = x; });
  if constexpr (HALF) row_setup(q0 + NFULL * 32 + sh_row, sh_c2 * 2, roffh, [&](int j, float x) { mch[j] = x; });

  u32x4 rr4[NFULL > 0 ? NFULL : 1][6];
  u32x2 rr2[6];
  float4 rpb4;
  float2 rpb2;
  auto load_rows = [&](int ci0b) {
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

  // LDS write addresses of a thread's 8-byte (4 bf16) / 4-byte (2 bf16) pieces inside a plane
  int a_wr4[NFULL > 0 ? NFULL : 1];
#pragma unroll
  for (int i = 0; i < NFULL; ++i) {
    const int row = st_row + i * 32;
    a_wr4[i] = row * ROWB + (((st_c4 >> 1) ^ swz64(row)) << 4) + (st_c4 & 1) * 8;
  }
  const int rowh = NFULL * 32 + sh_row;
  const int a_wrh = rowh * ROWB + (((sh_c2 >> 2) ^ swz64(rowh)) << 4) + (sh_c2 & 3) * 4;

  // One Winograd component from the raw rows of a K chunk -> its three bf16 planes in LDS. Shared terms as in wino43_gate16.hip:
  //   A = r4 - 4 r2, B = r3 - 4 r1 (kept from c1 for c2), C = r4 - r2, D = r3 - r1 (kept from c3 for c4)
  //   c0 = 4 r0 - 5 r2 + r4, c1 = A + B, c2 = A - B | c3 = C + 2 D, c4 = C - 2 D, c5 = 4 r1 - 5 r3 + r5
  // dstep enters every term with the sum of its coefficients over the valid rows (mc). The LDS holds two halves, (c0, c1, c2) and
  // (c3, c4, c5): while the matrix cores work on one half, the other half (of this K chunk, or the first half of the next) is built.
  float4 tP4[NFULL > 0 ? NFULL : 1], tQ4[NFULL > 0 ? NFULL : 1];
  float2 tPh, tQh;
  auto comp_value = [&](auto jtag, const auto& pb, const auto& m, auto R, auto& tP, auto& tQ) {
    constexpr int J = decltype(jtag)::value;
    if constexpr (J == 0) return vfma(m(0), pb, vfma(-5.f, R(2), vfma(4.f, R(0), R(4))));
    else if constexpr (J == 1) {
      tP = vfma(m(1), pb, vfma(-4.f, R(2), R(4)));
      tQ = vfma(m(2), pb, vfma(-4.f, R(1), R(3)));
      return vadd(tP, tQ);
    } else if constexpr (J == 2) return vsub(tP, tQ);
    else if constexpr (J == 3) {
      tP = vfma(m(3), pb, vsub(R(4), R(2)));
      tQ = vfma(m(4), pb, vsub(R(3), R(1)));
      return vfma(2.f, tQ, tP);
    } else if constexpr (J == 4) return vfma(-2.f, tQ, tP);
    else return vfma(m(5), pb, vfma(-5.f, R(3), vfma(4.f, R(1), R(5))));
  };
  auto build = [&](auto jtag) {
    constexpr int J = decltype(jtag)::value;
#pragma unroll
    for (int i = 0; i < NFULL; ++i) {
      const float4 v = comp_value(jtag, rpb4, [&](int t) { return mc4[t][i]; }, [&](int q) { return __builtin_bit_cast(float4, rr4[i][q]); },
                                  tP4[i], tQ4[i]);
      uint32_t h0, m0, l0, h1, m1, l1;
      split3(v.x, v.y, h0, m0, l0);
      split3(v.z, v.w, h1, m1, l1);
      char* p = smem_x + J * 3 * PLANE + a_wr4[i];
      *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + PLANE) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(p + 2 * PLANE) = u32x2{l0, l1};
    }
    if constexpr (HALF) {
      const float2 v = comp_value(jtag, rpb2, [&](int t) { return mch[t]; }, [&](int q) { return __builtin_bit_cast(float2, rr2[q]); }, tPh, tQh);
      uint32_t h0, m0, l0;
      split3(v.x, v.y, h0, m0, l0);
      char* p = smem_x + J * 3 * PLANE + a_wrh;
      *reinterpret_cast<uint32_t*>(p) = h0;
      *reinterpret_cast<uint32_t*>(p + PLANE) = m0;
      *reinterpret_cast<uint32_t*>(p + 2 * PLANE) = l0;
    }
  };

  // ---- matrix side: lane (lc, kg) of wave w; column pc, K elements [8 kg, 8 kg + 8) of the chunk
  int w_voff, a_rd;
  {
    const int lane = tid & 63;
    const int lc = lane & 15, kg = lane >> 4;
    w_voff = (nt * wtile + wave * (wtile / 4)) * 2 + lane * 16;
    a_rd = lc * ROWB + ((kg ^ swz64(lc)) << 4);
  }   // + (j * 3 + p) * PLANE + m * 16 * ROWB (16 m rows leave row & 8 unchanged)
  // weights: one register slot per component (3 planes x 16 B). The slots of a half are refilled while the OTHER half is on the matrix
  // cores - half a K chunk of cover for the fetch - and are dead between their last use and that refill.
  bf16x8 bst[NC][3];
  auto load_b = [&](auto jtag, int k) {
    constexpr int J = decltype(jtag)::value;
    const int cb = __builtin_amdgcn_readfirstlane(((k * NC + J) * 3) * 1024);   // 1 KB per (K chunk, component, plane) and wave
#pragma unroll
    for (int p = 0; p < 3; ++p)
      bst[J][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, cb + p * 1024, 0));
  };
  f32x4 acc[NC][MT];
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[j][m][r] = 0.f;

  // Component J on the matrix cores (three plane fragments per row tile, six products smallest first; the MT MFMAs of a product are
  // independent, consecutive products chain on the same accumulators - the order is pinned: left alone the scheduler strings up to five
  // dependent MFMAs behind each other). Between its two halves of three products runs `mid`: builds of the other LDS half and fetches.
  auto comp = [&](auto jtag, auto&& mid) {
    constexpr int J = decltype(jtag)::value;
    bf16x8 af[3][MT];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int m = 0; m < MT; ++m) af[p][m] = *reinterpret_cast<const bf16x8*>(smem_x + (J * 3 + p) * PLANE + m * 16 * ROWB + a_rd);
    __builtin_amdgcn_sched_barrier(0);
    auto mm = [&](int pa, int pb_) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[J][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pa][m], bst[J][pb_], acc[J][m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    mm(0, 2);
    mm(2, 0);
    mm(1, 1);
    mid();
    __builtin_amdgcn_sched_barrier(0);
    mm(0, 1);
    mm(1, 0);
    mm(0, 0);
  };
  const int cs = BK * 4;   // bytes of one K chunk of a raw fp32 row
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  using J4 = std::integral_constant<int, 4>;
  using J5 = std::integral_constant<int, 5>;
  // prologue: rows(0) -> first half of chunk 0; weights of that half
  load_rows(0);
  load_b(J0{}, 0);
  load_b(J1{}, 0);
  load_b(J2{}, 0);
  build(J0{});
  build(J1{});
  build(J2{});
  __syncthreads();
  // Schedule of a K chunk (every fetch gets as much matrix time as the registers allow before its first use):
  //   half 0 on the matrix cores: c0 | build c5 (needs r1, r3, r5), fetch the weights of c3..c5   c1 | build c3 (last use of the raw
  //                               rows), fetch rows(k+1)                                          c2 | build c4 (from the kept C, D)
  //   half 1:                     c3 | fetch the weights of c0..c2 of chunk k+1                   c4 | build c0, c1 of chunk k+1   c5 | build c2
  auto first_half = [&](int k, bool more) {
    comp(J0{}, [&]() {
      build(J5{});
      __builtin_amdgcn_sched_barrier(0);
      load_b(J3{}, k);
      load_b(J4{}, k);
      load_b(J5{}, k);
    });
    comp(J1{}, [&]() {
      build(J3{});
      __builtin_amdgcn_sched_barrier(0);   // stores first, then the fetch into the same registers
      if (more) load_rows((k + 1) * cs);
    });
    comp(J2{}, [&]() { build(J4{}); });
    __syncthreads();
  };
  for (int k = 0; k + 1 < kchunks; ++k) {
    first_half(k, true);
    comp(J3{}, [&]() {
      load_b(J0{}, k + 1);
      load_b(J1{}, k + 1);
      load_b(J2{}, k + 1);
    });
    comp(J4{}, [&]() {
      build(J0{});
      build(J1{});
    });
    comp(J5{}, [&]() { build(J2{}); });
    __syncthreads();
  }
  first_half(kchunks - 1, false);
  // ---- epilogue (wino43_gate16.hip): output transform, conditioner addend, gate; accumulator (m, r) = quad 16 m + 4 kg + r, column lc
  // (lane coordinates derived afresh: nothing lane-dependent of the prologue has to stay in a register across the K loop for this)
  {
  const int lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int lc = lane & 15, kg = lane >> 4;
  const int c7 = lc & 7, chi = lc >> 3;
  const int pc = n0 + 8 * wave + c7 + 32 * chi;
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Eb ? (const void*)Eb : (const void*)Wx), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
  const int oc = (n0 >> 1) + 8 * wave + c7;
  const bool col_ok = oc < a.N;
  const int oob = col_ok ? 0 : (int)0x80000000;
  const int lde4 = a.lde * 4, ldc4 = a.ldc * 4;
  // the conditioner addend (16 MT values per lane, a 40 KB row stride: one L2 / HBM miss per element) is fetched under the last half step,
  // into the registers the raw rows no longer need
  float pe[MT][4][4];
  comp(J3{}, [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int qm = q0 + 16 * m + 4 * kg;
      const int tm = qm + 3 * (qm & ~(d - 1));
      const int e_base = tm * lde4 + (pc * 4 + oob);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int dr = r + 3 * (r & ~(d - 1));
#pragma unroll
        for (int o = 0; o < 4; ++o)
          pe[m][r][o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, e_base, (dr + o * d) * lde4, 0));
      }
    }
  });
  comp(J4{}, []() {});
  comp(J5{}, []() {});
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);
  const bool use_sig = (chi == 0) == (a.gate_mode == 0);
  const float am = (use_sig ? -1.0f : -2.0f) * 1.44269504088896340736f, as = use_sig ? 1.0f : 2.0f, ah = use_sig ? 0.0f : -1.0f;
  auto act = [&](float x) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * am)), as, ah); };
  auto partner = [](float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xf, 0xf, true));
  };
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const int my_first = chi ? 2 * d : 0;
  const float bs = (a.bias && col_ok) ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc] : 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int qm = q0 + 16 * m + 4 * kg;
    const int tm = qm + 3 * (qm & ~(d - 1));
    const int c_base = (tm + my_first) * ldc4 + (oc * 4 + oob);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dr = r + 3 * (r & ~(d - 1));
      const float a0 = acc[0][m][r], a5 = acc[5][m][r];
      const float s12 = acc[1][m][r] + acc[2][m][r] + bs, d12 = acc[1][m][r] - acc[2][m][r] + bs;
      const float s34 = acc[3][m][r] + acc[4][m][r], d34 = acc[3][m][r] - acc[4][m][r];
      const float u0 = act(a0 + s12 + s34 + pe[m][r][0]);
      const float u1 = act(fmaf(2.0f, d34, d12) + pe[m][r][1]);
      const float u2 = act(fmaf(4.0f, s34, s12) + pe[m][r][2]);
      const float u3 = act(fmaf(8.0f, d34, d12) + a5 + pe[m][r][3]);
      const float g0 = u0 * partner(u0), g1 = u1 * partner(u1), g2 = u2 * partner(u2), g3 = u3 * partner(u3);
      float ga = chi ? g2 : g0, gb = chi ? g3 : g1;
      const int ta = tm + dr + my_first;
      if (ta >= row_lim) ga = 0.f;
      if (ta + d >= row_lim) gb = 0.f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ga), rsrc_c, c_base, dr * ldc4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gb), rsrc_c, c_base, (dr + d) * ldc4, 0);
    }
  }
  }
}

template <int MT>
void launch16x(const ss_conv_gemm_args& a, const uint16_t* Wx, int dilation, int log2d, hipStream_t stream) {
  constexpr int BQ = 16 * MT;
  const int quads_per_item = ss_cdiv(a.T, 4 * dilation) * dilation;
  const int q_tiles_per_item = ss_cdiv(quads_per_item, BQ);
  const int q_tiles = q_tiles_per_item * a.B;
  const int n_tiles = a.Np / BN;
  const int grid = ss_cdiv(q_tiles, 8) * 8 * n_tiles;
  const size_t lds = (size_t)NC * 3 * BQ * ROWB;
  hipLaunchKernelGGL(wino43_gate16x_kernel<MT>, dim3(grid), dim3(256), lds, stream, a, Wx, q_tiles_per_item, q_tiles, n_tiles, log2d);
}

// packed F(4,3) weights [Np][6][Kp] fp32 -> the three bf16 terms of every element in the order the kernel fetches them:
// [n tile (64 columns)][wave][K chunk][component][plane][lane][8 bf16], lane = kg * 16 + chi * 8 + c7 holding column 64 nt + 8 w + c7 + 32 chi,
// K elements 32 k + 8 kg + (0..7)
__global__ void split3_pack_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int Np, int Kp) {
  const int64_t n = (int64_t)Np * NC * Kp;
  const int kchunks = Kp / BK;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % Kp);
    const int j = (int)((i / Kp) % NC);
    const int col = (int)(i / ((int64_t)Kp * NC));
    const int nt = col / BN, cl = col % BN;
    const int chi = cl / 32, w = (cl % 32) / 8, c7 = cl % 8;
    const int k = kk / BK, kg = (kk % BK) / 8, e = kk % 8;
    uint32_t h, m, l;
    split3(src[i], 0.f, h, m, l);
    const int lane = kg * 16 + chi * 8 + c7;
    const int64_t base = ((((int64_t)(nt * 4 + w) * kchunks + k) * NC + j) * 3) * 512 + lane * 8 + e;
    dst[base] = (uint16_t)h;
    dst[base + 512] = (uint16_t)m;
    dst[base + 1024] = (uint16_t)l;
  }
}

}  // namespace

// packed F(4,3) weights [Np][6 * Kp] fp32 (ss_pack_conv_weight of the transformed taps) -> [Np * 18 * Kp] bf16 in the fetch order of
// wino43_gate16x_kernel (Np a multiple of 64, Kp a multiple of 32)
extern "C" int ss_split3_weights(const float* src, void* dst, int Np, int Kp, void* stream) {
  SS_CHECK_ARG(src && dst && Np > 0 && (Np % BN) == 0 && Kp > 0 && (Kp % BK) == 0, "ss_split3_weights: Np %% 64, Kp %% 32");
  const int64_t n = (int64_t)Np * NC * Kp;
  const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(split3_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, Np, Kp);
  SS_CHECK_LAUNCH("ss_split3_weights");
  return SS_OK;
}

extern "C" int ss_wino43_gate16x(const ss_conv_gemm_args* args, const void* Wx, int dilation, int mt, void* stream) {
  SS_CHECK_ARG(args != nullptr && Wx != nullptr, "ss_wino43_gate16x: null args / weights");
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.C, "ss_wino43_gate16x: null A/C");
  SS_CHECK_ARG(dilation >= 1 && (dilation & (dilation - 1)) == 0 && dilation <= 64, "ss_wino43_gate16x: dilation %d must be a power of two <= 64", dilation);
  SS_CHECK_ARG((a.Cin % BK) == 0 && a.Kp == a.Cin && (a.lda & 3) == 0, "ss_wino43_gate16x: Cin=%d must be a multiple of 32 and Kp == Cin", a.Cin);
  SS_CHECK_ARG((a.Np % 64) == 0 && 2 * a.N <= a.Np, "ss_wino43_gate16x: Np=%d must be a multiple of 64 and >= 2*N", a.Np);
  SS_CHECK_ARG((int64_t)a.T * a.lda * 4 < (1ll << 31) && (!a.E || (int64_t)a.T * a.lde * 4 < (1ll << 31)) &&
                   (int64_t)a.Np * NC * 3 * a.Kp * 2 < (1ll << 31) && (int64_t)a.T * a.ldc * 4 < (1ll << 31),
               "ss_wino43_gate16x: item too large for 32-bit offsets");
  SS_CHECK_ARG(mt == 0 || mt == 2 || mt == 3, "ss_wino43_gate16x: mt=%d must be 0 (auto), 2 or 3", mt);
  int log2d = 0;
  while ((1 << log2d) < dilation) ++log2d;
  if (mt == 0) {
    mt = ss_wino43_gate16_pick(a.B, a.T, a.Np, dilation);
    if (mt == 0) mt = 3;   // many rounds per launch: the larger tile
  }
  if (mt == 2) launch16x<2>(a, (const uint16_t*)Wx, dilation, log2d, (hipStream_t)stream);
  else launch16x<3>(a, (const uint16_t*)Wx, dilation, log2d, (hipStream_t)stream);
  SS_CHECK_LAUNCH("ss_wino43_gate16x");
  return SS_OK;
}
