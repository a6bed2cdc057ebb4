// "bf16x3" form of ss_gemm16_store (opt-in precision mode, see wino43_gate16x.hip): C = act(A . W^T + bias) for the K = L*C skip GEMM of the
// deferred-skip denoiser loops with every fp32 product computed on the BF16 matrix cores from operands split into three bf16 terms
// (a = hi + mid + lo, round-to-nearest each; six exact partial products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid accumulated in fp32,
// smallest first): 6 x v_mfma_f32_16x16x32_bf16 (16 cycles) where the exact form issues 8 x v_mfma_f32_16x16x4_f32 (32 cycles).
//   * W is pre-split at pack time (ss_split3_gemm16_weights) and stored in fetch order [n tile][wave][K chunk][plane][lane][8 bf16]: a
//     wave fetches its 16 columns of a K chunk with three 1-KB-contiguous loads, straight into registers (2 stages);
//   * A (the gate outputs of all layers, fp32 in HBM) is fetched two K chunks ahead into registers, split when it is staged:
//     LDS image [2 buffers][3 planes][16 MT rows][32 bf16] with the 64-byte-row slot swizzle of wino43_gate16x.hip; one barrier per chunk.
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
constexpr int ROWB = BK * 2;

__device__ __forceinline__ int swz64(int row) { return (row & 8) ? 3 : 0; }

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

template <int MT>
__global__ __launch_bounds__(256, 3) void gemm16x_store_kernel(const ss_conv_gemm_args a, const uint16_t* __restrict__ Wx, int m_tiles_per_item,
                                                                int m_tiles, int n_tiles) {
  constexpr int BM = 16 * MT;
  constexpr int NP = BM / 32;            // staging passes of 32 rows x 8 four-float slots
  static_assert(BM % 32 == 0, "row tile must be a multiple of 32");
  constexpr int PLANE = BM * ROWB;       // bytes of one bf16 plane
  __shared__ __attribute__((aligned(16))) char As[2 * 3 * PLANE];

  const int id = blockIdx.x;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int mt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (mt >= m_tiles) return;
  const int b = __builtin_amdgcn_readfirstlane(mt / m_tiles_per_item);
  const int t0 = __builtin_amdgcn_readfirstlane((mt % m_tiles_per_item) * BM);
  const int n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, kg = lane >> 4;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const int kchunks = a.Kp / BK;

  auto uniform_ptr = [](const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const int np64 = (a.Np + BN - 1) / BN;   // 64-column tiles in the split weights
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Wx + (int64_t)grp_w * a.w_group_stride), 0, __builtin_amdgcn_readfirstlane(np64 * BN * 3 * a.Kp * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);

  // A staging: thread -> (row tid >> 3 (+ 32 per pass), K slot tid & 7 of four floats); rows >= len are out of range -> 0
  const int st_c4 = tid & 7, st_row = tid >> 3;
  int a_voff[NP], a_wr[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = st_row + 32 * i;
    a_voff[i] = ((t0 + row) * a.lda + st_c4 * 4) * 4;
    a_wr[i] = row * ROWB + (((st_c4 >> 1) ^ swz64(row)) << 4) + (st_c4 & 1) * 8;
  }
  u32x4 ar[2][NP];
  auto load_a = [&](auto stag, int c) {
    constexpr int S = decltype(stag)::value;
    const int so = __builtin_amdgcn_readfirstlane(c * (BK * 4));
#pragma unroll
    for (int i = 0; i < NP; ++i) ar[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_voff[i], so, 0);
  };
  auto stage_a = [&](auto stag, char* buf) {   // registers -> three bf16 planes
    constexpr int S = decltype(stag)::value;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const float4 v = __builtin_bit_cast(float4, ar[S][i]);
      uint32_t h0, m0, l0, h1, m1, l1;
      split3(v.x, v.y, h0, m0, l0);
      split3(v.z, v.w, h1, m1, l1);
      char* p = buf + a_wr[i];
      *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + PLANE) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(p + 2 * PLANE) = u32x2{l0, l1};
    }
  };
  // weights: [n tile][wave][K chunk][plane][lane][8 bf16]
  const int w_voff = ((nt * 4 + wave) * kchunks * 3 * 512) * 2 + lane * 16;
  bf16x8 bst[2][3];
  auto load_b = [&](auto stag, int c) {
    constexpr int S = decltype(stag)::value;
    const int so = __builtin_amdgcn_readfirstlane(c * 3 * 1024);
#pragma unroll
    for (int p = 0; p < 3; ++p) bst[S][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, so + p * 1024, 0));
  };
  const int a_rd = lc * ROWB + ((kg ^ swz64(lc)) << 4);

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // chunk c (parity PAR = c & 1): LDS buffer PAR holds A(c) split; registers ar[PAR ^ 1] hold the raw A(c+1), bst[PAR] the weights of c.
  auto chunk = [&](auto ptag, int c) {
    constexpr int PAR = decltype(ptag)::value;
    using SP = std::integral_constant<int, PAR>;
    using SN = std::integral_constant<int, PAR ^ 1>;
    __syncthreads();   // A(c) staged by everyone; everyone done reading A(c-1) (the buffer A(c+1) goes to)
    const char* Ac = As + PAR * 3 * PLANE;
    if (c + 2 < kchunks) load_a(SP{}, c + 2);   // raw A two chunks ahead, into the registers A(c) was staged from
    constexpr int MH = (MT + 1) / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int m0 = half * MH;
      bf16x8 af[3][MH];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < MH; ++m)
          if (m0 + m < MT) af[p][m] = *reinterpret_cast<const bf16x8*>(Ac + p * PLANE + (m0 + m) * 16 * ROWB + a_rd);
      __builtin_amdgcn_sched_barrier(0);
      auto mm = [&](int pa, int pb_) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
          if (m0 + m < MT) acc[m0 + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[pa][m], bst[PAR][pb_], acc[m0 + m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      mm(0, 2);
      mm(2, 0);
      mm(1, 1);
      mm(0, 1);
      mm(1, 0);
      mm(0, 0);
    }
    if (c + 1 < kchunks) stage_a(SN{}, As + (PAR ^ 1) * 3 * PLANE);   // A(c+1): registers -> the other LDS buffer
    __builtin_amdgcn_sched_barrier(0);
    if (c + 2 < kchunks) load_b(SP{}, c + 2);
  };
  load_a(S0{}, 0);
  load_b(S0{}, 0);
  if (kchunks > 1) {
    load_a(S1{}, 1);
    load_b(S1{}, 1);
  }
  stage_a(S0{}, As);
  int c = 0;
  for (; c + 2 <= kchunks; c += 2) {
    chunk(S0{}, c);
    chunk(S1{}, c + 1);
  }
  if (c < kchunks) chunk(S0{}, c);

  const int col = n0 + 16 * wave + lc;
  const bool col_ok = col < a.N;
  const int oob = col_ok ? 0 : (int)0x80000000;
  const float bs = (a.bias && col_ok) ? a.bias[(int64_t)grp_w * a.bias_group_stride + col] : 0.f;
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const int c_base = ((t0 + 4 * kg) * a.ldc + col) * 4 + oob;
  const bool relu = a.act == SS_ACT_RELU;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float o = acc[m][r] + bs;
      if (relu) o = fmaxf(o, 0.f);
      if (t0 + 16 * m + 4 * kg + r >= row_lim) o = 0.f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsrc_c, c_base, (16 * m + r) * a.ldc * 4, 0);
    }
}

// [Np][Kp] fp32 -> the three bf16 terms of every element in fetch order [n tile (64 columns, zero padded)][wave][K chunk][plane][lane][8]:
// lane = kg * 16 + lc holds column 64 nt + 16 w + lc, K elements 32 c + 8 kg + (0..7)
__global__ void split3_gemm16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int Np, int Kp) {
  const int np64 = (Np + BN - 1) / BN * BN;
  const int64_t n = (int64_t)np64 * Kp;
  const int kch = Kp / BK;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % Kp), col = (int)(i / Kp);
    const int nt = col / BN, w = (col % BN) / 16, lc = col % 16;
    const int c = kk / BK, kg = (kk % BK) / 8, e = kk % 8;
    uint32_t h, m, l;
    split3(col < Np ? src[(int64_t)col * Kp + kk] : 0.f, 0.f, h, m, l);
    const int64_t base = (((int64_t)(nt * 4 + w) * kch + c) * 3) * 512 + (kg * 16 + lc) * 8 + e;
    dst[base] = (uint16_t)h;
    dst[base + 512] = (uint16_t)m;
    dst[base + 1024] = (uint16_t)l;
  }
}

template <int MT>
void launch_x(const ss_conv_gemm_args& a, const uint16_t* Wx, hipStream_t stream) {
  const int m_tiles_per_item = ss_cdiv(a.T, 16 * MT), m_tiles = m_tiles_per_item * a.B, n_tiles = ss_cdiv(a.N, BN);
  hipLaunchKernelGGL(gemm16x_store_kernel<MT>, dim3(ss_cdiv(m_tiles, 8) * 8 * n_tiles), dim3(256), 0, stream, a, Wx, m_tiles_per_item, m_tiles, n_tiles);
}

}  // namespace

// bf16 elements of ss_split3_gemm16_weights' output for an [Np][Kp] weight
extern "C" int64_t ss_split3_gemm16_elems(int Np, int Kp) { return (int64_t)((Np + BN - 1) / BN * BN) * 3 * Kp; }

extern "C" int ss_split3_gemm16_weights(const float* src, void* dst, int Np, int Kp, void* stream) {
  SS_CHECK_ARG(src && dst && Np > 0 && Kp > 0 && (Kp % BK) == 0, "ss_split3_gemm16_weights: Kp %% 32");
  const int64_t n = (int64_t)((Np + BN - 1) / BN * BN) * Kp;
  const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(split3_gemm16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst, Np, Kp);
  SS_CHECK_LAUNCH("ss_split3_gemm16_weights");
  return SS_OK;
}

extern "C" int ss_gemm16x_store(const ss_conv_gemm_args* args, const void* Wx, int mt, void* stream) {
  SS_CHECK_ARG(args != nullptr && Wx != nullptr, "ss_gemm16x_store: null args / weights");
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.C, "ss_gemm16x_store: null A/C");
  SS_CHECK_ARG(a.ntaps == 1 && a.tap_off[0] == 0, "ss_gemm16x_store: one tap at offset 0 only");
  SS_CHECK_ARG(a.Kp == a.Cin && (a.Kp % BK) == 0 && (a.lda & 3) == 0, "ss_gemm16x_store: K=%d must equal Kp and be a multiple of 32, lda %% 4 == 0", a.Cin);
  SS_CHECK_ARG(a.N > 0 && a.N <= a.Np, "ss_gemm16x_store: bad N=%d Np=%d", a.N, a.Np);
  SS_CHECK_ARG(a.a_scale == 1.0f && a.a_lrelu == 1.0f && a.a_bias == nullptr && a.mfma_bf16 == 0 && a.pre_scale == 1.0f && a.post_scale == 1.0f &&
                   a.R == nullptr && !a.accumulate && (a.act == SS_ACT_NONE || a.act == SS_ACT_RELU),
               "ss_gemm16x_store: plain C = act(A.W^T + bias) only (act none | relu)");
  SS_CHECK_ARG((int64_t)a.T * a.lda * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 4 < (1ll << 31) && ss_split3_gemm16_elems(a.Np, a.Kp) * 2 < (1ll << 31),
               "ss_gemm16x_store: item too large for 32-bit offsets");
  SS_CHECK_ARG(mt == 0 || mt == 4 || mt == 6 || mt == 8, "ss_gemm16x_store: mt=%d must be 0 (auto), 4, 6 or 8", mt);
  if (mt == 0) mt = ss_gemm16_pick(a.B, a.T, a.N);
  hipStream_t s = (hipStream_t)stream;
  switch (mt) {
    case 4: launch_x<4>(a, (const uint16_t*)Wx, s); break;
    case 6: launch_x<6>(a, (const uint16_t*)Wx, s); break;
    default: launch_x<8>(a, (const uint16_t*)Wx, s); break;
  }
  SS_CHECK_LAUNCH("ss_gemm16x_store");
  return SS_OK;
}
