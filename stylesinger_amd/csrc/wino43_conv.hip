// Grouped Winograd F(4,3) form of the HiFi-GAN ResBlock convs (modules/hifigan/hifigan_nsf.py:54-61 via hifigan.py ResBlock1: k = 3 / 7 / 11
// taps, dilation 1 / 3 / 5, C -> C channels, leaky-relu in front, bias + residual behind), exact-fp32 MFMA.
//
// A k-tap conv is split into G = ceil(k / 3) groups of three taps (the last group zero padded). Group g reads the frames
// t + s_g + i d (s_g = (3 g - (k - 1) / 2) d, i = 0..5) of a quad of output frames (t, t + d, t + 2d, t + 3d) and contributes six products
// m_j += c_j(g) . G_j(g) with the F(4,3) transforms of wino43_gate.hip; the six accumulators run over all groups and all input channels,
// ONE output transform at the end:
//   z[t] = m0+m1+m2+m3+m4   z[t+d] = (m1-m2) + 2(m3-m4)   z[t+2d] = (m1+m2) + 4(m3+m4)   z[t+3d] = (m1-m2) + 8(m3-m4) + m5
// 6 G matrix products per 4 output frames instead of 4 k:  k = 3: 0.50, k = 7: 0.64, k = 11: 0.55 of the direct form's matrix work.
// Numerics (oracle/wino_vocoder_numerics.py, CPU emulation of this arithmetic inside the oracle's vocoder, against the REAL reference's
// waveforms): wav max error 1.1e-7 / 1.3e-7 on the two golden cases (direct fp32: 0.9e-7); tests keep WAV_TOL = 1e-5.
//
// Quads are formed inside groups of 4 d frames (t = (q / d) 4 d + q % d) for ANY dilation (D is a template parameter: the vocoder's 1, 3, 5).
// Tile = 64 quads (256 frames) x 64 output channels, 4 waves x 6 accumulators of 32 x 32; K loop over (group, 32-channel chunk), six
// components per step, one LDS tile + barrier per component as in wino43_gate.hip; components are consumed in the order c1, c0, c5, c2, c3,
// c4 with the shared terms of wino43_gate16.hip (18 VALU per element instead of 28; the raw rows are dead after the third build, so the
// rows of the next step are fetched three components ahead). Rows outside [0, len) are out of the buffer range -> read 0: the row offset
// lives in a VGPR (negative = out of range), the K-chunk offset in an SGPR; moving to the next tap group adds 3 d rows to the VGPRs.
// The leaky-relu of the input is applied ONCE per fetched row register (max(x, slope x) == x >= 0 ? x : slope x for 0 < slope < 1): one
// packed multiply per two elements + one bare v_max_f32 per element (ss_lrelu_max).
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LD = BK;
constexpr int BQ = 64;  // quads per tile (= 256 output frames)
constexpr int BN = 64;
constexpr int NC = 6;

__device__ __forceinline__ int lds_slot(int row, int slot) { return row * LD + ((slot ^ ((row >> 1) & 7)) << 2); }

__device__ __forceinline__ float4 vfma(float c, const float4& r, const float4& v) {
  return make_float4(fmaf(c, r.x, v.x), fmaf(c, r.y, v.y), fmaf(c, r.z, v.z), fmaf(c, r.w, v.w));
}
__device__ __forceinline__ float4 vadd(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 vsub(const float4& a, const float4& b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

template <int D>
__device__ __forceinline__ int frame_of_quad(int q) {
  if constexpr (D == 1) return 4 * q;
  else return (q / D) * (4 * D) + (q % D);
}

template <int D, bool LRELU>
__global__ __launch_bounds__(256, 2) void wino43_conv_kernel(const ss_conv_gemm_args a, int q_tiles_per_item, int q_tiles, int n_tiles,
                                                             int groups, int lo_rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BQ][LD]
  float* Bs = smem + 2 * BQ * LD;    // [2][BN][LD]

  // 8 consecutive workgroups share a column tile and walk 8 row tiles: the weight slice stays hot in the XCD's L2
  const int id = blockIdx.x;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int qt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (qt >= q_tiles) return;
  const int b = __builtin_amdgcn_readfirstlane(qt / q_tiles_per_item);
  const int q0 = __builtin_amdgcn_readfirstlane((qt % q_tiles_per_item) * BQ);
  const int n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int len = ss_uniform_len(a.lens, b, a.T);
  const int kchunks = a.Kp / BK;
  const int ldw = groups * NC * a.Kp;

  auto uniform_ptr = [](const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.W), 0, __builtin_amdgcn_readfirstlane(a.Np * ldw * 4), 0x00020000);

  const int st_c4 = tid & 7;
  const int st_row = tid >> 3;  // 0..31; two passes cover the 64 quad rows / 64 weight rows
  // roff[i][r] = byte offset of raw row r of quad row i for the CURRENT tap group: frame t - lo + r d (+ 3 d per group). A negative
  // offset is >= 2^31 as the unsigned offset the buffer check sees -> out of range -> 0, like every frame >= len.
  int roff[2][6];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = frame_of_quad<D>(q0 + st_row + i * 32);
#pragma unroll
    for (int r = 0; r < 6; ++r) roff[i][r] = ((t - lo_rows + r * D) * a.lda + st_c4 * 4) * 4;
  }
  const int group_step = 3 * D * a.lda * 4;
  int w_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w_voff[i] = ((n0 + st_row + i * 32) * ldw + st_c4 * 4) * 4;

  u32x4 rr[2][6], rb[2];
  auto load_rows = [&](int ci0b) {  // ci0b = byte offset of the K chunk inside a row (wave-uniform -> SGPR soffset)
    ci0b = __builtin_amdgcn_readfirstlane(ci0b);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 6; ++r) rr[i][r] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, roff[i][r], ci0b, 0);
  };
  auto load_b = [&](int cb) {  // cb = byte offset of the weight chunk inside a packed row (wave-uniform)
    cb = __builtin_amdgcn_readfirstlane(cb);
#pragma unroll
    for (int i = 0; i < 2; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[i], cb, 0);
  };
  const float slope = a.a_lrelu;
  auto act_rows = [&]() {
    if constexpr (LRELU) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          const f32x4 v = __builtin_bit_cast(f32x4, rr[i][r]);
          const f32x4 sv = v * slope;   // two v_pk_mul_f32
          f32x4 o;
          o.x = ss_lrelu_max(v.x, sv.x);
          o.y = ss_lrelu_max(v.y, sv.y);
          o.z = ss_lrelu_max(v.z, sv.z);
          o.w = ss_lrelu_max(v.w, sv.w);
          rr[i][r] = __builtin_bit_cast(u32x4, o);
        }
    }
  };
  const int a_wr[2] = {lds_slot(st_row, st_c4), lds_slot(st_row + 32, st_c4)};
  // position p builds component ORD[p] (c1, c0, c5, c2, c3, c4) from the raw rows / the kept terms
  //   A = r4 - 4 r2, B = r3 - 4 r1, C = r4 - r2, Dd = r3 - r1   ->   c1 = A + B, c2 = A - B, c3 = C + 2 Dd, c4 = C - 2 Dd
  //   c0 = 4 r0 - 5 r2 + r4, c5 = 4 r1 - 5 r3 + r5
  float4 tA[2], tB[2], tC[2], tD[2];
  auto store_a = [&](float* Ad, auto ptag) {
    constexpr int P = decltype(ptag)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      auto R = [&](int q) { return __builtin_bit_cast(float4, rr[i][q]); };
      float4 v;
      if constexpr (P == 0) {
        tA[i] = vfma(-4.f, R(2), R(4));
        tB[i] = vfma(-4.f, R(1), R(3));
        v = vadd(tA[i], tB[i]);
      } else if constexpr (P == 1) {
        tC[i] = vsub(R(4), R(2));
        v = vfma(-5.f, R(2), vfma(4.f, R(0), R(4)));
      } else if constexpr (P == 2) {
        tD[i] = vsub(R(3), R(1));
        v = vfma(-5.f, R(3), vfma(4.f, R(1), R(5)));
      } else if constexpr (P == 3) {
        v = vsub(tA[i], tB[i]);
      } else if constexpr (P == 4) {
        v = vfma(2.f, tD[i], tC[i]);
      } else {
        v = vfma(-2.f, tD[i], tC[i]);
      }
      *reinterpret_cast<float4*>(Ad + a_wr[i]) = v;
    }
  };
  auto store_b = [&](float* Bd) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<float4*>(Bd + lds_slot(st_row + i * 32, st_c4)) = __builtin_bit_cast(float4, rb[i]);
  };

  f32x16 acc[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  using P5 = std::integral_constant<int, 5>;
  constexpr int ORD[6] = {1, 0, 5, 2, 3, 4};
  const int kb = a.Kp * 4;   // bytes of one component in a packed weight row
  const int cs = BK * 4;     // bytes of one K chunk
  // step (g, k) = one tap group x one 32-channel chunk; position p of it uses the weights at (g * 6 + ORD[p]) * kb + k * cs
  const int steps = groups * kchunks;

  const int swz = (l31 >> 1) & 7;
  const int a_row = (wm * 32 + l31) * LD;
  const int b_row = (wn * 32 + l31) * LD;
  auto read_frags = [&](const float* Ac, const float* Bc, int q, float4& af, float4& bf) {
    const int so = ((2 * q + lh) ^ swz) << 2;
    af = *reinterpret_cast<const float4*>(Ac + a_row + so);
    bf = *reinterpret_cast<const float4*>(Bc + b_row + so);
  };
  auto mfma4 = [&](f32x16& c, const float4& af, const float4& bf) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, c, 0, 0, 0);
  };
  // position P of a step: MFMAs from LDS buffer P & 1 into acc[ORD[P]]; in their shadow the component of position P+1 (or position 0 of
  // the next step) is built and stored with the weight registers, then the weights two positions ahead are fetched into the same
  // registers and - at position 2, the raw rows being dead - the raw rows of the next step.
  //   wb  = weight byte offset of the position two ahead;  rows_cb = K-chunk byte offset of the next step's rows
  auto chunk = [&](auto ptag, auto stage_tag, auto fetch_b_tag, auto fetch_rows_tag, int wb, int rows_cb, bool new_group) {
    constexpr int P = decltype(ptag)::value;
    constexpr int CUR = P & 1;
    constexpr int PN = (P + 1) % 6;
    const float* Ac = As + CUR * BQ * LD;
    const float* Bc = Bs + CUR * BN * LD;
    float4 af0, af1, bf0, bf1;
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(acc[ORD[P]], af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(acc[ORD[P]], af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(stage_tag)::value) {
      if constexpr (PN == 0) act_rows();   // first use of the next step's raw rows
      store_a(As + (CUR ^ 1) * BQ * LD, std::integral_constant<int, PN>{});
      store_b(Bs + (CUR ^ 1) * BN * LD);
    }
    __builtin_amdgcn_sched_barrier(0);  // stores first, then the fetches into the SAME registers
    if constexpr (decltype(fetch_b_tag)::value) load_b(wb);
    if constexpr (decltype(fetch_rows_tag)::value) {
      if (new_group) {   // wave-uniform: the next step starts the next tap group, 3 d frames further
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 6; ++r) roff[i][r] += group_step;
      }
      load_rows(rows_cb);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma4(acc[ORD[P]], af0, bf0);
    mfma4(acc[ORD[P]], af1, bf1);
    __syncthreads();
  };
  using Yes = std::true_type;
  using No = std::false_type;

  load_rows(0);
  load_b(ORD[0] * kb);
  act_rows();
  store_a(As, P0{});
  store_b(Bs);
  load_b(ORD[1] * kb);   // weights of position 1: in flight across the barrier
  __syncthreads();

  int g = 0, k = 0;   // current step
  for (int s = 0; s + 1 < steps; ++s) {
    const bool wrap = k + 1 == kchunks;
    const int gn = wrap ? g + 1 : g, kn = wrap ? 0 : k + 1;   // next step
    const int wcur = g * NC * kb + k * cs, wnext = gn * NC * kb + kn * cs;
    chunk(P0{}, Yes{}, Yes{}, No{}, wcur + ORD[2] * kb, 0, false);
    chunk(P1{}, Yes{}, Yes{}, No{}, wcur + ORD[3] * kb, 0, false);
    chunk(P2{}, Yes{}, Yes{}, Yes{}, wcur + ORD[4] * kb, kn * cs, wrap);
    chunk(P3{}, Yes{}, Yes{}, No{}, wcur + ORD[5] * kb, 0, false);
    chunk(P4{}, Yes{}, Yes{}, No{}, wnext + ORD[0] * kb, 0, false);
    chunk(P5{}, Yes{}, Yes{}, No{}, wnext + ORD[1] * kb, 0, false);
    g = gn;
    k = kn;
  }
  {
    const int wcur = g * NC * kb + k * cs;
    chunk(P0{}, Yes{}, Yes{}, No{}, wcur + ORD[2] * kb, 0, false);
    chunk(P1{}, Yes{}, Yes{}, No{}, wcur + ORD[3] * kb, 0, false);
    chunk(P2{}, Yes{}, Yes{}, No{}, wcur + ORD[4] * kb, 0, false);
    chunk(P3{}, Yes{}, Yes{}, No{}, wcur + ORD[5] * kb, 0, false);
    chunk(P4{}, Yes{}, No{}, No{}, 0, 0, false);
    chunk(P5{}, No{}, No{}, No{}, 0, 0, false);
  }

  // ---- epilogue: output transform, then the STORE epilogue of ss_conv_gemm: v = act((z + bias) * pre_scale); v = (v + R) * post_scale
  // (+ C when accumulating); rows >= len are written as 0 (mask_rows), rows >= T dropped by the buffer range check.
  // accumulator row r of this lane -> quad (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32 quads; column n0 + 32 wn + l31
  const int col = n0 + wn * 32 + l31;
  const bool col_ok = col < a.N;
  const int oob = col_ok ? 0 : (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.R ? const_cast<float*>(a.R) + (int64_t)b * a.r_batch_stride : a.C), 0,
      __builtin_amdgcn_readfirstlane(a.R ? (int)((int64_t)a.T * a.ldr * 4) : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(   // previous value of C (accumulate), else an empty range -> 0
      uniform_ptr(a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane(a.accumulate ? (int)((int64_t)a.T * a.ldc * 4) : 0),
      0x00020000);
  const float bs = (a.bias && col_ok) ? a.bias[col] : 0.f;
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const int ldc4 = a.ldc * 4, ldr4 = a.ldr * 4;
  const int qbase = q0 + wm * 32 + 4 * lh;
  const bool lrelu_out = a.act == SS_ACT_LRELU_;
#pragma unroll
  for (int rb4 = 0; rb4 < 4; ++rb4) {   // four quads (16 output frames) at a time: all loads first, then compute + store
    float rv[4][4], pv[4][4];
    int tq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tq[e] = frame_of_quad<D>(qbase + e + 8 * rb4);
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int t = tq[e] + o * D;
        rv[e][o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, t * ldr4 + (col * 4 + oob), 0, 0));
        pv[e][o] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_p, t * ldc4 + (col * 4 + oob), 0, 0));
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = e + 4 * rb4;
      const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
      const float s34 = acc[3][r] + acc[4][r], d34 = acc[3][r] - acc[4][r];
      float z[4];
      z[0] = acc[0][r] + s12 + s34;
      z[1] = fmaf(2.0f, d34, d12);
      z[2] = fmaf(4.0f, s34, s12);
      z[3] = fmaf(8.0f, d34, d12) + acc[5][r];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int t = tq[e] + o * D;
        float v = (z[o] + bs) * a.pre_scale;
        if (lrelu_out) v = ss_lrelu(v, a.act_slope);
        v = (v + rv[e][o]) * a.post_scale + pv[e][o];
        if (t >= row_lim) v = 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_c, t * ldc4 + (col * 4 + oob), 0, 0);
      }
    }
  }
}

template <int D>
void launch_conv(const ss_conv_gemm_args& a, int k, hipStream_t stream) {
  const int quads_per_item = ss_cdiv(a.T, 4 * D) * D;
  const int q_tiles_per_item = ss_cdiv(quads_per_item, BQ);
  const int q_tiles = q_tiles_per_item * a.B;
  const int n_tiles = a.Np / BN;
  const int grid = ss_cdiv(q_tiles, 8) * 8 * n_tiles;
  const size_t lds = (size_t)2 * (BQ + BN) * LD * sizeof(float);
  const int groups = (k + 2) / 3, lo = (k - 1) / 2 * D;
  if (a.a_lrelu != 1.0f)
    hipLaunchKernelGGL((wino43_conv_kernel<D, true>), dim3(grid), dim3(256), lds, stream, a, q_tiles_per_item, q_tiles, n_tiles, groups, lo);
  else
    hipLaunchKernelGGL((wino43_conv_kernel<D, false>), dim3(grid), dim3(256), lds, stream, a, q_tiles_per_item, q_tiles, n_tiles, groups, lo);
}

}  // namespace

extern "C" int ss_wino43_conv_ok(int C, int k, int dilation) {
  return (C % 64 == 0) && (k == 3 || k == 7 || k == 11) && (dilation == 1 || dilation == 3 || dilation == 5) ? 1 : 0;
}

extern "C" int ss_wino43_conv(const ss_conv_gemm_args* args, int k, int dilation, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_wino43_conv: null args");
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "ss_wino43_conv: null A/W/C");
  SS_CHECK_ARG(ss_wino43_conv_ok(a.Cin, k, dilation), "ss_wino43_conv: C=%d (multiple of 64), k=%d (3|7|11), dilation=%d (1|3|5)", a.Cin, k, dilation);
  SS_CHECK_ARG(a.Kp == a.Cin && a.Np == a.Cin && a.N == a.Cin && (a.lda & 3) == 0, "ss_wino43_conv: square conv with Kp == Np == N == Cin");
  SS_CHECK_ARG(a.epi == SS_EPI_STORE && (a.act == SS_ACT_NONE_ || a.act == SS_ACT_LRELU_) && !a.a_bias && a.a_scale == 1.0f && !a.mfma_bf16 &&
                   a.group_size == 0,
               "ss_wino43_conv: STORE epilogue, act none|lrelu, no input bias/scale, fp32, one weight set");
  SS_CHECK_ARG(a.a_lrelu > 0.0f && a.a_lrelu <= 1.0f, "ss_wino43_conv: input leaky-relu slope %g must be in (0, 1]", (double)a.a_lrelu);
  SS_CHECK_ARG(((int64_t)a.T + 1024) * a.lda * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 4 < (1ll << 31) &&
                   (!a.R || (int64_t)a.T * a.ldr * 4 < (1ll << 31)) && (int64_t)a.Np * ((k + 2) / 3) * NC * a.Kp * 4 < (1ll << 31),
               "ss_wino43_conv: item too large for 32-bit offsets");
  if (dilation == 1) launch_conv<1>(a, k, (hipStream_t)stream);
  else if (dilation == 3) launch_conv<3>(a, k, (hipStream_t)stream);
  else launch_conv<5>(a, k, (hipStream_t)stream);
  SS_CHECK_LAUNCH("ss_wino43_conv");
  return SS_OK;
}
