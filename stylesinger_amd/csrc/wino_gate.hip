// Winograd F(2,3) version of the denoisers' hot layer: 3-tap dilated conv + conditioner addend + gate
// (modules/diff/net.py:66-73), exact-fp32 MFMA, 1.5x fewer matrix ops than the direct form.
//
//   y[t] = x[t] + dstep   (0 outside [0,len))          z[t] = w0.y[t-d] + w1.y[t] + w2.y[t+d] + E[t]
//   g[t] = sigmoid(z[t][:C]) * tanh(z[t][C:])
//
// Frames t and t+d share three of their four inputs, so the pair (t, t+d) is computed from 4 products instead of 6:
//   m0 = (y[t-d]-y[t+d]).w0     m1 = (y[t]+y[t+d]).(w0+w1+w2)/2     m2 = (y[t+d]-y[t]).(w0-w1+w2)/2     m3 = (y[t]-y[t+2d]).w2
//   z[t] = m0+m1+m2             z[t+d] = m1-m2-m3
// Pairs are formed inside groups of 2d frames (t = g*2d + s, s < d), which works for any power-of-two dilation.
// GEMM view: rows = pairs, 4 "components" each a [pairs x C] x [C x 2C] product with its own accumulator; the input
// transform is fused into the global->LDS stage (two row fetches + one add per element), the output transform into
// the epilogue.  Tile = 64 pairs x 64 packed columns, 4 waves of 32x32 (x4 components = 64 accumulator registers);
// the two N-waves of a tile hold the sigmoid half and the tanh half of the same 32 channels and swap activations
// through LDS.  Main-loop skeleton (LDS swizzle, buffer-resource fetch, MFMA-shadow scheduling) = conv_gemm_kernel.h.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <stdlib.h>
#include <type_traits>
// SS_TRACE (debug builds only, tools/wave_trace.py): every wave of wino_gate_kernel_v2<1> sums, over its K chunks, the shader-clock
// time spent in each phase of a chunk and writes the sums at exit.
#ifdef SS_TRACE
__device__ unsigned long long* g_wino_trace = nullptr;
extern "C" int ss_debug_set_wino_trace(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino_trace), &p, sizeof(p));
}
#endif
#ifndef SS_ABL
#define SS_ABL 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LD = BK;
constexpr int BP = 64;  // pairs per tile (= 128 output frames)

__device__ __forceinline__ int lds_slot(int row, int slot) { return row * LD + ((slot ^ ((row >> 1) & 7)) << 2); }

// TN = 32-column blocks per wave. TN=1: tile 64 pairs x 64 cols (fine-grained: best balance for the f0 pair), the two
// N-waves swap activations through LDS. TN=2: tile 64 x 128, each wave owns both gate operands (twice the MFMAs per
// barrier: best for the mel net where 384 tiles still fill the chip).
template <int TN>
__global__ __launch_bounds__(256) void wino_gate_kernel(const ss_conv_gemm_args a, int p_tiles_per_item, int p_tiles,
                                                        int n_tiles, int log2d, int prio_mode) {
  constexpr int BN = 64 * TN;
  ss_apply_wave_prio(prio_mode);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BP][LD]
  float* Bs = smem + 2 * BP * LD;    // [2][BN][LD]

  const int id = blockIdx.x;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int pt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (pt >= p_tiles) return;
  const int b = pt / p_tiles_per_item;
  const int p0 = (pt % p_tiles_per_item) * BP;
  const int n0 = nt * BN;
  const int d = 1 << log2d;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int len = a.lens ? a.lens[b] : a.T;
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const float* Wg = a.W + (int64_t)grp_w * a.w_group_stride;
  const float* abiasg = a.a_bias ? a.a_bias + (int64_t)grp_w * a.a_bias_group_stride : nullptr;
  const int kchunks = a.Kp / BK;
  const int ldw = 4 * a.Kp;

  auto uniform_ptr = [](const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Wg), 0, __builtin_amdgcn_readfirstlane(a.Np * ldw * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(abiasg ? abiasg : Wg), 0, __builtin_amdgcn_readfirstlane(abiasg ? a.Cin * 4 : 0), 0x00020000);

  const int st_c4 = tid & 7;
  const int st_row = tid >> 3;  // 0..31; two passes cover the 64 pair rows / 64 weight rows
  // frame of pair p: t = (p >> log2d) * 2d + (p & (d-1))
  int t_of[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = p0 + st_row + i * 32;
    t_of[i] = ((p >> log2d) << (log2d + 1)) + (p & (d - 1));
  }
  const int w_off0 = ((n0 + st_row) * ldw + st_c4 * 4) * 4;
  const int w_pass = 32 * ldw * 4;
  const int col_b = st_c4 * 4 * 4;

  // component j -> (row offset of operand A, of operand B, sign of B) in units of d
  //   j0: y[t-d] - y[t+d]   j1: y[t] + y[t+d]   j2: y[t+d] - y[t]   j3: y[t] - y[t+2d]
  u32x4 ra[2][2], rb[2 * TN];
  float4 rpb;
  // description of the chunk being fetched / written (all wave-uniform scalars; selected, never branched on)
  struct Stage { int oa, ob, ci0; float sgn; };
  auto stage_of = [&](int j, int ci0) {
    Stage st;
    st.oa = (j == 0) ? -d : (j == 2) ? d : 0;
    st.ob = (j == 3) ? 2 * d : (j == 2) ? 0 : d;
    st.sgn = (j == 1) ? 1.0f : -1.0f;
    st.ci0 = ci0;
    return st;
  };
  auto load_a = [&](const Stage& k) {
    const int ci = k.ci0 + st_c4 * 4;
    const int oob = ci < a.Cin ? 0 : (int)0x80000000;
    rpb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_bias, ci * 4, 0, 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (((t_of[i] + k.oa) * a.lda + k.ci0) * 4 + col_b) | oob, 0, 0);
      ra[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, (((t_of[i] + k.ob) * a.lda + k.ci0) * 4 + col_b) | oob, 0, 0);
    }
  };
  auto load_b = [&](int c) {
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off0 + i * w_pass + c * (BK * 4), 0, 0);
  };
  auto store_a = [&](int buf, const Stage& k) {
    float* Ad = As + buf * BP * LD;
    const bool c_ok = k.ci0 + st_c4 * 4 < a.Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 va = __builtin_bit_cast(float4, ra[i][0]);
      float4 vb = __builtin_bit_cast(float4, ra[i][1]);
      const bool oka = c_ok && (unsigned)(t_of[i] + k.oa) < (unsigned)len;
      const bool okb = c_ok && (unsigned)(t_of[i] + k.ob) < (unsigned)len;
      // y = x + dstep on real frames, exact 0 on padding (the fetch already returned 0 there)
      const float ma = oka ? 1.0f : 0.0f, mb = okb ? k.sgn : 0.0f;
      float4 v;
      v.x = (va.x + ma * rpb.x) + (k.sgn * vb.x + mb * rpb.x);
      v.y = (va.y + ma * rpb.y) + (k.sgn * vb.y + mb * rpb.y);
      v.z = (va.z + ma * rpb.z) + (k.sgn * vb.z + mb * rpb.z);
      v.w = (va.w + ma * rpb.w) + (k.sgn * vb.w + mb * rpb.w);
      *reinterpret_cast<float4*>(Ad + lds_slot(st_row + i * 32, st_c4)) = v;
    }
  };
  auto store_b = [&](int buf) {
    float* Bd = Bs + buf * BN * LD;
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i)
      *reinterpret_cast<float4*>(Bd + lds_slot(st_row + i * 32, st_c4)) = __builtin_bit_cast(float4, rb[i]);
  };

  f32x16 acc[4][TN];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

  {
    const Stage s0 = stage_of(0, 0);
    load_a(s0);
    load_b(0);
    store_a(0, s0);
    store_b(0);
  }
  __syncthreads();

  const int swz = (l31 >> 1) & 7;
  const int a_row = (wm * 32 + l31) * LD;
  const int b_row = (wn * 32 * TN + l31) * LD;
  struct BF { float4 v[TN]; };
  auto read_frags = [&](const float* Ac, const float* Bc, int q, float4& af, BF& bf) {
    const int so = ((2 * q + lh) ^ swz) << 2;
    af = *reinterpret_cast<const float4*>(Ac + a_row + so);
#pragma unroll
    for (int n = 0; n < TN; ++n) bf.v[n] = *reinterpret_cast<const float4*>(Bc + b_row + n * 32 * LD + so);
  };
  auto mfma4 = [&](f32x16 (&c)[TN], const float4& af, const BF& bf) {
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.v[n].x, c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.v[n].y, c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.v[n].z, c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.v[n].w, c[n], 0, 0, 0);
  };

  // One K chunk of component J accumulated into its own accumulator; the next chunk (same component, or the first
  // chunk of component J+1 at the wrap) is fetched / transformed / written in the shadow of the MFMAs.
  int c = 0;
  auto chunk = [&](auto jtag, f32x16 (&cacc)[TN], int k) {
    constexpr int J = decltype(jtag)::value;
    const bool wrap = (k + 1 >= kchunks);
    const Stage cur_s = stage_of(J, 0), nxt_s = stage_of(J + 1, 0);
    Stage nx;
    nx.oa = wrap ? nxt_s.oa : cur_s.oa;
    nx.ob = wrap ? nxt_s.ob : cur_s.ob;
    nx.sgn = wrap ? nxt_s.sgn : cur_s.sgn;
    nx.ci0 = wrap ? 0 : (k + 1) * BK;
    const int cur = c & 1;
    const float* Ac = As + cur * BP * LD;
    const float* Bc = Bs + cur * BN * LD;
    float4 af0, af1;
    BF bf0, bf1;
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    load_a(nx);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(cacc, af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    load_b(c + 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(cacc, af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    mfma4(cacc, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    store_a(cur ^ 1, nx);
    store_b(cur ^ 1);
    mfma4(cacc, af1, bf1);
    __syncthreads();
    ++c;
  };
  for (int k = 0; k < kchunks; ++k) chunk(std::integral_constant<int, 0>{}, acc[0], k);
  for (int k = 0; k < kchunks; ++k) chunk(std::integral_constant<int, 1>{}, acc[1], k);
  for (int k = 0; k < kchunks; ++k) chunk(std::integral_constant<int, 2>{}, acc[2], k);
  for (int k = 0; k + 1 < kchunks; ++k) chunk(std::integral_constant<int, 3>{}, acc[3], k);
  // Epilogue operands from HBM (the hoisted conditioner slab, 40 KB row stride -> every element is its own miss) are
  // fetched BEFORE the last chunk's MFMAs, into the staging registers that chunk no longer needs, so the ~2 us miss
  // latency hides under 1024+ MFMA cycles instead of stalling every block of the (single) round at the same time.
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  auto frame_of = [&](int pl) {
    const int p = p0 + wm * 32 + pl;
    return ((p >> log2d) << (log2d + 1)) + (p & (d - 1));
  };
  float pe[TN == 2 ? 64 : 32];
  if constexpr (TN == 2) {  // frame t of both gate operands through a buffer resource (1 address register per load);
    // frame t+d is fetched in the epilogue: 228 of the 256 registers of a 2-waves/SIMD kernel are taken
    const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(Eb ? Eb : Wg), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
    const int pc0 = n0 + wn * 64 + l31;
    const int dead = (((pc0 >> 6) * 32 + l31) < a.N) ? 0 : (int)0x80000000;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = frame_of((r & 3) + 8 * (r >> 2) + 4 * lh);
      const int off = ((t * a.lde + pc0) * 4) | dead;
      pe[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off, 0, 0));
      pe[16 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off + 128, 0, 0));
      const int off2 = (((t + d) * a.lde + pc0) * 4) | dead;
      pe[32 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off2, 0, 0));
      pe[48 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off2 + 128, 0, 0));
    }
  } else {  // frames t and t+d of this wave's operand
    const int pc = n0 + wn * 32 + l31;
    const bool col_ok = ((n0 >> 1) + l31) < a.N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = frame_of((r & 3) + 8 * (r >> 2) + 4 * lh);
      pe[r] = (Eb && col_ok && t < a.T) ? Eb[(int64_t)t * a.lde + pc] : 0.f;
      pe[16 + r] = (Eb && col_ok && t + d < a.T) ? Eb[(int64_t)(t + d) * a.lde + pc] : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  {  // last chunk: nothing left to fetch
    const int cur = c & 1;
    const float* Ac = As + cur * BP * LD;
    const float* Bc = Bs + cur * BN * LD;
    float4 af0, af1;
    BF bf0, bf1;
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    mfma4(acc[3], af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    mfma4(acc[3], af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    mfma4(acc[3], af0, bf0);
    mfma4(acc[3], af1, bf1);
  }

  // ---- epilogue: output transform z[t] = m0+m1+m2, z[t+d] = m1-m2-m3, conditioner addend, gate ----
  float* Cb = a.C + (int64_t)b * a.c_batch_stride;
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  if constexpr (TN == 2) {
    // the wave owns both gate operands of channels [oc0, oc0+32): columns pc0 (first operand) and pc0+32 (second)
    const int pc0 = n0 + wn * 64 + l31;
    const int oc = (pc0 >> 6) * 32 + l31;
    if (oc >= a.N) return;
    const float b0 = a.bias ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc0] : 0.f;
    const float b1 = a.bias ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc0 + 32] : 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // half 0: frame t, half 1: frame t+d
      float e0[16], e1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        e0[r] = pe[32 * half + r];  // prefetched under the last chunk
        e1[r] = pe[32 * half + 16 + r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = frame_of((r & 3) + 8 * (r >> 2) + 4 * lh) + half * d;
        if (t >= a.T) continue;
        const float z0 = half == 0 ? acc[0][0][r] + acc[1][0][r] + acc[2][0][r] : acc[1][0][r] - acc[2][0][r] - acc[3][0][r];
        const float z1 = half == 0 ? acc[0][1][r] + acc[1][1][r] + acc[2][1][r] : acc[1][1][r] - acc[2][1][r] - acc[3][1][r];
        const float v0 = z0 + b0 + e0[r], v1 = z1 + b1 + e1[r];
        float g = (a.gate_mode == 0) ? ss_sigmoid_fast(v0) * ss_tanh_fast(v1) : ss_tanh_fast(v0) * ss_sigmoid_fast(v1);
        if (t >= row_lim) g = 0.f;
        Cb[(int64_t)t * a.ldc + oc] = g;
      }
    }
  } else {
    // wave wn=0 holds the first gate operand of channels [oc0, oc0+32), wave wn=1 the second; wn=0 finishes frame t,
    // wn=1 frame t+d, and the partners' activations travel through LDS.
    __syncthreads();  // every wave is done with the operand tiles: reuse LDS as the exchange buffer
    float* X0 = smem;                // [2 wm][32 pairs][33]: second-operand activation of frame t   (written by wn=1)
    float* X1 = smem + 2 * 32 * 33;  // [2 wm][32 pairs][33]: first-operand activation of frame t+d  (written by wn=0)
    const int pc = n0 + wn * 32 + l31;  // packed column
    const int oc = (n0 >> 1) + l31;     // output channel
    const bool col_ok = oc < a.N;
    const float bs = (a.bias && col_ok) ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc] : 0.f;
    const bool use_sig = (wn == 0) == (a.gate_mode == 0);
    float mine[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pl = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int t = frame_of(pl);
      const float z0 = acc[0][0][r] + acc[1][0][r] + acc[2][0][r];
      const float z1 = acc[1][0][r] - acc[2][0][r] - acc[3][0][r];
      const float e0 = bs + pe[r], e1 = bs + pe[16 + r];  // prefetched under the last chunk
      (void)t;
      const float u0 = use_sig ? ss_sigmoid_fast(z0 + e0) : ss_tanh_fast(z0 + e0);
      const float u1 = use_sig ? ss_sigmoid_fast(z1 + e1) : ss_tanh_fast(z1 + e1);
      if (wn == 0) {
        mine[r] = u0;
        X1[(wm * 32 + pl) * 33 + l31] = u1;
      } else {
        mine[r] = u1;
        X0[(wm * 32 + pl) * 33 + l31] = u0;
      }
    }
    __syncthreads();
    if (!col_ok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pl = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int t = frame_of(pl) + (wn == 0 ? 0 : d);
      if (t >= a.T) continue;
      const float other = (wn == 0) ? X0[(wm * 32 + pl) * 33 + l31] : X1[(wm * 32 + pl) * 33 + l31];
      float g = mine[r] * other;
      if (t >= row_lim) g = 0.f;
      Cb[(int64_t)t * a.ldc + oc] = g;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Version 2 of the same kernel: identical tiles, staging layout and arithmetic; the instruction stream is on a VALU diet.
// Measured (tools/ubench/mfma_valu.hip): a VALU instruction issued beside v_mfma_f32_32x32x2_f32 is NOT free on gfx950 - beyond
// ~2 per MFMA each one costs ~2.8 cycles of the SIMD the matrix op runs on (SQ_VALU_MFMA_COEXEC_CYCLES reads 0 for this
// kernel). Version 1 issues 62 VALU per 16-MFMA chunk plus ~1450 in prologue/epilogue (6.7 per MFMA overall). Here:
//   * every fetch address is a per-thread byte offset held in a register + a wave-uniform SGPR offset (buffer soffset): the
//     K-chunk and weight-chunk walks cost no VALU at all; the row offsets / padding masks of a Winograd component are
//     computed once per component (between the component loops), not once per chunk;
//   * the input transform is 2 ops per element:  (va +- vb) + mc*dstep  with mc = [row a valid] +- [row b valid];
//   * chunks are processed in pairs so that the LDS double-buffer index is a compile-time constant (immediate offsets);
//   * the epilogue computes one transcendental pair per activation (tanh(x) = 2*sigmoid(2x) - 1, selected per wave by a
//     multiplier instead of evaluating both branches), takes the frame index of every accumulator row once, and fetches the
//     conditioner addend through a buffer resource with 32-bit offsets.
// Requires Cin % 32 == 0 and an even number of K chunks (true for both denoisers: C = 256 / 192); the launcher falls back to
// version 1 otherwise.
template <int TN>
__global__ __launch_bounds__(256) void wino_gate_kernel_v2(const ss_conv_gemm_args a, int p_tiles_per_item, int p_tiles,
                                                           int n_tiles, int log2d, int prio_mode, unsigned long long* clock_probe) {
  constexpr int BN = 64 * TN;
  ss_apply_wave_prio(prio_mode);
  // ss_set_clock_probe: workgroup 0 reports how many shader cycles and 100 MHz ticks its first wave lived (-> sustained clock)
  const bool probing = clock_probe != nullptr && blockIdx.x == 0;
  unsigned long long probe_c0 = 0, probe_r0 = 0;
  if (probing) {
    probe_c0 = __builtin_readcyclecounter();
    probe_r0 = __builtin_amdgcn_s_memrealtime();
  }
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                  // [2][BP][LD]
  float* Bs = smem + 2 * BP * LD;    // [2][BN][LD]

  const int id = blockIdx.x;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int pt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (pt >= p_tiles) return;
  const int b = pt / p_tiles_per_item;
  const int p0 = (pt % p_tiles_per_item) * BP;
  const int n0 = nt * BN;
  const int d = 1 << log2d;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int len = a.lens ? a.lens[b] : a.T;
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const float* Wg = a.W + (int64_t)grp_w * a.w_group_stride;
  const float* abiasg = a.a_bias ? a.a_bias + (int64_t)grp_w * a.a_bias_group_stride : nullptr;
  const int kchunks = a.Kp / BK;
  const int ldw = 4 * a.Kp;

  auto uniform_ptr = [](const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Wg), 0, __builtin_amdgcn_readfirstlane(a.Np * ldw * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_bias = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(abiasg ? abiasg : Wg), 0, __builtin_amdgcn_readfirstlane(abiasg ? a.Cin * 4 : 0), 0x00020000);

  const int st_c4 = tid & 7;
  const int st_row = tid >> 3;  // 0..31; two passes cover the 64 pair rows / 64 weight rows
  // frame of pair p: t = p + (p & ~(d-1))   (= (p >> log2d) * 2d + (p & (d-1)))
  int t_of[2], rowoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = p0 + st_row + i * 32;
    t_of[i] = p + (p & ~(d - 1));
    rowoff[i] = (t_of[i] * a.lda + st_c4 * 4) * 4;
  }
  int w_voff[2 * TN];
#pragma unroll
  for (int i = 0; i < 2 * TN; ++i) w_voff[i] = ((n0 + st_row + i * 32) * ldw + st_c4 * 4) * 4;
  const int bias_voff = st_c4 * 16;
  const int lda4 = a.lda * 4;

  // Winograd component j: staged row = y[t + oa] + sgn * y[t + ob],  y = x + dstep on real frames, 0 on padding
  //   j0: y[t-d] - y[t+d]   j1: y[t] + y[t+d]   j2: y[t+d] - y[t]   j3: y[t] - y[t+2d]
  struct Comp { int va[2], vb[2]; float mc[2]; float sgn; };
  auto comp_of = [&](int j) {
    Comp c;
    const int oa = (j == 0) ? -d : (j == 2) ? d : 0;
    const int ob = (j == 3) ? 2 * d : (j == 2) ? 0 : d;
    c.sgn = (j == 1) ? 1.0f : -1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      c.va[i] = rowoff[i] + oa * lda4;
      c.vb[i] = rowoff[i] + ob * lda4;
      const bool oka = (unsigned)(t_of[i] + oa) < (unsigned)len, okb = (unsigned)(t_of[i] + ob) < (unsigned)len;
      c.mc[i] = (oka ? 1.0f : 0.0f) + (okb ? c.sgn : 0.0f);  // dstep enters once per valid fetched row, with that row's sign
    }
    return c;
  };

  u32x4 ra[2][2], rb[2 * TN];
  float4 rpb;
  auto load_a = [&](const Comp& c, int ci0b) {  // ci0b = byte offset of the K chunk inside a row (wave-uniform -> SGPR soffset)
    rpb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_bias, bias_voff, ci0b, 0));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, c.va[i], ci0b, 0);
      ra[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, c.vb[i], ci0b, 0);
    }
  };
  auto load_b = [&](int cb) {  // cb = byte offset of the weight chunk inside a packed row (wave-uniform)
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff[i], cb, 0);
  };
  const int a_wr = lds_slot(st_row, st_c4), a_wr1 = lds_slot(st_row + 32, st_c4);  // (row >> 1) & 7 differs by 0 mod 8 for +32
  auto store_a = [&](float* Ad, const Comp& c) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 va = __builtin_bit_cast(float4, ra[i][0]);
      const float4 vb = __builtin_bit_cast(float4, ra[i][1]);
      float4 v;
      v.x = fmaf(c.mc[i], rpb.x, fmaf(c.sgn, vb.x, va.x));
      v.y = fmaf(c.mc[i], rpb.y, fmaf(c.sgn, vb.y, va.y));
      v.z = fmaf(c.mc[i], rpb.z, fmaf(c.sgn, vb.z, va.z));
      v.w = fmaf(c.mc[i], rpb.w, fmaf(c.sgn, vb.w, va.w));
      *reinterpret_cast<float4*>(Ad + (i == 0 ? a_wr : a_wr1)) = v;
    }
  };
  auto store_b = [&](float* Bd) {
#pragma unroll
    for (int i = 0; i < 2 * TN; ++i)
      *reinterpret_cast<float4*>(Bd + lds_slot(st_row + i * 32, st_c4)) = __builtin_bit_cast(float4, rb[i]);
  };

  f32x16 acc[4][TN];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

#ifdef SS_TRACE
  unsigned tr_sum[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tr_t0 = __builtin_readcyclecounter();
  const unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz counter
#endif
  Comp cc = comp_of(0);
  load_a(cc, 0);
  load_b(0);
  store_a(As, cc);
  store_b(Bs);
  load_a(cc, BK * 4);   // chunk 1 stays in flight across the barrier: fetches run TWO chunks ahead of the MFMAs
  load_b(BK * 4);
  __syncthreads();

  const int swz = (l31 >> 1) & 7;
  const int a_row = (wm * 32 + l31) * LD;
  const int b_row = (wn * 32 * TN + l31) * LD;
  struct BF { float4 v[TN]; };
  auto read_frags = [&](const float* Ac, const float* Bc, int q, float4& af, BF& bf) {
    const int so = ((2 * q + lh) ^ swz) << 2;
    af = *reinterpret_cast<const float4*>(Ac + a_row + so);
#pragma unroll
    for (int n = 0; n < TN; ++n) bf.v[n] = *reinterpret_cast<const float4*>(Bc + b_row + n * 32 * LD + so);
  };
  auto mfma4 = [&](f32x16 (&c)[TN], const float4& af, const BF& bf) {
#if SS_ABL == 3
    asm volatile("" ::"v"(af.x), "v"(af.y), "v"(af.z), "v"(af.w), "v"(bf.v[0].x), "v"(bf.v[0].y), "v"(bf.v[0].z), "v"(bf.v[0].w));
    return;
#endif
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.v[n].x, c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.v[n].y, c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.v[n].z, c[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < TN; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.v[n].w, c[n], 0, 0, 0);
  };

  // One K chunk g: MFMAs from LDS buffer CUR into `cacc`. In their shadow, after the second MFMA group: the registers (chunk g+1,
  // fetched during chunk g-1; component parameters `st`) are transformed and written to buffer CUR^1, and the fetch of chunk g+2
  // (component parameters `ld`, K byte offset `ci0b`, weight byte offset `cb`) is issued right behind them into the same
  // registers. A fetch therefore has two groups of this chunk + two of the next (>= 1024 MFMA cycles of this wave, ~3x that of
  // wall time with 3 waves per SIMD) before its data is needed, and stays in flight across the barrier. (Timing ablations,
  // tools/ablate.sh: with the fetch issued at the top of the chunk that consumes it the loop waited ~10 us per launch for L2.)
  auto chunk = [&](auto cur_tag, auto store_tag, auto load_tag, f32x16 (&cacc)[TN], const Comp& st, const Comp& ld, int ci0b, int cb) {
    constexpr int CUR = decltype(cur_tag)::value;
    const float* Ac = As + CUR * BP * LD;
    const float* Bc = Bs + CUR * BN * LD;
    float4 af0, af1;
    BF bf0, bf1;
    [[maybe_unused]] unsigned ta, tb, tc, td, te, tf;
    SS_CLK(ta);
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    SS_CLK(tb);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(cacc, af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(cacc, af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    SS_CLK(tc);
    SS_CLK_VM(td);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(store_tag)::value) {
#if SS_ABL != 2
      store_a(As + (CUR ^ 1) * BP * LD, st);
      store_b(Bs + (CUR ^ 1) * BN * LD);
#else
      const u32x4 s0 = ra[0][0] + ra[0][1] + ra[1][0] + ra[1][1] + rb[0] + rb[1];
      asm volatile("" ::"v"(s0[0]), "v"(s0[1]), "v"(s0[2]), "v"(s0[3]), "v"(rpb.x));
#endif
    }
    __builtin_amdgcn_sched_barrier(0);  // stores first, then the fetch into the SAME registers (a second register set would cost a wave per SIMD)
    if constexpr (decltype(load_tag)::value) {
#if SS_ABL != 1
      load_a(ld, ci0b);
      load_b(cb);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma4(cacc, af0, bf0);
    mfma4(cacc, af1, bf1);
    SS_CLK(te);
#if SS_ABL != 4
    __syncthreads();
#endif
    SS_CLK(tf);
#ifdef SS_TRACE
    tr_sum[0] += tb - ta; tr_sum[1] += tc - tb; tr_sum[2] += td - tc; tr_sum[3] += te - td; tr_sum[4] += tf - te; tr_sum[5] += 1;
#endif
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;
  // Component J = kchunks chunks (even), processed in (buffer 0, buffer 1) pairs. Chunk (J,k) stages chunk (J,k+1) and fetches chunk
  // (J,k+2); the last two chunks of a component reach into component J+1, whose parameters are computed here, between the loops.
  const int kb = a.Kp * 4;   // bytes of one component in a packed weight row
  const int cs = BK * 4;     // bytes of one K chunk
  auto component = [&](f32x16 (&cacc)[TN], int j) {
    const Comp nxt = comp_of(j + 1);
    const int wbase = j * kb;
    int k = 0;
    for (; k + 2 < kchunks; k += 2) {
      chunk(C0{}, Yes{}, Yes{}, cacc, cc, cc, (k + 2) * cs, wbase + (k + 2) * cs);
      chunk(C1{}, Yes{}, Yes{}, cacc, cc, cc, (k + 3) * cs, wbase + (k + 3) * cs);   // (k+3 <= kchunks-1 inside this loop)
    }
    // k = kchunks - 2: stage (J, kchunks-1), fetch (J+1, 0); then k = kchunks - 1: stage (J+1, 0), fetch (J+1, 1)
    chunk(C0{}, Yes{}, Yes{}, cacc, cc, nxt, 0, wbase + kb);
    chunk(C1{}, Yes{}, Yes{}, cacc, nxt, nxt, cs, wbase + kb + cs);
    cc = nxt;
  };
  component(acc[0], 0);
  component(acc[1], 1);
  component(acc[2], 2);
  {  // component 3: nothing left to fetch for its last two chunks, nothing to stage for the very last one
    const int wbase = 3 * kb;
    int k = 0;
    for (; k + 2 < kchunks; k += 2) {
      chunk(C0{}, Yes{}, Yes{}, acc[3], cc, cc, (k + 2) * cs, wbase + (k + 2) * cs);
      chunk(C1{}, Yes{}, Yes{}, acc[3], cc, cc, (k + 3) * cs, wbase + (k + 3) * cs);
    }
    chunk(C0{}, Yes{}, No{}, acc[3], cc, cc, 0, 0);
  }

#ifdef SS_TRACE
  const unsigned long long tr_t1 = __builtin_readcyclecounter();
#endif
  // ---- frame index / byte offsets of the accumulator rows this lane owns: row r -> pair (r&3) + 8*(r>>2) + 4*lh of the wave tile
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Eb ? Eb : Wg), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
  const int pbase = p0 + wm * 32 + 4 * lh;
  int tfr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int p = pbase + (r & 3) + 8 * (r >> 2);
    tfr[r] = p + (p & ~(d - 1));
  }
  const int lde4 = a.lde * 4;
  // Conditioner addend (hoisted E slab, 40 KB row stride) fetched BEFORE the last chunk's MFMAs: its miss latency hides under them.
  float pe[32];
  if constexpr (TN == 2) {
    // (the 64 x 128 tile holds 128 accumulator registers: the addend is fetched in the epilogue, 32 values at a time, to stay
    //  under 256 registers = 2 waves per SIMD; with many rounds per launch other workgroups cover the latency)
  } else {
    const int pc = n0 + wn * 32 + l31;
    const int colb = pc * 4 + ((((n0 >> 1) + l31) < a.N) ? 0 : (int)0x80000000);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = tfr[r] * lde4 + colb;
      pe[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off, 0, 0));
      pe[16 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off + d * lde4, 0, 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  {  // last chunk (buffer 1: 4*kchunks chunks in all, kchunks even)
    const float* Ac = As + BP * LD;
    const float* Bc = Bs + BN * LD;
    float4 af0, af1;
    BF bf0, bf1;
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    mfma4(acc[3], af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    mfma4(acc[3], af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    mfma4(acc[3], af0, bf0);
    mfma4(acc[3], af1, bf1);
  }

  // ---- epilogue: output transform z[t] = m0+m1+m2, z[t+d] = m1-m2-m3, conditioner addend, gate ----
  float* Cb = a.C + (int64_t)b * a.c_batch_stride;
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  // sigmoid(x) = rcp(1 + exp(-x)); tanh(x) = 2*sigmoid(2x) - 1: one exp + one rcp either way, selected by (mul, scale, shift)
  auto act = [](float x, float mul, float sc, float sh) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __expf(x * mul)), sc, sh); };
  if constexpr (TN == 2) {
    const int pc0 = n0 + wn * 64 + l31;
    const int oc = (pc0 >> 6) * 32 + l31;
    if (oc >= a.N) return;
    const float b0 = a.bias ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc0] : 0.f;
    const float b1 = a.bias ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc0 + 32] : 0.f;
    const bool sig_first = a.gate_mode == 0;
    const float m0 = sig_first ? -1.0f : -2.0f, s0 = sig_first ? 1.0f : 2.0f, h0 = sig_first ? 0.0f : -1.0f;
    const float m1 = sig_first ? -2.0f : -1.0f, s1 = sig_first ? 2.0f : 1.0f, h1 = sig_first ? -1.0f : 0.0f;
    const int colb = pc0 * 4;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // half 0: frame t, half 1: frame t+d
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = (tfr[r] + half * d) * lde4 + colb;
        pe[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off, 0, 0));
        pe[16 + r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off, 128, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t = tfr[r] + half * d;
        if (t >= a.T) continue;
        const float z0 = half == 0 ? acc[0][0][r] + acc[1][0][r] + acc[2][0][r] : acc[1][0][r] - acc[2][0][r] - acc[3][0][r];
        const float z1 = half == 0 ? acc[0][1][r] + acc[1][1][r] + acc[2][1][r] : acc[1][1][r] - acc[2][1][r] - acc[3][1][r];
        const float v0 = z0 + b0 + pe[r], v1 = z1 + b1 + pe[16 + r];
        float g = act(v0, m0, s0, h0) * act(v1, m1, s1, h1);
        if (t >= row_lim) g = 0.f;
        Cb[(int64_t)t * a.ldc + oc] = g;
      }
    }
  } else {
    // wave wn=0 holds the first gate operand of channels [oc0, oc0+32), wave wn=1 the second; wn=0 finishes frame t,
    // wn=1 frame t+d, and the partners' activations travel through LDS.
    __syncthreads();  // every wave is done with the operand tiles: reuse LDS as the exchange buffer
    float* X0 = smem;                // [2 wm][32 pairs][33]: second-operand activation of frame t   (written by wn=1)
    float* X1 = smem + 2 * 32 * 33;  // [2 wm][32 pairs][33]: first-operand activation of frame t+d  (written by wn=0)
    const int pc = n0 + wn * 32 + l31;  // packed column
    const int oc = (n0 >> 1) + l31;     // output channel
    const bool col_ok = oc < a.N;
    const float bs = (a.bias && col_ok) ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc] : 0.f;
    const bool use_sig = (wn == 0) == (a.gate_mode == 0);  // wave-uniform
    const float am = use_sig ? -1.0f : -2.0f, as = use_sig ? 1.0f : 2.0f, ah = use_sig ? 0.0f : -1.0f;
    float* xw = (wn == 0 ? X1 : X0) + (wm * 32 + 4 * lh) * 33 + l31;  // what this wave publishes
    float mine[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pl = (r & 3) + 8 * (r >> 2);
      const float z0 = acc[0][0][r] + acc[1][0][r] + acc[2][0][r];
      const float z1 = acc[1][0][r] - acc[2][0][r] - acc[3][0][r];
      const float u0 = act(z0 + (bs + pe[r]), am, as, ah);
      const float u1 = act(z1 + (bs + pe[16 + r]), am, as, ah);
      mine[r] = wn == 0 ? u0 : u1;
      xw[pl * 33] = wn == 0 ? u1 : u0;
    }
    __syncthreads();
    if (!col_ok) return;
    const float* xr = (wn == 0 ? X0 : X1) + (wm * 32 + 4 * lh) * 33 + l31;
    const int tsh = wn == 0 ? 0 : d;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pl = (r & 3) + 8 * (r >> 2);
      const int t = tfr[r] + tsh;
      if (t >= a.T) continue;
      float g = mine[r] * xr[pl * 33];
      if (t >= row_lim) g = 0.f;
      Cb[(int64_t)t * a.ldc + oc] = g;
    }
  }
  if (probing && threadIdx.x == 0) {
    atomicAdd(clock_probe, (unsigned long long)__builtin_readcyclecounter() - probe_c0);
    atomicAdd(clock_probe + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - probe_r0);
  }
#ifdef SS_TRACE
  if (g_wino_trace && lane == 0) {
    unsigned long long* o = g_wino_trace + ((size_t)blockIdx.x * 4 + wave) * 16;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = tr_sum[q];
    o[6] = tr_t0;
    o[7] = tr_t1;
    o[8] = __builtin_readcyclecounter();
    o[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
    o[10] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    o[11] = tr_r0;
    o[12] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// g0 = w0, g1 = (w0+w1+w2)/2, g2 = (w0-w1+w2)/2, g3 = w2   (src [rows][3] -> dst [rows][4], rows = Cout*Cin)
__global__ void wino_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    const float w0 = src[i * 3 + 0], w1 = src[i * 3 + 1], w2 = src[i * 3 + 2];
    dst[i * 4 + 0] = w0;
    dst[i * 4 + 1] = (w0 + w1 + w2) * 0.5f;
    dst[i * 4 + 2] = (w0 - w1 + w2) * 0.5f;
    dst[i * 4 + 3] = w2;
  }
}

}  // namespace

extern "C" int ss_wino_weight_transform(const float* src, float* dst, int Cout, int Cin, void* stream) {
  SS_CHECK_ARG(src && dst && Cout > 0 && Cin > 0, "ss_wino_weight_transform: bad args");
  const int64_t rows = (int64_t)Cout * Cin;
  const int grid = (int)((rows + 255) / 256 < 4096 ? (rows + 255) / 256 : 4096);
  hipLaunchKernelGGL(wino_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, rows);
  SS_CHECK_LAUNCH("ss_wino_weight_transform");
  return SS_OK;
}

extern "C" int ss_wino_gate(const ss_conv_gemm_args* args, int dilation, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_wino_gate: null args");
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "ss_wino_gate: null A/W/C");
  SS_CHECK_ARG(dilation >= 1 && (dilation & (dilation - 1)) == 0 && dilation <= 64, "ss_wino_gate: dilation %d must be a power of two", dilation);
  SS_CHECK_ARG((a.Kp % BK) == 0 && a.Kp >= a.Cin && (a.Cin & 3) == 0 && (a.lda & 3) == 0, "ss_wino_gate: bad K dims");
  SS_CHECK_ARG((a.Np % 64) == 0 && 2 * a.N <= a.Np, "ss_wino_gate: Np=%d must be a multiple of 64 and >= 2*N", a.Np);
  SS_CHECK_ARG((int64_t)a.T * a.lda * 4 < (1ll << 31) && (!a.E || (int64_t)a.T * a.lde * 4 < (1ll << 31)),
               "ss_wino_gate: item too large for 32-bit offsets");
  int log2d = 0;
  while ((1 << log2d) < dilation) ++log2d;
  const int pairs_per_item = ss_cdiv(a.T, 2 * dilation) * dilation;
  const int p_tiles_per_item = ss_cdiv(pairs_per_item, BP);
  const int p_tiles = p_tiles_per_item * a.B;
  // tile choice: 64x128 (TN=2) has twice the MFMAs per barrier but 256 registers (2 blocks/CU = 512 slots); 64x64 (TN=1)
  // runs 3 blocks/CU (768 slots). Measured with tools/kbench.py (graph-timed, after the addend prefetch): mel C2
  // (752 / 376 blocks) TN=1 84.0 us; f0 pair (1152 / 576 blocks) TN=1 104.7 vs TN=2 112.0 us -> TN=1 while its grid is
  // at most two rounds; beyond that a makespan model picks (TN=2 wins once there are many rounds).
  int tn = a.tile == SS_TILE_64x128 ? 2 : a.tile == SS_TILE_64x64 ? 1 : 0;
  const int env_tn = g_ss_tuning.wino_tn;  // experiments: force the tile of every auto launch
  if (tn == 0 && (env_tn == 1 || env_tn == 2)) tn = env_tn;
  if (tn == 0) {
    const long b2 = (long)p_tiles * ss_cdiv(a.Np, 128), b1 = (long)p_tiles * (a.Np / 64);
    const double t2 = (double)ss_cdiv(b2, 256) * 2.0 / 0.92, t1 = (double)ss_cdiv(b1, 256) * 1.0 / 0.75;
    tn = ((a.Np % 128) == 0 && b1 > 2 * 768 && t2 <= t1) ? 2 : 1;
  }
  // version 2 (VALU diet) needs whole 32-channel K chunks and an even chunk count; the "wino_v1" knob forces the first version (A/B)
  const bool force_v1 = g_ss_tuning.wino_v1 != 0;
  const bool v2 = !force_v1 && (a.Cin % BK) == 0 && a.Kp == a.Cin && ((a.Kp / BK) % 2) == 0;
  if (tn == 2) {
    SS_CHECK_ARG((a.Np % 128) == 0, "ss_wino_gate: TN=2 needs Np multiple of 128");
    const int n_tiles = a.Np / 128;
    const int grid = ss_cdiv(p_tiles, 8) * 8 * n_tiles;
    const size_t lds = (size_t)2 * (BP + 128) * LD * sizeof(float);
    if (v2) hipLaunchKernelGGL(wino_gate_kernel_v2<2>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, p_tiles_per_item, p_tiles, n_tiles, log2d, g_ss_tuning.wave_prio, g_ss_tuning.clock_probe);
    else hipLaunchKernelGGL(wino_gate_kernel<2>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, p_tiles_per_item, p_tiles, n_tiles, log2d, g_ss_tuning.wave_prio);
  } else {
    const int n_tiles = a.Np / 64;
    const int grid = ss_cdiv(p_tiles, 8) * 8 * n_tiles;
#ifdef SS_EXPERIMENT_KNOBS   // occupancy experiments of the ablation builds only (tools/ablate.sh)
    static const size_t lds_pad = getenv("SS_WINO_LDS_PAD") ? (size_t)atoi(getenv("SS_WINO_LDS_PAD")) : 0;
#else
    const size_t lds_pad = 0;
#endif
    const size_t lds = (size_t)2 * (BP + 64) * LD * sizeof(float) + lds_pad;
    if (lds_pad) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_gate_kernel_v2<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (v2) hipLaunchKernelGGL(wino_gate_kernel_v2<1>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, p_tiles_per_item, p_tiles, n_tiles, log2d, g_ss_tuning.wave_prio, g_ss_tuning.clock_probe);
    else hipLaunchKernelGGL(wino_gate_kernel<1>, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, p_tiles_per_item, p_tiles, n_tiles, log2d, g_ss_tuning.wave_prio);
  }
  SS_CHECK_LAUNCH("ss_wino_gate");
  return SS_OK;
}
