// Winograd F(4,3) version of the denoisers' hot layer: 3-tap dilated conv + conditioner addend + gate
// (modules/diff/net.py:66-73), exact-fp32 MFMA, 2x fewer matrix ops than the direct form (F(2,3) in wino_gate.hip: 1.5x).
//
//   y[t] = x[t] + dstep   (0 outside [0,len))          z[t] = w0.y[t-d] + w1.y[t] + w2.y[t+d] + E[t]
//   g[t] = sigmoid(z[t][:C]) * tanh(z[t][C:])
//
// The four frames (t, t+d, t+2d, t+3d) read the six rows r_i = y[t+(i-1)d], i = 0..5. With the standard F(4,3) matrices
// (Lavin & Gray, interpolation points 0, +-1, +-2, inf) they come from 6 products instead of 12:
//   c0 = 4 r0 - 5 r2 + r4              g0 =  w0 / 4
//   c1 = -4 r1 - 4 r2 + r3 + r4        g1 = -(w0 + w1 + w2) / 6
//   c2 =  4 r1 - 4 r2 - r3 + r4        g2 = -(w0 - w1 + w2) / 6
//   c3 = -2 r1 - r2 + 2 r3 + r4        g3 =  w0/24 + w1/12 + w2/6
//   c4 =  2 r1 - r2 - 2 r3 + r4        g4 =  w0/24 - w1/12 + w2/6
//   c5 =  4 r1 - 5 r3 + r5             g5 =  w2
//   m_j = c_j . g_j
//   z[t] = m0+m1+m2+m3+m4   z[t+d] = (m1-m2) + 2(m3-m4)   z[t+2d] = (m1+m2) + 4(m3+m4)   z[t+3d] = (m1-m2) + 8(m3-m4) + m5
// Quads are formed inside groups of 4d frames (t = g*4d + s, s < d): any power-of-two dilation.
// GEMM view: rows = quads, 6 components, each a [quads x C] x [C x 2C] product with its own accumulator. Tile = 64 quads
// (256 frames) x 64 packed columns, 4 waves of 32x32 x 6 components = 96 accumulator registers -> 2 workgroups per CU.
//
// Loop order is K-outer / component-inner: the six raw rows of a K chunk are fetched ONCE into registers (12 x 16 B per thread for
// its two quad rows) and the six component tiles of that chunk are produced from them, one per MFMA chunk, in the MFMA shadow.
// Chunk g = 6k + j uses LDS buffer j & 1 (compile time) and accumulator j. Skeleton (LDS swizzle, buffer-resource fetch with SGPR
// chunk offsets, stores-then-fetch scheduling) as wino_gate_kernel_v2.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// SS_TRACE (debug builds only, tools/wave_trace.py --kernel wino43): every wave sums, over its chunks, the shader-clock time of the
// phases of a chunk and writes the sums + its entry/exit stamps at exit (same record layout as wino_gate_kernel_v2).
#ifdef SS_TRACE
__device__ unsigned long long* g_wino43_trace = nullptr;
extern "C" int ss_debug_set_wino43_trace(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wino43_trace), &p, sizeof(p));
}
#endif

namespace {

constexpr int BK = 32;
constexpr int LD = BK;
constexpr int BQ = 64;  // quads per tile (= 256 output frames)
constexpr int BN = 64;
constexpr int NC = 6;   // components

__device__ __forceinline__ int lds_slot(int row, int slot) { return row * LD + ((slot ^ ((row >> 1) & 7)) << 2); }

// input-transform coefficients of raw row q for component J (0 = row not used)
template <int J, int Q>
struct Coef {
  static constexpr float v = (J == 0) ? (Q == 0 ? 4.f : Q == 2 ? -5.f : Q == 4 ? 1.f : 0.f)
                           : (J == 1) ? (Q == 1 ? -4.f : Q == 2 ? -4.f : Q == 3 ? 1.f : Q == 4 ? 1.f : 0.f)
                           : (J == 2) ? (Q == 1 ? 4.f : Q == 2 ? -4.f : Q == 3 ? -1.f : Q == 4 ? 1.f : 0.f)
                           : (J == 3) ? (Q == 1 ? -2.f : Q == 2 ? -1.f : Q == 3 ? 2.f : Q == 4 ? 1.f : 0.f)
                           : (J == 4) ? (Q == 1 ? 2.f : Q == 2 ? -1.f : Q == 3 ? -2.f : Q == 4 ? 1.f : 0.f)
                                      : (Q == 1 ? 4.f : Q == 3 ? -5.f : Q == 5 ? 1.f : 0.f);
};

__global__ __launch_bounds__(256, 2) void wino43_gate_kernel(const ss_conv_gemm_args a, int q_tiles_per_item, int q_tiles, int n_tiles,
                                                             int log2d, unsigned long long* clock_probe) {
  const bool probing = clock_probe != nullptr && blockIdx.x == 0;
  unsigned long long probe_c0 = 0, probe_r0 = 0;
  if (probing) {
    probe_c0 = __builtin_readcyclecounter();
    probe_r0 = __builtin_amdgcn_s_memrealtime();
  }
#ifdef SS_TRACE
  unsigned tr_sum[6] = {0, 0, 0, 0, 0, 0};
  const unsigned long long tr_t0 = __builtin_readcyclecounter();
  const unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime();
#endif
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
  const int b = qt / q_tiles_per_item;
  const int q0 = (qt % q_tiles_per_item) * BQ;
  const int n0 = nt * BN;
  const int d = 1 << log2d;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const float* Wg = a.W + (int64_t)grp_w * a.w_group_stride;
  const float* abiasg = a.a_bias ? a.a_bias + (int64_t)grp_w * a.a_bias_group_stride : nullptr;
  const int kchunks = a.Kp / BK;
  const int ldw = NC * a.Kp;

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
  const int st_row = tid >> 3;  // 0..31; two passes cover the 64 quad rows / 64 weight rows
  const int lda4 = a.lda * 4;
  // frame of quad q: t = q + 3 * (q & ~(d-1))   (= (q >> log2d) * 4d + (q & (d-1)))
  // roff[i][r] = byte offset of raw row r (frame t + (r-1)d) of quad row i, or out of range (-> the fetch returns 0) when that frame
  // is outside [0, len); mc[j][i] = sum of component j's coefficients over the VALID rows: what dstep enters the component with.
  int roff[2][6];
  float mc[NC][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = q0 + st_row + i * 32;
    const int t = q + 3 * (q & ~(d - 1));
    float v[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int tr = t + (r - 1) * d;
      const bool ok = (unsigned)tr < (unsigned)len;
      v[r] = ok ? 1.0f : 0.0f;
      roff[i][r] = ok ? (tr * a.lda + st_c4 * 4) * 4 : (int)0x80000000;
    }
    mc[0][i] = 4.f * v[0] - 5.f * v[2] + v[4];
    mc[1][i] = -4.f * v[1] - 4.f * v[2] + v[3] + v[4];
    mc[2][i] = 4.f * v[1] - 4.f * v[2] - v[3] + v[4];
    mc[3][i] = -2.f * v[1] - v[2] + 2.f * v[3] + v[4];
    mc[4][i] = 2.f * v[1] - v[2] - 2.f * v[3] + v[4];
    mc[5][i] = 4.f * v[1] - 5.f * v[3] + v[5];
  }
  int w_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) w_voff[i] = ((n0 + st_row + i * 32) * ldw + st_c4 * 4) * 4;
  const int bias_voff = st_c4 * 16;

  u32x4 rr[2][6], rb[2];
  float4 rpb;
  auto load_rows = [&](int ci0b) {  // ci0b = byte offset of the K chunk inside a row (wave-uniform -> SGPR soffset)
    ci0b = __builtin_amdgcn_readfirstlane(ci0b);
    rpb = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_bias, bias_voff, ci0b, 0));
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
  const int a_wr = lds_slot(st_row, st_c4), a_wr1 = lds_slot(st_row + 32, st_c4);
  auto store_a = [&](float* Ad, auto jtag) {
    constexpr int J = decltype(jtag)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v;
      v.x = mc[J][i] * rpb.x;
      v.y = mc[J][i] * rpb.y;
      v.z = mc[J][i] * rpb.z;
      v.w = mc[J][i] * rpb.w;
      auto add_row = [&](auto qtag) {
        constexpr int Q = decltype(qtag)::value;
        constexpr float c = Coef<J, Q>::v;
        if constexpr (c != 0.f) {
          const float4 r = __builtin_bit_cast(float4, rr[i][Q]);
          v.x = fmaf(c, r.x, v.x);
          v.y = fmaf(c, r.y, v.y);
          v.z = fmaf(c, r.z, v.z);
          v.w = fmaf(c, r.w, v.w);
        }
      };
      add_row(std::integral_constant<int, 0>{});
      add_row(std::integral_constant<int, 1>{});
      add_row(std::integral_constant<int, 2>{});
      add_row(std::integral_constant<int, 3>{});
      add_row(std::integral_constant<int, 4>{});
      add_row(std::integral_constant<int, 5>{});
      *reinterpret_cast<float4*>(Ad + (i == 0 ? a_wr : a_wr1)) = v;
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

  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  using J4 = std::integral_constant<int, 4>;
  using J5 = std::integral_constant<int, 5>;
  const int kb = a.Kp * 4;   // bytes of one component in a packed weight row
  const int cs = BK * 4;     // bytes of one K chunk
  // chunk (k, j): weight bytes start at j*kb + k*cs
  load_rows(0);
  load_b(0);
  store_a(As, J0{});
  store_b(Bs);
  load_b(kb);   // weights of chunk (0,1): in flight across the barrier
  __syncthreads();

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
  // chunk (k, J): MFMAs from buffer J&1 into acc[J]. In their shadow: the component tile of chunk g+1 (component JN of K chunk k, or
  // component 0 of k+1) is built from the raw rows in registers and stored with the weight registers; then the weights of chunk g+2
  // are fetched into the same registers and - when JN was the last user of the raw rows (JN = 5) - the raw rows of K chunk k+1.
  auto chunk = [&](auto jtag, auto jn_tag, auto stage_tag, auto fetch_b_tag, auto fetch_rows_tag, int cb2, int rows_ci0b) {
    constexpr int J = decltype(jtag)::value;
    constexpr int CUR = J & 1;
    const float* Ac = As + CUR * BQ * LD;
    const float* Bc = Bs + CUR * BN * LD;
    float4 af0, af1, bf0, bf1;
    [[maybe_unused]] unsigned ta, tb, tc, td, te, tf;
    SS_CLK(ta);
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    SS_CLK(tb);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(acc[J], af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(acc[J], af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    SS_CLK(tc);
    SS_CLK_VM(td);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(stage_tag)::value) {
      store_a(As + (CUR ^ 1) * BQ * LD, jn_tag);
      store_b(Bs + (CUR ^ 1) * BN * LD);
    }
    __builtin_amdgcn_sched_barrier(0);  // stores first, then the fetches into the SAME registers
    if constexpr (decltype(fetch_b_tag)::value) load_b(cb2);
    if constexpr (decltype(fetch_rows_tag)::value) load_rows(rows_ci0b);
    __builtin_amdgcn_sched_barrier(0);
    mfma4(acc[J], af0, bf0);
    mfma4(acc[J], af1, bf1);
    SS_CLK(te);
    __syncthreads();
    SS_CLK(tf);
#ifdef SS_TRACE
    tr_sum[0] += tb - ta; tr_sum[1] += tc - tb; tr_sum[2] += td - tc; tr_sum[3] += te - td; tr_sum[4] += tf - te; tr_sum[5] += 1;
#endif
  };
  using Yes = std::true_type;
  using No = std::false_type;
  for (int k = 0; k + 1 < kchunks; ++k) {
    const int kc = k * cs;
    chunk(J0{}, J1{}, Yes{}, Yes{}, No{}, 2 * kb + kc, 0);          // stage (k,1); fetch weights (k,2)
    chunk(J1{}, J2{}, Yes{}, Yes{}, No{}, 3 * kb + kc, 0);          // stage (k,2); fetch weights (k,3)
    chunk(J2{}, J3{}, Yes{}, Yes{}, No{}, 4 * kb + kc, 0);          // stage (k,3); fetch weights (k,4)
    chunk(J3{}, J4{}, Yes{}, Yes{}, No{}, 5 * kb + kc, 0);          // stage (k,4); fetch weights (k,5)
    chunk(J4{}, J5{}, Yes{}, Yes{}, Yes{}, kc + cs, kc + cs);       // stage (k,5) = last use of the rows; fetch weights (k+1,0), rows k+1
    chunk(J5{}, J0{}, Yes{}, Yes{}, No{}, kb + kc + cs, 0);         // stage (k+1,0); fetch weights (k+1,1)
  }
  {
    const int kc = (kchunks - 1) * cs;
    chunk(J0{}, J1{}, Yes{}, Yes{}, No{}, 2 * kb + kc, 0);
    chunk(J1{}, J2{}, Yes{}, Yes{}, No{}, 3 * kb + kc, 0);
    chunk(J2{}, J3{}, Yes{}, Yes{}, No{}, 4 * kb + kc, 0);
    chunk(J3{}, J4{}, Yes{}, Yes{}, No{}, 5 * kb + kc, 0);
    chunk(J4{}, J5{}, Yes{}, No{}, No{}, 0, 0);
  }
#ifdef SS_TRACE
  const unsigned long long tr_t1 = __builtin_readcyclecounter();
#endif
  {  // last chunk (component 5, buffer 1)
    const float* Ac = As + BQ * LD;
    const float* Bc = Bs + BN * LD;
    float4 af0, af1, bf0, bf1;
    read_frags(Ac, Bc, 0, af0, bf0);
    read_frags(Ac, Bc, 1, af1, bf1);
    mfma4(acc[5], af0, bf0);
    read_frags(Ac, Bc, 2, af0, bf0);
    mfma4(acc[5], af1, bf1);
    read_frags(Ac, Bc, 3, af1, bf1);
    mfma4(acc[5], af0, bf0);
    mfma4(acc[5], af1, bf1);
  }

  // ---- epilogue: output transform, conditioner addend, gate ----
  // accumulator row r of this lane -> quad (r&3) + 8*(r>>2) + 4*lh of the wave tile
  const float* Eb = a.E ? a.E + (int64_t)b * a.e_batch_stride : nullptr;
  const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(Eb ? Eb : Wg), 0, __builtin_amdgcn_readfirstlane(Eb ? (int)((int64_t)a.T * a.lde * 4) : 0), 0x00020000);
  const int qbase = q0 + wm * 32 + 4 * lh;
  int tfr[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = qbase + (r & 3) + 8 * (r >> 2);
    tfr[r] = q + 3 * (q & ~(d - 1));
  }
  const int lde4 = a.lde * 4;
  float* Cb = a.C + (int64_t)b * a.c_batch_stride;
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  // sigmoid(x) = rcp(1 + exp(-x)); tanh(x) = 2*sigmoid(2x) - 1: one exp + one rcp either way, selected by (mul, scale, shift)
  auto act = [](float x, float mul, float sc, float sh) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __expf(x * mul)), sc, sh); };
  // wave wn=0 holds the first gate operand of channels [oc0, oc0+32), wave wn=1 the second; wn=0 finishes frames t and t+d,
  // wn=1 frames t+2d and t+3d, and the partners' activations travel through LDS.
  __syncthreads();  // every wave is done with the operand tiles: reuse LDS as the exchange buffer
  // X[o][wm][32 quads][33]: o = 0,1: second-operand activations of frames t, t+d (written by wn=1);
  //                         o = 2,3: first-operand activations of frames t+2d, t+3d (written by wn=0)
  const int pc = n0 + wn * 32 + l31;  // packed column
  const int oc = (n0 >> 1) + l31;     // output channel
  const bool col_ok = oc < a.N;
  const int colb = pc * 4 + (col_ok ? 0 : (int)0x80000000);
  const float bs = (a.bias && col_ok) ? a.bias[(int64_t)grp_w * a.bias_group_stride + pc] : 0.f;
  const bool use_sig = (wn == 0) == (a.gate_mode == 0);  // wave-uniform
  const float am = use_sig ? -1.0f : -2.0f, as = use_sig ? 1.0f : 2.0f, ah = use_sig ? 0.0f : -1.0f;
  constexpr int XS = 2 * 32 * 33;  // floats per frame slot
  float* xbase = smem + (wm * 32 + 4 * lh) * 33 + l31;
  float mine[2][16];
#pragma unroll
  for (int half = 0; half < 2; ++half) {  // half 0: frames t, t+d; half 1: frames t+2d, t+3d
    float pe[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = (tfr[r] + 2 * half * d) * lde4 + colb;
      pe[0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off, 0, 0));
      pe[1][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_e, off + d * lde4, 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ql = (r & 3) + 8 * (r >> 2);
      const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
      const float s34 = acc[3][r] + acc[4][r], d34 = acc[3][r] - acc[4][r];
      float za, zb;
      if (half == 0) {
        za = acc[0][r] + s12 + s34;
        zb = fmaf(2.0f, d34, d12);
      } else {
        za = fmaf(4.0f, s34, s12);
        zb = fmaf(8.0f, d34, d12) + acc[5][r];
      }
      const float ua = act(za + (bs + pe[0][r]), am, as, ah);
      const float ub = act(zb + (bs + pe[1][r]), am, as, ah);
      const bool keep = (wn == 0) == (half == 0);  // wave-uniform: this wave finishes the frames of this half
      if (keep) {
        mine[0][r] = ua;
        mine[1][r] = ub;
      } else {
        xbase[(2 * half) * XS + ql * 33] = ua;
        xbase[(2 * half + 1) * XS + ql * 33] = ub;
      }
    }
  }
  __syncthreads();
  if (col_ok) {
    const int o0 = wn == 0 ? 0 : 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = (r & 3) + 8 * (r >> 2);
        const int t = tfr[r] + (o0 + o) * d;
        if (t >= a.T) continue;
        float g = mine[o][r] * xbase[(o0 + o) * XS + ql * 33];
        if (t >= row_lim) g = 0.f;
        Cb[(int64_t)t * a.ldc + oc] = g;
      }
    }
  }
  if (probing && threadIdx.x == 0) {
    atomicAdd(clock_probe, (unsigned long long)__builtin_readcyclecounter() - probe_c0);
    atomicAdd(clock_probe + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - probe_r0);
  }
#ifdef SS_TRACE
  if (g_wino43_trace && lane == 0) {
    unsigned long long* o = g_wino43_trace + ((size_t)blockIdx.x * 4 + wave) * 16;
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

// src [rows][3] -> dst [rows][6] (rows = Cout*Cin): the G matrix of F(4,3)
__global__ void wino43_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t rows) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    const float w0 = src[i * 3 + 0], w1 = src[i * 3 + 1], w2 = src[i * 3 + 2];
    dst[i * 6 + 0] = w0 * 0.25f;
    dst[i * 6 + 1] = -(w0 + w1 + w2) / 6.0f;
    dst[i * 6 + 2] = -(w0 - w1 + w2) / 6.0f;
    dst[i * 6 + 3] = w0 / 24.0f + w1 / 12.0f + w2 / 6.0f;
    dst[i * 6 + 4] = w0 / 24.0f - w1 / 12.0f + w2 / 6.0f;
    dst[i * 6 + 5] = w2;
  }
}

}  // namespace

extern "C" int ss_wino43_weight_transform(const float* src, float* dst, int Cout, int Cin, void* stream) {
  SS_CHECK_ARG(src && dst && Cout > 0 && Cin > 0, "ss_wino43_weight_transform: bad args");
  const int64_t rows = (int64_t)Cout * Cin;
  const int grid = (int)((rows + 255) / 256 < 4096 ? (rows + 255) / 256 : 4096);
  hipLaunchKernelGGL(wino43_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, rows);
  SS_CHECK_LAUNCH("ss_wino43_weight_transform");
  return SS_OK;
}

extern "C" int ss_wino43_gate(const ss_conv_gemm_args* args, int dilation, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_wino43_gate: null args");
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "ss_wino43_gate: null A/W/C");
  SS_CHECK_ARG(dilation >= 1 && (dilation & (dilation - 1)) == 0 && dilation <= 64, "ss_wino43_gate: dilation %d must be a power of two", dilation);
  SS_CHECK_ARG((a.Cin % BK) == 0 && a.Kp == a.Cin && (a.lda & 3) == 0, "ss_wino43_gate: Cin=%d must be a multiple of 32 and Kp == Cin", a.Cin);
  SS_CHECK_ARG((a.Np % 64) == 0 && 2 * a.N <= a.Np, "ss_wino43_gate: Np=%d must be a multiple of 64 and >= 2*N", a.Np);
  SS_CHECK_ARG((int64_t)a.T * a.lda * 4 < (1ll << 31) && (!a.E || (int64_t)a.T * a.lde * 4 < (1ll << 31)) &&
                   (int64_t)a.Np * NC * a.Kp * 4 < (1ll << 31),
               "ss_wino43_gate: item too large for 32-bit offsets");
  int log2d = 0;
  while ((1 << log2d) < dilation) ++log2d;
  const int quads_per_item = ss_cdiv(a.T, 4 * dilation) * dilation;
  const int q_tiles_per_item = ss_cdiv(quads_per_item, BQ);
  const int q_tiles = q_tiles_per_item * a.B;
  const int n_tiles = a.Np / BN;
  const int grid = ss_cdiv(q_tiles, 8) * 8 * n_tiles;
  const size_t lds_ops = (size_t)2 * (BQ + BN) * LD * sizeof(float), lds_xchg = (size_t)4 * 2 * 32 * 33 * sizeof(float);
  const size_t lds = lds_ops > lds_xchg ? lds_ops : lds_xchg;  // operand double buffers, reused as the gate exchange buffer
  hipLaunchKernelGGL(wino43_gate_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, a, q_tiles_per_item, q_tiles, n_tiles, log2d,
                     g_ss_tuning.clock_probe);
  SS_CHECK_LAUNCH("ss_wino43_gate");
  return SS_OK;
}
