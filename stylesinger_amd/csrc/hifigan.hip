// HiFi-GAN-NSF generator (modules/hifigan/hifigan_nsf.py:105-169) + NSF harmonic source
// (modules/parallel_wavegan/models/source.py:311-531), channels-last, on the fp32-MFMA conv kernel.
//
//   * ConvTranspose1d(k=2u, stride u) is run as its polyphase decomposition: two 2-tap convs whose
//     N dimension is (phases x Cout), written straight into the [T*u][Cout] output (no zero-stuffing);
//   * every ResBlock1 conv fuses leaky-relu (A prologue), bias, residual add and the 1/num_kernels
//     accumulation (epilogue);
//   * the NSF phase accumulation reproduces torch's CPU cumsum (double accumulator, fp32 output)
//     with a hierarchical scan: per-frame closed form for the first cumsum (the per-sample phase
//     increment is constant inside a hop), a per-chain scan over frames, and an in-block double scan
//     for the wrapped second cumsum.
#include "common.h"
#include "../../include/stylesinger_hip.h"

namespace {

constexpr int NH = 9;      // fundamental + 8 overtones (hifigan_nsf.py:112)
constexpr int HOP_MAX = 1024;

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline int round_up32(int x) { return (x + 31) / 32 * 32; }

struct HgWs {
  int32_t* lens;  // [(n_ups+1)][B]
  double* base;   // [B][NH][T]
  double* q;      // [B][NH][T]
  float* r0;      // [B][NH]  first-sample increment (rad + rand_ini, fp32)
  float* har;     // [B][L]
  float* pre;     // [B][T][c0]
  float* x;       // big
  float* xs;      // big
  float* ta;      // big
  float* tb;      // big
  int64_t bytes;
};

int hop_of(const ss_hifigan* hg) {
  int hop = 1;
  for (int i = 0; i < hg->n_ups; ++i) hop *= hg->up_rate[i];
  return hop;
}

HgWs hg_layout(const ss_hifigan* hg, int B, int T, void* basep) {
  HgWs w;
  int64_t off = 0;
  char* p = (char*)basep;
  auto take = [&](int64_t bytes) {
    void* r = p + off;
    off = align_up(off + bytes, 256);
    return r;
  };
  const int hop = hop_of(hg);
  int64_t big = 0;
  int R = 1;
  for (int i = 0; i < hg->n_ups; ++i) {
    R *= hg->up_rate[i];
    const int64_t c = hg->c0 >> (i + 1);
    const int64_t sz = (int64_t)B * T * R * c;
    if (sz > big) big = sz;
  }
  w.lens = (int32_t*)take((int64_t)(hg->n_ups + 1) * B * 4);
  w.base = (double*)take((int64_t)B * NH * T * 8);
  w.q = (double*)take((int64_t)B * NH * T * 8);
  w.r0 = (float*)take((int64_t)B * NH * 4);
  w.har = (float*)take((int64_t)B * T * hop * 4);
  w.pre = (float*)take((int64_t)B * T * hg->c0 * 4);
  w.x = (float*)take(big * 4);
  w.xs = (float*)take(big * 4);
  w.ta = (float*)take(big * 4);
  w.tb = (float*)take(big * 4);
  w.bytes = off;
  return w;
}

__global__ void stage_lens_kernel(const int32_t* __restrict__ lens, int32_t* __restrict__ out, int B, int T, int n_stage,
                                  int r0, int r1, int r2, int r3, int r4, int r5) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const int rates[6] = {r0, r1, r2, r3, r4, r5};
  int l = lens ? lens[i] : T;
  if (l > T) l = T;
  out[i] = l;
  for (int s = 0; s < n_stage; ++s) {
    l *= rates[s];
    out[(s + 1) * B + i] = l;
  }
}

// per-sample phase increment of harmonic h in frame f (source.py:355): (f0*(h+1)/sr) % 1
__device__ __forceinline__ float rad_of(float f0, int h, float sr) {
  const float fh = (h == 0) ? f0 : f0 * (float)(h + 1);
  const float r = fh / sr;
  return r - floorf(r);
}

// base[b][h][f] = sum_{f'<f} hop*rad_{f'} + (r0' - rad_0): everything the first cumsum has accumulated
// before frame f, in double like torch's CPU cumsum (acc_type<float> = double).
// One workgroup per (item, harmonic) chain: the per-frame increments are computed by all threads into LDS, ONE lane then
// walks them in order (the same sequential double additions as torch's cumsum: 6 us for 1500 frames instead of the 260 us
// a thread-per-chain loop over global memory took), and all threads write the result back coalesced.
constexpr int SCAN_CHUNK = 2048;
__global__ __launch_bounds__(256) void src_base_kernel(const float* __restrict__ f0, const float* __restrict__ rand_ini, uint64_t seed,
                                                       double* __restrict__ base, float* __restrict__ r0_out, int B, int T, int hop, float sr) {
  __shared__ double inc[SCAN_CHUNK];
  __shared__ double carry;
  const int i = blockIdx.x;  // chain = b * NH + h
  const int b = i / NH, h = i % NH;
  const int tid = threadIdx.x;
  if (tid == 0) {
    float ini = 0.f;
    if (h > 0) {
      if (rand_ini) ini = rand_ini[b * NH + h];
      else {
        const SsPhilox rng(seed);
        uint32_t o[4];
        rng.gen((uint32_t)i, 0u, 0x52494e49u, 0x4e534631u, o);
        ini = (float)(o[0] >> 8) * (1.0f / 16777216.0f);
      }
    }
    const float rad0 = rad_of(f0[(int64_t)b * T], h, sr);
    const float r0p = rad0 + ini;  // fp32 add at sample 0 (source.py:361)
    r0_out[i] = r0p;
    carry = (double)r0p - (double)rad0;
  }
  double* bp = base + (int64_t)i * T;
  for (int f0i = 0; f0i < T; f0i += SCAN_CHUNK) {
    const int n = T - f0i < SCAN_CHUNK ? T - f0i : SCAN_CHUNK;
    __syncthreads();
    for (int f = tid; f < n; f += 256) inc[f] = (double)hop * (double)rad_of(f0[(int64_t)b * T + f0i + f], h, sr);
    __syncthreads();
    if (tid == 0) {
      double acc = carry;
      for (int f = 0; f < n; ++f) {
        const double v = inc[f];
        inc[f] = acc;
        acc += v;
      }
      carry = acc;
    }
    __syncthreads();
    for (int f = tid; f < n; f += 256) bp[f0i + f] = inc[f];
  }
}

__device__ __forceinline__ double block_scan_incl(double v, double* sh, int tid, int nthreads) {
  // inclusive scan across the block (nthreads multiple of 64)
  const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  if (lane == 63) sh[wv] = v;
  __syncthreads();
  double add = 0.0;
  for (int w = 0; w < wv; ++w) add += sh[w];
  __syncthreads();
  return v + add;
}

// addend of the second cumsum for sample k of frame f: rad + cumsum_shift (source.py:370-374)
__device__ __forceinline__ float second_addend(double basef, float rad, float r0p, int f, int k) {
  // c1 at this sample and at the previous one (double accumulate, fp32 output, then % 1 in fp32)
  const bool first = (f == 0 && k == 0);
  double c1d, c1p;
  float inc = rad;
  if (f == 0) {
    // frame 0: sample 0 contributes r0p, the rest rad; base(0) already holds (r0p - rad)
    c1d = basef + (double)(k + 1) * (double)rad;
    c1p = basef + (double)k * (double)rad;
    if (k == 0) inc = r0p;
  } else {
    c1d = basef + (double)(k + 1) * (double)rad;
    c1p = basef + (double)k * (double)rad;  // k == 0 -> base == cumsum at the last sample of the previous frame
  }
  const float c1 = (float)c1d, cp = (float)c1p;
  const float t1 = c1 - floorf(c1), t0 = cp - floorf(cp);
  float shift = 0.f;
  if (!first && (t1 - t0) < 0.f) shift = -1.0f;
  return inc + shift;
}

// S2[b][h][f] = sum over the frame of the second-cumsum addends (double)
__global__ void src_frame_sum_kernel(const float* __restrict__ f0, const double* __restrict__ base,
                                     const float* __restrict__ r0, double* __restrict__ s2, int B, int T, int hop,
                                     float sr) {
  __shared__ double sh[HOP_MAX / 64];
  const int bf = blockIdx.x;
  const int b = bf / T, f = bf % T;
  const int k = threadIdx.x;
  const float f0v = f0[(int64_t)b * T + f];
  for (int h = 0; h < NH; ++h) {
    const float rad = rad_of(f0v, h, sr);
    const double basef = base[((int64_t)b * NH + h) * T + f];
    const float a = second_addend(basef, rad, r0[b * NH + h], f, k);
    const double tot = block_scan_incl((double)a, sh, k, hop);
    if (k == hop - 1) s2[((int64_t)b * NH + h) * T + f] = tot;
  }
}

// exclusive scan over the frames of one chain, in place (same structure as src_base_kernel: ordered double additions by one lane)
__global__ __launch_bounds__(256) void src_scan_kernel(double* __restrict__ s2, int chains, int T) {
  __shared__ double v[SCAN_CHUNK];
  __shared__ double carry;
  const int tid = threadIdx.x;
  double* p = s2 + (int64_t)blockIdx.x * T;
  if (tid == 0) carry = 0.0;
  for (int f0i = 0; f0i < T; f0i += SCAN_CHUNK) {
    const int n = T - f0i < SCAN_CHUNK ? T - f0i : SCAN_CHUNK;
    __syncthreads();
    for (int f = tid; f < n; f += 256) v[f] = p[f0i + f];
    __syncthreads();
    if (tid == 0) {
      double acc = carry;
      for (int f = 0; f < n; ++f) {
        const double x = v[f];
        v[f] = acc;
        acc += x;
      }
      carry = acc;
    }
    __syncthreads();
    for (int f = tid; f < n; f += 256) p[f0i + f] = v[f];
  }
}

// har_source[b][i] = tanh(l_linear(sine_waves))  (source.py:408-441,518-531)
__global__ void src_final_kernel(const float* __restrict__ f0, const double* __restrict__ base, const float* __restrict__ r0,
                                 const double* __restrict__ q, const float* __restrict__ sine_noise, uint64_t seed,
                                 const float* __restrict__ lw, const float* __restrict__ lb, float* __restrict__ har, int B,
                                 int T, int hop, float sr) {
  __shared__ double sh[HOP_MAX / 64];
  const int bf = blockIdx.x;
  const int b = bf / T, f = bf % T;
  const int k = threadIdx.x;
  const int64_t L = (int64_t)T * hop;
  const int64_t i = (int64_t)f * hop + k;
  const float f0v = f0[(int64_t)b * T + f];
  const float uv = f0v > 0.f ? 1.0f : 0.0f;
  const float noise_amp = uv * 0.003f + (1.0f - uv) * 0.1f / 3.0f;
  const SsPhilox rng(seed);
  float merged = lb[0];
  for (int h = 0; h < NH; ++h) {
    const float rad = rad_of(f0v, h, sr);
    const double basef = base[((int64_t)b * NH + h) * T + f];
    const float a = second_addend(basef, rad, r0[b * NH + h], f, k);
    const double incl = block_scan_incl((double)a, sh, k, hop);
    const float c2 = (float)(q[((int64_t)b * NH + h) * T + f] + incl);
    const float s = sinf(c2 * 2.0f * 3.14159274101257324f) * 0.1f;
    float z;
    if (sine_noise) z = sine_noise[((int64_t)b * L + i) * NH + h];
    else {
      uint32_t o[4];
      const uint64_t ctr = ((uint64_t)b * L + i) * NH + h;
      rng.gen((uint32_t)ctr, (uint32_t)(ctr >> 32), 0x53494e45u, 0x4e534632u, o);
      float z1;
      ss_boxmuller(o[0], o[1], z, z1);
    }
    const float sw = s * uv + noise_amp * z;
    merged += lw[h] * sw;
  }
  har[(int64_t)b * L + i] = tanhf(merged);
}

// x[b][n][c] += bias[c] + sum_j w[c][j] * har[b][n*s - pad + j]      (noise_convs, hifigan_nsf.py:124-130,155-157)
__global__ __launch_bounds__(256) void noise_conv_kernel(const float* __restrict__ har, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ x,
                                                         const int32_t* __restrict__ lens_out,
                                                         const int32_t* __restrict__ lens_samples, int B, int Tout,
                                                         int64_t L, int C, int k, int s, int pad, int npos) {
  extern __shared__ float smem[];
  float* wsh = smem;             // [k][C]
  float* hsh = smem + k * C;     // [npos*s + k]
  const int tiles = (Tout + npos - 1) / npos;
  const int b = blockIdx.x / tiles;
  const int n0 = (blockIdx.x % tiles) * npos;
  const int tid = threadIdx.x;
  for (int i = tid; i < k * C; i += 256) {
    const int j = i / C, c = i % C;
    wsh[i] = w[c * k + j];
  }
  const int64_t valid = lens_samples ? (int64_t)lens_samples[b] : L;
  const int seg = npos * s + k;
  for (int i = tid; i < seg; i += 256) {
    const int64_t idx = (int64_t)n0 * s - pad + i;
    hsh[i] = (idx >= 0 && idx < valid) ? har[(int64_t)b * L + idx] : 0.f;
  }
  __syncthreads();
  const int groups = 256 / C > 0 ? 256 / C : 1;
  const int c = tid % C;
  const int g = tid / C;
  if (g >= groups) return;
  const int len_out = lens_out ? lens_out[b] : Tout;
  const float bs = bias[c];
  for (int p = g; p < npos; p += groups) {
    const int n = n0 + p;
    if (n >= Tout || n >= len_out) continue;
    float acc = bs;
    for (int j = 0; j < k; ++j) acc += wsh[j * C + c] * hsh[p * s + j];
    x[((int64_t)b * Tout + n) * C + c] += acc;
  }
}

// wav[b][i] = tanh(conv_post(leaky_relu(x, 0.01)))   (hifigan_nsf.py:165-167)
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ wav,
                                                        const int32_t* __restrict__ lens, int B, int64_t L, int C, int k) {
  extern __shared__ float smem[];
  const int ldx = C + 1;
  float* xs = smem;                          // [256 + k - 1][C+1]
  float* wsh = smem + (256 + k - 1) * ldx;   // [k][C]
  const int tiles = (int)((L + 255) / 256);
  const int b = blockIdx.x / tiles;
  const int64_t i0 = (int64_t)(blockIdx.x % tiles) * 256;
  const int tid = threadIdx.x;
  const int64_t valid = lens ? (int64_t)lens[b] : L;
  const int half = (k - 1) / 2;
  for (int i = tid; i < k * C; i += 256) {
    const int j = i / C, c = i % C;
    wsh[i] = w[c * k + j];
  }
  const int rows = 256 + k - 1;
  for (int i = tid; i < rows * C; i += 256) {
    const int r = i / C, c = i % C;
    const int64_t idx = i0 - half + r;
    float v = 0.f;
    if (idx >= 0 && idx < valid && idx < L) {
      v = x[((int64_t)b * L + idx) * C + c];
      v = v >= 0.f ? v : v * 0.01f;
    }
    xs[r * ldx + c] = v;
  }
  __syncthreads();
  const int64_t i = i0 + tid;
  if (i >= L) return;
  float acc = bias[0];
  for (int j = 0; j < k; ++j)
    for (int c = 0; c < C; ++c) acc += wsh[j * C + c] * xs[(tid + j) * ldx + c];
  wav[(int64_t)b * L + i] = (i < valid) ? tanhf(acc) : 0.f;
}

inline ss_conv_gemm_args base_args(int B, int T, const int32_t* lens) {
  ss_conv_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.B = B;
  a.T = T;
  a.lens = lens;
  a.ntaps = 1;
  a.a_scale = 1.0f;
  a.a_lrelu = 1.0f;
  a.pre_scale = 1.0f;
  a.post_scale = 1.0f;
  a.mask_rows = 1;
  return a;
}

// W_wino != NULL: the grouped Winograd F(4,3) kernel (fp32 mode; the caller checked ss_wino43_conv_ok). act_out = 1: leaky-relu(0.1)
// of the OUTPUT in the epilogue - the value is only ever read through the next conv's input leaky-relu, which then passes lrelu = 1.
int conv_same(const float* A, int B, int Trows, int C, const int32_t* lens, const float* W, const float* W_wino, const float* bias, int k,
              int d, float lrelu, int act_out, const float* R, float post_scale, int accumulate, float* out, int bf16, hipStream_t stream) {
  ss_conv_gemm_args a = base_args(B, Trows, lens);
  a.mfma_bf16 = bf16;
  a.A = A;
  a.lda = C;
  a.a_batch_stride = (int64_t)Trows * C;
  a.Cin = C;
  a.ntaps = k;
  for (int j = 0; j < k; ++j) a.tap_off[j] = (j - (k - 1) / 2) * d;
  a.a_lrelu = lrelu;
  a.W = W;
  a.N = C;
  a.Np = round_up32(C);
  a.Kp = round_up32(C);
  a.epi = SS_EPI_STORE;
  a.bias = bias;
  a.R = R;
  a.ldr = C;
  a.r_batch_stride = (int64_t)Trows * C;
  a.post_scale = post_scale;
  a.accumulate = accumulate;
  a.C = out;
  a.ldc = C;
  a.c_batch_stride = (int64_t)Trows * C;
  if (act_out) {
    a.act = SS_ACT_LRELU_;
    a.act_slope = 0.1f;
  }
  // 32-bit byte offsets inside an item: longer items take the direct kernel ("voc_wino_max_mb" knob, default 2048 MiB = the real limit)
  if (W_wino && ((int64_t)Trows + 1024) * C * 4 < ((int64_t)g_ss_tuning.voc_wino_max_mb << 20)) {
    a.W = W_wino;
    return ss_wino43_conv(&a, k, d, stream);
  }
  return ss_conv_gemm(&a, stream);
}

}  // namespace

extern "C" int64_t ss_hifigan_workspace_bytes(const ss_hifigan* hg, int B, int T) {
  if (!hg || B <= 0 || T <= 0) return -1;
  return hg_layout(hg, B, T, nullptr).bytes;
}

extern "C" int ss_hifigan_source(const ss_hifigan* hg, const float* f0, int B, int T, const float* rand_ini,
                                 const float* sine_noise, uint64_t seed, float* har, void* ws, int64_t ws_bytes,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(hg && f0 && har && ws, "ss_hifigan_source: null pointer");
  SS_CHECK_ARG(hg->harmonics + 1 == NH, "ss_hifigan_source: harmonic_num must be 8");
  const int hop = hop_of(hg);
  SS_CHECK_ARG(hop % 64 == 0 && hop <= HOP_MAX, "ss_hifigan_source: hop=%d must be a multiple of 64 and <= %d", hop, HOP_MAX);
  const HgWs w = hg_layout(hg, B, T, ws);
  SS_CHECK_ARG(ws_bytes >= w.bytes, "ss_hifigan_source: workspace too small");
  const float sr = (float)hg->sr;
  hipLaunchKernelGGL(src_base_kernel, dim3(B * NH), dim3(256), 0, stream, f0, rand_ini, seed, w.base, w.r0, B, T, hop, sr);
  SS_CHECK_LAUNCH("src_base_kernel");
  hipLaunchKernelGGL(src_frame_sum_kernel, dim3(B * T), dim3(hop), 0, stream, f0, w.base, w.r0, w.q, B, T, hop, sr);
  SS_CHECK_LAUNCH("src_frame_sum_kernel");
  hipLaunchKernelGGL(src_scan_kernel, dim3(B * NH), dim3(256), 0, stream, w.q, B * NH, T);
  SS_CHECK_LAUNCH("src_scan_kernel");
  hipLaunchKernelGGL(src_final_kernel, dim3(B * T), dim3(hop), 0, stream, f0, w.base, w.r0, w.q, sine_noise, seed,
                     hg->src_w, hg->src_b, har, B, T, hop, sr);
  SS_CHECK_LAUNCH("src_final_kernel");
  return SS_OK;
}

extern "C" int ss_hifigan_forward(const ss_hifigan* hg, const float* mel, const float* f0, const int32_t* lens, int B, int T,
                                  const float* rand_ini, const float* sine_noise, uint64_t seed, float* wav,
                                  float* har_source_out, void* ws, int64_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(hg && mel && f0 && wav && ws, "ss_hifigan_forward: null pointer");
  SS_CHECK_ARG(hg->n_ups >= 1 && hg->n_ups <= SS_HG_MAX_UPS && hg->n_kernels >= 1 && hg->n_kernels <= SS_HG_MAX_KERNELS,
               "ss_hifigan_forward: bad config");
  const HgWs w = hg_layout(hg, B, T, ws);
  SS_CHECK_ARG(ws_bytes >= w.bytes, "ss_hifigan_forward: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)w.bytes);
  const int hop = hop_of(hg);
  const int64_t L = (int64_t)T * hop;

  hipLaunchKernelGGL(stage_lens_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, lens, w.lens, B, T, hg->n_ups,
                     hg->up_rate[0], hg->up_rate[1], hg->up_rate[2], hg->up_rate[3], hg->up_rate[4], hg->up_rate[5]);
  SS_CHECK_LAUNCH("stage_lens_kernel");

  // harmonic source (hifigan_nsf.py:145-149)
  float* har = har_source_out ? har_source_out : w.har;
  SS_PROPAGATE(ss_hifigan_source(hg, f0, B, T, rand_ini, sine_noise, seed, har, ws, ws_bytes, stream_));

  // conv_pre (k=7)
  {
    ss_conv_gemm_args a = base_args(B, T, w.lens);
    a.A = mel;
    a.lda = 80;
    a.a_batch_stride = (int64_t)T * 80;
    a.Cin = 80;
    a.ntaps = 7;
    for (int j = 0; j < 7; ++j) a.tap_off[j] = j - 3;
    a.W = hg->w_pre;
    a.N = hg->c0;
    a.Np = round_up32(hg->c0);
    a.Kp = 96;
    a.epi = SS_EPI_STORE;
    a.bias = hg->b_pre;
    a.C = w.pre;
    a.ldc = hg->c0;
    a.c_batch_stride = (int64_t)T * hg->c0;
    a.mfma_bf16 = hg->mfma_bf16;
    SS_PROPAGATE(ss_conv_gemm(&a, stream));
  }

  const float* cur = w.pre;
  int rows_in = T, cin = hg->c0, R = 1;
  for (int i = 0; i < hg->n_ups; ++i) {
    const int u = hg->up_rate[i], ku = hg->up_k[i];
    const int cout = hg->c0 >> (i + 1);
    const int pad = (ku - u) / 2;
    const int nph0 = u - pad;
    const int rows_out = rows_in * u;
    const int32_t* lens_in = w.lens + (int64_t)i * B;
    const int32_t* lens_out = w.lens + (int64_t)(i + 1) * B;
    // x = ups[i](leaky_relu(x, 0.1)) as two polyphase GEMMs
    for (int g = 0; g < 2; ++g) {
      const int nph = g == 0 ? nph0 : u - nph0;
      ss_conv_gemm_args a = base_args(B, rows_in, lens_in);
      a.A = cur;
      a.lda = cin;
      a.a_batch_stride = (int64_t)rows_in * cin;
      a.Cin = cin;
      a.ntaps = 2;
      a.tap_off[0] = g == 0 ? 0 : 1;
      a.tap_off[1] = g == 0 ? -1 : 0;
      a.a_lrelu = 0.1f;
      a.W = hg->w_up[i][g];
      a.N = nph * cout;
      a.Np = round_up32(nph * cout);
      a.Kp = round_up32(cin);
      a.epi = SS_EPI_STORE;
      a.bias = hg->b_up[i];
      a.C = w.x + (g == 0 ? 0 : (int64_t)nph0 * cout);
      a.ldc = u * cout;
      a.c_batch_stride = (int64_t)rows_in * u * cout;
      a.mfma_bf16 = hg->mfma_bf16;
      SS_PROPAGATE(ss_conv_gemm(&a, stream));
    }
    R *= u;
    // x += noise_convs[i](har_source)
    {
      int s = 1;
      for (int j = i + 1; j < hg->n_ups; ++j) s *= hg->up_rate[j];
      const int kk = (i + 1 < hg->n_ups) ? 2 * s : 1;
      const int pd = (i + 1 < hg->n_ups) ? s / 2 : 0;
      const int npos = 64;
      const size_t lds = ((size_t)kk * cout + (size_t)npos * s + kk) * sizeof(float);
      const int tiles = (rows_out + npos - 1) / npos;
      hipLaunchKernelGGL(noise_conv_kernel, dim3(B * tiles), dim3(256), lds, stream, har, hg->w_noise[i], hg->b_noise[i],
                         w.x, lens_out, w.lens + (int64_t)hg->n_ups * B, B, rows_out, L, cout, kk, s, pd, npos);
      SS_CHECK_LAUNCH("noise_conv_kernel");
    }
    // xs = mean_j ResBlock1_j(x)   (hifigan_nsf.py:54-61,158-164)
    const float inv = 1.0f / (float)hg->n_kernels;
    for (int j = 0; j < hg->n_kernels; ++j) {
      const int k = hg->rb_k[j];
      const float* xin = w.x;
      for (int m = 0; m < 3; ++m) {
        const int d = hg->rb_d[j][m];
        const bool wino = hg->wino && !hg->mfma_bf16;
        const float* ww1 = wino && ss_wino43_conv_ok(cout, k, d) ? hg->w_rb1_wino[i][j][m] : nullptr;
        const float* ww2 = wino && ss_wino43_conv_ok(cout, k, 1) ? hg->w_rb2_wino[i][j][m] : nullptr;
        // xt = c1(lrelu(x)) is only read as lrelu(xt): the first conv stores lrelu(xt), the second needs no input activation
        SS_PROPAGATE(conv_same(xin, B, rows_out, cout, lens_out, hg->w_rb1[i][j][m], ww1, hg->b_rb1[i][j][m], k, d, 0.1f, 1, nullptr, 1.0f,
                               0, w.ta, hg->mfma_bf16, stream));
        if (m < 2) {
          SS_PROPAGATE(conv_same(w.ta, B, rows_out, cout, lens_out, hg->w_rb2[i][j][m], ww2, hg->b_rb2[i][j][m], k, 1, 1.0f, 0, xin, 1.0f,
                                 0, w.tb, hg->mfma_bf16, stream));
          xin = w.tb;
        } else {
          SS_PROPAGATE(conv_same(w.ta, B, rows_out, cout, lens_out, hg->w_rb2[i][j][m], ww2, hg->b_rb2[i][j][m], k, 1, 1.0f, 0, xin, inv,
                                 j > 0, w.xs, hg->mfma_bf16, stream));
        }
      }
    }
    cur = w.xs;
    rows_in = rows_out;
    cin = cout;
  }
  // leaky_relu(0.01) -> conv_post -> tanh
  {
    const int k = 7;
    const size_t lds = ((size_t)(256 + k - 1) * (cin + 1) + (size_t)k * cin) * sizeof(float);
    const int tiles = (int)((L + 255) / 256);
    hipLaunchKernelGGL(conv_post_kernel, dim3(B * tiles), dim3(256), lds, stream, cur, hg->w_post, hg->b_post, wav,
                       w.lens + (int64_t)hg->n_ups * B, B, L, cin, k);
    SS_CHECK_LAUNCH("conv_post_kernel");
  }
  return SS_OK;
}
