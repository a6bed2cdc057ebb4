// Row-wise and elementwise kernels (HBM-bound): wave-per-row LayerNorm, embedding gathers,
// fairseq positions, length regulator, masks.  One 64-lane wave owns one row; reductions are
// wavefront shuffles; loads are coalesced along the channel dim.
#include "common.h"
#include "../../include/stylesinger_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, C <= 64*MAXV.
// ---------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int B, int T, int C, int ldx, int ldy, int64_t xbs, int64_t ybs,
                                                        float eps, const int32_t* __restrict__ lens, int mask_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * T) return;
  const int b = row / T, t = row % T;
  const float* xr = x + (int64_t)b * xbs + (int64_t)t * ldx;
  float* yr = y + (int64_t)b * ybs + (int64_t)t * ldy;
  if (mask_rows && lens && t >= lens[b]) {
    for (int c = lane; c < C; c += 64) yr[c] = 0.f;
    return;
  }
  float v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float var = wave_sum(q) / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < C) yr[c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
}

template <typename IdxT>
__global__ void embedding_kernel(const IdxT* __restrict__ idx, const float* __restrict__ table, float* __restrict__ out,
                                 int rows, int C, int n_table, float scale, int accumulate) {
  const int64_t total = (int64_t)rows * (C / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (C / 4));
    const int c4 = (int)(i % (C / 4));
    int64_t id = (int64_t)idx[r];
    if (id < 0) id = 0;
    if (id >= n_table) id = n_table - 1;
    float4 v = *reinterpret_cast<const float4*>(table + id * C + c4 * 4);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    float4* o = reinterpret_cast<float4*>(out + (int64_t)r * C + c4 * 4);
    if (accumulate) {
      float4 p = *o;
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    *o = v;
  }
}

// positions = cumsum(nz) * nz (+ padding_idx = 0); one wave per item, ballot prefix over 64-wide chunks.
__global__ __launch_bounds__(64) void make_positions_kernel(const int64_t* __restrict__ probe_i64,
                                                            const float* __restrict__ probe_f32, int ldp, int64_t pbs,
                                                            int32_t* __restrict__ pos, int B, int T) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int running = 0;
  for (int t0 = 0; t0 < T; t0 += 64) {
    const int t = t0 + lane;
    bool nz = false;
    if (t < T) nz = probe_i64 ? (probe_i64[(int64_t)b * T + t] != 0) : (probe_f32[(int64_t)b * pbs + (int64_t)t * ldp] != 0.0f);
    const unsigned long long m = __ballot(nz);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (t < T) pos[(int64_t)b * T + t] = nz ? running + before + 1 : 0;
    running += __popcll(m);
  }
}

// out[b][t][c] (+)= alpha * table[pos[b][t]][c]
__global__ void table_add_kernel(const int32_t* __restrict__ pos, const float* __restrict__ table, int table_rows,
                                 float* __restrict__ out, int ldo, int64_t obs, int B, int T, int C,
                                 const float* __restrict__ alpha_dev, float alpha_host, int accumulate) {
  const float alpha = alpha_dev ? alpha_dev[0] * alpha_host : alpha_host;
  const int64_t total = (int64_t)B * T * (C / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4));
    const int64_t r = i / (C / 4);
    const int b = (int)(r / T), t = (int)(r % T);
    int p = pos[r];
    if (p >= table_rows) p = table_rows - 1;
    float4 v = *reinterpret_cast<const float4*>(table + (int64_t)p * C + c4 * 4);
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    float4* o = reinterpret_cast<float4*>(out + (int64_t)b * obs + (int64_t)t * ldo + c4 * 4);
    if (accumulate) {
      float4 q = *o;
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *o = v;
  }
}

// out = ((((x + v1[b]) + y1) + v2[b]) + y2) * (t < lens[b])   (every addend optional; order as in stylesinger.py:139-166)
// x and out may be the SAME buffer (the callers mask in place): neither is __restrict__
__global__ void add_bcast_mask_kernel(const float* x, const float* __restrict__ v1,
                                      const float* __restrict__ y1, const float* __restrict__ v2,
                                      const float* __restrict__ y2, float* out, int B, int T, int C,
                                      const int32_t* __restrict__ lens) {
  const int64_t total = (int64_t)B * T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int b = (int)(r / T), t = (int)(r % T);
    float v = x[i];
    if (v1) v += v1[(int64_t)b * C + c];
    if (y1) v += y1[i];
    if (v2) v += v2[(int64_t)b * C + c];
    if (y2) v += y2[i];
    if (lens && t >= lens[b]) v = 0.f;
    out[i] = v;
  }
}

__global__ void gather_expand_kernel(const float* __restrict__ src, const int64_t* __restrict__ mel2ph,
                                     float* __restrict__ out, int B, int Tsrc, int T, int C) {
  const int64_t total = (int64_t)B * T * (C / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (C / 4));
    const int64_t r = i / (C / 4);
    const int b = (int)(r / T);
    const int64_t m = mel2ph[r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m > 0 && m <= Tsrc) v = *reinterpret_cast<const float4*>(src + ((int64_t)b * Tsrc + (m - 1)) * C + c4 * 4);
    *reinterpret_cast<float4*>(out + r * C + c4 * 4) = v;
  }
}

__global__ void gather_expand_i64_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ mel2ph,
                                         int64_t* __restrict__ out, int B, int Tsrc, int T) {
  const int64_t total = (int64_t)B * T;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / T);
    const int64_t m = mel2ph[i];
    out[i] = (m > 0 && m <= Tsrc) ? src[(int64_t)b * Tsrc + (m - 1)] : 0;
  }
}

__global__ void note_dur_add_kernel(const float* __restrict__ dur, const float* __restrict__ w, const float* __restrict__ b,
                                    float* __restrict__ out, int rows, int C) {
  const int64_t total = (int64_t)rows * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int r = (int)(i / C);
    out[i] += dur[r] * w[c] + b[c];
  }
}

// DurationPredictor.out2dur + LengthRegulator: one wave per item (Tp is small).
__global__ __launch_bounds__(64) void length_regulate_kernel(const float* __restrict__ logdur,
                                                             const int64_t* __restrict__ tokens,
                                                             int64_t* __restrict__ dur_out, int64_t* __restrict__ mel2ph,
                                                             int32_t* __restrict__ lens, int B, int Tp, int Tmax) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  // 1) durations
  for (int i = lane; i < Tp; i += 64) {
    const float x = logdur[(int64_t)b * Tp + i];
    float d = rintf(expf(x) - 1.0f);  // torch.round = round-half-even
    if (!(d > 0.f)) d = 0.f;
    int64_t di = (int64_t)d;
    if (tokens[(int64_t)b * Tp + i] == 0) di = 0;
    dur_out[(int64_t)b * Tp + i] = di;
  }
  __syncthreads();
  // 2) sequential cumsum by lane 0 (Tp <= a few hundred), then fill
  if (lane == 0) {
    int64_t cum = 0;
    for (int i = 0; i < Tp; ++i) {
      const int64_t d = dur_out[(int64_t)b * Tp + i];
      const int64_t s = cum, e = cum + d;
      for (int64_t t = s; t < e && t < Tmax; ++t) mel2ph[(int64_t)b * Tmax + t] = i + 1;
      cum = e;
    }
    const int tot = (int)((Tmax == 0 || cum < Tmax) ? cum : Tmax);
    for (int t = tot; t < Tmax; ++t) mel2ph[(int64_t)b * Tmax + t] = 0;
    lens[b] = tot;
  }
}

__global__ __launch_bounds__(64) void count_nonzero_kernel(const int64_t* __restrict__ x, int32_t* __restrict__ lens, int B,
                                                           int T) {
  const int b = blockIdx.x;
  int cnt = 0;
  for (int t = threadIdx.x; t < T; t += 64) cnt += x[(int64_t)b * T + t] > 0 ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (threadIdx.x == 0) lens[b] = cnt;
}

// lens[b] = 1 + last t with ref[b][t][0] != 0 (padding mask of the style reference, lse.py:109)
__global__ __launch_bounds__(64) void ref_lens_kernel(const float* __restrict__ ref, int B, int T, int C,
                                                      int32_t* __restrict__ lens) {
  const int b = blockIdx.x;
  int last = 0;
  for (int t = threadIdx.x; t < T; t += 64)
    if (ref[((int64_t)b * T + t) * C] != 0.0f) last = t + 1 > last ? t + 1 : last;
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(last, o);
    last = other > last ? other : last;
  }
  if (threadIdx.x == 0) lens[b] = last;
}

__global__ void add_rowscalar_kernel(float* __restrict__ x, const float* __restrict__ s, int B, int T, int C,
                                     const int32_t* __restrict__ lens) {
  const int64_t total = (int64_t)B * T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int b = (int)(r / T), t = (int)(r % T);
    if (!lens || t < lens[b]) x[i] += s[r];
  }
}

__global__ void mask_rows_by_ref_kernel(float* __restrict__ x, const float* __restrict__ ref, int ldref, int rows, int C) {
  const int64_t total = (int64_t)rows * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    if (ref[r * ldref] == 0.0f) x[i] = 0.f;
  }
}

__global__ void clip_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float lo, float hi) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fminf(fmaxf(x[i], lo), hi);
}

__global__ void fill_normal_kernel(float* __restrict__ x, int64_t n, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                   uint64_t offset) {
  const SsPhilox rng(seed + (seed_dev ? seed_dev[0] : 0ull));
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i * 4 < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t o[4];
    const uint64_t ctr = offset + (uint64_t)i;
    rng.gen((uint32_t)ctr, (uint32_t)(ctr >> 32), 0x46494c4cu, 0u, o);
    float z[4];
    ss_boxmuller(o[0], o[1], z[0], z[1]);
    ss_boxmuller(o[2], o[3], z[2], z[3]);
    for (int k = 0; k < 4; ++k)
      if (i * 4 + k < n) x[i * 4 + k] = z[k];
  }
}

__global__ void fill_normal_rows_kernel(float* __restrict__ x, int B, int T, int ld, uint64_t seed, const uint64_t* __restrict__ seed_dev) {
  const SsPhilox rng(seed + (seed_dev ? seed_dev[0] : 0ull));
  const int q = (T + 3) / 4;
  const int64_t n = (int64_t)B * q;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / q), t4 = (int)(i % q);
    uint32_t o[4];
    rng.gen((uint32_t)t4, (uint32_t)b, 0x46494c4cu, 1u, o);
    float z[4];
    ss_boxmuller(o[0], o[1], z[0], z[1]);
    ss_boxmuller(o[2], o[3], z[2], z[3]);
    for (int k = 0; k < 4; ++k)
      if (t4 * 4 + k < T) x[(int64_t)b * ld + t4 * 4 + k] = z[k];
  }
}

inline int grid_for(int64_t work_items, int block = 256, int cap = 8192) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g < cap ? g : cap);
}

}  // namespace

extern "C" int ss_layernorm(const float* x, float* y, const float* gamma, const float* beta, int B, int T, int C, int ldx,
                            int ldy, int64_t xbs, int64_t ybs, float eps, const int32_t* lens, int mask_rows,
                            void* stream) {
  SS_CHECK_ARG(x && y && gamma && beta, "ss_layernorm: null pointer");
  SS_CHECK_ARG(B > 0 && T > 0 && C > 0 && C <= 512, "ss_layernorm: bad dims B=%d T=%d C=%d (C<=512)", B, T, C);
  const int rows = B * T;
  dim3 grid((rows + 3) / 4), block(256);
  if (C <= 128)
    hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, (hipStream_t)stream, x, y, gamma, beta, B, T, C, ldx, ldy, xbs,
                       ybs, eps, lens, mask_rows);
  else if (C <= 256)
    hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, (hipStream_t)stream, x, y, gamma, beta, B, T, C, ldx, ldy, xbs,
                       ybs, eps, lens, mask_rows);
  else
    hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, (hipStream_t)stream, x, y, gamma, beta, B, T, C, ldx, ldy, xbs,
                       ybs, eps, lens, mask_rows);
  SS_CHECK_LAUNCH("ss_layernorm");
  return SS_OK;
}

extern "C" int ss_embedding(const int64_t* idx, const float* table, float* out, int rows, int C, int n_table, float scale,
                            int accumulate, void* stream) {
  SS_CHECK_ARG(idx && table && out && rows > 0 && (C & 3) == 0 && n_table > 0, "ss_embedding: bad args");
  hipLaunchKernelGGL(embedding_kernel<int64_t>, dim3(grid_for((int64_t)rows * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     idx, table, out, rows, C, n_table, scale, accumulate);
  SS_CHECK_LAUNCH("ss_embedding");
  return SS_OK;
}

extern "C" int ss_make_positions(const int64_t* probe_i64, const float* probe_f32, int ldp, int64_t probe_batch_stride,
                                 int32_t* pos, int B, int T, void* stream) {
  SS_CHECK_ARG((probe_i64 != nullptr) != (probe_f32 != nullptr), "ss_make_positions: exactly one probe must be given");
  SS_CHECK_ARG(pos && B > 0 && T > 0, "ss_make_positions: bad args");
  hipLaunchKernelGGL(make_positions_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, probe_i64, probe_f32, ldp,
                     probe_batch_stride, pos, B, T);
  SS_CHECK_LAUNCH("ss_make_positions");
  return SS_OK;
}

extern "C" int ss_table_add(const int32_t* pos, const float* table, int table_rows, float* out, int ldo,
                            int64_t out_batch_stride, int B, int T, int C, const float* alpha_dev, float alpha,
                            int accumulate, void* stream) {
  SS_CHECK_ARG(pos && table && out && (C & 3) == 0 && (ldo & 3) == 0, "ss_table_add: bad args");
  hipLaunchKernelGGL(table_add_kernel, dim3(grid_for((int64_t)B * T * C / 4)), dim3(256), 0, (hipStream_t)stream, pos,
                     table, table_rows, out, ldo, out_batch_stride, B, T, C, alpha_dev, alpha, accumulate);
  SS_CHECK_LAUNCH("ss_table_add");
  return SS_OK;
}

extern "C" int ss_add_bcast_mask(const float* x, const float* v1, const float* y1, const float* v2, const float* y2,
                                 float* out, int B, int T, int C, const int32_t* lens, void* stream) {
  SS_CHECK_ARG(x && out, "ss_add_bcast_mask: null pointer");
  hipLaunchKernelGGL(add_bcast_mask_kernel, dim3(grid_for((int64_t)B * T * C)), dim3(256), 0, (hipStream_t)stream, x, v1,
                     y1, v2, y2, out, B, T, C, lens);
  SS_CHECK_LAUNCH("ss_add_bcast_mask");
  return SS_OK;
}

extern "C" int ss_gather_expand(const float* src, const int64_t* mel2ph, float* out, int B, int Tsrc, int T, int C,
                                void* stream) {
  SS_CHECK_ARG(src && mel2ph && out && (C & 3) == 0, "ss_gather_expand: bad args");
  hipLaunchKernelGGL(gather_expand_kernel, dim3(grid_for((int64_t)B * T * C / 4)), dim3(256), 0, (hipStream_t)stream, src,
                     mel2ph, out, B, Tsrc, T, C);
  SS_CHECK_LAUNCH("ss_gather_expand");
  return SS_OK;
}

extern "C" int ss_gather_expand_i64(const int64_t* src, const int64_t* mel2ph, int64_t* out, int B, int Tsrc, int T,
                                    void* stream) {
  SS_CHECK_ARG(src && mel2ph && out, "ss_gather_expand_i64: bad args");
  hipLaunchKernelGGL(gather_expand_i64_kernel, dim3(grid_for((int64_t)B * T)), dim3(256), 0, (hipStream_t)stream, src,
                     mel2ph, out, B, Tsrc, T);
  SS_CHECK_LAUNCH("ss_gather_expand_i64");
  return SS_OK;
}

extern "C" int ss_note_dur_add(const float* dur, const float* w, const float* b, float* out, int rows, int C,
                               void* stream) {
  SS_CHECK_ARG(dur && w && b && out, "ss_note_dur_add: null pointer");
  hipLaunchKernelGGL(note_dur_add_kernel, dim3(grid_for((int64_t)rows * C)), dim3(256), 0, (hipStream_t)stream, dur, w, b,
                     out, rows, C);
  SS_CHECK_LAUNCH("ss_note_dur_add");
  return SS_OK;
}

extern "C" int ss_length_regulate(const float* logdur, const int64_t* tokens, int64_t* dur_out, int64_t* mel2ph,
                                  int32_t* lens, int B, int Tp, int Tmax, void* stream) {
  SS_CHECK_ARG(logdur && tokens && dur_out && lens && (mel2ph || Tmax == 0), "ss_length_regulate: null pointer");
  hipLaunchKernelGGL(length_regulate_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logdur, tokens, dur_out, mel2ph,
                     lens, B, Tp, Tmax);
  SS_CHECK_LAUNCH("ss_length_regulate");
  return SS_OK;
}

extern "C" int ss_count_nonzero_i64(const int64_t* x, int32_t* lens, int B, int T, void* stream) {
  SS_CHECK_ARG(x && lens, "ss_count_nonzero_i64: null pointer");
  hipLaunchKernelGGL(count_nonzero_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, lens, B, T);
  SS_CHECK_LAUNCH("ss_count_nonzero_i64");
  return SS_OK;
}

extern "C" int ss_ref_lens(const float* ref_mels, int B, int T, int C, int32_t* lens, void* stream) {
  SS_CHECK_ARG(ref_mels && lens, "ss_ref_lens: null pointer");
  hipLaunchKernelGGL(ref_lens_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, ref_mels, B, T, C, lens);
  SS_CHECK_LAUNCH("ss_ref_lens");
  return SS_OK;
}

extern "C" int ss_add_rowscalar(float* x, const float* s, int B, int T, int C, const int32_t* lens, void* stream) {
  SS_CHECK_ARG(x && s, "ss_add_rowscalar: null pointer");
  hipLaunchKernelGGL(add_rowscalar_kernel, dim3(grid_for((int64_t)B * T * C)), dim3(256), 0, (hipStream_t)stream, x, s, B,
                     T, C, lens);
  SS_CHECK_LAUNCH("ss_add_rowscalar");
  return SS_OK;
}

extern "C" int ss_mask_rows_by_ref(float* x, const float* ref, int ldref, int rows, int C, void* stream) {
  SS_CHECK_ARG(x && ref, "ss_mask_rows_by_ref: null pointer");
  hipLaunchKernelGGL(mask_rows_by_ref_kernel, dim3(grid_for((int64_t)rows * C)), dim3(256), 0, (hipStream_t)stream, x, ref,
                     ldref, rows, C);
  SS_CHECK_LAUNCH("ss_mask_rows_by_ref");
  return SS_OK;
}

extern "C" int ss_clip(const float* x, float* y, int64_t n, float lo, float hi, void* stream) {
  SS_CHECK_ARG(x && y && n > 0, "ss_clip: bad args");
  hipLaunchKernelGGL(clip_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, lo, hi);
  SS_CHECK_LAUNCH("ss_clip");
  return SS_OK;
}

// |X| of a DFT stored as two column blocks (real part at [0,nbins), imaginary part at [sin_off, sin_off+nbins)):
// P[r][f] = sqrt(re^2 + im^2), columns [nbins, ldp) are written as 0 (they are the K padding of the mel GEMM).
__global__ void spec_mag_kernel(const float* __restrict__ S, float* __restrict__ P, int64_t rows, int lds, int ldp, int nbins,
                                int sin_off) {
  const int64_t n = rows * ldp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ldp;
    const int f = (int)(i - r * ldp);
    float v = 0.f;
    if (f < nbins) {
      const float re = S[r * lds + f], im = S[r * lds + sin_off + f];
      v = sqrtf(re * re + im * im);
    }
    P[i] = v;
  }
}

__global__ void log10_floor_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float eps) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = log10f(fmaxf(eps, x[i]));
}

extern "C" int ss_spec_magnitude(const float* S, float* P, int64_t rows, int lds, int ldp, int nbins, int sin_off, void* stream) {
  SS_CHECK_ARG(S && P && rows > 0 && nbins > 0 && nbins <= ldp && sin_off + nbins <= lds, "ss_spec_magnitude: bad args");
  hipLaunchKernelGGL(spec_mag_kernel, dim3(grid_for(rows * ldp)), dim3(256), 0, (hipStream_t)stream, S, P, rows, lds, ldp, nbins, sin_off);
  SS_CHECK_LAUNCH("ss_spec_magnitude");
  return SS_OK;
}

extern "C" int ss_log10_floor(const float* x, float* y, int64_t n, float eps, void* stream) {
  SS_CHECK_ARG(x && y && n > 0, "ss_log10_floor: bad args");
  hipLaunchKernelGGL(log10_floor_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, eps);
  SS_CHECK_LAUNCH("ss_log10_floor");
  return SS_OK;
}

// int16 PCM exactly as numpy's `(wav * 32767).astype(np.int16)` computes it for in-range samples (utils/audio.py:12-17):
// fp32 multiply, then truncation toward zero. Out-of-range products saturate instead of wrapping.
__global__ void pcm16_kernel(const float* __restrict__ x, int16_t* __restrict__ y, int64_t n, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = x[i] * scale;
    v = fminf(fmaxf(v, -32768.0f), 32767.0f);
    y[i] = (int16_t)(int)v;
  }
}

extern "C" int ss_wav_to_pcm16(const float* wav, int16_t* pcm, int64_t n, float scale, void* stream) {
  SS_CHECK_ARG(wav && pcm && n > 0, "ss_wav_to_pcm16: bad args");
  hipLaunchKernelGGL(pcm16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, wav, pcm, n, scale);
  SS_CHECK_LAUNCH("ss_wav_to_pcm16");
  return SS_OK;
}

extern "C" int ss_fill_normal_rows(float* x, int B, int T, int ld, uint64_t seed, const uint64_t* seed_dev, void* stream) {
  SS_CHECK_ARG(x && B > 0 && T > 0 && ld >= T, "ss_fill_normal_rows: bad args");
  hipLaunchKernelGGL(fill_normal_rows_kernel, dim3(grid_for((int64_t)B * ((T + 3) / 4))), dim3(256), 0, (hipStream_t)stream, x, B, T, ld,
                     seed, seed_dev);
  SS_CHECK_LAUNCH("ss_fill_normal_rows");
  return SS_OK;
}

extern "C" int ss_fill_normal(float* x, int64_t n, uint64_t seed, const uint64_t* seed_dev, uint64_t offset, void* stream) {
  SS_CHECK_ARG(x && n > 0, "ss_fill_normal: bad args");
  hipLaunchKernelGGL(fill_normal_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, n, seed,
                     seed_dev, offset);
  SS_CHECK_LAUNCH("ss_fill_normal");
  return SS_OK;
}
