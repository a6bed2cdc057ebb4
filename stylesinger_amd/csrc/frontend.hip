// Input-producer kernels of `preprocess_input` that run on the device: the f0 conditioning of
// utils/pitch_utils.py:34-62 and the small element-wise steps of the emotion encoder's 40-mel front end
// (data_gen/tts/emotion/audio.py:43-55). All HBM-bound, off the hot path; the GEMM-shaped steps reuse ss_conv_gemm.
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <limits.h>

namespace {

// inclusive max-scan (dir = +1) / min-scan (dir = -1 means "suffix minimum") helpers over one 256-thread block
__device__ __forceinline__ int wave_scan_max(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(v, o);
    if (lane >= o) v = max(v, n);
  }
  return v;
}
__device__ __forceinline__ int wave_scan_min_down(int v, int lane) {  // suffix minimum inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_down(v, o);
    if (lane + o < 64) v = min(v, n);
  }
  return v;
}

// One workgroup per utterance. Pass A: prev[t] = last voiced frame <= t (or -1); pass B: next[t] = first voiced frame >= t (or
// INT_MAX); both kept as integer bit patterns in the two output rows. Pass C: the value.
//   voiced:            log2(f0 + 1e-8)
//   no voiced frame:   0
//   unvoiced:          np.interp(t, voiced_idx, voiced_val): slope * (t - xp[j]) + fp[j] in double (multiply, then add), flat beyond
//                      the first / last voiced frame.
__global__ __launch_bounds__(256) void norm_interp_f0_kernel(const float* __restrict__ f0, const int32_t* __restrict__ lens,
                                                             float* __restrict__ out, float* __restrict__ uv, int T) {
  const int b = blockIdx.x;
  const int n = lens ? min(max(lens[b], 0), T) : T;
  const float* x = f0 + (int64_t)b * T;
  int* prev = reinterpret_cast<int*>(out + (int64_t)b * T);
  int* next = reinterpret_cast<int*>(uv + (int64_t)b * T);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int part[4];
  __shared__ int carry_s;
  if (tid == 0) carry_s = -1;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int t = base + tid;
    int v = (t < n && x[t] != 0.0f) ? t : -1;
    v = wave_scan_max(v, lane);
    if (lane == 63) part[wave] = v;
    __syncthreads();
    int pre = carry_s;
    for (int w = 0; w < wave; ++w) pre = max(pre, part[w]);
    v = max(v, pre);
    if (t < n) prev[t] = v;
    __syncthreads();
    if (tid == 255) carry_s = v;
    __syncthreads();
  }
  if (tid == 0) carry_s = INT_MAX;
  __syncthreads();
  const int nchunks = (n + 255) / 256;
  for (int c = nchunks - 1; c >= 0; --c) {
    const int t = c * 256 + tid;
    int v = (t < n && x[t] != 0.0f) ? t : INT_MAX;
    v = wave_scan_min_down(v, lane);
    if (lane == 0) part[wave] = v;
    __syncthreads();
    int post = carry_s;
    for (int w = wave + 1; w < 4; ++w) post = min(post, part[w]);
    v = min(v, post);
    if (t < n) next[t] = v;
    __syncthreads();
    if (tid == 0) carry_s = v;
    __syncthreads();
  }
  for (int t = tid; t < T; t += 256) {
    float y = 0.f, u = 0.f;
    if (t < n) {
      const int p = prev[t], q = next[t];
      if (p == t) {
        y = (float)log2((double)x[t] + 1e-8);
      } else {
        u = 1.f;
        if (p < 0 && q == INT_MAX) {
          y = 0.f;
        } else if (p < 0) {
          y = (float)log2((double)x[q] + 1e-8);
        } else if (q == INT_MAX) {
          y = (float)log2((double)x[p] + 1e-8);
        } else {
          // numpy keeps the contour in the input's dtype (fp32 here): fp = float32(log2(.)); np.interp then works in double
          const double fp0 = (double)(float)log2((double)x[p] + 1e-8), fp1 = (double)(float)log2((double)x[q] + 1e-8);
          const double slope = __ddiv_rn(__dsub_rn(fp1, fp0), (double)(q - p));
          y = (float)__dadd_rn(__dmul_rn(slope, (double)(t - p)), fp0);
        }
      }
    }
    out[(int64_t)b * T + t] = y;
    uv[(int64_t)b * T + t] = u;
  }
}

// power spectrum of a DFT stored as two column blocks (as ss_spec_magnitude): P[r][f] = re^2 + im^2, K padding written as 0
__global__ void spec_power_kernel(const float* __restrict__ S, float* __restrict__ P, int64_t rows, int lds, int ldp, int nbins,
                                  int sin_off) {
  const int64_t n = rows * ldp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ldp;
    const int f = (int)(i - r * ldp);
    float v = 0.f;
    if (f < nbins) {
      const float re = S[r * lds + f], im = S[r * lds + sin_off + f];
      v = fmaf(re, re, im * im);
    }
    P[i] = v;
  }
}

// y[b][i] = x[b][reflect(i - pad)] for i in [0, n[b] + 2 pad), 0 beyond: numpy.pad(mode="reflect") of each item's first n[b] samples
__global__ void reflect_pad_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens, float* __restrict__ y, int B,
                                   int Lx, int Ly, int pad) {
  const int64_t total = (int64_t)B * Ly;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / Ly);
    const int j = (int)(i - (int64_t)b * Ly);
    const int n = lens ? min(lens[b], Lx) : Lx;
    float v = 0.f;
    if (n > 0 && j < n + 2 * pad) {
      int s = j - pad;
      // reflect without repeating the edge sample, period 2(n-1)
      if (n == 1) {
        s = 0;
      } else {
        const int period = 2 * (n - 1);
        s %= period;
        if (s < 0) s += period;
        if (s >= n) s = period - s;
      }
      v = x[(int64_t)b * Lx + s];
    }
    y[i] = v;
  }
}

inline int fe_grid(int64_t work, int cap = 8192) {
  const int64_t g = (work + 255) / 256;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int ss_norm_interp_f0(const float* f0_hz, const int32_t* lens, float* out, float* uv, int B, int T, void* stream) {
  SS_CHECK_ARG(f0_hz && out && uv && B > 0 && T > 0, "ss_norm_interp_f0: bad args");
  SS_CHECK_ARG(f0_hz != out && f0_hz != uv && out != uv, "ss_norm_interp_f0: input and outputs must not alias");
  hipLaunchKernelGGL(norm_interp_f0_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, f0_hz, lens, out, uv, T);
  SS_CHECK_LAUNCH("ss_norm_interp_f0");
  return SS_OK;
}

extern "C" int ss_spec_power(const float* S, float* P, int64_t rows, int lds, int ldp, int nbins, int sin_off, void* stream) {
  SS_CHECK_ARG(S && P && rows > 0 && nbins > 0 && nbins <= ldp && sin_off + nbins <= lds, "ss_spec_power: bad args");
  hipLaunchKernelGGL(spec_power_kernel, dim3(fe_grid(rows * ldp)), dim3(256), 0, (hipStream_t)stream, S, P, rows, lds, ldp, nbins, sin_off);
  SS_CHECK_LAUNCH("ss_spec_power");
  return SS_OK;
}

extern "C" int ss_reflect_pad(const float* x, const int32_t* lens, float* y, int B, int Lx, int Ly, int pad, void* stream) {
  SS_CHECK_ARG(x && y && B > 0 && Lx > 0 && Ly > 0 && pad >= 0, "ss_reflect_pad: bad args");
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(fe_grid((int64_t)B * Ly)), dim3(256), 0, (hipStream_t)stream, x, lens, y, B, Lx, Ly, pad);
  SS_CHECK_LAUNCH("ss_reflect_pad");
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// trim_long_silences (data_gen/tts/emotion/audio.py:58-100) around the caller's VAD decisions. The reference asks webrtcvad (an un-vendored C
// library: a fixed-point GMM, no published text to restate) for one flag per 30 ms window; everything AROUND that decision is restated here and
// pinned against the real function run with injected flags (tests/golden/vad_trim.pt): the waveform cut to whole windows, the flags smoothed by
// a moving average of `avg_width` (zero padded (w - 1) / 2 left, w / 2 right; np.round = half to even, so a window stays voiced iff MORE than
// half of its neighbourhood is), dilated by `max_silence` windows to both sides (binary_dilation with ones(max_silence + 1), origin centred),
// the kept windows compacted. Integer logic + copies: bit-exact.
namespace {

__global__ __launch_bounds__(256) void vad_mask_kernel(const uint8_t* __restrict__ flags, int flags_stride, const int32_t* __restrict__ n_samples,
                                                       int spw, int avg_width, int dil, int max_windows, int32_t* __restrict__ win_dst,
                                                       int32_t* __restrict__ out_lens) {
  extern __shared__ int32_t vad_smem[];   // [max_windows] smoothed mask, then [max_windows] kept mask
  int32_t* m1 = vad_smem;
  int32_t* m2 = vad_smem + max_windows;
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint8_t* f = flags + (int64_t)b * flags_stride;
  const int nw = n_samples[b] / spw < max_windows ? n_samples[b] / spw : max_windows;
  const int lpad = (avg_width - 1) / 2, rpad = avg_width / 2;
  for (int i = tid; i < nw; i += 256) {
    int cnt = 0;
    for (int j = i - lpad; j <= i + rpad; ++j) cnt += (j >= 0 && j < nw && f[j]) ? 1 : 0;
    m1[i] = 2 * cnt > avg_width ? 1 : 0;   // np.round(cnt / width): exactly one half rounds to the even 0
  }
  __syncthreads();
  const int o = (dil + 1) / 2;   // structure ones(dil + 1), origin at its centre (scipy.ndimage.binary_dilation's default)
  for (int i = tid; i < nw; i += 256) {
    int any = 0;
    // binary_dilation: out[i] = OR_k structure[k] & in[i - k + origin], k = 0 .. dil
    for (int k = 0; k <= dil; ++k) {
      const int j = i - k + o;
      any |= (j >= 0 && j < nw) ? m1[j] : 0;
    }
    m2[i] = any;
  }
  __syncthreads();
  if (tid == 0) {   // windows per item are a few hundred: a serial prefix is cheaper than its own launch
    int kept = 0;
    for (int i = 0; i < nw; ++i) {
      win_dst[(int64_t)b * max_windows + i] = m2[i] ? kept : -1;
      kept += m2[i];
    }
    for (int i = nw; i < max_windows; ++i) win_dst[(int64_t)b * max_windows + i] = -1;
    out_lens[b] = kept * spw;
  }
}

__global__ __launch_bounds__(256) void vad_copy_kernel(const float* __restrict__ wav, int64_t wav_stride, const int32_t* __restrict__ win_dst, int max_windows,
                                                       int spw, float* __restrict__ out, int64_t out_stride, const int32_t* __restrict__ out_lens) {
  const int b = blockIdx.y, w = blockIdx.x;
  if (w < max_windows) {
    const int dst = win_dst[(int64_t)b * max_windows + w];
    if (dst >= 0)
      for (int i = threadIdx.x; i < spw; i += 256) out[(int64_t)b * out_stride + (int64_t)dst * spw + i] = wav[(int64_t)b * wav_stride + (int64_t)w * spw + i];
  }
  // zero the tail [out_lens[b], out_stride): block w owns the slice w, w + gridDim.x, ... of it
  const int64_t n = out_lens[b];
  for (int64_t i = n + (int64_t)w * 256 + threadIdx.x; i < out_stride; i += (int64_t)gridDim.x * 256) out[(int64_t)b * out_stride + i] = 0.f;
}

}  // namespace

extern "C" int ss_vad_trim(const float* wav, int64_t wav_stride, const int32_t* n_samples, const uint8_t* flags, int flags_stride, int B, int max_windows,
                           int samples_per_window, int avg_width, int max_silence, float* out, int64_t out_stride, int32_t* out_lens, int32_t* win_dst,
                           void* stream_) {
  SS_CHECK_ARG(wav && n_samples && flags && out && out_lens && win_dst, "ss_vad_trim: null argument");
  SS_CHECK_ARG(B > 0 && max_windows > 0 && samples_per_window > 0 && avg_width > 0 && max_silence >= 0 && flags_stride >= max_windows &&
                   out_stride >= (int64_t)max_windows * samples_per_window && wav_stride >= (int64_t)max_windows * samples_per_window,
               "ss_vad_trim: bad dims (B=%d windows=%d spw=%d)", B, max_windows, samples_per_window);
  SS_CHECK_ARG((size_t)max_windows * 8 <= 64 * 1024, "ss_vad_trim: more than 8192 windows per item (%d)", max_windows);
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(vad_mask_kernel, dim3(B), dim3(256), (size_t)max_windows * 8, stream, flags, flags_stride, n_samples, samples_per_window, avg_width,
                     max_silence, max_windows, win_dst, out_lens);
  SS_CHECK_LAUNCH("vad_mask_kernel");
  hipLaunchKernelGGL(vad_copy_kernel, dim3(max_windows, B), dim3(256), 0, stream, wav, wav_stride, win_dst, max_windows, samples_per_window, out, out_stride, out_lens);
  SS_CHECK_LAUNCH("vad_copy_kernel");
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Two element-wise producers of `preprocess_input` that used to be torch expressions on the host side of the binding:
//   ss_normalize_volume : audio.normalize_volume(wav, -30 dBFS, increase_only=True) (data_gen/tts/emotion/audio.py:109-115) per item of a zero-padded
//                         batch: gain = 10^((target - 10 log10(mean(wav^2))) / 20) when that change is >= 0, else 1 (mean over the item's own samples,
//                         accumulated in float64);
//   ss_round_f16_rows   : what `process_audio` hands the speaker encoder and the f0 tracker (inference/StyleSinger.py:86-88): the waveform zero-padded
//                         to n_out[b] samples and rounded to float16 (`.astype(np.float16)`, RNE), returned as fp32 values.
namespace {

__global__ __launch_bounds__(256) void normalize_volume_kernel(const float* __restrict__ wav, const int32_t* __restrict__ lens, float* __restrict__ out, int L,
                                                               float target_dbfs) {
  __shared__ double red[256];
  __shared__ float gain_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = wav + (int64_t)b * L;
  int n = lens ? lens[b] : L;
  n = n < 0 ? 0 : (n > L ? L : n);
  double s = 0.0;
  for (int i = tid; i < n; i += 256) s += (double)x[i] * (double)x[i];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    const double ms = red[0] / (n > 0 ? n : 1);
    const double change = (double)target_dbfs - 10.0 * log10(ms);
    gain_s = (ms > 0.0 && change >= 0.0) ? (float)pow(10.0, change / 20.0) : 1.0f;
  }
  __syncthreads();
  const float g = gain_s;
  for (int i = tid; i < L; i += 256) out[(int64_t)b * L + i] = x[i] * g;
}

__global__ void round_f16_rows_kernel(const float* __restrict__ x, int64_t ldx, int Lx, const int32_t* __restrict__ n_in, const int32_t* __restrict__ n_out,
                                      float* __restrict__ y, int64_t ldy, int B) {
  const int64_t total = (int64_t)B * ldy;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / ldy);
    const int64_t t = i % ldy;
    float v = 0.f;
    // samples past the item's own length are padding whatever the batch buffer holds there (MelFrontendHIP.wav2mel masks the same samples)
    if (t < n_out[b] && t < (n_in ? min(n_in[b], Lx) : Lx)) v = (float)(_Float16)x[(int64_t)b * ldx + t];   // fp32 -> fp16 rounds to nearest even (v_cvt_f16_f32)
    y[i] = v;
  }
}

}  // namespace

extern "C" int ss_normalize_volume(const float* wav, const int32_t* lens, float* out, int B, int L, float target_dbfs, void* stream) {
  SS_CHECK_ARG(wav && out && B > 0 && L > 0, "ss_normalize_volume: bad arguments");
  hipLaunchKernelGGL(normalize_volume_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, wav, lens, out, L, target_dbfs);
  SS_CHECK_LAUNCH("normalize_volume_kernel");
  return SS_OK;
}

extern "C" int ss_round_f16_rows(const float* x, int64_t ldx, int Lx, const int32_t* n_in, const int32_t* n_out, float* y, int64_t ldy, int B, void* stream) {
  SS_CHECK_ARG(x && n_out && y && B > 0 && Lx > 0 && ldx >= Lx && ldy > 0, "ss_round_f16_rows: bad arguments");
  const int64_t total = (int64_t)B * ldy;
  const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(round_f16_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, Lx, n_in, n_out, y, ldy, B);
  SS_CHECK_LAUNCH("round_f16_rows_kernel");
  return SS_OK;
}
