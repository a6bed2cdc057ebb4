// f0 tracker of the reference's preprocessing: inference/StyleSinger.py:125-127 calls
//   parselmouth.Sound(wav, sr).to_pitch_ac(time_step, voicing_threshold=0.6, pitch_floor=80, pitch_ceiling=800).selected_array['frequency']
// parselmouth / Praat are UN-VENDORED (requirements.txt: praat-parselmouth==0.3.3): this restates the PUBLISHED algorithm - P. Boersma (1993),
// "Accurate short-term analysis of the fundamental frequency and the harmonics-to-noise ratio of a sampled sound" - with a Hanning window of
// three floor periods, as oracle/praat_pitch.py does on the CPU (parity UNPINNED; pinned by analytic known answers only).
//
// All arithmetic in float64 (the paper's quantities are ratios of nearly equal sums; MI355X has full-rate fp64 vector units and the whole
// tracker is < 1 % of a forward). Five launches per batch of waveforms, no host round trip:
//   stats       per item: mean, global peak |x - mean|                                   (workgroup per item)
//   autocorr    per frame: local mean, Hanning window, local peak, r(k) = sum_j f_j f_{j+k} for k <= nsamp_window / 2 by direct summation
//               from LDS (the paper goes through an FFT of the zero-padded frame: the same numbers), / r(0) / window autocorrelation
//   candidates  per frame: local maxima above half the voicing threshold, parabolic lag + windowed-sinc strength (depth 30), <= 14 kept
//   refine      per candidate: Brent maximisation of the depth-70 sinc interpolation over [lag - 1, lag + 1]; strengths > 1 reflected
//   viterbi     per item: path over candidates + the unvoiced one with octave / octave-jump / voiced-unvoiced costs; writes the contour
//               at its place on the mel frame grid (the reference's left pad of 2 * pad_size frames, zeros elsewhere)
#include "common.h"
#include "../../include/stylesinger_hip.h"

namespace {

constexpr int F0T_MAXC = 15;   // max_number_of_candidates of to_pitch_ac, the unvoiced candidate included
constexpr double F0T_PI = 3.14159265358979323846;

__global__ __launch_bounds__(256) void f0t_stats_kernel(const float* __restrict__ wav, const int32_t* __restrict__ n_samples, int64_t stride,
                                                        double* __restrict__ gpeak) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = wav + (int64_t)b * stride;
  const int n = n_samples[b];
  double s = 0.0;
  for (int i = tid; i < n; i += 256) s += (double)x[i];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double mean = n > 0 ? red[0] / n : 0.0;
  __syncthreads();
  double m = 0.0;
  for (int i = tid; i < n; i += 256) m = fmax(m, fabs((double)x[i] - mean));
  red[tid] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
    __syncthreads();
  }
  if (tid == 0) gpeak[b] = red[0];
}

// one workgroup per (frame, item). LDS: the windowed frame as doubles.
__global__ __launch_bounds__(256) void f0t_autocorr_kernel(const float* __restrict__ wav, const int32_t* __restrict__ n_frames,
                                                           const int32_t* __restrict__ left0, int64_t stride, const double* __restrict__ window,
                                                           const double* __restrict__ window_r, const double* __restrict__ gpeak, int hop, int nw,
                                                           int halfw, int nper, int halfper, int nlag, int max_frames, double* __restrict__ R,
                                                           double* __restrict__ intensity) {
  extern __shared__ __attribute__((aligned(16))) double f0t_smem[];
  double* f = f0t_smem;          // [nw]
  double* red = f0t_smem + nw;   // [256]
  const int b = blockIdx.y, i = blockIdx.x, tid = threadIdx.x;
  if (i >= n_frames[b] || i >= max_frames) return;
  const float* x = wav + (int64_t)b * stride;
  const int64_t row = (int64_t)b * max_frames + i;
  double* r = R + row * (nlag + 1);
  const double gp = gpeak[b];
  if (gp == 0.0) {   // a silent item: every frame keeps only its unvoiced candidate
    if (tid == 0) intensity[row] = 0.0;
    return;
  }
  const int right = left0[b] + 1 + i * hop;   // 0-based index of the sample right of the frame centre
  const int ms = right - nper, ws = right - halfw;
  double s = 0.0;
  for (int j = tid; j < 2 * nper; j += 256) s += (double)x[ms + j];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double mean = red[0] / (2 * nper);
  __syncthreads();
  for (int j = tid; j < nw; j += 256) f[j] = ((double)x[ws + j] - mean) * window[j];
  __syncthreads();
  // local peak: half a floor period to both sides of the centre, on the windowed frame
  int lo = halfw + 1 - halfper, hi = halfw + halfper;   // 1-based, inclusive
  lo = lo < 1 ? 1 : lo;
  hi = hi > nw ? nw : hi;
  double m = 0.0;
  for (int j = lo - 1 + tid; j < hi; j += 256) m = fmax(m, fabs(f[j]));
  red[tid] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
    __syncthreads();
  }
  const double lpeak = red[0];
  __syncthreads();
  if (tid == 0) intensity[row] = lpeak > gp ? 1.0 : lpeak / gp;
  if (lpeak == 0.0) {
    if (tid == 0) intensity[row] = -1.0;   // "local peak 0": no voiced candidates for this frame (the value itself is then unused: 0 / gp)
    return;
  }
  // lags tid, tid + 256, tid + 512, tid + 768 (<= nlag): one broadcast read of f[j] feeds four products
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int k0 = tid, k1 = tid + 256, k2 = tid + 512, k3 = tid + 768;
  for (int j = 0; j < nw; ++j) {
    const double fj = f[j];
    if (j + k0 < nw) a0 = fma(fj, f[j + k0], a0);
    if (j + k1 < nw) a1 = fma(fj, f[j + k1], a1);
    if (j + k2 < nw) a2 = fma(fj, f[j + k2], a2);
    if (j + k3 < nw) a3 = fma(fj, f[j + k3], a3);
  }
  if (tid == 0) red[0] = a0;
  __syncthreads();
  const double r0 = red[0];
  if (k0 <= nlag) r[k0] = k0 == 0 ? 1.0 : a0 / (r0 * window_r[k0]);
  if (k1 <= nlag) r[k1] = a1 / (r0 * window_r[k1]);
  if (k2 <= nlag) r[k2] = a2 / (r0 * window_r[k2]);
  if (k3 <= nlag) r[k3] = a3 / (r0 * window_r[k3]);
}

// windowed-sinc interpolation of the symmetric sequence y(pos) = r[|pos - off|], pos = 1 .. 2 nlag + 1, at the real position x
__device__ double f0t_sinc(const double* __restrict__ r, int nlag, double x, int depth) {
  const int nx = 2 * nlag + 1, off = nlag + 1;
  auto y = [&](int pos) { const int k = pos - off; return r[k < 0 ? -k : k]; };
  const int midleft = (int)floor(x), midright = midleft + 1;
  if (x > nx) return y(nx);
  if (x < 1) return y(1);
  if (x == (double)midleft) return y(midleft);
  if (depth > midright - 1) depth = midright - 1;
  if (depth > nx - midleft) depth = nx - midleft;
  if (depth <= 0) return y((int)floor(x + 0.5));
  if (depth == 1) return y(midleft) + (x - midleft) * (y(midright) - y(midleft));
  const int left = midright - depth, right = midleft + depth;
  double res = 0.0;
  double a = F0T_PI * (x - midleft), halfsina = 0.5 * sin(a), aa = a / (x - left + 1), daa = F0T_PI / (x - left + 1);
  for (int ix = midleft; ix >= left; --ix) {
    res += y(ix) * (halfsina / a * (1.0 + cos(aa)));
    a += F0T_PI;
    aa += daa;
    halfsina = -halfsina;
  }
  a = F0T_PI * (midright - x);
  halfsina = 0.5 * sin(a);
  aa = a / (right - x + 1);
  daa = F0T_PI / (right - x + 1);
  for (int ix = midright; ix <= right; ++ix) {
    res += y(ix) * (halfsina / a * (1.0 + cos(aa)));
    a += F0T_PI;
    aa += daa;
    halfsina = -halfsina;
  }
  return res;
}

// thread per frame: the candidate list (slot 0 = unvoiced)
__global__ __launch_bounds__(64) void f0t_candidates_kernel(const double* __restrict__ R, const double* __restrict__ intensity,
                                                            const int32_t* __restrict__ n_frames, int max_frames, int B, int nlag, int maximum_lag,
                                                            double sr, double floor_hz, double voicing_threshold, double octave_cost,
                                                            double* __restrict__ cand_f, double* __restrict__ cand_s, int32_t* __restrict__ cand_i,
                                                            int32_t* __restrict__ n_cand) {
  const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (row >= (int64_t)B * max_frames) return;
  const int b = (int)(row / max_frames), i = (int)(row % max_frames);
  if (i >= n_frames[b]) return;
  double* cf = cand_f + row * F0T_MAXC;
  double* cs = cand_s + row * F0T_MAXC;
  int32_t* ci = cand_i + row * F0T_MAXC;
  cf[0] = 0.0;
  cs[0] = 0.0;
  ci[0] = 0;
  int nc = 1;
  const double it = intensity[row];
  if (it > 0.0) {   // 0: silent item, -1: silent frame
    const double* r = R + row * (nlag + 1);
    const int lim = maximum_lag < nlag ? maximum_lag : nlag;
    for (int k = 2; k < lim; ++k) {
      const double rk = r[k], rm = r[k - 1], rp = r[k + 1];
      if (rk > 0.5 * voicing_threshold && rk > rm && rk >= rp) {
        const double dr = 0.5 * (rp - rm), d2r = 2.0 * rk - rm - rp;
        const double fm = sr / (k + dr / d2r);
        double sm = f0t_sinc(r, nlag, sr / fm + (nlag + 1), 30);
        if (sm > 1.0) sm = 1.0 / sm;
        int place = 0;
        if (nc < F0T_MAXC) place = nc++;
        else {
          double weakest = 2.0;
          for (int w = 1; w < F0T_MAXC; ++w) {
            const double local = cs[w] - octave_cost * log2(floor_hz / cf[w]);
            if (local < weakest) {
              weakest = local;
              place = w;
            }
          }
          if (sm - octave_cost * log2(floor_hz / fm) <= weakest) place = 0;
        }
        if (place) {
          cf[place] = fm;
          cs[place] = sm;
          ci[place] = k;
        }
      }
    }
  }
  n_cand[row] = nc;
}

// thread per (frame, voiced candidate): Brent's minimiser of -sinc over [lag - 1, lag + 1] (tol 1e-10, <= 60 iterations)
__global__ __launch_bounds__(64) void f0t_refine_kernel(const double* __restrict__ R, const int32_t* __restrict__ n_frames, int max_frames, int B,
                                                        int nlag, double sr, double* __restrict__ cand_f, double* __restrict__ cand_s,
                                                        const int32_t* __restrict__ cand_i, const int32_t* __restrict__ n_cand) {
  const int64_t idx = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t row = idx / (F0T_MAXC - 1);
  const int c = (int)(idx % (F0T_MAXC - 1)) + 1;
  if (row >= (int64_t)B * max_frames) return;
  const int b = (int)(row / max_frames), i = (int)(row % max_frames);
  if (i >= n_frames[b] || c >= n_cand[row]) return;
  const double* r = R + row * (nlag + 1);
  const int off = nlag + 1;
  const int depth = cand_f[row * F0T_MAXC + c] > 0.3 * sr ? 700 : 70;
  auto fn = [&](double x) { return -f0t_sinc(r, nlag, x, depth); };
  double a = cand_i[row * F0T_MAXC + c] + off - 1, bb = a + 2.0;
  const double golden = 1.0 - 0.6180339887498948482045868343656381177203, sqrt_eps = 1.4901161193847656e-08, tol = 1e-10;
  double v = a + golden * (bb - a), fv = fn(v), x = v, w = v, fx = fv, fw = fv;
  for (int iter = 0; iter < 60; ++iter) {
    const double middle = 0.5 * (a + bb), tol_act = sqrt_eps * fabs(x) + tol / 3.0;
    if (fabs(x - middle) + 0.5 * (bb - a) <= 2.0 * tol_act) break;
    double new_step = golden * (x >= middle ? a - x : bb - x);
    if (fabs(x - w) >= tol_act) {
      const double t = (x - w) * (fx - fv);
      double q = (x - v) * (fx - fw);
      double p = (x - v) * q - (x - w) * t;
      q = 2.0 * (q - t);
      if (q > 0.0) p = -p;
      else q = -q;
      if (fabs(p) < fabs(new_step * q) && p > q * (a - x + 2.0 * tol_act) && p < q * (bb - x - 2.0 * tol_act)) new_step = p / q;
    }
    if (fabs(new_step) < tol_act) new_step = new_step > 0.0 ? tol_act : -tol_act;
    const double t = x + new_step, ft = fn(t);
    if (ft <= fx) {
      if (t < x) bb = x;
      else a = x;
      v = w; w = x; x = t;
      fv = fw; fw = fx; fx = ft;
    } else {
      if (t < x) a = t;
      else bb = t;
      if (ft <= fw || w == x) {
        v = w; w = t;
        fv = fw; fw = ft;
      } else if (ft <= fv || v == x || v == w) {
        v = t;
        fv = ft;
      }
    }
  }
  const double ymid = -fx;
  cand_f[row * F0T_MAXC + c] = sr / (x - off);
  cand_s[row * F0T_MAXC + c] = ymid > 1.0 ? 1.0 / ymid : ymid;
}

// one wave per item: lane c2 owns candidate c2 of the current frame
__global__ __launch_bounds__(64) void f0t_viterbi_kernel(const double* __restrict__ cand_f, const double* __restrict__ cand_s,
                                                         const int32_t* __restrict__ n_cand, const double* __restrict__ intensity,
                                                         const int32_t* __restrict__ n_frames, int max_frames, double ceiling, double time_step,
                                                         double voicing_threshold, double silence_threshold, double octave_cost,
                                                         double octave_jump_cost, double voiced_unvoiced_cost, uint8_t* __restrict__ psi,
                                                         float* __restrict__ f0_out, int ld_out, int lpad) {
  __shared__ double dprev[F0T_MAXC], fprev[F0T_MAXC], dcur[F0T_MAXC];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = n_frames[b] < max_frames ? n_frames[b] : max_frames;
  float* out = f0_out + (int64_t)b * ld_out;
  for (int t = lane; t < ld_out; t += 64) out[t] = 0.f;
  if (n <= 0) return;
  const double corr = 0.01 / time_step, ojc = octave_jump_cost * corr, vuc = voiced_unvoiced_cost * corr;
  int nprev = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t row = (int64_t)b * max_frames + i;
    int nc = n_cand[row];
    nc = nc < 1 ? 1 : (nc > F0T_MAXC ? F0T_MAXC : nc);
    double f2 = 0.0, d2 = 0.0;
    bool v2 = false;
    if (lane < nc) {
      f2 = cand_f[row * F0T_MAXC + lane];
      v2 = f2 > 0.0 && f2 < ceiling;
      const double it = fmax(intensity[row], 0.0);
      double unv = silence_threshold <= 0.0 ? 0.0 : 2.0 - it / (silence_threshold / (1.0 + voicing_threshold));
      unv = voicing_threshold + (unv > 0.0 ? unv : 0.0);
      d2 = v2 ? cand_s[row * F0T_MAXC + lane] - octave_cost * log2(ceiling / f2) : unv;
      if (i > 0) {
        double best = -1e30;
        int place = 0;
        for (int c1 = 0; c1 < nprev; ++c1) {
          const double f1 = fprev[c1];
          const bool v1 = f1 > 0.0 && f1 < ceiling;
          const double cost = !v2 ? (v1 ? vuc : 0.0) : (!v1 ? vuc : ojc * fabs(log2(f1 / f2)));
          const double val = dprev[c1] - cost + d2;
          if (val > best) {
            best = val;
            place = c1;
          }
        }
        d2 = best;
        psi[row * F0T_MAXC + lane] = (uint8_t)place;
      }
    }
    __syncthreads();   // everyone has read dprev / fprev
    if (lane < nc) {
      dprev[lane] = d2;
      fprev[lane] = f2;
      if (i == n - 1) dcur[lane] = d2;
    }
    nprev = nc;
    __syncthreads();
  }
  if (lane == 0) {
    int place = 0;
    double best = -1e30;
    for (int c = 0; c < nprev; ++c)
      if (dcur[c] > best) {
        best = dcur[c];
        place = c;
      }
    for (int i = n - 1; i >= 0; --i) {
      const int64_t row = (int64_t)b * max_frames + i;
      if (lpad + i < ld_out) out[lpad + i] = (float)cand_f[row * F0T_MAXC + place];
      if (i > 0) place = psi[row * F0T_MAXC + place];
    }
  }
}

}  // namespace

extern "C" int64_t ss_f0track_workspace_bytes(int B, int max_frames, int nlag) {
  if (B <= 0 || max_frames <= 0 || nlag <= 0) return 0;
  const int64_t rows = (int64_t)B * max_frames;
  // gpeak [B] | R [rows][nlag + 1] | intensity [rows] | cand_f, cand_s [rows][15] doubles | cand_i [rows][15], n_cand [rows] int32 | psi [rows][15] bytes
  return 8 * ((int64_t)B + rows * (nlag + 1) + rows + 2 * rows * F0T_MAXC) + 4 * (rows * F0T_MAXC + rows) + rows * 16 + 256;
}

extern "C" int ss_f0track(const float* wav, int64_t wav_stride, const int32_t* n_samples, const int32_t* n_frames, const int32_t* left0, int B,
                          int max_frames, const ss_f0track_params* prm, const double* window, const double* window_r, float* f0_out, int ld_out,
                          int lpad, void* workspace, int64_t workspace_bytes, void* stream_) {
  SS_CHECK_ARG(wav && n_samples && n_frames && left0 && prm && window && window_r && f0_out && workspace, "ss_f0track: null argument");
  const ss_f0track_params& p = *prm;
  SS_CHECK_ARG(B > 0 && max_frames > 0 && ld_out > 0 && lpad >= 0, "ss_f0track: bad dims B=%d max_frames=%d ld_out=%d lpad=%d", B, max_frames, ld_out, lpad);
  SS_CHECK_ARG(p.nsamp_window >= 8 && p.nsamp_window == 2 * p.halfnsamp_window && p.nlag >= 4 && p.nlag <= p.nsamp_window / 2 && p.nlag < 1024 &&
                   p.hop > 0 && p.nsamp_period > 0 && p.halfnsamp_period > 0 && p.maximum_lag >= 3,
               "ss_f0track: bad geometry (window %d, nlag %d, hop %d)", p.nsamp_window, p.nlag, p.hop);
  SS_CHECK_ARG(p.sample_rate > 0 && p.pitch_floor > 0 && p.pitch_ceiling > p.pitch_floor && p.time_step > 0 && p.voicing_threshold > 0,
               "ss_f0track: bad analysis parameters");
  SS_CHECK_ARG(workspace_bytes >= ss_f0track_workspace_bytes(B, max_frames, p.nlag), "ss_f0track: workspace too small (%lld < %lld bytes)",
               (long long)workspace_bytes, (long long)ss_f0track_workspace_bytes(B, max_frames, p.nlag));
  SS_CHECK_ARG((((uintptr_t)workspace) & 7) == 0, "ss_f0track: workspace must be 8-byte aligned");
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t rows = (int64_t)B * max_frames;
  double* gpeak = (double*)workspace;
  double* R = gpeak + B;
  double* intensity = R + rows * (p.nlag + 1);
  double* cand_f = intensity + rows;
  double* cand_s = cand_f + rows * F0T_MAXC;
  int32_t* cand_i = (int32_t*)(cand_s + rows * F0T_MAXC);
  int32_t* n_cand = cand_i + rows * F0T_MAXC;
  uint8_t* psi = (uint8_t*)(n_cand + rows);
  hipLaunchKernelGGL(f0t_stats_kernel, dim3(B), dim3(256), 0, stream, wav, n_samples, wav_stride, gpeak);
  SS_CHECK_LAUNCH("f0t_stats_kernel");
  const size_t lds = (size_t)(p.nsamp_window + 256) * sizeof(double);
  SS_CHECK_ARG(lds <= 64 * 1024, "ss_f0track: analysis window of %d samples does not fit the LDS plan", p.nsamp_window);
  hipLaunchKernelGGL(f0t_autocorr_kernel, dim3(max_frames, B), dim3(256), lds, stream, wav, n_frames, left0, wav_stride, window, window_r, gpeak, p.hop,
                     p.nsamp_window, p.halfnsamp_window, p.nsamp_period, p.halfnsamp_period, p.nlag, max_frames, R, intensity);
  SS_CHECK_LAUNCH("f0t_autocorr_kernel");
  hipLaunchKernelGGL(f0t_candidates_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, stream, R, intensity, n_frames, max_frames, B, p.nlag,
                     p.maximum_lag, p.sample_rate, p.pitch_floor, p.voicing_threshold, p.octave_cost, cand_f, cand_s, cand_i, n_cand);
  SS_CHECK_LAUNCH("f0t_candidates_kernel");
  const int64_t nref = rows * (F0T_MAXC - 1);
  hipLaunchKernelGGL(f0t_refine_kernel, dim3((unsigned)((nref + 63) / 64)), dim3(64), 0, stream, R, n_frames, max_frames, B, p.nlag, p.sample_rate, cand_f,
                     cand_s, cand_i, n_cand);
  SS_CHECK_LAUNCH("f0t_refine_kernel");
  hipLaunchKernelGGL(f0t_viterbi_kernel, dim3(B), dim3(64), 0, stream, cand_f, cand_s, n_cand, intensity, n_frames, max_frames,
                     fmin(p.pitch_ceiling, 0.5 * p.sample_rate), p.time_step, p.voicing_threshold, p.silence_threshold, p.octave_cost,
                     p.octave_jump_cost, p.voiced_unvoiced_cost, psi, f0_out, ld_out, lpad);
  SS_CHECK_LAUNCH("f0t_viterbi_kernel");
  return SS_OK;
}
