// One-time weight relayout for the MFMA conv kernel (device -> device).
//   torch conv1d weight [Cout][Cin][k]  ->  [Np][k][Kp]   (K-contiguous per output column, zero padded)
// plus weight-norm folding and the 32-row gate interleave.  See include/stylesinger_hip.h.
#include "common.h"
#include "../../include/stylesinger_hip.h"

namespace {

__device__ __forceinline__ int packed_to_orig(int np, int n, int interleave_half) {
  if (interleave_half > 0) {
    const int nb = np >> 5, jj = np & 31;
    const int p = nb >> 1, which = nb & 1;
    const int c = p * 32 + jj;
    return (c < interleave_half) ? which * interleave_half + c : -1;
  }
  return np < n ? np : -1;
}

// scale[r] = g[r] / ||v[r,:]||  (torch.nn.utils.weight_norm with dim=0)
__global__ void weight_norm_scale_kernel(const float* __restrict__ v, const float* __restrict__ g, float* scale, int rows,
                                         int cols) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int i = threadIdx.x; i < cols; i += blockDim.x) {
    const float x = v[(int64_t)r * cols + i];
    s += x * x;
  }
  __shared__ float red[4];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    scale[r] = g[r] / sqrtf(t);
  }
}

__global__ void pack_conv_kernel(const float* __restrict__ src, const float* __restrict__ scale0, float* __restrict__ dst,
                                 int Cout, int Cin, int k, int Np, int Kp, int interleave_half, float row_scale) {
  const int64_t total = (int64_t)Np * k * Kp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Kp);
    const int j = (int)((i / Kp) % k);
    const int np = (int)(i / ((int64_t)Kp * k));
    const int o = packed_to_orig(np, Cout, interleave_half);
    float v = 0.f;
    if (o >= 0 && ci < Cin) {
      v = src[((int64_t)o * Cin + ci) * k + j];
      if (scale0) v *= scale0[o];
      v *= row_scale;
    }
    dst[i] = v;
  }
}

// ConvTranspose1d weight [Cin][Cout][k], stride u, pad=(k-u)/2, k==2u. scale0 is per Cin (weight_norm dim=0).
__global__ void pack_convtr_kernel(const float* __restrict__ src, const float* __restrict__ scale0, float* __restrict__ dst,
                                   int Cin, int Cout, int k, int u, int group, int Np, int Kp) {
  const int pad = (k - u) / 2;
  const int nph0 = u - pad;
  const int nph = group == 0 ? nph0 : u - nph0;
  const int64_t total = (int64_t)Np * 2 * Kp;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Kp);
    const int tap = (int)((i / Kp) % 2);
    const int np = (int)(i / ((int64_t)Kp * 2));
    float v = 0.f;
    if (np < nph * Cout && ci < Cin) {
      const int pl = np / Cout, co = np % Cout;
      const int p = group == 0 ? pl : pl + nph0;
      // group 0: tap0 <-> x[t]   kernel index p+pad     ; tap1 <-> x[t-1] kernel index p+pad+u
      // group 1: tap0 <-> x[t+1] kernel index p+pad-u   ; tap1 <-> x[t]   kernel index p+pad
      const int j = group == 0 ? (tap == 0 ? p + pad : p + pad + u) : (tap == 0 ? p + pad - u : p + pad);
      v = src[((int64_t)ci * Cout + co) * k + j];
      if (scale0) v *= scale0[ci];
    }
    dst[i] = v;
  }
}

__global__ void pack_bias_kernel(const float* __restrict__ src, const float* __restrict__ src2, float* __restrict__ dst,
                                 int n, int Np, int interleave_half, int repeat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Np) return;
  float v = 0.f;
  if (repeat > 1) {
    if (i < repeat * n) v = src[i % n] + (src2 ? src2[i % n] : 0.f);
  } else {
    const int o = packed_to_orig(i, n, interleave_half);
    if (o >= 0) v = src[o] + (src2 ? src2[o] : 0.f);
  }
  dst[i] = v;
}

}  // namespace

extern "C" int ss_weight_norm_scale(const float* v, const float* g, float* scale, int rows, int cols, void* stream) {
  SS_CHECK_ARG(v && g && scale && rows > 0 && cols > 0, "ss_weight_norm_scale: bad args");
  hipLaunchKernelGGL(weight_norm_scale_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, scale, rows, cols);
  SS_CHECK_LAUNCH("ss_weight_norm_scale");
  return SS_OK;
}

extern "C" int ss_pack_conv_weight(const float* src, const float* scale0, float* dst, int Cout, int Cin, int k, int Np,
                                   int Kp, int interleave_half, float row_scale, void* stream) {
  SS_CHECK_ARG(src && dst, "ss_pack_conv_weight: null pointer");
  SS_CHECK_ARG(Cout > 0 && Cin > 0 && k > 0 && (Np & 31) == 0 && (Kp & 31) == 0 && Kp >= Cin,
               "ss_pack_conv_weight: bad dims Cout=%d Cin=%d k=%d Np=%d Kp=%d", Cout, Cin, k, Np, Kp);
  if (interleave_half > 0)
    SS_CHECK_ARG(Np >= 2 * ((interleave_half + 31) / 32) * 32 && 2 * interleave_half == Cout,
                 "ss_pack_conv_weight: interleave_half=%d inconsistent with Cout=%d Np=%d", interleave_half, Cout, Np);
  else
    SS_CHECK_ARG(Np >= Cout, "ss_pack_conv_weight: Np=%d < Cout=%d", Np, Cout);
  const int64_t total = (int64_t)Np * k * Kp;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, scale0, dst, Cout, Cin, k, Np,
                     Kp, interleave_half, row_scale);
  SS_CHECK_LAUNCH("ss_pack_conv_weight");
  return SS_OK;
}

extern "C" int ss_pack_convtr_weight(const float* src, const float* scale0, float* dst, int Cin, int Cout, int k, int u,
                                     int group, int Np, int Kp, void* stream) {
  SS_CHECK_ARG(src && dst, "ss_pack_convtr_weight: null pointer");
  SS_CHECK_ARG(k == 2 * u && (u % 2) == 0, "ss_pack_convtr_weight: only k == 2*stride, even stride (k=%d u=%d)", k, u);
  const int pad = (k - u) / 2, nph = group == 0 ? u - pad : pad;
  SS_CHECK_ARG((Np & 31) == 0 && Np >= nph * Cout && (Kp & 31) == 0 && Kp >= Cin, "ss_pack_convtr_weight: bad Np/Kp");
  const int64_t total = (int64_t)Np * 2 * Kp;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_convtr_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, scale0, dst, Cin, Cout, k, u,
                     group, Np, Kp);
  SS_CHECK_LAUNCH("ss_pack_convtr_weight");
  return SS_OK;
}

extern "C" int ss_pack_bias(const float* src, const float* src2, float* dst, int n, int Np, int interleave_half,
                            int repeat, void* stream) {
  SS_CHECK_ARG(src && dst && n > 0 && Np > 0, "ss_pack_bias: bad args");
  hipLaunchKernelGGL(pack_bias_kernel, dim3((Np + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, src2, dst, n, Np,
                     interleave_half, repeat);
  SS_CHECK_LAUNCH("ss_pack_bias");
  return SS_OK;
}
