// Emotion encoder recurrence (SURVEY.md §8f-1 "input producers on GPU"): the reference's 3-layer LSTM-256 over 40-mel
// partials (data_gen/tts/emotion/model.py:11-78, torch.nn.LSTM gate order i, f, g, o).
//
// Per layer the input contribution  x_t . W_ih^T + b_ih + b_hh  of ALL time steps is one fp32-MFMA GEMM (ss_conv_gemm,
// host side); what is left is the sequential part, run here as ONE persistent launch per layer: one workgroup per
// partial utterance walks the n time steps with h in LDS and c in registers.  Thread j owns hidden unit j and all four of
// its gates, so the cell update needs no exchange; the recurrent weights are packed [k][j][gate] so that every lane
// streams one 16-byte load per k (coalesced, L2 resident: 1 MB per layer) and the previous hidden state is an LDS
// broadcast.  Exact fp32 (k-ordered fmaf chain); transcendental functions = the libm forms torch's CPU kernels use.
#include "common.h"
#include "../../include/stylesinger_hip.h"

namespace {

template <int H>
__global__ __launch_bounds__(H) void lstm_layer_kernel(const float4* __restrict__ xproj,  // [P][n][H] x (i,f,g,o)
                                                       const float4* __restrict__ whh,    // [H (k)][H (j)] x (i,f,g,o)
                                                       float* __restrict__ h_seq,         // [P][n][H] or null
                                                       float* __restrict__ h_last,        // [P][H] or null
                                                       int n) {
  __shared__ float hs[2][H];
  const int p = blockIdx.x, j = threadIdx.x;
  hs[0][j] = 0.f;
  float c = 0.f, h = 0.f;
  __syncthreads();
  const float4* xp = xproj + (int64_t)p * n * H + j;
  for (int t = 0; t < n; ++t) {
    const float* hp = hs[t & 1];
    float4 acc = xp[(int64_t)t * H];
#pragma unroll 16
    for (int k = 0; k < H; ++k) {
      const float4 w = whh[k * H + j];
      const float hk = hp[k];
      acc.x = fmaf(w.x, hk, acc.x);
      acc.y = fmaf(w.y, hk, acc.y);
      acc.z = fmaf(w.z, hk, acc.z);
      acc.w = fmaf(w.w, hk, acc.w);
    }
    const float ig = ss_sigmoid(acc.x), fg = ss_sigmoid(acc.y), gg = tanhf(acc.z), og = ss_sigmoid(acc.w);
    c = fg * c + ig * gg;
    h = og * tanhf(c);
    hs[(t + 1) & 1][j] = h;
    if (h_seq) h_seq[((int64_t)p * n + t) * H + j] = h;
    __syncthreads();
  }
  if (h_last) h_last[(int64_t)p * H + j] = h;
}

// out[c] = mean_r x[r][c] (rows added in order, like numpy's axis-0 reduction), then L2-normalised over c
__global__ void mean_l2norm_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int C) {
  __shared__ float red[256];
  float loc = 0.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += x[(int64_t)r * C + c];
    s /= (float)rows;
    out[c] = s;
    loc += s * s;
  }
  red[threadIdx.x] = loc;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float nrm = sqrtf(red[0]);
  for (int c = threadIdx.x; c < C; c += blockDim.x) out[c] = out[c] / nrm;
}

__global__ void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int C) {
  const int r = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = x[(int64_t)r * C + c];
    s += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float nrm = sqrtf(s);
  for (int c = lane; c < C; c += 64) y[(int64_t)r * C + c] = x[(int64_t)r * C + c] / nrm;
}

}  // namespace

extern "C" int ss_lstm_layer(const float* xproj, const float* w_hh_packed, float* h_seq, float* h_last, int P, int n, int H,
                             void* stream) {
  SS_CHECK_ARG(xproj && w_hh_packed && (h_seq || h_last), "ss_lstm_layer: null pointer");
  SS_CHECK_ARG(P > 0 && n > 0, "ss_lstm_layer: bad dims P=%d n=%d", P, n);
  SS_CHECK_ARG(H == 256, "ss_lstm_layer: hidden size %d not built (the reference's model_hidden_size is 256)", H);
  hipLaunchKernelGGL(lstm_layer_kernel<256>, dim3(P), dim3(256), 0, (hipStream_t)stream, (const float4*)xproj,
                     (const float4*)w_hh_packed, h_seq, h_last, n);
  SS_CHECK_LAUNCH("ss_lstm_layer");
  return SS_OK;
}

extern "C" int ss_mean_l2norm(const float* x, float* out, int rows, int C, void* stream) {
  SS_CHECK_ARG(x && out && rows > 0 && C > 0, "ss_mean_l2norm: bad args");
  hipLaunchKernelGGL(mean_l2norm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, rows, C);
  SS_CHECK_LAUNCH("ss_mean_l2norm");
  return SS_OK;
}

extern "C" int ss_l2norm_rows(const float* x, float* y, int rows, int C, void* stream) {
  SS_CHECK_ARG(x && y && rows > 0 && C > 0, "ss_l2norm_rows: bad args");
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, rows, C);
  SS_CHECK_LAUNCH("ss_l2norm_rows");
  return SS_OK;
}
