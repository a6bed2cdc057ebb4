// Residual-quantisation codebook lookup of the Residual Style Adaptor
// (modules/StyleSinger/RQ.py: VQEmbedding.compute_distances :30-47, find_nearest_embedding :49-55,
//  RQBottleneck.quantize :226-260, forward :262-270).
//
//   r <- x ; agg <- 0
//   for d in 0..depth-1:  dist_c = (|r|^2 + |c|^2) - 2 r.c  over the n_embed real codes (pad row excluded)
//                         k = argmin_c dist_c (first minimum) ; q = C_d[k] ; r -= q ; agg += q
//   out = x + (agg - x)
//
// One 128-thread block (2 waves) owns ROWS rows: thread c owns code c and streams its 256-float code
// row once per depth while the ROWS residual rows sit in LDS (broadcast reads).  The argmin is a
// wavefront (dist, idx) min-reduction with the smaller index winning ties == torch.argmin.
#include "common.h"
#include "../../include/stylesinger_hip.h"

namespace {

constexpr int ROWS = 8;

template <int C>
__global__ __launch_bounds__(128) void rq_lookup_kernel(const float* __restrict__ x, const float* __restrict__ codebooks,
                                                        float* __restrict__ out, int64_t* __restrict__ codes, int rows,
                                                        int n_embed, int depth) {
  __shared__ __attribute__((aligned(16))) float res[ROWS][C];
  __shared__ float agg[ROWS][C];
  __shared__ float rnorm[ROWS];
  __shared__ float wmin_d[2][ROWS];
  __shared__ int wmin_i[2][ROWS];
  __shared__ int best[ROWS];

  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * ROWS;
  for (int i = tid; i < ROWS * C; i += 128) {
    const int r = i / C, c = i % C;
    res[r][c] = (r0 + r < rows) ? x[(int64_t)(r0 + r) * C + c] : 0.f;
    agg[r][c] = 0.f;
  }
  __syncthreads();

  for (int d = 0; d < depth; ++d) {
    const float* cb = codebooks + (int64_t)d * (n_embed + 1) * C;
    // |r|^2 per row (wave 0 handles rows 0..3, wave 1 rows 4..7)
    {
      const int w = tid >> 6, lane = tid & 63;
      for (int r = w * (ROWS / 2); r < (w + 1) * (ROWS / 2); ++r) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += res[r][c] * res[r][c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) rnorm[r] = s;
      }
    }
    __syncthreads();
    float dist[ROWS];
    if (tid < n_embed) {
      float dot[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) dot[r] = 0.f;
      float cn = 0.f;
      const float4* cr = reinterpret_cast<const float4*>(cb + (int64_t)tid * C);
      for (int k4 = 0; k4 < C / 4; ++k4) {
        const float4 cv = cr[k4];
        cn += cv.x * cv.x + cv.y * cv.y + cv.z * cv.z + cv.w * cv.w;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const float4 rv = *reinterpret_cast<const float4*>(&res[r][k4 * 4]);
          dot[r] += rv.x * cv.x + rv.y * cv.y + rv.z * cv.z + rv.w * cv.w;
        }
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) dist[r] = (rnorm[r] + cn) - 2.0f * dot[r];
    } else {
#pragma unroll
      for (int r = 0; r < ROWS; ++r) dist[r] = INFINITY;
    }
    // argmin over codes with first-min tie-break
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float dv = dist[r];
      int di = tid;
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(dv, o);
        const int oi = __shfl_xor(di, o);
        if (ov < dv || (ov == dv && oi < di)) { dv = ov; di = oi; }
      }
      if ((tid & 63) == 0) { wmin_d[tid >> 6][r] = dv; wmin_i[tid >> 6][r] = di; }
    }
    __syncthreads();
    if (tid < ROWS) {
      const float d0 = wmin_d[0][tid], d1 = wmin_d[1][tid];
      const int i0 = wmin_i[0][tid], i1 = wmin_i[1][tid];
      const int k = (d1 < d0 || (d1 == d0 && i1 < i0)) ? i1 : i0;
      best[tid] = k;
      if (codes && r0 + tid < rows) codes[(int64_t)(r0 + tid) * depth + d] = k;
    }
    __syncthreads();
    for (int i = tid; i < ROWS * C; i += 128) {
      const int r = i / C, c = i % C;
      const float qv = cb[(int64_t)best[r] * C + c];
      res[r][c] -= qv;
      agg[r][c] += qv;
    }
    __syncthreads();
  }
  for (int i = tid; i < ROWS * C; i += 128) {
    const int r = i / C, c = i % C;
    if (r0 + r < rows) {
      const float xv = x[(int64_t)(r0 + r) * C + c];
      out[(int64_t)(r0 + r) * C + c] = xv + (agg[r][c] - xv);
    }
  }
}

}  // namespace

extern "C" int ss_rq_lookup(const float* x, const float* codebooks, float* out, int64_t* codes, int rows, int C,
                            int n_embed, int depth, void* stream) {
  SS_CHECK_ARG(x && codebooks && out, "ss_rq_lookup: null pointer");
  SS_CHECK_ARG(C == 256, "ss_rq_lookup: only C=256 (got %d)", C);
  SS_CHECK_ARG(n_embed > 0 && n_embed <= 128, "ss_rq_lookup: n_embed=%d must be in 1..128", n_embed);
  SS_CHECK_ARG(rows > 0 && depth > 0, "ss_rq_lookup: bad dims");
  hipLaunchKernelGGL(rq_lookup_kernel<256>, dim3((rows + ROWS - 1) / ROWS), dim3(128), 0, (hipStream_t)stream, x,
                     codebooks, out, codes, rows, n_embed, depth);
  SS_CHECK_LAUNCH("ss_rq_lookup");
  return SS_OK;
}
