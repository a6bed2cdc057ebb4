// Flash-style multi-head attention core in exact fp32 on the matrix cores (gfx950).
//
//   O[b,q,h,:] = softmax_k( scale * Q[b,q,h,:] . K[b,k,h,:]  (k < klens[b]) ) V[b,k,h,:]      D = 128
//
// Layout trick (MI355X-first): the wave computes the TRANSPOSED score tile S^T = K Q^T with
// v_mfma_f32_32x32x2_f32, so that in the MFMA C/D layout each LANE owns one query column
// (lane&31) and its 16 registers are 16 keys.  Then
//   * the softmax reductions over keys are in-register (+ one lane^32 exchange) — no LDS, no
//     cross-lane butterfly per row;
//   * P^T is ALREADY in the B-operand layout of the second MFMA (O^T += V^T P^T): B wants
//     lane (q, h) to hold P^T[key(h,step)][q], which is exactly accumulator register `step` of
//     that lane when the contraction order over keys is key(h,r) = (r&3)+8(r>>2)+4h.  The sum over
//     keys does not care about the order, so no data movement at all between the two GEMMs;
//   * the running rescale exp(m_old-m_new) is a per-lane scalar on the O^T accumulators.
// The [T,T] score matrix is never materialised (needed for T=5625: 2*B*127 MB).
//
// One block = 4 waves = 128 queries of one (b, head); K/V tiles of 32 keys staged in LDS.
#include "common.h"
#include "../../include/stylesinger_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int D = 128;
constexpr int KT = 32;            // keys per tile
constexpr int K_LD = D + 4;       // floats: 528-B rows, 16-B aligned, conflict-free b128 column reads
constexpr int V_LD = D + 4;

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                        const float* __restrict__ V, float* __restrict__ O, int B, int H,
                                                        int Tq, int Tk, int ldq, int ldk, int ldv, int ldo, int64_t qbs,
                                                        int64_t kbs, int64_t vbs, int64_t obs,
                                                        const int32_t* __restrict__ qlens,
                                                        const int32_t* __restrict__ klens, float scale, int q_tiles) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * K_LD];
  __shared__ __attribute__((aligned(16))) float Vs[KT * V_LD];

  const int bh = blockIdx.x / q_tiles;
  const int qt = blockIdx.x % q_tiles;
  const int b = bh / H, h = bh % H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int qlen = qlens ? min(qlens[b], Tq) : Tq;
  const int klen = klens ? min(klens[b], Tk) : Tk;
  const int q0 = qt * 128 + wave * 32;
  if (qt * 128 >= qlen) return;  // whole block past the valid queries (uniform)

  // ---- Q fragment (B operand of S^T = K Q^T): lane (q=l31, lh) holds Q[q][8g + 4lh + s] * scale ----
  const int q = q0 + l31;
  const bool q_ok = q < qlen;
  const float* Qr = Q + (int64_t)b * qbs + (int64_t)(q_ok ? q : 0) * ldq + h * D;
  float4 qf[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    float4 v = q_ok ? *reinterpret_cast<const float4*>(Qr + g * 8 + lh * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    qf[g] = v;
  }

  f32x16 o_acc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* Kb = K + (int64_t)b * kbs + h * D;
  const float* Vb = V + (int64_t)b * vbs + h * D;
  const int n_ktiles = (klen + KT - 1) / KT;

  for (int kt = 0; kt < n_ktiles; ++kt) {
    const int k0 = kt * KT;
    __syncthreads();  // previous tile fully consumed
    // ---- stage K and V tiles: 32 rows x 128 floats each = 1024 float4 per tensor, 4 per thread ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + i * 256;
      const int row = f >> 5, c4 = f & 31;
      const int kr = k0 + row;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (kr < klen) {
        kv = *reinterpret_cast<const float4*>(Kb + (int64_t)kr * ldk + c4 * 4);
        vv = *reinterpret_cast<const float4*>(Vb + (int64_t)kr * ldv + c4 * 4);
      }
      *reinterpret_cast<float4*>(Ks + row * K_LD + c4 * 4) = kv;
      *reinterpret_cast<float4*>(Vs + row * V_LD + c4 * 4) = vv;
    }
    __syncthreads();

    // ---- S^T[key][q] = sum_d K[key][d] * Qs[q][d] ----
    f32x16 s_acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_acc[r] = 0.f;
    const float* Kl = Ks + l31 * K_LD + lh * 4;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float4 kf = *reinterpret_cast<const float4*>(Kl + g * 8);
      s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[g].x, s_acc, 0, 0, 0);
      s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[g].y, s_acc, 0, 0, 0);
      s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[g].z, s_acc, 0, 0, 0);
      s_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[g].w, s_acc, 0, 0, 0);
    }
    // ---- online softmax over keys: this lane's keys are key(lh, r) = (r&3) + 8*(r>>2) + 4*lh ----
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (key >= klen) s_acc[r] = -INFINITY;
      mx = fmaxf(mx, s_acc[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);  // finite: every staged tile has >= 1 valid key
    const float alpha = expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = expf(s_acc[r] - m_new);
      s_acc[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;

    // ---- O^T[dd][q] += sum_key V[key][dd] * P^T[key][q];  A = V^T: lane (dd=l31, lh) -> V[key(lh,r)][dblk*32+dd] ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float* Vl = Vs + key * V_LD + l31;
      const float p = s_acc[r];
#pragma unroll
      for (int d = 0; d < 4; ++d) o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vl[d * 32], p, o_acc[d], 0, 0, 0);
    }
  }

  // ---- epilogue: O[q][h*D + dblk*32 + (r&3)+8*(r>>2)+4*lh] = o_acc / l ----
  if (q_ok) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    float* Or = O + (int64_t)b * obs + (int64_t)q * ldo + h * D;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float4 v;
        v.x = o_acc[d][rq * 4 + 0] * inv;
        v.y = o_acc[d][rq * 4 + 1] * inv;
        v.z = o_acc[d][rq * 4 + 2] * inv;
        v.w = o_acc[d][rq * 4 + 3] * inv;
        *reinterpret_cast<float4*>(Or + d * 32 + 8 * rq + 4 * lh) = v;
      }
  }
}

}  // namespace

extern "C" int ss_attention(const float* Q, const float* K, const float* V, float* O, int B, int H, int Dh, int Tq, int Tk,
                            int ldq, int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                            const int32_t* qlens, const int32_t* klens, float scale, void* stream) {
  SS_CHECK_ARG(Q && K && V && O, "ss_attention: null pointer");
  SS_CHECK_ARG(Dh == D, "ss_attention: head dim %d unsupported (only 128)", Dh);
  SS_CHECK_ARG(B > 0 && H > 0 && Tq > 0 && Tk > 0, "ss_attention: bad dims");
  SS_CHECK_ARG(((ldq | ldk | ldv | ldo) & 3) == 0 && ((q_bs | k_bs | v_bs | o_bs) & 3) == 0,
               "ss_attention: strides must be multiples of 4 floats");
  const int q_tiles = (Tq + 127) / 128;
  hipLaunchKernelGGL(attention_kernel, dim3(B * H * q_tiles), dim3(256), 0, (hipStream_t)stream, Q, K, V, O, B, H, Tq, Tk,
                     ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, qlens, klens, scale, q_tiles);
  SS_CHECK_LAUNCH("ss_attention");
  return SS_OK;
}
