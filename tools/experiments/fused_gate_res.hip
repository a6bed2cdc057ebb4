// EXPERIMENT (round 5, the review's "flag-based form"): the F(4,3) gate launch and the residual-half projection of ONE layer as ONE launch -
// a dataflow grid instead of two dependent kernels. Workgroups [0, n_gate) run the gate kernel's body unchanged (wino43_gate16_body), workgroups
// [n_gate, n_gate + n_res) the projection's (gemm16_res_body); a projection workgroup starts as soon as the gate workgroups of the row tile(s) it
// reads have published, not when the whole gate grid has drained:
//   gate workgroup, after its stores:  __syncthreads -> lane 0: release fence (agent) -> s_waitcnt vmcnt(0) -> relaxed atomic add on counter[row tile]
//   projection workgroup, first thing: lane 0: relaxed poll of its 1-2 counters (s_sleep between polls, BOUNDED: gives up and raises *error
//                                      after ~2^20 polls instead of hanging the box) -> acquire fence (agent) -> __syncthreads -> body
// (the valid forms of MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"; per-XCD L2s are not coherent, so the
// fences are agent scope). Progress: a waiting workgroup only waits for workgroups with LOWER ids of the same launch, which the dispatcher has
// started before it (observed in-order dispatch; with one such launch in flight the lowest unfinished workgroup can always run).
// Same bodies, same arithmetic, same order: the results are bit-identical to the two-launch form (tools/kbench.py --which fused checks it).
// What it prices: one kernel boundary (1.7-1.9 us) and the drain of the gate grid against a release per gate workgroup, a poll + acquire per
// projection workgroup, and the projection running with the gate's register / LDS footprint. Measured: DESIGN.md 7 (round 5).
// NOT part of libstylesinger_hip.so: an experiment lives under tools/ and is built into its own shared object by tools/kbench_fused.py
// (hipcc, a few seconds); the product library ships no kernel its default paths cannot reach.
#define SS_FUSED_TU 1
#include "../../stylesinger_amd/csrc/common.h"
#include "../../include/stylesinger_hip.h"
#include <stdarg.h>
#include <type_traits>

namespace fz_g {
#include "../../stylesinger_amd/csrc/wino43_gate16.hip"
}
namespace fz_r {
#include "../../stylesinger_amd/csrc/gemm16.hip"
}

// the product library's error sink, local to this object (common.h declares it; the SS_CHECK_* macros call it)
static char g_fz_error[512];
void ss_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_fz_error, sizeof(g_fz_error), fmt, ap);
  va_end(ap);
}
extern "C" const char* ssx_fused_last_error(void) { return g_fz_error; }

namespace {

// MT = 2 gate tiles (32 quads = 128 frames, 48 KB of LDS, K-staged, weights in fetch order) + 96-row projection tiles (36 KB): BASELINE configs[1]
// WT = false: plain stores + agent-scope release fence / acquire fence (form 1). WT = true: the gate writes its outputs THROUGH (sc1 stores), waits for
// them (every wave's s_waitcnt vmcnt(0) inside __syncthreads) and raises the counter; the projection polls and reads the gate outputs with sc1
// loads - no fence on either side (the guide's cheaper valid form: "sc1 loads may replace the acquire only when the producer stored sc1").
template <int GMT, int RMT, int KCH, bool WT>
__global__ __launch_bounds__(256, 3) void fused_gate_res_kernel(const ss_conv_gemm_args g, const float* __restrict__ W16g, int q_tiles_per_item, int q_tiles,
                                                                int n_tiles_g, int log2d, const ss_conv_gemm_args r, const float* __restrict__ W16r,
                                                                int m_tiles_per_item, int m_tiles, int n_tiles_r, int n_gate_blocks,
                                                                unsigned* __restrict__ counters, unsigned target, int* __restrict__ error) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = (int)blockIdx.x;
  if (bid < n_gate_blocks) {
    const bool real = fz_g::wino43_gate16_body<GMT, true, true, WT ? 16 : 0>(g, W16g, q_tiles_per_item, q_tiles, n_tiles_g, log2d, nullptr, bid, smem);
    if (!real) return;
    __syncthreads();   // every wave's stores are issued
    if (threadIdx.x == 0) {
      const int grp = bid / (8 * n_tiles_g), rem = bid % (8 * n_tiles_g);
      const int qt = grp * 8 + (rem & 7);
      if constexpr (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(counters + qt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  const int rb = bid - n_gate_blocks;
  {
    const int grp = rb / (8 * n_tiles_r), rem = rb % (8 * n_tiles_r);
    const int mt = grp * 8 + (rem & 7);
    if (mt >= m_tiles) return;
    if (threadIdx.x == 0) {
      constexpr int BM = 16 * RMT, BF = 64 * GMT;   // projection rows per tile; frames per gate row tile (16 GMT quads of 4 frames)
      const int b = mt / m_tiles_per_item, t0 = (mt % m_tiles_per_item) * BM;
      int t1 = t0 + BM - 1;
      if (t1 > r.T - 1) t1 = r.T - 1;
      const int q_lo = b * q_tiles_per_item + t0 / BF, q_hi = b * q_tiles_per_item + t1 / BF;
      for (int qt = q_lo; qt <= q_hi; ++qt) {
        int spins = 0;
        while (__hip_atomic_load(counters + qt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(8);
          if (++spins > (1 << 20)) {   // never hang the box: flag the failure and carry on with whatever is there
            *error = 1;
            break;
          }
        }
      }
      if constexpr (!WT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  fz_r::gemm16_res_body<RMT, KCH, true, WT ? 16 : 0>(r, W16r, m_tiles_per_item, m_tiles, n_tiles_r, rb, smem);
}

}  // namespace

extern "C" int ssx_fused_gate_res_counters(int B, int T, int dilation) {
  if (B <= 0 || T <= 0 || dilation <= 0) return 0;
  return ss_cdiv(ss_cdiv(T, 4 * dilation) * dilation, 32) * B;
}

// gate args / res args exactly as ss_wino43_gate16w / ss_gemm16_resw take them (the gate's output C must be the projection's A); mt_gate = 2,
// mt_res = 6 only. counters: one zeroed uint32 per gate row tile (ceil(quads / 32) * B); error: one int32, set to 1 if a wait gave up.
extern "C" int ssx_fused_gate_res(const ss_conv_gemm_args* gate, const float* W16g, int dilation, const ss_conv_gemm_args* res, const float* W16r,
                                 uint32_t* counters, int32_t* error, int write_through, void* stream_) {
  SS_CHECK_ARG(gate && res && W16g && W16r && counters && error, "ssx_fused_gate_res: null argument");
  const ss_conv_gemm_args& g = *gate;
  const ss_conv_gemm_args& r = *res;
  SS_CHECK_ARG(dilation >= 1 && dilation <= 32 && (dilation & (dilation - 1)) == 0, "ssx_fused_gate_res: dilation must be a power of two <= 32");
  SS_CHECK_ARG(g.B == r.B && g.T == r.T && g.Kp == 256 && r.Kp == 256 && g.Np == 512 && r.N == 256 && g.epi == SS_EPI_GATE && g.e_tiled,
               "ssx_fused_gate_res: the mel denoiser's layer shape only (C = 256, addend in fetch order)");
  SS_CHECK_ARG(g.group_size == 0 && r.group_size == 0 && g.C == r.A && g.ldc == r.lda, "ssx_fused_gate_res: the gate's output must be the projection's operand");
  int log2d = 0;
  while ((1 << log2d) < dilation) ++log2d;
  constexpr int GMT = 2, RMT = 6, KCH = 8;
  const int quads_per_item = ss_cdiv(g.T, 4 * dilation) * dilation;
  const int q_tiles_per_item = ss_cdiv(quads_per_item, 16 * GMT);
  const int q_tiles = q_tiles_per_item * g.B;
  const int n_tiles_g = g.Np / 64;
  const int n_gate = ss_cdiv(q_tiles, 8) * 8 * n_tiles_g;
  const int m_tiles_per_item = ss_cdiv(r.T, 16 * RMT);
  const int m_tiles = m_tiles_per_item * r.B;
  const int n_tiles_r = ss_cdiv(r.N, 64);
  const int n_res = ss_cdiv(m_tiles, 8) * 8 * n_tiles_r;
  const size_t lds = (size_t)12 * 16 * GMT * 32 * sizeof(float);   // the gate's K-staged image (48 KB) >= the projection's ring (36 KB)
  if (write_through)
    hipLaunchKernelGGL((fused_gate_res_kernel<GMT, RMT, KCH, true>), dim3(n_gate + n_res), dim3(256), lds, (hipStream_t)stream_, g, W16g, q_tiles_per_item,
                       q_tiles, n_tiles_g, log2d, r, W16r, m_tiles_per_item, m_tiles, n_tiles_r, n_gate, counters, (unsigned)n_tiles_g, error);
  else
    hipLaunchKernelGGL((fused_gate_res_kernel<GMT, RMT, KCH, false>), dim3(n_gate + n_res), dim3(256), lds, (hipStream_t)stream_, g, W16g, q_tiles_per_item,
                       q_tiles, n_tiles_g, log2d, r, W16r, m_tiles_per_item, m_tiles, n_tiles_r, n_gate, counters, (unsigned)n_tiles_g, error);
  SS_CHECK_LAUNCH("fused_gate_res_kernel");
  return SS_OK;
}
