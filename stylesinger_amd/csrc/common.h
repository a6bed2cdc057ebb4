// Shared device/host helpers for libstylesinger_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>

#define SS_OK 0
#define SS_ERR_ARG (-1)
#define SS_ERR_HIP (-2)
#define SS_ERR_UNSUPPORTED (-3)

void ss_set_error(const char* fmt, ...);

#define SS_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ss_set_error(__VA_ARGS__);           \
      return SS_ERR_ARG;                   \
    }                                      \
  } while (0)

#define SS_CHECK_LAUNCH(name)                                                   \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      ss_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return SS_ERR_HIP;                                                        \
    }                                                                           \
  } while (0)

#define SS_PROPAGATE(expr)        \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != SS_OK) return rc__; \
  } while (0)

static inline int ss_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Run-time tuning knobs (ss_set_tuning): process-wide, read at launch time, never change results.
//   wave_prio: 0 = all waves at priority 0; 1 = static s_setprio((blockIdx / 256) % 3); 2 = s_setprio(blockIdx % 3).
//     The blocks sharing a CU then differ in priority, so the matrix pipe of a SIMD serves them one after the other instead
//     of round-robin: their non-MFMA phases (staging, barrier) stop coinciding (launch structure).
//   gate16: tiling of the Winograd F(4,3) gate launch. 1 (default) = pick per launch (ss_wino43_gate16_pick: 16x16x4 tiles of
//     16*MT quads when that fills the chip better, else the 32x32x2 kernel); 0 = always the 32x32x2 kernel; 2 / 3 = force MT.
//   res_tile / skip_tile: SS_TILE_* override of the residual-half projection / the K = L*C skip GEMM of the denoiser loops
//     (0 = the built-in choice); validated by ss_set_tuning.
//   res16: 1 (default) = the residual-half projection of the deferred-skip loops runs on ss_gemm16_res (16x16x4 tiles, LDS-DMA);
//     0 = ss_conv_gemm; 4 / 6 / 8 = force the row tile.
//   skip16: the same switch for the K = L*C skip GEMM (ss_gemm16_store).
//   gate16_ks: 1 (default) = the 16x16x4 gate kernel stages all six Winograd components of a K chunk at once (one barrier per K chunk, 48 / 72 KB
//     of LDS); 0 = one component per barrier (8 / 12 KB).
//   gate256: 1 (default) = bf16 GATE launches that qualify (ss_gemm_bf16_gate256_ok) run on the 256x256 LDS-DMA kernel; 0 = always the
//     generic bf16 kernel.
//   htile: row tile of the generic bf16 kernel (0 = built-in choice, 64 | 128 = force); wino_tn: column tile of the F(2,3) gate (0 = pick,
//     1 = 64, 2 = 128 columns); wino_v1: 1 = the round-1 F(2,3) kernel (A/B against v2).
//   e16: 1 (default) = the fp32 denoiser loops re-lay the conditioner addend of every 16x16x4 gate launch in that kernel's fetch order once per
//     forward (ss_gate16_tile_addend); 0 = the gate reads the row-major slab (A/B; results are bit-identical).
//   mel_tail: 1 (default) = for launches of at most 8 frames per CU (one short utterance) the mel sampler runs output projection + DDPM update +
//     the next input projection as one VALU launch (mel_tail_kernel); 0 = always the two matrix-core launches.
//   voc_wino_max_mb: the vocoder's grouped-Winograd convs address an item with 32-bit byte offsets; items whose stage panel (+ halo) reaches
//     this many MiB take the direct kernel instead (default 2048 = the real limit; tests lower it to force that fallback).
//   layer512: 1 (default) = the fp16x2 mel stack runs ONE ss_layer512 launch per layer (gate + residual projection, G kept in LDS) when the
//     net carries the fragment-order packs and ss_layer512_ok(B, T, ...) holds; 0 = the gate + residual-projection launch pair; 2 = also below the
//     chip-filling size (parity tests).
//   layer512_tail: 1 (default) = ss_layer512 cuts the tiles of its last round into half tiles when that round would keep at most half of the
//     workgroups busy (1408 tiles on 256 CUs: makespan 5.6 instead of 6 tile periods), and the EVEN workgroups run their half tile first - the two
//     halves of the chip are then half a tile period out of phase, one streams through HBM while the other multiplies; 2 = every half tile last (the
//     round-6 schedule before the phase shift); 0 = whole tiles only (A/B; results are identical in all three).
//   skip_dense: 1 (default) = the fp16sd skip GEMM with both operands compact runs 64 channels per step (tile256s_kernel<.., DENSE>); 0 = 32-channel steps
//     with dead DMA lanes (A/B; results are identical: the same products enter every accumulator in the same order).
//   q4_force: 0 (default) = the fp16q4 kernels take only launches that fill the chip (their _ok rules); 1 = any launch they can compute (parity tests run
//     one 30 s item through them).
struct SsTuning { int wave_prio; unsigned long long* clock_probe; int gate16; int res_tile; int skip_tile; int res16; int skip16; int gate256; int gate16_ks;
                  int htile; int wino_tn; int wino_v1; int voc_wino_max_mb; int e16; int mel_tail; int gate128; int q4_force; int layer512; int layer512_tail; int skip_dense; };
extern SsTuning g_ss_tuning;
// fp16q4 range guard (ss_set_q4_guard): while non-null, every fp16q4 launch first reduces max |a| / (6 q_scale) over the fp16 operand it is about
// to convert to fp4 into guard[which] (which = 0 gate, 1 skip GEMM; float bits, atomicMax): > 1 means the fixed scale saturates
extern unsigned int* g_ss_q4_guard;
struct ss_gemm_bf16_args;
int ss_q4_guard_launch(const ss_gemm_bf16_args* a, int which, void* stream);
// compute units of the current device (cached per device; 256 when no device can be queried): the tiling picks model a launch as
// workgroup layers per CU, so the count must be the device's, not MI355X's
int ss_n_cu();
struct ss_conv_gemm_args;
// gemm16.hip, library-internal: split-K skip GEMM without its reduction launch (the tail kernels of diffusion.hip add the slices)
int ss_gemm16_store_partials(const ss_conv_gemm_args* args, int mt, int ksplit, float* partials, void* stream);

// static per-block wave priority (wave-uniform; s_setprio takes an immediate)
__device__ __forceinline__ void ss_apply_wave_prio(int mode) {
  if (mode == 0) return;
  const int p = mode == 1 ? (int)((blockIdx.x >> 8) % 3u) : (int)(blockIdx.x % 3u);
  if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
}

// ----------------------------------------------------------------------------------------------
// Device math. Activations follow the torch CPU definitions the reference relies on.
// ----------------------------------------------------------------------------------------------
// lens[b] for a block-uniform b as a SCALAR load (constant address space): left to the compiler it may become a vector load whose
// s_waitcnt vmcnt(0) sits in front of the kernel's first operand fetch - one full memory round trip per launch.
__device__ __forceinline__ int ss_uniform_len(const int* lens, int b, int T) {
  if (!lens) return T;
  const auto* p = reinterpret_cast<const __attribute__((address_space(4))) int*>(reinterpret_cast<uintptr_t>(lens));
  const int n = p[__builtin_amdgcn_readfirstlane(b)];
  return n < T ? (n < 0 ? 0 : n) : T;   // a length beyond the buffer is clamped: the A-operand buffer range of the kernels is len * lda bytes
}
__device__ __forceinline__ float ss_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// Gate nonlinearities on the hardware exp2/rcp units (v_exp_f32 / v_rcp_f32, ~1 ulp each): absolute error
// <= ~3e-7 on outputs in [-1,1], far inside the 1e-4 mel budget, and ~10x fewer VALU ops than ocml tanhf/expf.
// (v_rcp_f32 directly: __frcp_rn expands to the 10-instruction correctly-rounded division sequence)
__device__ __forceinline__ float ss_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ss_tanh_fast(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
// nn.GELU() default = erf form (modules/commons/common_layers.py:574, modules/StyleSinger/lse.py:183)
__device__ __forceinline__ float ss_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// F.softplus(beta=1, threshold=20) then tanh (modules/diff/diffusion.py:64-66)
__device__ __forceinline__ float ss_mish(float x) {
  float sp = (x > 20.0f) ? x : log1pf(expf(x));
  return x * tanhf(sp);
}
__device__ __forceinline__ float ss_lrelu(float x, float slope) { return x >= 0.0f ? x : x * slope; }
// The same value for 0 < slope < 1 from the product sx = slope * x in ONE instruction: max(x, sx). fmaxf(x, sx) - and fmed3(x, sx, inf), which
// the compiler folds back into it - costs two on this target (IEEE mode: a possibly signalling x is quieted by v_max x, x before the v_max),
// and VALU instructions take matrix time; the asm is the bare v_max_f32.
__device__ __forceinline__ float ss_lrelu_max(float x, float sx) {
  float o;
  asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(x), "v"(sx));
  return o;
}

enum SsAct { SS_ACT_NONE = 0, SS_ACT_RELU = 1, SS_ACT_GELU = 2, SS_ACT_MISH = 3, SS_ACT_TANH = 4, SS_ACT_LRELU = 5 };

__device__ __forceinline__ float ss_apply_act(float v, int act, float slope) {
  switch (act) {
    case SS_ACT_RELU: return fmaxf(v, 0.0f);
    case SS_ACT_GELU: return ss_gelu(v);
    case SS_ACT_MISH: return ss_mish(v);
    case SS_ACT_TANH: return tanhf(v);
    case SS_ACT_LRELU: return ss_lrelu(v, slope);
    default: return v;
  }
}

// ----------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (production noise; parity tests inject a tape instead).
// ----------------------------------------------------------------------------------------------
struct SsPhilox {
  uint32_t k0, k1;
  __device__ __forceinline__ SsPhilox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t ka, uint32_t kb) const {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ ka, n1 = lo1, n2 = hi0 ^ c[3] ^ kb, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  // 4 x 32 random bits for counter (c0,c1,c2,c3)
  __device__ __forceinline__ void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) const {
    uint32_t c[4] = {c0, c1, c2, c3};
    uint32_t ka = k0, kb = k1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, ka, kb);
      ka += 0x9E3779B9u;
      kb += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
};
// uniform in (0,1]
__device__ __forceinline__ float ss_u01(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }
// two standard normals from two 32-bit words (Box-Muller)
__device__ __forceinline__ void ss_boxmuller(uint32_t a, uint32_t b, float& z0, float& z1) {
  float u1 = ss_u01(a), u2 = ss_u01(b);
  float r = sqrtf(-2.0f * logf(u1));
  float s, c;
  sincosf(6.28318530717958647692f * u2, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

// The mel sampler's Gaussian draw (SS_EPI_DDPM epilogue of ss_conv_gemm, mel_tail_kernel): element (frame t, bin n) of item b at step `step` is output t & 3
// of the Philox block with counter ((t >> 2) * N + n, b) - a block's four words become four normals (two Box-Muller pairs), and a lane of the MFMA
// epilogue holds four consecutive frames of a bin: one block per four elements (round 6; before, every element ran its own block and used one of its
// four words: 7 of the 23 us of the C2 launch). The draw depends on (item, frame, bin, step) only - not on T padding, batching or tiling.
__device__ __forceinline__ void ss_mel_draw4(const SsPhilox& rng, uint32_t t4, uint32_t N, uint32_t n, uint32_t b, uint32_t step, float (&z)[4]) {
  uint32_t o[4];
  rng.gen(t4 * N + n, b, step, 0x4d454c44u, o);
  ss_boxmuller(o[0], o[1], z[0], z[1]);
  ss_boxmuller(o[2], o[3], z[2], z[3]);
}

// SS_TRACE (debug builds only, tools/wave_trace.py): phase stamps on the shader clock. SS_CLK waits for the LDS/scalar queue (lgkmcnt 0),
// SS_CLK_VM for the vector-memory queue (vmcnt 0) before reading s_memtime; both pin the schedule around them.
#ifdef SS_TRACE
#define SS_CLK(var)                                  \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_waitcnt(0xc07f);              \
    var = (unsigned)__builtin_readcyclecounter();    \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#define SS_CLK_VM(var)                               \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_waitcnt(0x0f70);              \
    var = (unsigned)__builtin_readcyclecounter();    \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#else
#define SS_CLK(var) do { } while (0)
#define SS_CLK_VM(var) do { } while (0)
#endif

