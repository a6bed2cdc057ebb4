// Round 5: SKELETON of the "row-owner" fused residual layer the round-4 review asked to price on hardware (DESIGN.md 7 lead 1) - the
// instruction skeleton of such a kernel, NOT a kernel that computes the layer (operands are synthetic; nothing checks the numbers):
//   one 512-thread workgroup (8 waves, one per CU) owns 16 Winograd quads = 64 frames across ALL 512 packed gate columns:
//   gate phase   per wave 16 quads x 64 columns x 6 F(4,3) components = 24 accumulator tiles of 16x16 (96 registers), K = 256 in 64 k-steps:
//                1536 v_mfma_f32_16x16x4_f32 per wave; the transformed A tile comes from LDS (built once per 32-channel chunk by all 512
//                threads from raw rows fetched from global memory: 6 loads + ~20 VALU + 6 LDS writes per thread and chunk, one barrier);
//                the wave's weights (its 64 columns x 6 components: 393 KB per wave, 3.1 MB per workgroup, the SAME 3.1 MB for every
//                workgroup -> L2 / MALL resident) stream global -> registers in fetch order, 1 KB per instruction, a ring of three
//                quarter-steps (6 x 16 B per lane each) fetched two quarter-steps ahead (~3000 cycles of MFMA issue per SIMD);
//   gate epilogue output transform 6 -> 4 frames, conditioner addend (16 x 16-byte loads per lane), exp / rcp gate, G -> LDS (64 KB) and -> HBM;
//   projection   G (LDS) x W_res (256 KB from L2, fetch order): per wave 64 rows x 32 columns = 8 tiles, 512 MFMAs; epilogue x <- (x + . + b) / sqrt 2.
// The MFMA count (12 288 + 4 096 per workgroup), the bytes each workgroup pulls (3.1 MB + 0.26 MB of weights, 64 KB of raw rows with halo x 6/4,
// 128 KB of addend, 64 KB x in / out, 64 KB G out) and the LDS traffic are those of the real thing; what is missing can only make the real
// kernel slower. Printed: time per launch for 188 workgroups (BASELINE configs[1]: 12 000 frames) and for 376 / 752 (many rounds), to be read
// against the two-launch form's 65-68 us per layer at C2 (tools/kbench_fused.py prints it in the same session).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/rowowner_skeleton.hip -o /tmp/rowowner && /tmp/rowowner
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KCH = 8;            // 32-channel chunks of K = 256
constexpr int A_FLOATS = 6 * 16 * 32;   // one staged chunk: 6 components x 16 quads x 32 channels = 12 KB

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

__global__ __launch_bounds__(512, 1) void rowowner_skeleton(const float* __restrict__ X, const float* __restrict__ Wg, const float* __restrict__ E,
                                                            const float* __restrict__ Wr, float* __restrict__ Xout, float* __restrict__ G, int rows) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][A_FLOATS]  24 KB
  float* Gs = smem + 2 * A_FLOATS;        // [64 frames][256 channels] fp32 = 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, kq = lane >> 4;
  const int t0 = blockIdx.x * 64;         // first frame of this workgroup
  // ---- raw-row staging role: thread -> (quad tid >> 5, channel tid & 31) of a chunk; six raw rows t - d .. t + 4 d (d = 1 here)
  const int sq = tid >> 5, sc = tid & 31;
  auto stage = [&](int chunk, float* dst) {
    float r[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      int t = t0 + 4 * sq + j - 1;
      t = t < 0 ? 0 : (t >= rows ? rows - 1 : t);
      r[j] = X[(size_t)t * 256 + chunk * 32 + sc];
    }
    // F(4,3) input transform with the shared sub-expressions of the product kernel (18 VALU)
    const float a = r[4] - 4.f * r[2], b = r[3] - 4.f * r[1], c = r[4] - r[2], d = r[3] - r[1];
    const float c0 = 4.f * r[0] - 5.f * r[2] + r[4], c5 = 4.f * r[1] - 5.f * r[3] + r[5];
    const float comp[6] = {c0, a + b, a - b, c + 2.f * d, c - 2.f * d, c5};
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[(j * 16 + sq) * 32 + (sc ^ ((sq & 7) << 2))] = comp[j];
  };
  f32x4 acc[6][4];
#pragma unroll
  for (int c = 0; c < 6; ++c)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[c][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weights of this wave: [k group 16][n tile 4][component 6][lane 64][4 floats]
  const float* wp = Wg + (size_t)wave * 16 * 24 * 256 + lane * 4;
  f32x4 bw[3][6];   // ring of three quarter-steps (72 registers), fetched two quarter-steps ahead (the loop is fully unrolled: static indices)
  auto load_q = [&](int t, f32x4 (&dst)[6]) {   // quarter-step t = 4 kg + n
#pragma unroll
    for (int c = 0; c < 6; ++c) dst[c] = ldg4(wp + (size_t)(t * 6 + c) * 256);
  };
  load_q(0, bw[0]);
  load_q(1, bw[1]);
  stage(0, As);
  __syncthreads();
#pragma unroll
  for (int ch = 0; ch < KCH; ++ch) {
    const float* Ab = As + (ch & 1) * A_FLOATS;
    if (ch + 1 < KCH) stage(ch + 1, As + ((ch + 1) & 1) * A_FLOATS);
#pragma unroll
    for (int g = 0; g < 2; ++g) {   // two k groups of 4 k-steps per 32-channel chunk
      f32x4 af[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) af[c] = *reinterpret_cast<const f32x4*>(Ab + (c * 16 + lc) * 32 + (((g * 4 + kq) * 4) ^ ((lc & 7) << 2)));
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int t = (ch * 2 + g) * 4 + n;
        if (t + 2 < 64) load_q(t + 2, bw[(t + 2) % 3]);
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].x, bw[t % 3][c].x, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].y, bw[t % 3][c].y, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].z, bw[t % 3][c].z, acc[c][n], 0, 0, 0);
          acc[c][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c].w, bw[t % 3][c].w, acc[c][n], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- gate epilogue: output transform 6 -> 4 frames, addend, gate (the partner operand by a lane exchange), G -> LDS + HBM
  const float* ep = E + ((size_t)blockIdx.x * 8 + wave) * 16 * 256 + lane * 4;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    f32x4 o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m0 = acc[0][n][r], m1 = acc[1][n][r], m2 = acc[2][n][r], m3 = acc[3][n][r], m4 = acc[4][n][r], m5 = acc[5][n][r];
      o[0][r] = m0 + m1 + m2 + m3 + m4;
      o[1][r] = m1 - m2 + 2.f * (m3 - m4);
      o[2][r] = m1 + m2 + 4.f * (m3 + m4);
      o[3][r] = m1 - m2 + 8.f * (m3 - m4) + m5;
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const f32x4 e = ldg4(ep + (size_t)(n * 4 + f) * 256);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = o[f][r] + e[r];
        const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-z));           // sigmoid or (through a multiplier) tanh: one exp + one rcp
        const float p = __shfl_xor(s, 8, 64);                               // the gate partner sits 8 lanes away in the product kernel
        const float gv = s * p;
        const int frame = 4 * (kq * 4 + r) + f, chn = wave * 32 + n * 8 + (lc & 7);
        if (lc < 8) {
          Gs[frame * 256 + chn] = gv;
          G[(size_t)(t0 + frame) * 256 + chn] = gv;
        }
      }
    }
  }
  __syncthreads();
  // ---- projection: G [64 x 256] (LDS) x W_res: this wave's 32 columns, 4 row tiles
  f32x4 pa[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) pa[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wr = Wr + (size_t)wave * 16 * 2 * 256 + lane * 4;   // [wave][k group 16][n 2][lane][4]
  f32x4 br[3][2];
  br[0][0] = ldg4(wr);
  br[0][1] = ldg4(wr + 256);
  br[1][0] = ldg4(wr + 512);
  br[1][1] = ldg4(wr + 768);
#pragma unroll
  for (int kg = 0; kg < 16; ++kg) {
    if (kg + 2 < 16) {
      br[(kg + 2) % 3][0] = ldg4(wr + (size_t)(kg + 2) * 512);
      br[(kg + 2) % 3][1] = ldg4(wr + (size_t)(kg + 2) * 512 + 256);
    }
    f32x4 ga[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) ga[m] = *reinterpret_cast<const f32x4*>(Gs + (16 * m + lc) * 256 + kg * 16 + kq * 4);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        pa[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m].x, br[kg % 3][n].x, pa[m][n], 0, 0, 0);
        pa[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m].y, br[kg % 3][n].y, pa[m][n], 0, 0, 0);
        pa[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m].z, br[kg % 3][n].z, pa[m][n], 0, 0, 0);
        pa[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[m].w, br[kg % 3][n].w, pa[m][n], 0, 0, 0);
      }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t idx = (size_t)(t0 + 16 * m + 4 * kq + r) * 256 + wave * 32 + n * 16 + lc;
        Xout[idx] = (X[idx] + pa[m][n][r] + 0.01f) * 0.70710678f;
      }
}

int main() {
  const int rows_max = 752 * 64;
  float *X, *Wg, *E, *Wr, *Xo, *G;
  hipMalloc(&X, (size_t)rows_max * 256 * 4);
  hipMalloc(&Xo, (size_t)rows_max * 256 * 4);
  hipMalloc(&G, (size_t)rows_max * 256 * 4);
  hipMalloc(&E, (size_t)752 * 8 * 16 * 256 * 4);
  hipMalloc(&Wg, (size_t)8 * 16 * 24 * 256 * 4);   // 3.1 MB
  hipMalloc(&Wr, (size_t)8 * 16 * 2 * 256 * 4);    // 0.26 MB
  std::vector<float> h((size_t)rows_max * 256, 0.001f);
  hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(E, 0, (size_t)752 * 8 * 16 * 256 * 4);
  hipMemcpy(Wg, h.data(), (size_t)8 * 16 * 24 * 256 * 4, hipMemcpyHostToDevice);
  hipMemcpy(Wr, h.data(), (size_t)8 * 16 * 2 * 256 * 4, hipMemcpyHostToDevice);
  const size_t lds = (size_t)(2 * A_FLOATS + 64 * 256) * 4;   // 24 + 64 KB
  hipFuncSetAttribute(reinterpret_cast<const void*>(&rowowner_skeleton), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs : {188, 376, 752}) {
    const int rows = wgs * 64;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rowowner_skeleton, dim3(wgs), dim3(512), lds, 0, X, Wg, E, Wr, Xo, G, rows);
    hipDeviceSynchronize();
    const int iters = 40;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(rowowner_skeleton, dim3(wgs), dim3(512), lds, 0, X, Wg, E, Wr, Xo, G, rows);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double flop = (double)wgs * (12288.0 + 4096.0) * 8 * 2048.0 / 8;   // MFMAs per workgroup x 2048 flop
    printf("row-owner skeleton: %4d workgroups (%6d frames): %7.1f us per launch back to back = %.2f us per 188-workgroup round; %.1f TF/s executed "
           "(%.2f of the fp32 MFMA roof)%s\n", wgs, rows, us, us * 188.0 / wgs, flop / (us * 1e-6) * 1e-12, flop / (us * 1e-6) / 157.3e12,
           hipGetLastError() == hipSuccess ? "" : "  [launch error]");
  }
  return 0;
}
