// Micro-benchmark: do VALU instructions co-execute with v_mfma_f32_32x32x2_f32 on gfx950, or do they take matrix-pipe time?
// (decides how much the non-MFMA instruction count of the fp32 GEMM kernels matters; result recorded in DESIGN.md)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V, bool BF16, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(seed + i); bb[i] = (__bf16)(seed - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (BF16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int idx = (j * V + q) & 15;
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[idx]) : "v"(b), "v"(a));
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, bool BF16, int WAVES>
double run(float* d_out, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * (12 / WAVES);  // 12 waves per CU = 3 per SIMD (the production occupancy)
  hipLaunchKernelGGL((k<V, BF16, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d_out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, BF16, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d_out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3;
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 12 * 64 * 4 * 4);
  const int iters = 20000;
  const double n_mfma = 4.0 * iters;  // per wave
#define ROW(V)                                                                                                     \
  {                                                                                                                \
    const double t = run<V, false, 4>(d, iters), tb = run<V, true, 4>(d, iters);                                   \
    printf("VALU per MFMA %2d: fp32 32x32x2 %7.1f cycles/MFMA/SIMD (at 2.4 GHz; 3 waves/SIMD)   bf16 32x32x16 %7.1f\n", V, \
           t * 2.4e9 / (n_mfma * 3), tb * 2.4e9 / (n_mfma * 3));                                                   \
  }
  ROW(0) ROW(1) ROW(2) ROW(4) ROW(8) ROW(16)
  return 0;
}
