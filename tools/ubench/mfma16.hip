// Micro-benchmark for the 16x16-tile Winograd gate kernel (round 3): issue rate of v_mfma_f32_16x16x4_f32 with NACC independent
// accumulators, W waves per SIMD and V independent VALU ops per MFMA; plus what DPP row_ror:8 moves (the in-wave gate exchange).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma16.hip -o /tmp/mfma16 && /tmp/mfma16
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = seed + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < V; ++q) {
        const int idx = (j * V + q) & 15;
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[idx]) : "v"(b), "v"(a));
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int r = 0; r < 4; ++r) s += acc[j][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V, int NACC>
void run(float* d_out, int wg_per_cu) {
  const int iters = 20000 / NACC * 3;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 256 * wg_per_cu;
  hipLaunchKernelGGL((k<V, NACC>), dim3(blocks), dim3(256), 0, 0, d_out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, NACC>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * NACC * wg_per_cu;  // one wave of each workgroup per SIMD
  const double cyc = ms * 1e-3 * 2.4e9 / mfma_per_simd;
  printf("16x16x4 f32: %d waves/SIMD, %d accumulators, %d VALU/MFMA: %.1f cycles per MFMA per SIMD at 2.4 GHz (%.1f TF/s)\n", wg_per_cu,
         NACC, V, cyc, 2048.0 * mfma_per_simd * 1024 / (ms * 1e-3) * 1e-12);
}

__global__ void dpp_probe(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(0, lane, 0x128, 0xf, 0xf, false);  // row_ror:8
}

int main() {
  float* d;
  hipMalloc(&d, 256 * 4 * 256 * 4);
  for (int w = 1; w <= 3; ++w) {
    run<0, 1>(d, w);
    run<0, 3>(d, w);
    run<1, 3>(d, w);
    run<2, 3>(d, w);
    run<3, 3>(d, w);
    run<0, 6>(d, w);
  }
  int* o;
  hipMalloc(&o, 64 * 4);
  hipLaunchKernelGGL(dpp_probe, dim3(1), dim3(64), 0, 0, o);
  int h[64];
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("row_ror:8 -> lane i reads lane:");
  for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
  printf("\n");
  return 0;
}
