// Micro-benchmark: how many bytes per clock can ONE CU pull from its XCD's L2 (a) into VGPRs with buffer_load_dwordx4, (b) into LDS by
// LDS-DMA? Decides which tile shapes are feedable (round 3: the 256x256 bf16 tile needs 48 GB/s per CU of operand traffic).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l2bw.hip -o /tmp/l2bw && /tmp/l2bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

// every wave sweeps its own window of `win` bytes (a multiple of 8 KB), 8 x 1 KB per iteration, `iters` times
template <bool DMA>
__global__ __launch_bounds__(256) void k(const char* src, int win, int iters, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, 0x7fffffff, 0x00020000);
  const int base = ((blockIdx.x * 4 + wave) * win) & (8 * 1024 * 1024 - 1);   // stay inside 8 MB total
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    for (int off = 0; off < win; off += 8192) {
      if constexpr (DMA) {
#pragma unroll
        for (int j = 0; j < 8; ++j) glds16(rsrc, lds + wave * 8192 + j * 1024, lane * 16, base + off + j * 1024);
      } else {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, base + off + j * 1024, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
      }
    }
  }
  if constexpr (DMA) {
    __builtin_amdgcn_s_waitcnt(0x0f70);
    acc[0] = *reinterpret_cast<unsigned*>(lds + threadIdx.x * 4);
  }
  if (acc[0] == 0x12345678u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}

template <bool DMA>
void run(const char* d, unsigned* sink, int wg_per_cu, int win) {
  const int iters = 64 * 1024 * 1024 / (wg_per_cu * 4 * win) / 4;   // same total bytes per CU for every configuration
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<DMA>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, win, 2, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<DMA>, dim3(256 * wg_per_cu), dim3(256), 0, 0, d, win, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)256 * wg_per_cu * 4 * win * iters;
  printf("%s  %d waves/CU, %3d KB window per wave: %6.2f TB/s chip = %6.1f GB/s per CU = %5.1f B/clk/CU at 2.4 GHz\n",
         DMA ? "LDS-DMA       " : "load -> VGPR  ", wg_per_cu * 4, win / 1024, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 256 / 1e9,
         bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  char* d;
  unsigned* sink;
  hipMalloc(&d, 8 * 1024 * 1024);
  hipMemset(d, 1, 8 * 1024 * 1024);
  hipMalloc(&sink, 16);
  for (int wg : {1, 2, 4, 8}) {
    run<false>(d, sink, wg, 8 * 1024);      // 8 KB windows: everything L1 / L2 hot
    run<false>(d, sink, wg, 512 * 1024);    // 512 KB windows per wave: L2-resident, L1 thrashed
    run<true>(d, sink, wg, 8 * 1024);
    run<true>(d, sink, wg, 512 * 1024);
  }
  return 0;
}
