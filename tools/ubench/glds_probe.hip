// Probe of the LDS-DMA builtins on gfx950 (round 3, before building a kernel on them): where do the 64 lanes of
// __builtin_amdgcn_raw_ptr_buffer_load_lds(size 16) land in LDS, what does an out-of-range lane write, and does a counted vmcnt order it?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/glds_probe.hip -o /tmp/glds_probe && /tmp/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float* src, int n_bytes, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 64 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2 * 64 * 4; i += 64) lds[i] = -1.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_bytes, 0x00020000);
  // lane i fetches the 16 bytes at byte offset perm(i) * 16; lanes >= 48 point out of range
  const int voff = lane < 48 ? ((lane * 7) % 48) * 16 : (int)0x80000000;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  // second DMA into the second KB, all lanes in range, SGPR offset 256 bytes
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 64 * 4), 16, lane * 16, 256, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();
  for (int i = lane; i < 2 * 64 * 4; i += 64) out[i] = lds[i];
}

int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4);
  hipMalloc(&o, 2 * 64 * 4 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n * 4, o);
  std::vector<float> r(2 * 64 * 4);
  hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int lane = 0; lane < 64; ++lane) {
    const float want = lane < 48 ? (float)(((lane * 7) % 48) * 4) : 0.0f;
    if (r[lane * 4] != want || (lane < 48 && r[lane * 4 + 3] != want + 3)) ok = 0;
  }
  printf("DMA 1 (gather by per-lane offset, lane-linear LDS image, OOB lanes): lane0 %.0f lane1 %.0f lane47 %.0f lane48 %.0f lane63 %.0f -> %s\n",
         r[0], r[4], r[47 * 4], r[48 * 4], r[63 * 4], ok ? "as expected (OOB writes 0)" : "UNEXPECTED");
  int ok2 = 1;
  for (int lane = 0; lane < 64; ++lane)
    if (r[256 + lane * 4] != (float)(64 + lane * 4)) ok2 = 0;
  printf("DMA 2 (soffset 256 B): lane0 %.0f lane63 %.0f -> %s\n", r[256], r[256 + 63 * 4], ok2 ? "as expected" : "UNEXPECTED");
  return 0;
}
