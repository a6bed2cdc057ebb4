// Probe: is the SGPR offset of a raw buffer access part of the range check on gfx950? (round 3: several kernels put row offsets there)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float* buf, int n_bytes, float* out) {
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(buf, 0, n_bytes, 0x00020000);
  const int lane = threadIdx.x;
  // voffset inside the buffer, soffset pushes lanes >= 32 past num_records
  const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, n_bytes - 128, 0));
  out[lane] = v;
  // store with the same addressing: must be dropped for lanes >= 32
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 777.0f), rsrc, lane * 4, n_bytes - 128, 0);
  // negative voffset + positive soffset that would land in range if the sum wrapped
  const float w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, -64 + lane * 4, 64, 0));
  out[64 + lane] = w;
}
int main() {
  const int n = 1024;  // floats inside the descriptor; the allocation is twice as large
  float *d, *o, h[2 * n], r[128];
  for (int i = 0; i < 2 * n; ++i) h[i] = (float)i;
  hipMalloc(&d, 2 * n * 4); hipMalloc(&o, 128 * 4);
  hipMemcpy(d, h, 2 * n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, n * 4, o);
  hipMemcpy(r, o, 128 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(h, d, 2 * n * 4, hipMemcpyDeviceToHost);
  printf("load  voffset in range + soffset: lane 31 -> %.0f (expect %d), lane 32 -> %.0f (0 = soffset IS range checked, %d = it is not)\n", r[31], n - 1, r[32], n);
  printf("store voffset in range + soffset: buf[%d] = %.0f (777 expected), buf[%d] = %.0f (%d = dropped, 777 = written past num_records)\n", n - 1, h[n - 1], n, h[n], n);
  printf("load  negative voffset + positive soffset: lane 0 -> %.0f, lane 16 -> %.0f (0 0 = no wrap: out of range; 0 0.. / 0 %d = wrapped)\n", r[64], r[64 + 16], 0);
  return 0;
}
