// Probe: semantics of gfx950's v_cvt_scalef32_pk_fp4_f16 / _f32 (two values -> two e2m1 nibbles in byte `sel` of the destination) - what the
// "fp16q4" precision mode needs to convert a wave's own A fragment registers for the block-scaled second product:
//   * scale direction: is the result fp4(v / scale) or fp4(v * scale)?   * rounding at the grid's midpoints (0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5)
//   * saturation above 6   * which source lands in the low nibble   * are the other bytes of the destination preserved?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/cvt_fp4_probe.hip -o /tmp/cvt_fp4_probe && /tmp/cvt_fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

__global__ void k(const float* x, float scale, unsigned* out) {
  const int i = threadIdx.x;
  h2 v;
  v.x = (_Float16)x[2 * i];
  v.y = (_Float16)x[2 * i + 1];
  out[3 * i] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0u, v, scale, 0);
  out[3 * i + 1] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(0xffffffffu, v, scale, 2);
  out[3 * i + 2] = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(0u, x[2 * i], x[2 * i + 1], scale, 1);
}

int main() {
  static const float GRID[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  const float vals[] = {0.1f, 0.25f, 0.26f, 0.5f, 0.75f, 1.0f, 1.25f, 1.5f, 1.75f, 2.0f, 2.5f, 3.0f, 3.5f, 4.0f, 5.0f, 6.0f, 7.0f, 100.f, -0.75f, -1.25f, -3.0f, -9.0f, 0.0f, 0.24f};
  const int n = sizeof(vals) / sizeof(vals[0]);   // 24 -> 12 lanes; (v[2i], v[2i+1]) = (x, y)
  float* dx;
  unsigned* dout;
  hipMalloc(&dx, 64 * 2 * 4);
  hipMalloc(&dout, 64 * 3 * 4);
  std::vector<float> hx(128, 0.f);
  for (int i = 0; i < n; ++i) hx[i] = vals[i];
  hipMemcpy(dx, hx.data(), 128 * 4, hipMemcpyHostToDevice);
  for (float scale : {1.0f, 2.0f, 0.25f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, scale, dout);
    std::vector<unsigned> o(192);
    hipMemcpy(o.data(), dout, 192 * 4, hipMemcpyDeviceToHost);
    printf("scale %g\n", scale);
    for (int i = 0; i < n / 2; ++i) {
      const unsigned r0 = o[3 * i], r1 = o[3 * i + 1], r2 = o[3 * i + 2];
      const unsigned lo = r0 & 15, hi = (r0 >> 4) & 15;
      printf("  f16 (x = %-6g y = %-6g) -> word %08x: low nibble %x = %s%g, high nibble %x = %s%g | sel=2 into ffffffff: %08x | f32 form sel=1: %08x\n", vals[2 * i],
             vals[2 * i + 1], r0, lo, (lo & 8) ? "-" : "", GRID[lo & 7], hi, (hi & 8) ? "-" : "", GRID[hi & 7], r1, r2);
    }
  }
  return 0;
}
