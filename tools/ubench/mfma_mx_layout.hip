// Micro-benchmark / probe: operand semantics of gfx950's block-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4, the path to running
// the SECOND product of the fp16x2 precision mode (activation x weight-lo: needs 3-4 significant bits, oracle/second_product_numerics.py) at 2x
// (fp8) or 4x (fp4) the fp16 matrix rate. Before a kernel is built on it, three things have to be facts, not readings of a manual:
//   1. pairing: does lane (i, h) of A meet lane (j, h) of B ELEMENT BY ELEMENT (D[i][j] = sum_h sum_e A_lane(i,h)[e] * B_lane(j,h)[e], e < 32)?
//      Then any K order is fine as long as both operands are packed the same way, and the C/D map is the 32x32 one.
//   2. formats: fp8 e4m3 (cbsz / blgp = 0) in 8 VGPRs, fp4 e2m1 (= 4) in the first 4 VGPRs, mixed fp8 x fp4;
//   3. scales: one E8M0 byte per lane (its 32 elements), value 2^(byte - 127), byte picked by op_sel.
// Plus the issue rate of the three forms against v_mfma_f32_32x32x16_f16.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_mx_layout.hip -o /tmp/mfma_mx_layout && /tmp/mfma_mx_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FA, int FB>
__global__ void probe(const v8i* a, const v8i* b, const int* sa, const int* sb, v16f* d, int opsel_a) {
  v16f acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int l = threadIdx.x;
  if (opsel_a == 0) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, 0, sa[l], 0, sb[l]);
  else acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, 1, sa[l], 0, sb[l]);
  d[l] = acc;
}

// issue rate: 4 independent accumulators, `iters` rounds; WAVES waves per workgroup, one workgroup per CU
template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters, int seed) {
  v16f acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = 0x38383838 + seed;   // fp8 ones / harmless fp4 values
    b[i] = 0x38383838 - seed;
  }
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) {
    ah[i] = (_Float16)(1.0f + seed);
    bh[i] = (_Float16)(1.0f - seed);
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (MODE == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
      else if constexpr (MODE == 1) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, 127, 0, 127);
      else if constexpr (MODE == 2) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 4, 4, 0, 127, 0, 127);
      else acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 4, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static const float FP4_GRID[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static uint8_t enc_fp8(int v) {   // small integers, e4m3 (bias 7): 1 = 0x38, 2 = 0x40, 3 = 0x44, 4 = 0x48
  static const uint8_t mag[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
  return (uint8_t)(mag[std::abs(v)] | (v < 0 ? 0x80 : 0));
}

struct Operand {
  std::vector<float> val;     // [64 lanes][32 elements]
  std::vector<v8i> regs;      // packed
};
static Operand make(int fmt, unsigned seed) {
  Operand o;
  o.val.resize(64 * 32);
  o.regs.resize(64);
  srand(seed);
  for (int l = 0; l < 64; ++l) {
    uint8_t bytes[32] = {0};
    for (int e = 0; e < 32; ++e) {
      if (fmt == 0) {
        const int v = rand() % 9 - 4;
        o.val[l * 32 + e] = (float)v;
        bytes[e] = enc_fp8(v);
      } else {
        const int code = rand() % 16;   // sign | 3-bit magnitude code; element e in nibble (e & 1) of byte e >> 1
        o.val[l * 32 + e] = (code & 8 ? -1.f : 1.f) * FP4_GRID[code & 7];
        bytes[e >> 1] |= (uint8_t)(code << (4 * (e & 1)));
      }
    }
    for (int i = 0; i < 8; ++i) o.regs[l][i] = (int)(bytes[4 * i] | (bytes[4 * i + 1] << 8) | (bytes[4 * i + 2] << 16) | ((uint32_t)bytes[4 * i + 3] << 24));
  }
  return o;
}

template <int FA, int FB>
static void run_probe(const char* name, const std::vector<int>& sa, const std::vector<int>& sb, int opsel_a, const std::vector<float>& row_scale) {
  Operand A = make(FA == 4 ? 4 : 0, 1234 + FA), B = make(FB == 4 ? 4 : 0, 99 + FB);
  v8i *da, *db;
  int *dsa, *dsb;
  v16f* dd;
  hipMalloc(&da, 64 * sizeof(v8i));
  hipMalloc(&db, 64 * sizeof(v8i));
  hipMalloc(&dsa, 64 * 4);
  hipMalloc(&dsb, 64 * 4);
  hipMalloc(&dd, 64 * sizeof(v16f));
  hipMemcpy(da, A.regs.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice);
  hipMemcpy(db, B.regs.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice);
  hipMemcpy(dsa, sa.data(), 64 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsb, sb.data(), 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((probe<FA, FB>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd, opsel_a);
  std::vector<v16f> D(64);
  hipMemcpy(D.data(), dd, 64 * sizeof(v16f), hipMemcpyDeviceToHost);
  // hypothesis: D[i][j] = row_scale[i] * sum_h sum_e A_lane(i + 32 h)[e] * B_lane(j + 32 h)[e]; C/D map of the 32x32 shapes
  double worst = 0, worst_swapped = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      double want = 0, want_sw = 0;
      for (int h = 0; h < 2; ++h)
        for (int e = 0; e < 32; ++e) {
          want += (double)A.val[(i + 32 * h) * 32 + e] * B.val[(j + 32 * h) * 32 + e];
          want_sw += (double)A.val[(j + 32 * h) * 32 + e] * B.val[(i + 32 * h) * 32 + e];   // row <-> column swapped
        }
      worst = fmax(worst, fabs(D[l][r] - want * row_scale[i]));
      worst_swapped = fmax(worst_swapped, fabs(D[l][r] - want_sw * row_scale[j]));
    }
  printf("%-46s max |D - elementwise pairing| = %-10.4g (row<->col swapped: %.4g)  D[0][0..3] = %g %g %g %g\n", name, worst, worst_swapped, D[0][0], D[1][0],
         D[2][0], D[3][0]);
  hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
}

template <int MODE>
static void run_rate(const char* name, double flops_per_inst) {
  int dev = 0;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount, iters = 20000;
  float* out;
  hipMalloc(&out, cus * 256 * 4);
  hipLaunchKernelGGL((rate<MODE>), dim3(cus), dim3(256), 0, 0, out, 100, 0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate<MODE>), dim3(cus), dim3(256), 0, 0, out, iters, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)cus * 4 * iters * 4;   // 4 waves (one per SIMD) x 4 MFMAs per round
  printf("%-46s %8.3f ms  %6.1f ns per instruction and SIMD  %8.1f TFLOP/s\n", name, ms, ms * 1e6 / ((double)iters * 4), insts * flops_per_inst / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  std::vector<int> one(64, 127), two_odd(64), bytes(64);
  std::vector<float> rs1(32, 1.f), rs2(32);
  for (int l = 0; l < 64; ++l) {
    two_odd[l] = (l & 1) ? 128 : 127;                 // rows i odd scaled by 2 (both lane halves of the row)
    bytes[l] = 127 | (((l & 1) ? 129 : 127) << 8);    // byte 0 = 1.0; byte 1 = 4.0 for odd rows: op_sel = 1 must pick byte 1
  }
  for (int i = 0; i < 32; ++i) rs2[i] = (i & 1) ? 2.f : 1.f;
  std::vector<float> rs4(32);
  for (int i = 0; i < 32; ++i) rs4[i] = (i & 1) ? 4.f : 1.f;
  run_probe<0, 0>("fp8 x fp8, scales 2^0", one, one, 0, rs1);
  run_probe<4, 4>("fp4 x fp4, scales 2^0", one, one, 0, rs1);
  run_probe<0, 4>("fp8 (A) x fp4 (B), scales 2^0", one, one, 0, rs1);
  run_probe<0, 0>("fp8 x fp8, scale A = 2 on odd rows", two_odd, one, 0, rs2);
  run_probe<0, 0>("fp8 x fp8, op_sel A = 1 -> byte 1 (4 on odd rows)", bytes, one, 1, rs4);
  run_probe<4, 4>("fp4 x fp4, scale A = 2 on odd rows", two_odd, one, 0, rs2);
  run_rate<0>("rate: v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16);
  run_rate<1>("rate: mfma_scale 32x32x64 fp8 x fp8", 2.0 * 32 * 32 * 64);
  run_rate<2>("rate: mfma_scale 32x32x64 fp4 x fp4", 2.0 * 32 * 32 * 64);
  run_rate<3>("rate: mfma_scale 32x32x64 fp8 x fp4", 2.0 * 32 * 32 * 64);
  return 0;
}
