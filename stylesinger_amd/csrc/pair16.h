// 16-bit operand terms of the split-operand matrix-core modes (ss_gemm_bf16_args.split), shared by gemm_bf16*.hip:
//   split = 1 ("bf16x2"): bf16 terms, every operand a (hi, mid) pair, three products hi*hi + hi*mid + mid*hi;
//   split = 2 ("fp16x2"): fp16 terms, only the WEIGHTS are pairs (of w * 2^shift, the shift undone by args.out_scale after the fp32
//                          accumulation), two products a*hi + a*lo; activations keep the pair LAYOUT (pairs interleaved by 32) but the matrix
//                          cores read their hi term only - the second term exists for the residual stream (22 significant bits).
// Both term types are 2 bytes, so every fetch / LDS / DMA plan is shared; only the MFMA opcode and the float <-> term conversions differ.
#pragma once
#include <stdint.h>

typedef float ss_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ss_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ss_f16x8 __attribute__((ext_vector_type(8)));

// one 32x32x16 matrix product on 8-term fragments held as raw 128-bit registers
template <bool F16>
__device__ __forceinline__ ss_f32x16 ss_mfma_32x32x16(ss_bf16x8 a, ss_bf16x8 b, ss_f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ss_f16x8, a), __builtin_bit_cast(ss_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// float -> term (round to nearest even, like torch's .bfloat16() / .half()) and back
template <bool F16>
__device__ __forceinline__ uint16_t ss_f2t(float x) {
  if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)x);
  else return __builtin_bit_cast(uint16_t, (__bf16)x);
}
template <bool F16>
__device__ __forceinline__ float ss_t2f(uint16_t h) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, h);
  else return __builtin_bit_cast(float, (uint32_t)h << 16);
}
// the two terms packed in one 32-bit word (element k = 0: low half, 1: high half)
template <bool F16>
__device__ __forceinline__ float ss_t2f_packed(uint32_t w, int k) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(k ? (w >> 16) : (w & 0xffffu)));
  else return __builtin_bit_cast(float, k ? (w & 0xffff0000u) : (w << 16));
}
