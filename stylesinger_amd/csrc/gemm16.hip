// Small-K fp32 GEMM on 16x16x4 MFMA tiles for the SINGLE-ROUND launches of the denoiser loops: the residual half of the output
// projection  x <- (x + g . W_res^T + b) / sqrt(2)  (modules/diff/net.py:75-77, deferred-skip form: N = C columns), 3000 launches per
// BASELINE-config-2 step with K = N = 256 (mel) or 192 (f0 pair).
//
// Why not the generic conv_gemm_kernel<64,64> it replaces there (round 2: 22.6 us per launch for 10 us of matrix time):
//   * tile count: 64-row tiles give 768 workgroups for the mel launch and 1152 for the f0 pair - on 768 resident slots the f0 launch
//     runs 1.5 rounds. Here a workgroup is 16*MT rows x 64 columns (4 waves x 16 columns); MT = 6 (96 rows) makes an 8 s utterance
//     exactly 16 row tiles: mel 512 workgroups (2 per CU), f0 pair 768 (3 per CU), one balanced round each.
//   * a wave's weight slice (16 columns x K) goes global -> registers directly, two chunks ahead through 3 register stages, issued
//     right behind the A pieces of the same chunk (the VMEM counter retires in order: fetched up front they all had to land before
//     the first MFMA - 3.3 of 22 us in the ablation); no LDS for B.
//   * the A tile goes global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPRs, no ds_write, no VALU): each DMA instruction
//     moves 8 rows x 128 B (full lines); the LDS image is lane-linear, so the XOR slot swizzle is applied on the SOURCE address.
//     A ring of 3 K chunks is in flight with counted s_waitcnt vmcnt + a raw s_barrier per chunk (a __syncthreads() would drain the
//     DMA queue: cdna_hip_programming.md "Pipelining across barriers").
//   * epilogue operands (the residual stream tile) are fetched at kernel entry, addresses are one VGPR base + SGPR offsets, stores are
//     buffer stores (rows >= T dropped by the range check).
// Arithmetic: exact fp32 products, fp32 accumulation; the K order inside an accumulator differs from the 32x32x2 kernel
// (tests/test_gpu_round3.py compares with torch fp32 and with ss_conv_gemm).
#include "common.h"
#include "../../include/stylesinger_hip.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// SS_R16_ABL (debug builds only, tools/ablate_r16.sh; results are wrong by design), gemm16_res_kernel: 1 = no weight preload, 2 = no A
// DMA after the first two chunks, 3 = no MFMAs, 4 = no residual-stream loads, 5 = no output stores, 6 = no barriers in the loop
#ifndef SS_R16_ABL
#define SS_R16_ABL 0
#endif

namespace {

constexpr int BK = 32;
constexpr int LD = BK;
constexpr int BN = 64;
constexpr int NBUF = 3;

// same 16-byte slot swizzle as the 16x16 gate kernel (conflict-free ds_read_b128 for lanes (row = l & 15, slots 2(l>>4), 2(l>>4)+1))
__device__ __forceinline__ int swz16(int row) { return ((row >> 1) & 7) ^ ((((row >> 2) ^ (row >> 3)) & 1) << 1); }

// s_waitcnt vmcnt(N), other counters untouched (gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4, expcnt imm[6:4], lgkmcnt imm[11:8])
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// LDS-DMA of 64 x 16 bytes: lane i's 16 bytes land at lds_dst + 16 i (wave-uniform destination, per-lane source offset).
// (A __device__ helper on purpose: with the builtin written directly inside the templated __global__ body, the host pass of hipcc
//  (ROCm 7.2) silently drops the kernel's launch stub and the library no longer links.)
template <int AUX = 0>   // cache-policy bits (0 in the product; 16 = sc1: reads past this CU's L1, for operands another workgroup published write-through)
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, AUX);
}

// WL = weight layout: false = packed rows [Np][Kp]; true = the lane-contiguous repack of ss_pack_gemm16_weights
// ([n tile][wave][K chunk][half][lane][4 floats]): one fetch instruction of a wave = 1 KB contiguous instead of 16 columns x 64 B
// The kernel body as a device function of (workgroup id, LDS base of 3 * 16 MT * LD floats): the __global__ wrapper below passes blockIdx.x and its
// static LDS; the dataflow experiment of fused_gate_res.hip (round 5) calls the same body from a launch that also holds the gate's workgroups.
template <int MT, int KCH, bool WL, int A_AUX = 0>
__device__ __forceinline__ void gemm16_res_body(const ss_conv_gemm_args& a, const float* __restrict__ W16, int m_tiles_per_item, int m_tiles, int n_tiles,
                                                const int block_id, float* __restrict__ lds_) {
  constexpr int BM = 16 * MT;
  constexpr int GROUPS = BM / 8;                 // DMA instructions per chunk (8 rows x 128 B each)
  constexpr int DPW = (GROUPS + 3) / 4;          // ... per wave
  static_assert(GROUPS % 4 == 0, "row groups must split evenly over the 4 waves");
  float* const As0 = static_cast<float*>(__builtin_assume_aligned(lds_, 16));
  float* const As1 = As0 + BM * LD;
  float* const As2 = As1 + BM * LD;

  const int id = block_id;
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int mt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (mt >= m_tiles) return;
  const int b = mt / m_tiles_per_item;
  const int t0 = (mt % m_tiles_per_item) * BM;
  const int n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, kg = lane >> 4;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;

  auto uniform_ptr = [](const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr((WL ? W16 : a.W) + (int64_t)grp_w * a.w_group_stride), 0, __builtin_amdgcn_readfirstlane(a.Np * a.Kp * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<float*>(a.R) + (int64_t)b * a.r_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldr * 4)),
      0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);

  // ---- the wave's weight slice: column n0 + 16 w + lc, K floats [32 j + 8 kg, +8) of every chunk j -> registers, once
  const int col = n0 + 16 * wave + lc;
  const int w_voff = WL ? ((nt * 4 + wave) * KCH * 512) * 4 + lane * 16 : (col * a.Kp + kg * 8) * 4;
  // streamed two chunks ahead through a ring of 3 register stages, issued right after the A pieces of the same chunk: the VMEM
  // counter retires in order, so weights fetched up front would all have to land before the first MFMA (ablation: 3.3 of 22 us)
  float4 bw[3][2];
  auto load_w = [&](auto jtag) {
    constexpr int j = decltype(jtag)::value;
    if constexpr (SS_R16_ABL == 1) {
      bw[j % 3][0] = make_float4(0.01f * lane, 0.02f, 0.03f * j, 0.04f);
      bw[j % 3][1] = make_float4(0.05f, 0.06f * lane, 0.07f, 0.08f * j);
    } else {
      if constexpr (WL) {
        bw[j % 3][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, j * 2048, 0));
        bw[j % 3][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, j * 2048 + 1024, 0));
      } else {
        bw[j % 3][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff, j * (BK * 4), 0));
        bw[j % 3][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff + 16, j * (BK * 4), 0));
      }
    }
  };
  constexpr int WLD = SS_R16_ABL == 1 ? 0 : 2;   // VMEM instructions of one weight stage
  // ---- epilogue operands: residual-stream tile (rows 16 m + 4 kg + r, column col) and the bias
  const bool col_ok = col < a.N;
  const int oob = col_ok ? 0 : (int)0x80000000;
  const int r_base = ((t0 + 4 * kg) * a.ldr + col) * 4 + oob;
  float rv[MT][4];   // fetched inside the loop (chunk KCH-3)
  // the bias travels with them (a buffer load with an empty range when there is none): fetched through a pointer at kernel entry it
  // put a full memory round trip in front of the first A piece
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.bias ? a.bias + (int64_t)grp_w * a.bias_group_stride : a.W), 0, __builtin_amdgcn_readfirstlane(a.bias ? a.N * 4 : 0), 0x00020000);
  float bs = 0.f;

  // ---- A tile by LDS-DMA. DMA instruction (wave w, j): rows 8 (w + 4 j) .. +8 of the tile; lane i lands at byte 16 i of that 1-KiB
  // piece = (row i >> 3, physical slot i & 7), so it FETCHES logical slot (i & 7) ^ swz16(row).
  int a_voff[DPW];
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int row = 8 * (wave + 4 * j) + (lane >> 3);
    a_voff[j] = ((t0 + row) * a.lda + (((lane & 7) ^ swz16(row)) << 2)) * 4;   // rows >= len are out of range: the DMA writes zeros
  }
  auto dma = [&](float* buf, int c) {
#pragma unroll
    for (int j = 0; j < DPW; ++j)
      glds16<A_AUX>(rsrc_a, buf + (wave + 4 * j) * 8 * LD, a_voff[j], c * (BK * 4));
  };
  int a_rd[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h) a_rd[m][h] = (16 * m + lc) * LD + (((2 * kg + h) ^ swz16(16 * m + lc)) << 2);

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;

  float* const bufs[NBUF] = {As0, As1, As2};
  __builtin_amdgcn_sched_barrier(0);   // the counted waits below rely on the DMA pieces being issued in chunk order
  dma(bufs[0], 0);
  load_w(std::integral_constant<int, 0>{});
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KCH > 1) {
    dma(bufs[1], 1);
    load_w(std::integral_constant<int, 1>{});
  }
  __builtin_amdgcn_sched_barrier(0);
  // chunk c: wait for MY pieces of chunk c (the DPW instructions of chunk c+1 may stay in flight), barrier (everyone's pieces of chunk c
  // have landed; everyone is done reading chunk c-1), refill the slot chunk c-1 used with chunk c+2, fragments, MFMAs.
  auto chunk = [&](auto ctag) {
    constexpr int c = decltype(ctag)::value;
    // outstanding VMEM ops younger than my pieces and weights of chunk c: the DPW pieces + WL weight loads of chunk c+1, plus - in chunk
    // RC+1 - the 4 MT residual loads and the bias load
    constexpr int RC = KCH >= 3 ? KCH - 3 : 0;   // the chunk that issues the residual-stream loads (before its DMA)
    if constexpr (c + 1 >= KCH) wait_vmcnt<0>();
    else if constexpr (c == RC + 1 && KCH >= 3) wait_vmcnt<DPW + WLD + 1 + (SS_R16_ABL == 4 ? 0 : 4 * MT)>();
    else wait_vmcnt<DPW + WLD>();
    if constexpr (SS_R16_ABL != 6) __builtin_amdgcn_s_barrier();
    const float* Ac = bufs[c % NBUF];
    float4 af[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      af[m][0] = *reinterpret_cast<const float4*>(Ac + a_rd[m][0]);
      af[m][1] = *reinterpret_cast<const float4*>(Ac + a_rd[m][1]);
    }
    __builtin_amdgcn_sched_barrier(0);   // all fragment reads of the chunk are in flight before anything else issues
    if constexpr (c == RC) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          rv[m][r] = SS_R16_ABL == 4 ? 0.5f * r : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, r_base, (16 * m + r) * a.ldr * 4, 0));
      bs = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_b, col * 4, 0, 0));   // col >= N: out of range -> 0
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (c + 2 < KCH) {
      if constexpr (SS_R16_ABL != 2) dma(bufs[(c + 2) % NBUF], c + 2);
      load_w(std::integral_constant<int, c + 2>{});
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SS_R16_ABL == 3) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][0] += af[m][0].x * bw[c % 3][0].x + af[m][1].w * bw[c % 3][1].w;
      return;
    }
    // MT independent accumulators per K step: consecutive MFMAs never touch the same one (40-cycle dependent latency, 32-cycle issue)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4 bf = bw[c % 3][h];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].x, bf.x, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].y, bf.y, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].z, bf.z, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].w, bf.w, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto run = [&](auto... cs) { (chunk(cs), ...); };
  if constexpr (KCH == 6)
    run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
        std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
  else
    run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{},
        std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{});

  // ---- epilogue: x <- (x + v + bias) * post_scale, rows >= len written as 0 (mask_rows), rows >= T dropped by the range check
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const int c_base = ((t0 + 4 * kg) * a.ldc + col) * 4 + oob;
  const bool interior = t0 + BM <= row_lim;   // block-uniform
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float o = (rv[m][r] + (acc[m][r] + bs)) * a.post_scale;
      if (!interior && t0 + 16 * m + 4 * kg + r >= row_lim) o = 0.f;
      if (SS_R16_ABL == 5 && o != 123456.789f) continue;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsrc_c, c_base, (16 * m + r) * a.ldc * 4, 0);
    }
}

template <int MT, int KCH, bool WL>
__global__ __launch_bounds__(256, (MT >= 8 ? 2 : 3)) void gemm16_res_kernel(const ss_conv_gemm_args a, const float* __restrict__ W16, int m_tiles_per_item,
                                                                           int m_tiles, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float As[3 * 16 * MT * LD];
  gemm16_res_body<MT, KCH, WL>(a, W16, m_tiles_per_item, m_tiles, n_tiles, (int)blockIdx.x, As);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Streaming form for large K (the K = L*C skip GEMM of the deferred-skip loops: 5120 / 1920): C = act(A . W^T + bias).
// Both operands arrive by LDS-DMA, nothing is staged through VGPRs and the loop has no VALU work at all:
//   A ring: 2 buffers of [16 MT rows][32 floats], filled by all four waves (the tile is shared);
//   B ring: 2 buffers of [16 columns][32 floats] PER WAVE (a wave only ever reads its own 16 columns, so its pieces need no barrier).
// Chunk c: wait for my pieces of chunk c (vmcnt 0: nothing younger is in flight), barrier, issue the DMA of chunk c+1 into the buffers
// chunk c-1 used, read the fragments of chunk c, 8 MT MFMAs. One chunk of MFMA time (x the waves sharing the SIMD) covers the DMA latency.
// Split-K form (ksplit > 1; launches that leave most CUs without a workgroup - one short utterance, the K = L*C skip GEMM is then 48
// workgroups x 160 K chunks): workgroup (tile, s) runs K chunks [s * kchunks / ksplit, (s + 1) * kchunks / ksplit) and stores its raw partial
// sums to P[s][b][t][n] (no bias, no activation); splitk_reduce_kernel adds the ksplit partials in a FIXED order (deterministic: graph replay
// and eager runs stay bit-identical) and applies bias / activation / row mask.
template <int MT>
__global__ __launch_bounds__(256, 3) void gemm16_store_kernel(const ss_conv_gemm_args a, int m_tiles_per_item, int m_tiles, int n_tiles, int ksplit,
                                                              float* __restrict__ P) {
  constexpr int BM = 16 * MT;
  constexpr int GROUPS = BM / 8;
  constexpr int DPW = GROUPS / 4;
  static_assert(GROUPS % 4 == 0, "row groups must split evenly over the 4 waves");
  __shared__ __attribute__((aligned(16))) float A0[BM * LD];
  __shared__ __attribute__((aligned(16))) float A1[BM * LD];
  __shared__ __attribute__((aligned(16))) float B0[4 * 16 * LD];
  __shared__ __attribute__((aligned(16))) float B1[4 * 16 * LD];

  const int tiles_grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
  const int id = blockIdx.x % tiles_grid;
  const int ksi = blockIdx.x / tiles_grid;   // split-K slice (0 when ksplit == 1)
  const int grp = id / (8 * n_tiles);
  const int rem = id % (8 * n_tiles);
  const int mt = grp * 8 + (rem & 7);
  const int nt = rem >> 3;
  if (mt >= m_tiles) return;
  const int b = mt / m_tiles_per_item;
  const int t0 = (mt % m_tiles_per_item) * BM;
  const int n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lc = lane & 15, kg = lane >> 4;
  const int len = ss_uniform_len(a.lens, b, a.T);
  const int grp_w = a.group_size > 0 ? b / a.group_size : 0;
  const int kch_all = a.Kp / BK;
  const int kbeg = (int)((long)ksi * kch_all / ksplit), kchunks = (int)((long)(ksi + 1) * kch_all / ksplit);   // this slice: chunks [kbeg, kchunks)

  auto uniform_ptr = [](const float* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<float*>(((uint64_t)hi << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.A + (int64_t)b * a.a_batch_stride), 0, __builtin_amdgcn_readfirstlane(len * a.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.W + (int64_t)grp_w * a.w_group_stride), 0, __builtin_amdgcn_readfirstlane(a.Np * a.Kp * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.C + (int64_t)b * a.c_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldc * 4)), 0x00020000);

  int a_voff[DPW], b_voff[2];
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int row = 8 * (wave + 4 * j) + (lane >> 3);
    a_voff[j] = ((t0 + row) * a.lda + (((lane & 7) ^ swz16(row)) << 2)) * 4;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {   // the wave's 16 weight rows (= output columns) in two pieces of 8 rows x 128 B; rows >= Np read 0
    const int row = 8 * j + (lane >> 3);
    b_voff[j] = ((n0 + 16 * wave + row) * a.Kp + (((lane & 7) ^ swz16(row)) << 2)) * 4;
  }
  auto dma = [&](float* Abuf, float* Bbuf, int c) {
    const int so = c * (BK * 4);
#pragma unroll
    for (int j = 0; j < DPW; ++j) glds16(rsrc_a, Abuf + (wave + 4 * j) * 8 * LD, a_voff[j], so);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16(rsrc_w, Bbuf + (wave * 16 + 8 * j) * LD, b_voff[j], so);
  };
  const int rd0 = lc * LD + (((2 * kg) ^ swz16(lc)) << 2), rd1 = lc * LD + (((2 * kg + 1) ^ swz16(lc)) << 2);

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[m][r] = 0.f;

  auto chunk = [&](const float* Ac, const float* Bc, float* An, float* Bn, int c, bool more) {
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    float4 af[MT][2], bf[2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      af[m][0] = *reinterpret_cast<const float4*>(Ac + 16 * m * LD + rd0);
      af[m][1] = *reinterpret_cast<const float4*>(Ac + 16 * m * LD + rd1);
    }
    bf[0] = *reinterpret_cast<const float4*>(Bc + wave * 16 * LD + rd0);
    bf[1] = *reinterpret_cast<const float4*>(Bc + wave * 16 * LD + rd1);
    __builtin_amdgcn_sched_barrier(0);
    if (more) dma(An, Bn, c + 1);   // wave-uniform branch
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].x, bf[h].x, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].y, bf[h].y, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].z, bf[h].z, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][h].w, bf[h].w, acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  dma(A0, B0, kbeg);
  int c = kbeg;
  for (; c + 2 <= kchunks; c += 2) {
    chunk(A0, B0, A1, B1, c, true);
    chunk(A1, B1, A0, B0, c + 1, c + 2 < kchunks);
  }
  if (c < kchunks) chunk(A0, B0, A1, B1, c, false);

  const int col = n0 + 16 * wave + lc;
  const bool col_ok = col < a.N;
  const int oob = col_ok ? 0 : (int)0x80000000;
  if (ksplit > 1) {   // raw partial sums -> P[ksi][b][t][col] (rows >= T dropped by the range check)
    const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(P + ((int64_t)ksi * a.B + b) * a.T * a.N), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.N * 4)), 0x00020000);
    const int p_base = ((t0 + 4 * kg) * a.N + col) * 4 + oob;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[m][r];   // (bit_cast straight from the vector element stored element 0 four times: hipcc 7.2)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_p, p_base, (16 * m + r) * a.N * 4, 0);
      }
    return;
  }
  const float bs = (a.bias && col_ok) ? a.bias[(int64_t)grp_w * a.bias_group_stride + col] : 0.f;
  const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;
  const int c_base = ((t0 + 4 * kg) * a.ldc + col) * 4 + oob;
  const bool relu = a.act == SS_ACT_RELU;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float o = acc[m][r] + bs;
      if (relu) o = fmaxf(o, 0.f);
      if (t0 + 16 * m + 4 * kg + r >= row_lim) o = 0.f;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsrc_c, c_base, (16 * m + r) * a.ldc * 4, 0);
    }
}

// C[b][t][n] = act(bias[n] + sum_s P[s][b][t][n]) in slice order s = 0, 1, ...; rows >= lens[b] -> 0 when mask_rows
__global__ void splitk_reduce_kernel(const float* __restrict__ P, const ss_conv_gemm_args a, int ksplit) {
  const int64_t per = (int64_t)a.B * a.T * a.N;
  const int n4 = a.N / 4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per / 4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % n4) * 4;
    const int64_t r = i / n4;
    const int b = (int)(r / a.T), t = (int)(r % a.T);
    float4 v = *reinterpret_cast<const float4*>(P + i * 4);
    for (int s = 1; s < ksplit; ++s) {
      const float4 u = *reinterpret_cast<const float4*>(P + s * per + i * 4);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (a.bias) {
      const float* bp = a.bias + (a.group_size > 0 ? (int64_t)(b / a.group_size) * a.bias_group_stride : 0) + c4;
      v.x += bp[0]; v.y += bp[1]; v.z += bp[2]; v.w += bp[3];
    }
    if (a.act == SS_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    if (a.mask_rows && a.lens && t >= a.lens[b]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(a.C + (int64_t)b * a.c_batch_stride + (int64_t)t * a.ldc + c4) = v;
  }
}

template <int MT>
int launch_store(const ss_conv_gemm_args& a, int ksplit, float* P, hipStream_t stream, bool reduce = true) {
  constexpr int BM = 16 * MT;
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const int n_tiles = ss_cdiv(a.N, BN);
  const int grid = ss_cdiv(m_tiles, 8) * 8 * n_tiles;
  hipLaunchKernelGGL(gemm16_store_kernel<MT>, dim3(grid * ksplit), dim3(256), 0, stream, a, m_tiles_per_item, m_tiles, n_tiles, ksplit, P);
  if (ksplit > 1 && reduce) {
    const int64_t n = (int64_t)a.B * a.T * a.N / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, stream, P, a, ksplit);
  }
  return 0;
}

template <int MT, int KCH>
int launch_res(const ss_conv_gemm_args& a, const float* W16, hipStream_t stream) {
  constexpr int BM = 16 * MT;
  const int m_tiles_per_item = ss_cdiv(a.T, BM);
  const int m_tiles = m_tiles_per_item * a.B;
  const int n_tiles = ss_cdiv(a.N, BN);
  const int grid = ss_cdiv(m_tiles, 8) * 8 * n_tiles;
  if (W16) hipLaunchKernelGGL((gemm16_res_kernel<MT, KCH, true>), dim3(grid), dim3(256), 0, stream, a, W16, m_tiles_per_item, m_tiles, n_tiles);
  else hipLaunchKernelGGL((gemm16_res_kernel<MT, KCH, false>), dim3(grid), dim3(256), 0, stream, a, W16, m_tiles_per_item, m_tiles, n_tiles);
  return 0;
}

// [Np][Kp] fp32 -> [n tile (64 columns)][wave][K chunk][half][lane][4 floats]: lane = kg * 16 + lc holds column 64 nt + 16 w + lc,
// K elements 32 j + 8 kg + 4 half + (0..3)
__global__ void pack_gemm16_kernel(const float* __restrict__ src, float* __restrict__ dst, int Np, int Kp) {
  const int64_t n = (int64_t)Np * Kp;
  const int kch = Kp / BK;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % Kp), col = (int)(i / Kp);
    const int nt = col / BN, w = (col % BN) / 16, lc = col % 16;
    const int j = kk / BK, kg = (kk % BK) / 8, h = (kk % 8) / 4, e = kk % 4;
    dst[((((int64_t)(nt * 4 + w) * kch + j) * 2 + h) * 64 + kg * 16 + lc) * 4 + e] = src[i];
  }
}

}  // namespace

#ifndef SS_FUSED_TU   // fused_gate_res.hip includes this file for the kernel bodies above only

// row-tile count (16*mt rows per workgroup) for a launch of B items x T rows x N columns: fewest workgroup layers per CU x rows per tile
extern "C" int ss_gemm16_pick(int B, int T, int N) {
  const int n_tiles = ss_cdiv(N, BN);
  int best = 4;
  long best_cost = -1;
  // 32-row tiles only for launches whose 64-row grid leaves at least half the CUs without a workgroup (one short utterance)
  const bool tiny = (long)ss_cdiv(T, 64) * B * n_tiles * 2 <= ss_n_cu();
  for (int mt = 8; mt >= (tiny ? 2 : 4); mt -= 2) {   // 128, 96, 64 (, 32) rows
    const long wgs = (long)ss_cdiv(T, 16 * mt) * B * n_tiles;
    const long cost = (long)mt * ss_cdiv(wgs, ss_n_cu());
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = mt;
    }
  }
  return best;
}

static int gemm16_res_impl(const ss_conv_gemm_args* args, const float* W16, int mt, void* stream, const char* who) {
  SS_CHECK_ARG(args != nullptr, "%s: null args", who);
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C && a.R, "%s: null A/W/C/R", who);
  SS_CHECK_ARG(a.ntaps == 1 && a.tap_off[0] == 0, "%s: one tap at offset 0 only", who);
  SS_CHECK_ARG(a.Kp == a.Cin && (a.Kp == 192 || a.Kp == 256) && (a.lda & 3) == 0, "%s: K=%d must be 192 or 256 (= Kp), lda %% 4 == 0", who, a.Cin);
  SS_CHECK_ARG(a.N > 0 && a.N <= a.Np && (a.Np & 15) == 0, "%s: bad N=%d Np=%d", who, a.N, a.Np);
  SS_CHECK_ARG(!W16 || (a.N % BN) == 0, "%s: the fetch-order weights need N %% 64 == 0 (N=%d)", who, a.N);
  SS_CHECK_ARG(a.a_scale == 1.0f && a.a_lrelu == 1.0f && a.a_bias == nullptr && a.mfma_bf16 == 0, "%s: no A prologue, fp32 only", who);
  SS_CHECK_ARG((int64_t)a.T * a.lda * 4 < (1ll << 31) && (int64_t)a.T * a.ldr * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 4 < (1ll << 31) &&
                   (int64_t)a.Np * a.Kp * 4 < (1ll << 31), "%s: item too large for 32-bit offsets", who);
  SS_CHECK_ARG(mt == 0 || mt == 2 || mt == 4 || mt == 6 || mt == 8, "%s: mt=%d must be 0 (auto), 2, 4, 6 or 8", who, mt);
  if (mt == 0) mt = ss_gemm16_pick(a.B, a.T, a.N);
  hipStream_t s = (hipStream_t)stream;
  const bool k6 = a.Kp == 192;
  switch (mt) {
    case 2: k6 ? launch_res<2, 6>(a, W16, s) : launch_res<2, 8>(a, W16, s); break;
    case 4: k6 ? launch_res<4, 6>(a, W16, s) : launch_res<4, 8>(a, W16, s); break;
    case 6: k6 ? launch_res<6, 6>(a, W16, s) : launch_res<6, 8>(a, W16, s); break;
    default: k6 ? launch_res<8, 6>(a, W16, s) : launch_res<8, 8>(a, W16, s); break;
  }
  SS_CHECK_LAUNCH(who);
  return SS_OK;
}

extern "C" int ss_gemm16_res(const ss_conv_gemm_args* args, int mt, void* stream) { return gemm16_res_impl(args, nullptr, mt, stream, "ss_gemm16_res"); }

extern "C" int ss_gemm16_resw(const ss_conv_gemm_args* args, const float* W16, int mt, void* stream) {
  SS_CHECK_ARG(W16 != nullptr, "ss_gemm16_resw: null W16");
  return gemm16_res_impl(args, W16, mt, stream, "ss_gemm16_resw");
}

extern "C" int ss_pack_gemm16_weights(const float* src, float* dst, int Np, int Kp, void* stream) {
  SS_CHECK_ARG(src && dst && src != dst && Np > 0 && (Np % BN) == 0 && Kp > 0 && (Kp % BK) == 0, "ss_pack_gemm16_weights: Np %% 64, Kp %% 32, out of place");
  const int64_t n = (int64_t)Np * Kp;
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_gemm16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, Np, Kp);
  SS_CHECK_LAUNCH("ss_pack_gemm16_weights");
  return SS_OK;
}

static int gemm16_store_impl(const ss_conv_gemm_args* args, int mt, int ksplit, float* partials, void* stream, const char* who, bool reduce = true) {
  SS_CHECK_ARG(args != nullptr, "%s: null args", who);
  const ss_conv_gemm_args& a = *args;
  SS_CHECK_ARG(a.A && a.W && a.C, "%s: null A/W/C", who);
  SS_CHECK_ARG(a.ntaps == 1 && a.tap_off[0] == 0, "%s: one tap at offset 0 only", who);
  SS_CHECK_ARG(a.Kp == a.Cin && (a.Kp % BK) == 0 && (a.lda & 3) == 0, "%s: K=%d must equal Kp and be a multiple of 32, lda %% 4 == 0", who, a.Cin);
  SS_CHECK_ARG(a.N > 0 && a.N <= a.Np, "%s: bad N=%d Np=%d", who, a.N, a.Np);
  SS_CHECK_ARG(a.a_scale == 1.0f && a.a_lrelu == 1.0f && a.a_bias == nullptr && a.mfma_bf16 == 0 && a.pre_scale == 1.0f && a.post_scale == 1.0f &&
                   a.R == nullptr && !a.accumulate && (a.act == SS_ACT_NONE || a.act == SS_ACT_RELU),
               "%s: plain C = act(A.W^T + bias) only (act none | relu), fp32", who);
  SS_CHECK_ARG((int64_t)a.T * a.lda * 4 < (1ll << 31) && (int64_t)a.T * a.ldc * 4 < (1ll << 31) && (int64_t)a.Np * a.Kp * 4 < (1ll << 31),
               "%s: item too large for 32-bit offsets", who);
  SS_CHECK_ARG(mt == 0 || mt == 4 || mt == 6 || mt == 8, "%s: mt=%d must be 0 (auto), 4, 6 or 8", who, mt);
  SS_CHECK_ARG(ksplit >= 1 && ksplit <= 16 && ksplit <= a.Kp / BK && (ksplit == 1 || (partials && (a.N & 3) == 0 && (a.ldc & 3) == 0)),
               "%s: ksplit=%d needs a partials buffer of ksplit*B*T*N floats, N %% 4 == 0 and at most Kp/32 slices", who, ksplit);
  if (mt == 0) mt = ksplit > 1 ? 4 : ss_gemm16_pick(a.B, a.T, a.N);
  if (mt == 2) mt = 4;   // (the 32-row tile exists for the residual projection only)
  hipStream_t s = (hipStream_t)stream;
  switch (mt) {
    case 4: launch_store<4>(a, ksplit, partials, s, reduce); break;
    case 6: launch_store<6>(a, ksplit, partials, s, reduce); break;
    default: launch_store<8>(a, ksplit, partials, s, reduce); break;
  }
  SS_CHECK_LAUNCH(who);
  return SS_OK;
}

extern "C" int ss_gemm16_store(const ss_conv_gemm_args* args, int mt, void* stream) {
  return gemm16_store_impl(args, mt, 1, nullptr, stream, "ss_gemm16_store");
}

extern "C" int ss_gemm16_store_splitk(const ss_conv_gemm_args* args, int mt, int ksplit, float* partials, void* stream) {
  return gemm16_store_impl(args, mt, ksplit, partials, stream, "ss_gemm16_store_splitk");
}

// library-internal (diffusion.hip): the split-K launch WITHOUT the reduction - the consumer (f0_tail_kernel / mel_tail_kernel) adds the slices itself,
// in the same order and with the same bias / ReLU / row mask as splitk_reduce_kernel
int ss_gemm16_store_partials(const ss_conv_gemm_args* args, int mt, int ksplit, float* partials, void* stream) {
  return gemm16_store_impl(args, mt, ksplit, partials, stream, "ss_gemm16_store_partials", false);
}

// K slices for a long-K launch of B items x T rows x N columns x K: 1 unless 64-row tiles leave most CUs without a workgroup; then as many
// slices as keep >= 8 K chunks per slice and at most ~one workgroup per CU
extern "C" int ss_gemm16_ksplit_pick(int B, int T, int N, int K) {
  const long wgs = (long)ss_cdiv(T, 64) * B * ss_cdiv(N, BN);
  const int n_cu = ss_n_cu();
  if (wgs * 2 > n_cu || K < 512) return 1;
  int s = (int)(n_cu / wgs);
  const int smax = K / BK / 8;
  if (s > smax) s = smax;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}
#endif  // SS_FUSED_TU
