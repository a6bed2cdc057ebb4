// ONE launch per residual layer of the mel denoiser in "fp16x2" precision for many-round launches (BASELINE configs[3]: 32 x 5625 rows):
//   y = dilated_conv(x + dstep_l) + conditioner addend ; g = sigmoid(y[:C]) * tanh(y[C:]) ; x' = (x + W_res g + b) / sqrt(2)      (modules/diff/net.py:66-78)
// A workgroup owns 128 rows x ALL 512 pre-activation columns: the gate output G (128 x 256 fp16 = 64 KB) stays in LDS, the 1x1 residual
// projection runs from it, and only the skip operand (G, once) and the new residual stream leave the CU. Replaces gate128_kernel +
// tile256s_kernel<RESX> per layer (profiles/r05_bench_c4_fp16x2_20steps_kernel_stats.csv: 262 + 125 us): the projection launch's 461 MB of
// HBM traffic (G re-read, stream read + rewritten) and its 19 launches per step disappear.
//
// Shape of the kernel (what differs from the gate128 / gate256 family):
//   * 8 waves, wave w owns packed columns 64 w .. 64 w + 63 (= output channels 32 w .. + 31, both gate operands) of all 128 rows:
//     2 x 4 accumulator blocks of 32 x 32 (128 registers). The WEIGHT fragments of a wave are private to it, so they never touch LDS: they are
//     packed once in fragment order (ss_layer512_pack_gate / _res: one k-step of a wave = 4 KB contiguous) and stream L2 -> VGPR through a
//     register ring, 1 KB per instruction. Only the activations go through LDS.
//   * The WHOLE activation tile (128 + 2 x 8 halo rows x 256 channels, hi plane only: 72 KB) is staged once by LDS-DMA, so the 48 k-steps
//     (8 chunks x 3 taps x 2) run WITHOUT a barrier: the eight waves drift apart and each SIMD's two waves fill each other's issue gaps.
//   * Operand roles are swapped with respect to gate128: the weights are the matrix instruction's A operand, the activations its B operand.
//     An accumulator lane then holds ONE row and 16 channels in groups of 4 consecutive ones: G goes to LDS as 8-byte stores, the conditioner
//     addend arrives as 16-byte loads from a slab laid out in exactly this order (ss_layer512_tile_addend, once per forward), and the residual
//     epilogue reads / writes the stream's (hi, lo) fp16 pairs as 8-byte vectors straight from the accumulators (no staging pass).
//   * Persistent workgroups (one per CU, 144 KB of LDS = two 72 KB regions): region r holds A(tile i), then G(tile i) over it; the A tile of
//     tile i + 1 lands in the other region while tile i's epilogues run. Three barriers per tile.
//   * The residual stream has a layout of its own (nothing but this kernel and ss_layer512_entry touches it): H = fp16(x + dstep_l) - what the
//     conv's DMA fetches - in SLOT-MAJOR tiles [tile][slot 32][row 128] x 16 bytes (slot = 8 channels), and P = x itself in FP32 in accumulator
//     order (16 bytes per lane = 4 channels; 1 KB per wave instruction) for the epilogue's read-modify-write. The LDS image of a tile is
//     slot-major too, so the DMA moves contiguous kilobytes, the fragment reads are conflict-free without a swizzle, and the epilogue's 8-byte
//     H stores of a wave instruction tile 512 contiguous bytes. H is double buffered (Hin -> Hout): a tile reads 8 halo rows of its
//     neighbours, which another workgroup rewrites; P is updated in place. (History: v0 read and wrote ss_gemm_bf16's pair layout as 8-byte vectors, 32 rows per
//     instruction - the epilogue took 165 us per launch against the projection launch's 129; v1 kept the stream as an fp16 (hi, lo) pair in
//     accumulator order - 16 VALU instructions per element to unpack / re-split, 7.5 k cycles per tile, profiles/r06_trace_layer512_v1_fused.log.
//     The fp32 stream costs the same 4 bytes per element, 5 instructions, and is 2 bits MORE precise than the pair.)
// Arithmetic contract = ss_gemm_bf16 with split = 2: a * hi + a * lo of fp16 terms (weights = pairs of w * 2^s), fp32 accumulation scaled by
// out_scale; G = fp16(g) in the hi slots of the pair layout; the conv's operand fp16(x + dstep_l); the stream itself fp32 (the two-launch form
// keeps it as an fp16 pair = 22 bits). Results equal those of the two-launch form up to the
// fp32 summation order (tests/test_gpu_layer512.py: both against float64 of the same terms).
#include "common.h"
#include <stdlib.h>
#include "../../include/stylesinger_hip.h"
#include "pair16.h"
#include <type_traits>
#include <utility>

typedef ss_f32x16 f32x16;
typedef ss_bf16x8 bf16x8;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;                    // rows per tile
constexpr int HALO = 8;                    // dilations up to 8
constexpr int AROWS = BM + 2 * HALO;       // 144 staged rows
constexpr int SLOTB = AROWS * 16 + 16;     // bytes per LDS slot: 8 channels (16 B) of 144 rows + 16 B of padding (lanes that walk the slots of a row
                                           // are then 4 banks apart)
constexpr int REGION = 32 * SLOTB;         // 74 240 B; two regions
constexpr int H_TILE = BM * 512;           // 65 536 B of H per tile: [slot 32][row 128] x 16 B
constexpr int KSTEPS = 48;                 // 8 chunks x 3 taps x 2 k-steps of 16 channels
constexpr int RSTEPS = 16;                 // K = 256 of the residual projection
// NP = weight terms per element = matrix products per GEMM: 2 = "fp16x2" (hi, lo), 1 = "fp16sd" (one term; the rounding is noise-shaped over the
// loop's evaluations by cycling weight SETS, see ss_wavenet.n_wsets)
constexpr int wg_step(int NP) { return NP * 2048; }             // bytes of gate weights per wave and k-step: NP planes x 2 column blocks x 1 KB
constexpr int wg_wave(int NP) { return KSTEPS * wg_step(NP); }  // 196 608 B per wave (NP = 2)
constexpr int wr_step(int NP) { return NP * 1024; }
constexpr int wr_wave(int NP) { return RSTEPS * wr_step(NP); }  // 32 768 B per wave (NP = 2)
constexpr int E_TILE = BM * 512 * 4;       // 262 144 B of tiled addend per tile
constexpr int P_TILE = BM * 256 * 2;       // 65 536 B of the stream's residual term R per tile (fp16; the stream is x = (H - dstep_l) + R)
// E and P tiles: the eight waves' blocks of one (nb, m, q) step side by side ([..][q][wave 8][lane 64]: the workgroup reads 8 KB contiguous per step).
// (One 32 KB / 16 KB stream per wave, the first layout of round 6, measures the same: 255.5 / 351.6 against 256.4 / 350.7 us per launch,
// profiles/r06_kbench_layer512_phase_shift.log - the HBM channel hash copes with either.)
constexpr bool WAVE_MAJOR = false;
constexpr int nring(int NP) { return NP == 2 ? 3 : 5; }     // weight fragments of nring - 1 k-steps in flight (one product: 8 MFMAs per k-step, half the cover per step)
constexpr int nring_r(int NP) { return NP == 2 ? 5 : 8; }   // ... of the residual projection (4 NP MFMAs per k-step)

// -DSS_L512_TRACE (debug builds, tools/trace_layer512.py): lane 0 of every wave stamps the shader clock at 8 points of every tile into the
// buffer handed over through ss_set_clock_probe ([workgroup][wave][tile slot < 8][8]); the product build compiles none of it.
#ifdef SS_L512_TRACE
#define L512_STAMP(k)                                                                                                   \
  do {                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    if (clock_probe && it < 8 && (tid0 & 63) == 0)                                                                      \
      clock_probe[(((int64_t)blockIdx.x * 8 + wave) * 8 + it) * 8 + (k)] = (unsigned long long)__builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
  } while (0)
#else
#define L512_STAMP(k) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}
template <class F, int... I>
__device__ __forceinline__ void unrolled_steps(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
__device__ __forceinline__ void* uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ bf16x8 ldw(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}
__device__ __forceinline__ int acc_rr(int r) { return (r & 3) + 8 * (r >> 2); }

// One work item of a persistent workgroup: a whole 128-row tile (nm = 4 row blocks of 32) or, in the last round, half of one (nm = 2, rows r0 ..
// r0 + 63 of the tile) - see the schedule in layer512_kernel.
struct L512Item {
  int tile, r0, nm;
};

// EH: the addend slab holds fp16 values (ss_layer512_args.e_f16; one of the sigma-delta SETS the caller cycles over the evaluations): per lane and
// (m, q) one 16-byte entry = 4 + 4 values of the two gate operands - half the bytes of the launch's largest stream.
template <bool FUSE, int NP, bool EH = false>
__global__ __launch_bounds__(512, 2) void layer512_kernel(const ss_layer512_args a, int tiles_per_item, int n_tiles, int split_tail, unsigned long long* clock_probe) {
#ifdef SS_L512_TRACE
  const bool probing = false;
  // the workgroup's own life on the constant 100 MHz counter (one time axis for all XCDs; the shader-cycle counters are per XCD): start and end at
  // [gridDim.x * 512 + 4 * blockIdx.x] of the trace buffer, with the shader cycle counter beside them
  if (clock_probe && threadIdx.x == 0) {
    clock_probe[(int64_t)gridDim.x * 512 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    clock_probe[(int64_t)gridDim.x * 512 + 4 * blockIdx.x + 1] = __builtin_readcyclecounter();
  }
#else
  const bool probing = clock_probe != nullptr && blockIdx.x == 0;
#endif
  unsigned long long probe_c0 = 0, probe_r0 = 0;
  if (probing) {
    probe_c0 = __builtin_readcyclecounter();
    probe_r0 = __builtin_amdgcn_s_memrealtime();
  }
  extern __shared__ __attribute__((aligned(16))) char smem_l512[];   // 145 KB: two regions of 32 slots x 2320 B

  const int tid0 = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int d = a.d;

  constexpr int WG_STEP = wg_step(NP), WG_WAVE = wg_wave(NP), WR_STEP = wr_step(NP), WR_WAVE = wr_wave(NP), NRING = nring(NP), NRING_R = nring_r(NP);
  // (Measured, no effect: a deeper weight ring - 6 / 7 / 8 k-steps instead of 5, 4 instead of 3 with two products - and separate copies of the gate
  // weights per group of CUs (is the one 98 KB stream per wave an L2 hot spot? no: 2 / 4 copies cost 1-3 %): profiles/r06_kbench_layer512_phase_shift.log)
  const __amdgpu_buffer_rsrc_t rsrc_wg = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)a.Wg + (int64_t)wave * WG_WAVE), 0, WG_WAVE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_wr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(FUSE ? (const char*)a.Wr + (int64_t)wave * WR_WAVE : (const char*)a.Wg), 0,
                                                                         FUSE ? WR_WAVE : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_hi = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.Hin), 0, __builtin_amdgcn_readfirstlane(n_tiles * H_TILE), 0x00020000);

  // ---- schedule. Workgroup g of W takes tiles g, g + W, ... With n_tiles = k W + R the last round would keep R workgroups busy for a whole tile
  // period while the others idle (BASELINE configs[3]: 1408 tiles on 256 CUs = 5.5 rounds, makespan 6). When 0 < R <= W / 2 (split_tail, decided by
  // the launcher) the R tiles of that round are cut into 2 R HALF tiles of 64 rows - same code with two row blocks instead of four - so that the
  // round costs about 0.6 of a tile period.
  const int W = gridDim.x, g = blockIdx.x;
  // PHASES. The epilogues of a tile pull 0.46 MB per CU through HBM (addend slab, stream, next activation tile), its conv loop nothing. Workgroups
  // that start together stay in step for the 5.5 tiles of a launch, so the whole chip asks for its 117 MB at once - 17 us at HBM's pace for 3 us of
  // arithmetic, 38 k cycles against 17 k on a quarter of the CUs (profiles/r06_trace_layer512_warm.log) - and then leaves the memory idle during
  // the conv loops. With a split tail the EVEN workgroups therefore run their half tile FIRST and the odd ones last: the two halves of the chip
  // are half a tile period apart for the whole launch, one streams while the other multiplies; same items per workgroup, same end. 278 -> 255 us
  // (one product), 375 -> 351 us (two) per launch. Which bit of the index picks the class matters (0 and 4..7: 253-257 us, 1..3: 265-274):
  // profiles/r06_kbench_layer512_phase_shift.log; a start delay on top of it (four phases, paid for at the end) gains 3 % at best.
  // Knob layer512_tail = 2 keeps every half tile last (A/B).
  const int full_rounds = split_tail ? n_tiles / W : 0;
  const bool has_half = split_tail && (g >> 1) < n_tiles - full_rounds * W;
  const bool half_first = has_half && full_rounds > 0 && split_tail == 1 && (g & 1) == 0;
  const int n_items = split_tail ? full_rounds + (has_half ? 1 : 0) : (n_tiles - g + W - 1) / W;
  auto item = [&](int i) {
    L512Item r;
    if (has_half && i == (half_first ? 0 : full_rounds)) {
      r.tile = full_rounds * W + (g >> 1);
      r.r0 = 64 * (g & 1);
      r.nm = 2;
    } else {
      r.tile = g + (half_first ? i - 1 : i) * W;
      r.r0 = 0;
      r.nm = 4;
    }
    return r;
  };
  auto tile_coords = [&](int tile, int& b, int& ti) {
    b = tile / tiles_per_item;
    ti = tile - b * tiles_per_item;
  };
  // ---- DMA of an activation tile. H and the LDS image are both SLOT-MAJOR: slot s (8 channels = 16 bytes) of all rows, row after row. LDS row
  // L = tile row r0 + L - HALO. Wave w stages slots 4 w .. 4 w + 3 in pieces of 64 rows: LDS rows [0, 64), [64, 128) and [80, 144) of a whole
  // tile (the last one rewrites 48 rows with the same bytes: every piece is a full 1 KB, no lane masking), [0, 64) and [16, 80) of a half tile.
  // A piece reads H contiguously except where it crosses into the previous / next tile of the item (the halo); rows outside the item are out of
  // range: the DMA writes zeros (the conv's padding; rows in [len, T) hold zeros already - every producer of H masks them).
  auto dma_item = [&](const L512Item& it_, char* region, int lane) {
    int b, ti;
    tile_coords(it_.tile, b, ti);
    const int npiece = it_.nm == 4 ? 3 : 2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j >= npiece) break;
      const int L0 = it_.nm == 4 ? (j == 2 ? 80 : 64 * j) : 16 * j;
      const int rho = it_.r0 + L0 + lane - HALO;     // row relative to the tile
      const int dt = rho < 0 ? -1 : (rho >= BM ? 1 : 0);
      const bool ok = (unsigned)(ti + dt) < (unsigned)tiles_per_item;
      const int base = ok ? (it_.tile + dt) * H_TILE + (rho - dt * BM) * 16 : (int)0x80000000;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int s_ = 4 * wave + k;
        glds16(rsrc_hi, region + s_ * SLOTB + L0 * 16, base, s_ * (BM * 16));
      }
    }
  };

  // sigmoid(v0) * tanh(v1) = (1 - b) / ((1 + a) (1 + b)) with a = e^-v0, b = e^-2 v1: two exp2, ONE rcp. The addend slab arrives pre-multiplied
  // by -log2(e) (sigmoid columns) / -2 log2(e) (tanh columns) (tile_addend_kernel), so the exponents are one FMA from the accumulators.
  // b is capped at 2^30 (tanh = -1 to fp32 precision there) so that (1 - b) * 0 cannot become inf * 0 when (1 + a) overflows.
  const float L2E = 1.44269504088896340736f;
  const float ka = -L2E * a.out_scale, kbx = -2.0f * L2E * a.out_scale;

  // ---- one work item; NM = row blocks of 32 (4: a tile, 2: half of one)
  auto run_item = [&](auto nm_tag, const L512Item cur, const bool has_next, const L512Item nxt, const int it) {
    constexpr int NM = decltype(nm_tag)::value;
    // per-lane constants are recomputed per item from an opaque copy of the thread id: hoisted out of the item loop they would stay live across
    // the 128-accumulator conv loop and spill
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
    const int w_voff = lane * 16;
    const int tile = cur.tile, r0 = cur.r0, mb0 = cur.r0 >> 5;   // mb0: the item's first row block inside its tile (addend / stream / H layouts)
    // ---- activation fragments (the matrix instruction's B operand): lane (l31, lh) reads slot 4 cc + 2 ks + lh (channels 16 ks + 8 lh .. + 7
    // of chunk cc) of LDS row HALO + (tap - 1) d + 32 m + l31: 32 lanes read 512 contiguous bytes - conflict-free without a swizzle
    int a_off[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) a_off[j] = (HALO + (j - 1) * d + l31) * 16 + lh * SLOTB;
    // G tile (rows 0 .. 32 NM - 1, same slot-major form): fragment reads of the residual projection and the gate epilogue's 8-byte writes
    const int g_off = l31 * 16 + lh * SLOTB;
    char* const Rc = smem_l512 + (it & 1) * REGION;   // A(item), then G(item)
    char* const Rn = smem_l512 + ((it & 1) ^ 1) * REGION;
    int b, ti;
    tile_coords(tile, b, ti);
    const int t0 = ti * BM + r0;                      // first row of the item inside its utterance
    const int len = ss_uniform_len(a.lens, b, a.T);
    const int row_lim = a.mask_rows ? (len < a.T ? len : a.T) : a.T;

    f32x16 acc[2][NM];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

    // ---- the dilated conv: 48 k-steps, no barrier. Weight ring: fragments of k-step S + NRING - 1 are requested before the MFMAs of step S.
    bf16x8 wq[NRING][2 * NP];   // [plane * 2 + nb]
    bf16x8 act[2][NM];
    auto load_w = [&](bf16x8 (&dst)[2 * NP], int S) {
#pragma unroll
      for (int p = 0; p < 2 * NP; ++p) dst[p] = ldw(rsrc_wg, w_voff + p * 1024, S * WG_STEP);
    };
    auto read_act = [&](bf16x8 (&dst)[NM], int S) {
      const int cc = S / 6, tap = (S / 2) % 3, ks = S & 1;
      const int ao = a_off[tap] + (4 * cc + 2 * ks) * SLOTB;
#pragma unroll
      for (int m = 0; m < NM; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(Rc + ao + m * 512);
    };
#pragma unroll
    for (int s = 0; s < NRING - 1; ++s) load_w(wq[s], s);
    // my DMA pieces of this item have landed: only the ring's loads are younger. (Fused form: already waited for - a vmcnt wait HERE would also
    // wait for the stream epilogue's 32 stores, a full memory round trip per tile: 5.3 k cycles in the v1 trace.)
    if constexpr (!FUSE) wait_vmcnt<2 * NP * (NRING - 1)>();
    __builtin_amdgcn_s_barrier();    // [B1] everyone's pieces have
    L512_STAMP(0);
    read_act(act[0], 0);
    auto kstep = [&](auto stag) {
      constexpr int S = decltype(stag)::value;
      if constexpr (S + NRING - 1 < KSTEPS) load_w(wq[(S + NRING - 1) % NRING], S + NRING - 1);
      if constexpr (S + 1 < KSTEPS) read_act(act[(S + 1) & 1], S + 1);
      const bf16x8 (&w)[2 * NP] = wq[S % NRING];
      const bf16x8 (&x)[NM] = act[S & 1];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < NM; ++m) acc[n][m] = ss_mfma_32x32x16<true>(w[p * 2 + n], x[m], acc[n][m]);
      __builtin_amdgcn_sched_barrier(0);
    };
    unrolled_steps(kstep, std::make_integer_sequence<int, KSTEPS>{});
    L512_STAMP(1);

    // ---- gate epilogue. Addend slab in accumulator order: block (nb, m), quarter q -> one 16-byte load per lane, 1 KB per wave instruction.
    // Request order = order of need (vmcnt retires in order: whatever is waited for drags everything older with it): E(m = 0), E(1) | E(2) |
    // E(3) | the stream P | the next item's DMA pieces last.
    constexpr int E_BYTES = EH ? E_TILE / 2 : E_TILE;   // bytes of addend per tile
    const __amdgpu_buffer_rsrc_t rsrc_e = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr((const char*)a.E512 + (int64_t)tile * E_BYTES + (WAVE_MAJOR ? (int64_t)wave * (E_BYTES / 8) : 0)), 0, WAVE_MAJOR ? E_BYTES / 8 : E_BYTES, 0x00020000);
    constexpr int BLK = WAVE_MAJOR ? 1024 : 8192;            // bytes between consecutive (nb, m, q) blocks of a wave
    const int wave_off = WAVE_MAJOR ? 0 : wave * 1024;
    // TWO blocks ahead: the slab is 256 KB per tile and CU, and with one block (8 KB per wave) in flight it arrived at ~26 B per cycle and CU -
    // the gate epilogue took 20 k cycles for 5 k of VALU work (profiles/r06_trace_layer512_v3.log)
    using EvBlock = std::conditional_t<EH, u32x4[4], f32x4[2][4]>;   // EH: [q] = (4 fp16 of nb 0 | 4 fp16 of nb 1)
    constexpr int NEV = 3;   // (all four blocks of the fp16 addend requested up front - 64 registers - measure worse: 231.5 against 225.5 us per launch)
    EvBlock ev[NEV];
    auto load_e = [&](EvBlock& dst, int m) {
      if constexpr (EH) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_e, w_voff, ((mb0 + m) * 4 + q) * BLK + wave_off, 0);
      } else {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            dst[n][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_e, w_voff, ((n * 4 + mb0 + m) * 4 + q) * BLK + wave_off, 0));
      }
    };
    auto e_val = [&](const EvBlock& blk, int n, int q, int e) -> float {
      if constexpr (EH) return ss_t2f_packed<true>(blk[q][2 * n + (e >> 1)], e & 1);
      else return blk[n][q][e];
    };
    load_e(ev[0], 0);
    load_e(ev[1], 1);
    // (Measured, no gain: the next item's DMA requested HERE - the other region is free since [B1], and the wave that leaves the conv loop first idles
    // at [B2] for ~16 k cycles - instead of at the end of the gate epilogue: 216-220 against 220 us; the conv phase grows by what the epilogues lose.
    // With the fp16 addend and the fp16 stream remainder both epilogues are VALU-bound - 64 gate values x (10 VALU + 2 exp2 + 1 rcp) and 64 stream
    // values x 15 VALU per lane, two waves per SIMD in the same phase: 11.3 k + 7.7 k cycles - profiles/r06_trace_layer512_final_forms.log)
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): my reads of the A tile are done
    __builtin_amdgcn_s_barrier();         // [B2] everyone's are: G may overwrite the A tile; the other region is free since the last item ended
    L512_STAMP(2);
    // (Measured and dropped: the gate arithmetic AHEAD of [B2], into registers, so that the wave that leaves the conv loop first - the older wave of
    // a SIMD gets the matrix pipe, 25 k against 50 k cycles - works under its partner's MFMAs instead of waiting at the barrier. Its VALU stream
    // then competes with the partner's MFMA issue: conv loop 50 -> 62 k cycles, gate arithmetic 30 k; 403.6 against 397.7 us per launch -
    // profiles/r06_trace_layer512_v5_gate_math_before_b2.log. Work moved between the two waves of a SIMD is zero-sum, as the guide says.)
    // the stream of this item: x = (H - dstep_l) + R with H = fp16(x + dstep_l) - this tile's own rows of the conv operand, still in LDS (a wave's
    // channels are its own slots, which only its own G writes below will overwrite: read block by block ahead of them) - and R = the fp16 remainder
    // (accumulator order, 8 bytes per lane and (m, q)), requested once half of the conv accumulators are dead so that the loads fly under the rest
    // of this epilogue, [B3] and the G pass. 22 significant bits (as the two-launch form's pair), 2 bytes per element of HBM traffic each way
    // instead of 4: the fp32 copy of the stream was 256 of the 584 KB a tile moved.
    [[maybe_unused]] u32x2 hown[NM][4];
    [[maybe_unused]] u32x2 rv[NM][4];
    constexpr int BLKR = 4096;   // bytes between consecutive (m, q) blocks of R: 8 waves x 512
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(FUSE ? (const char*)a.P + (int64_t)tile * P_TILE : (const char*)a.Wg), 0, FUSE ? P_TILE : 0, 0x00020000);
    auto load_p = [&]() {
#pragma unroll
      for (int mm = 0; mm < NM; ++mm)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rv[mm][q] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc_p, lane * 8, ((mb0 + mm) * 4 + q) * BLKR + wave * 512, 0));
    };
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if constexpr (NEV == 3) {
        if (m + 2 < NM) load_e(ev[(m + 2) % 3], m + 2);
      }
      if constexpr (FUSE) {
        if (m == NM - 2) load_p();
      }
      if constexpr (FUSE) {
        // the H term of row block m, BEFORE this block's G goes over it: G rows 32 m .. 32 m + 31 land on activation rows 32 m - 8 .. 32 m + 23 of the same
        // slots (other LANES' rows of this block and the tail of block m - 1, read an iteration ago). The dependence runs across lanes, which the
        // compiler's per-thread alias analysis does not see (it would sink these loads below the stores): hence the wait + memory clobber.
#pragma unroll
        for (int q = 0; q < 4; ++q) hown[m][q] = *reinterpret_cast<const u32x2*>(Rc + (4 * wave + q) * SLOTB + (HALO + 32 * m + l31) * 16 + 8 * lh);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      const bool pad = t0 + 32 * m + l31 >= row_lim;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t pk[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          uint32_t v = 0;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int e = 2 * e2 + k, r = 4 * q + e;
            const float ea = __builtin_amdgcn_exp2f(fmaf(acc[0][m][r], ka, e_val(ev[m % NEV], 0, q, e)));
            const float eb = __builtin_amdgcn_exp2f(fminf(fmaf(acc[1][m][r], kbx, e_val(ev[m % NEV], 1, q, e)), 30.0f));
            float g_ = (1.0f - eb) * __builtin_amdgcn_rcpf((1.0f + ea) * (1.0f + eb));   // sigmoid(v0) * tanh(v1), net.py:72-73
            if (pad) g_ = 0.f;
            v |= (uint32_t)ss_f2t<true>(g_) << (16 * k);
          }
          pk[e2] = v;
        }
        // channels 32 w + 8 q + 4 lh .. + 3 of row 32 m + l31: slot 4 w + q, bytes 8 lh .. of the row's 16: a wave writes 512 contiguous bytes
        *reinterpret_cast<u32x2*>(Rc + (4 * wave + q) * SLOTB + (32 * m + l31) * 16 + 8 * lh) = u32x2{pk[0], pk[1]};
      }
    }
    // (Measured, no gain: the stream loads and this DMA after [B3] instead - 264.4 / 355.1 against 266.8 / 349.2 us, profiles/r06_kbench_layer512_phase_shift.log)
    if (has_next) dma_item(nxt, Rn, lane);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // my G writes are done
    L512_STAMP(3);
    __builtin_amdgcn_s_barrier();         // [B3] the G tile is complete
    L512_STAMP(4);

    // ---- G -> HBM (the skip GEMM's operand, ss_gemm_bf16's pair layout): 32 NM rows x 32 slots, 2 NM per thread; lanes walk the slots of a row
    // (LDS stride 2320 B = 4 banks x 16 B apart: conflict-free), the hi halves of the row's 128-byte pair lines in HBM
    {
      const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(
          uniform_ptr(a.G + (int64_t)b * a.g_batch_stride), 0, __builtin_amdgcn_readfirstlane((int)((int64_t)a.T * a.ldg * 2)), 0x00020000);
#pragma unroll
      for (int j = 0; j < 2 * NM; ++j) {
        const int p = tid + 512 * j;
        const int R = p >> 5, s_ = p & 31;
        const u32x4 v = *reinterpret_cast<const u32x4*>(Rc + s_ * SLOTB + R * 16);
        // pair layout: the hi half of chunk s_ >> 2's 128-byte line; compact: the row's 32 slots side by side (512 contiguous bytes per 32 lanes)
        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_g, (t0 + R) * a.ldg * 2 + (a.g_compact ? s_ * 16 : (s_ >> 2) * 128 + (s_ & 3) * 16), 0, 0);   // rows >= T dropped
      }
    }
    L512_STAMP(5);
    if constexpr (FUSE) {
      // ---- residual projection from the G tile: out^T[channel][row], wave w owns channels 32 w .. + 31; 16 k-steps of 2 NM MFMAs
      f32x16 acc2[NM];
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
      bf16x8 wr[NRING_R][NP];
      bf16x8 gf[2][NM];
      auto load_wr = [&](bf16x8 (&dst)[NP], int S) {
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p] = ldw(rsrc_wr, w_voff + p * 1024, S * WR_STEP);
      };
      auto read_g = [&](bf16x8 (&dst)[NM], int S) {
#pragma unroll
        for (int m = 0; m < NM; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(Rc + g_off + 2 * S * SLOTB + m * 512);
      };
#pragma unroll
      for (int s = 0; s < NRING_R - 1; ++s) load_wr(wr[s], s);
      read_g(gf[0], 0);
      // per-channel constants of this lane's 16 channels 32 w + 8 q + 4 lh + e
      f32x4 bs[4], nb[4], cb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * wave + 8 * q + 4 * lh;
        bs[q] = a.bias_r ? *reinterpret_cast<const f32x4*>(a.bias_r + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
        nb[q] = a.next_bias ? *reinterpret_cast<const f32x4*>(a.next_bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
        cb[q] = a.cur_bias ? *reinterpret_cast<const f32x4*>(a.cur_bias + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      auto rstep = [&](auto stag) {
        constexpr int S = decltype(stag)::value;
        if constexpr (S + NRING_R - 1 < RSTEPS) load_wr(wr[(S + NRING_R - 1) % NRING_R], S + NRING_R - 1);
        if constexpr (S + 1 < RSTEPS) read_g(gf[(S + 1) & 1], S + 1);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int m = 0; m < NM; ++m) acc2[m] = ss_mfma_32x32x16<true>(wr[S % NRING_R][p], gf[S & 1][m], acc2[m]);
        __builtin_amdgcn_sched_barrier(0);
      };
      unrolled_steps(rstep, std::make_integer_sequence<int, RSTEPS>{});
      L512_STAMP(6);

      // ---- stream update: x = (H - dstep_l) + R ; x' = (x + acc * out_scale + b) * post_scale in fp32; out: H' = fp16(x' + dstep_(l+1)) into Hout's
      // slot-major tile and R' = fp16(x' - (H' - dstep_(l+1))) in place - 8 bytes per lane each, the two lane halves fill a row's 16-byte slot, 32 rows
      // in a row: 512 contiguous bytes per instruction. Everything of mine that is in flight has to land first anyway (the stream loads) - and with
      // it the next item's DMA pieces, which [B1] then needs no memory wait for.
      wait_vmcnt<0>();
      const __amdgpu_buffer_rsrc_t rsrc_ho = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const char*)a.Hout + (int64_t)tile * H_TILE), 0, H_TILE, 0x00020000);
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const bool pad = t0 + 32 * m + l31 >= row_lim;
        const int ho = (r0 + 32 * m + l31) * 16 + 8 * lh;   // + slot (4 w + q) * 2048
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t hp[2] = {0, 0}, rp[2] = {0, 0};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // no contraction here: every term is rounded where the next layer's epilogue (and the other tile schedules) round it
#pragma clang fp contract(off)
            const float xo = (ss_t2f_packed<true>(hown[m][q][e >> 1], e & 1) - cb[q][e]) + ss_t2f_packed<true>(rv[m][q][e >> 1], e & 1);
            const float xn = (xo + fmaf(acc2[m][4 * q + e], a.out_scale, bs[q][e])) * a.post_scale;
            const uint16_t hh = ss_f2t<true>(pad ? 0.f : xn + nb[q][e]);
            const uint16_t rr = ss_f2t<true>(pad ? 0.f : xn - (ss_t2f<true>(hh) - nb[q][e]));
            hp[e >> 1] |= (uint32_t)hh << (16 * (e & 1));
            rp[e >> 1] |= (uint32_t)rr << (16 * (e & 1));
          }
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{rp[0], rp[1]}, rsrc_p, lane * 8, ((mb0 + m) * 4 + q) * BLKR + wave * 512, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{hp[0], hp[1]}, rsrc_ho, ho, (4 * wave + q) * (BM * 16), 0);
        }
      }
    }
    L512_STAMP(7);
    // (no barrier here: the next item's [B1] is reached by a wave only after its reads of this G tile, and this region is next written by
    // the DMA issued before the next item's [B3] - after its [B2], which every wave reaches only after [B1])
  };

  if (n_items > 0) dma_item(item(0), smem_l512, tid0 & 63);
  if constexpr (FUSE) wait_vmcnt<0>();   // (the fused form waits for the NEXT item's pieces before its stream epilogue, not at [B1]: see there)
  for (int it = 0; it < n_items; ++it) {
    const L512Item cur = item(it);
    const bool has_next = it + 1 < n_items;
    const L512Item nxt = item(has_next ? it + 1 : it);
    if (cur.nm == 4) run_item(std::integral_constant<int, 4>{}, cur, has_next, nxt, it);
    else run_item(std::integral_constant<int, 2>{}, cur, has_next, nxt, it);
  }
#ifdef SS_L512_TRACE
  if (clock_probe && threadIdx.x == 0) {
    clock_probe[(int64_t)gridDim.x * 512 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    clock_probe[(int64_t)gridDim.x * 512 + 4 * blockIdx.x + 3] = __builtin_readcyclecounter();
  }
#endif
  if (probing && tid0 == 0) {
    atomicAdd(clock_probe, (unsigned long long)__builtin_readcyclecounter() - probe_c0);
    atomicAdd(clock_probe + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - probe_r0);
  }
}

// ---- packers (device -> device, once per checkpoint / forward)
// gate weights: the ss_split_f16 pack of the interleaved dilated-conv weights, [512 packed columns][3 taps x 256 channels x (hi | lo)] with pairs
// interleaved by 32 (line (tap * 8 + cc) of a row = 32 hi + 32 lo terms) -> fragment order [wave 8][k-step 48][plane 2][nb 2][lane 64][8]:
// k-step S = (cc * 3 + tap) * 2 + ks; lane (l31, lh) holds channels 32 cc + 16 ks + 8 lh .. + 7 of packed column 64 wave + 32 nb + l31
__global__ void pack_gate_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int np) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte fragment each: 8 * 48 * 2 np * 64 (np = 2: 98 304)
  if (i >= 8 * KSTEPS * 2 * np * 64) return;
  const int lane = i & 63, pn = (i >> 6) % (2 * np), S = (i / (64 * 2 * np)) % KSTEPS, w = i / (64 * 2 * np * KSTEPS);
  const int plane = pn >> 1, nb = pn & 1, l31 = lane & 31, lh = lane >> 5;
  const int cc = S / 6, tap = (S / 2) % 3, ks = S & 1;
  const int col = 64 * w + 32 * nb + l31;
  const uint16_t* s = src + (int64_t)col * (3 * 256 * 2) + (tap * 8 + cc) * 64 + plane * 32 + 16 * ks + 8 * lh;
  *reinterpret_cast<uint4*>(dst + (int64_t)i * 8) = *reinterpret_cast<const uint4*>(s);
}
// residual weights: [>= 256 rows][256 channels x (hi | lo)] pairs interleaved by 32 -> [wave 8][k-step 16][plane 2][lane 64][8]: lane (l31, lh)
// holds channels 16 S + 8 lh .. + 7 of output channel 32 wave + l31
__global__ void pack_res_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int np) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 8 * 16 * np * 64 fragments (np = 2: 16 384)
  if (i >= 8 * RSTEPS * np * 64) return;
  const int lane = i & 63, plane = (i >> 6) % np, S = (i / (64 * np)) % RSTEPS, w = i / (64 * np * RSTEPS);
  const int l31 = lane & 31, lh = lane >> 5;
  const int row = 32 * w + l31, k = 16 * S + 8 * lh;
  const uint16_t* s = src + (int64_t)row * (256 * 2) + (k >> 5) * 64 + plane * 32 + (k & 31);
  *reinterpret_cast<uint4*>(dst + (int64_t)i * 8) = *reinterpret_cast<const uint4*>(s);
}
// conditioner addend E [B][T][lde] (this layer's 512 packed columns) -> [tile][nb 2][m 4][q 4][wave 8][lane 64][4]: lane (l31, lh) holds packed
// columns 64 wave + 32 nb + 8 q + 4 lh .. + 3 of row 32 m + l31 of the tile; rows >= T are zero. The values are stored as the gate's exp2
// arguments: times -log2(e) in the sigmoid blocks (nb = 0), times -2 log2(e) in the tanh blocks (nb = 1).
__global__ void tile_addend_kernel(const float* __restrict__ E, int lde, int64_t e_batch_stride, float* __restrict__ out, int T, int tiles_per_item, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 each
  if (i >= n) return;
  const int lane = (int)(i & 63);
  const int q = (int)(i >> (WAVE_MAJOR ? 6 : 9)) & 3, m = (int)(i >> (WAVE_MAJOR ? 8 : 11)) & 3, nb = (int)(i >> (WAVE_MAJOR ? 10 : 13)) & 1, w = (int)(i >> (WAVE_MAJOR ? 11 : 6)) & 7;
  const int64_t tile = i >> 14;
  const int b = (int)(tile / tiles_per_item), t = (int)(tile % tiles_per_item) * BM + 32 * m + (lane & 31);
  const int col = 64 * w + 32 * nb + 8 * q + 4 * (lane >> 5);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < T) v = *reinterpret_cast<const float4*>(E + (int64_t)b * e_batch_stride + (int64_t)t * lde + col);
  const float k = nb ? -2.0f * 1.44269504088896340736f : -1.44269504088896340736f;
  *reinterpret_cast<float4*>(out + i * 4) = make_float4(v.x * k, v.y * k, v.z * k, v.w * k);
}

// The same addend as N fp16 SETS (ss_layer512_args.e_f16): [set][tile][m 4][q 4][wave 8][lane 64] x (4 fp16 of nb 0 | 4 fp16 of nb 1) - 16 bytes per
// lane and (m, q), half the slab. The addend is the same in every evaluation of a sampling loop, so ONE fp16 rounding of it would be a fixed bias
// (9.5e-5 on the reference's 1000-step golden, oracle/dither_numerics.py --e-sets=1); like the fp16sd weights it is therefore stored as a first-order
// sigma-delta sequence of roundings of the scaled value (r_0 = 0, E_k = RNE16(e + r_k), r_(k+1) = r_k + (e - E_k)) that the caller cycles over the
// evaluations: 8 sets 2.5e-5, 16 sets 2.3e-5, exact fp32 2.2e-5.
__global__ void tile_addend_f16_kernel(const float* __restrict__ E, int lde, int64_t e_batch_stride, uint16_t* __restrict__ out, int n_sets, int64_t set_stride,
                                       int T, int tiles_per_item, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte entry each
  if (i >= n) return;
  const int lane = (int)(i & 63), w = (int)(i >> 6) & 7, q = (int)(i >> 9) & 3, m = (int)(i >> 11) & 3;
  const int64_t tile = i >> 13;
  const int b = (int)(tile / tiles_per_item), t = (int)(tile % tiles_per_item) * BM + 32 * m + (lane & 31);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, r[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (t < T) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float4 x = *reinterpret_cast<const float4*>(E + (int64_t)b * e_batch_stride + (int64_t)t * lde + 64 * w + 32 * nb + 8 * q + 4 * (lane >> 5));
      const float k = nb ? -2.0f * 1.44269504088896340736f : -1.44269504088896340736f;
      v[4 * nb] = x.x * k; v[4 * nb + 1] = x.y * k; v[4 * nb + 2] = x.z * k; v[4 * nb + 3] = x.w * k;
    }
  }
  for (int s_ = 0; s_ < n_sets; ++s_) {
    uint32_t pk[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      uint32_t word = 0;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma clang fp contract(off)
        const int e = 2 * e2 + kk;
        const uint16_t h = ss_f2t<true>(v[e] + r[e]);
        r[e] = r[e] + (v[e] - ss_t2f<true>(h));
        word |= (uint32_t)h << (16 * kk);
      }
      pk[e2] = word;
    }
    *reinterpret_cast<uint4*>(out + (int64_t)s_ * set_stride + i * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  }
}

// stack entry: X fp32 [B][T][ldx] -> the stream's two forms: P = R, the fp16 remainder x - (H - bias), in accumulator order ([tile][m 4][q 4][wave 8][lane 64] x 4 fp16; lane
// (l31, lh) of (wave, m, q) holds channels 32 wave + 8 q + 4 lh .. + 3 of row 32 m + l31) and H = fp16(x + bias) in slot-major tiles
// ([tile][slot 32][row 128] x 8 channels). Rows >= lens[b] are zero.
__global__ void entry_kernel(const float* __restrict__ X, int ldx, int64_t x_batch_stride, const float* __restrict__ bias, const int32_t* __restrict__ lens,
                             uint16_t* __restrict__ H, uint2* __restrict__ P, int T, int tiles_per_item, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one 8-byte R entry each
  if (i >= n) return;
  const int lane = (int)(i & 63), q = (int)(i >> (WAVE_MAJOR ? 6 : 9)) & 3, m = (int)(i >> (WAVE_MAJOR ? 8 : 11)) & 3, w = (int)(i >> (WAVE_MAJOR ? 10 : 6)) & 7;
  const int64_t tile = i >> 13;
  const int b = (int)(tile / tiles_per_item), t = (int)(tile % tiles_per_item) * BM + 32 * m + (lane & 31);
  const int c0 = 32 * w + 8 * q + 4 * (lane >> 5);
  const int len = lens ? min(max(lens[b], 0), T) : T;
  uint32_t hp[2] = {0, 0}, rp[2] = {0, 0};
  if (t < len) {
    const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)b * x_batch_stride + (int64_t)t * ldx + c0);
    const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float xv[4] = {x.x, x.y, x.z, x.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
      const uint16_t hh = ss_f2t<true>(xv[e] + bv[e]);
      const uint16_t rr = ss_f2t<true>(xv[e] - (ss_t2f<true>(hh) - bv[e]));
      hp[e >> 1] |= (uint32_t)hh << (16 * (e & 1));
      rp[e >> 1] |= (uint32_t)rr << (16 * (e & 1));
    }
  }
  P[i] = make_uint2(rp[0], rp[1]);
  *reinterpret_cast<uint2*>(H + tile * (H_TILE / 2) + ((4 * w + q) * BM + 32 * m + (lane & 31)) * 8 + 4 * (lane >> 5)) = make_uint2(hp[0], hp[1]);
}

}  // namespace

extern "C" int64_t ss_layer512_stream_bytes(int B, int T) { return (int64_t)B * ss_cdiv(T, BM) * P_TILE; }

extern "C" int64_t ss_layer512_h_elems(int B, int T) { return (int64_t)B * ss_cdiv(T, BM) * (H_TILE / 2); }

extern "C" int ss_layer512_entry(const float* X, int ldx, int64_t x_batch_stride, const float* bias, const int32_t* lens, uint16_t* H, void* P, int B, int T,
                                 void* stream) {
  SS_CHECK_ARG(X && H && P && B > 0 && T > 0 && ldx >= 256 && (ldx % 4) == 0 && (x_batch_stride % 4) == 0, "ss_layer512_entry: X [B][T][ldx >= 256, %% 4]");
  SS_CHECK_ARG((((uintptr_t)X) & 15) == 0 && (((uintptr_t)H) & 15) == 0 && (((uintptr_t)P) & 15) == 0 && (!bias || (((uintptr_t)bias) & 15) == 0),
               "ss_layer512_entry: X / H / P / bias must be 16-byte aligned");
  const int tpi = ss_cdiv(T, BM);
  const int64_t n = (int64_t)B * tpi * (P_TILE / 8);
  hipLaunchKernelGGL(entry_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, ldx, x_batch_stride, bias, lens, H, (uint2*)P, T, tpi, n);
  SS_CHECK_LAUNCH("ss_layer512_entry");
  return SS_OK;
}

extern "C" int64_t ss_layer512_addend_floats(int B, int T) { return (int64_t)B * ss_cdiv(T, BM) * (E_TILE / 4); }

extern "C" int ss_layer512_tile_addend(const float* E, int lde, int64_t e_batch_stride, float* out, int B, int T, void* stream) {
  SS_CHECK_ARG(E && out && B > 0 && T > 0 && (lde % 4) == 0 && lde >= 512 && (e_batch_stride % 4) == 0 && (((uintptr_t)E) & 15) == 0 && (((uintptr_t)out) & 15) == 0,
               "ss_layer512_tile_addend: E / out 16-byte aligned, lde %% 4 == 0 and >= 512");
  const int tpi = ss_cdiv(T, BM);
  const int64_t n = (int64_t)B * tpi * (E_TILE / 16);
  hipLaunchKernelGGL(tile_addend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, E, lde, e_batch_stride, out, T, tpi, n);
  SS_CHECK_LAUNCH("ss_layer512_tile_addend");
  return SS_OK;
}

extern "C" int ss_layer512_pack_gate(const uint16_t* w_pairs, uint16_t* out, int n_products, void* stream) {
  SS_CHECK_ARG(w_pairs && out && (((uintptr_t)w_pairs) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "ss_layer512_pack_gate: 16-byte aligned pointers");
  SS_CHECK_ARG(n_products == 1 || n_products == 2, "ss_layer512_pack_gate: n_products = 1 (hi terms only) | 2");
  hipLaunchKernelGGL(pack_gate_kernel, dim3(8 * KSTEPS * 2 * n_products * 64 / 256), dim3(256), 0, (hipStream_t)stream, w_pairs, out, n_products);
  SS_CHECK_LAUNCH("ss_layer512_pack_gate");
  return SS_OK;
}

extern "C" int ss_layer512_pack_res(const uint16_t* w_pairs, uint16_t* out, int n_products, void* stream) {
  SS_CHECK_ARG(w_pairs && out && (((uintptr_t)w_pairs) & 15) == 0 && (((uintptr_t)out) & 15) == 0, "ss_layer512_pack_res: 16-byte aligned pointers");
  SS_CHECK_ARG(n_products == 1 || n_products == 2, "ss_layer512_pack_res: n_products = 1 (hi terms only) | 2");
  hipLaunchKernelGGL(pack_res_kernel, dim3(8 * RSTEPS * n_products * 64 / 256), dim3(256), 0, (hipStream_t)stream, w_pairs, out, n_products);
  SS_CHECK_LAUNCH("ss_layer512_pack_res");
  return SS_OK;
}

extern "C" int64_t ss_layer512_addend_halfs(int B, int T) { return (int64_t)B * ss_cdiv(T, BM) * (E_TILE / 4); }   // fp16 elements per set

extern "C" int ss_layer512_tile_addend_f16(const float* E, int lde, int64_t e_batch_stride, uint16_t* out, int n_sets, int64_t set_stride, int B, int T, void* stream) {
  SS_CHECK_ARG(E && out && B > 0 && T > 0 && (lde % 4) == 0 && lde >= 512 && (e_batch_stride % 4) == 0 && (((uintptr_t)E) & 15) == 0 && (((uintptr_t)out) & 15) == 0,
               "ss_layer512_tile_addend_f16: E / out 16-byte aligned, lde %% 4 == 0 and >= 512");
  SS_CHECK_ARG(n_sets >= 1 && n_sets <= 64 && (n_sets == 1 || (set_stride >= ss_layer512_addend_halfs(B, T) && (set_stride % 8) == 0)),
               "ss_layer512_tile_addend_f16: 1 <= n_sets <= 64, set_stride >= ss_layer512_addend_halfs(B, T) and a multiple of 8");
  const int tpi = ss_cdiv(T, BM);
  const int64_t n = (int64_t)B * tpi * (E_TILE / 32);
  hipLaunchKernelGGL(tile_addend_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, E, lde, e_batch_stride, out, n_sets, set_stride, T, tpi, n);
  SS_CHECK_LAUNCH("ss_layer512_tile_addend_f16");
  return SS_OK;
}

// 1 if the fused layer launch can take this shape and is expected to pay: C = 256 (the kernel's fixed geometry), dilation <= 8, 32-bit offsets,
// and at least four rounds of 128-row tiles per CU (below that the single-round kernels win)
extern "C" int ss_layer512_ok(int B, int T, int C, int d_max, int ldg) {
  if (C != 256 || d_max < 1 || d_max > HALO || B < 1 || T < 1) return 0;
  if (ldg < 512 || (ldg % 8) != 0) return 0;
  if ((int64_t)T * ldg * 2 >= (1ll << 31) || (int64_t)B * ss_cdiv(T, BM) * H_TILE >= (1ll << 31)) return 0;
  return (g_ss_tuning.layer512 == 2 || (long)ss_cdiv(T, BM) * B >= 4l * ss_n_cu()) ? 1 : 0;   // knob = 2: any shape (parity tests run one item)
}

extern "C" int ss_layer512(const ss_layer512_args* args, void* stream) {
  SS_CHECK_ARG(args != nullptr, "ss_layer512: null args");
  const ss_layer512_args& a = *args;
  SS_CHECK_ARG(a.Hin && a.Wg && a.E512 && a.G, "ss_layer512: null Hin / Wg / E512 / G");
  SS_CHECK_ARG(a.B > 0 && a.T > 0 && a.d >= 1 && a.d <= HALO, "ss_layer512: B, T > 0 and 1 <= d <= 8");
  SS_CHECK_ARG(a.ldg >= (a.g_compact ? 256 : 512) && (a.ldg % 8) == 0 && (a.g_batch_stride % 8) == 0,
               "ss_layer512: ldg >= 512 (pair layout of 256 channels; 256 with g_compact), a multiple of 8, as the batch stride");
  SS_CHECK_ARG((int64_t)a.T * a.ldg * 2 < (1ll << 31) && (int64_t)a.B * ss_cdiv(a.T, BM) * H_TILE < (1ll << 31), "ss_layer512: too large for 32-bit offsets");
  SS_CHECK_ARG((((uintptr_t)a.Hin) & 15) == 0 && (((uintptr_t)a.Wg) & 15) == 0 && (((uintptr_t)a.E512) & 15) == 0 && (((uintptr_t)a.G) & 15) == 0,
               "ss_layer512: Hin / Wg / E512 / G must be 16-byte aligned");
  SS_CHECK_ARG(a.out_scale > 0.f && a.out_scale <= 1.f, "ss_layer512: 0 < out_scale <= 1");
  const bool fuse = a.Hout != nullptr;
  if (fuse) {
    SS_CHECK_ARG(a.Wr && a.P && a.Hout != a.Hin && (((uintptr_t)a.Hout) & 15) == 0 && (((uintptr_t)a.Wr) & 15) == 0 && (((uintptr_t)a.P) & 15) == 0,
                 "ss_layer512: the fused form needs Wr, P and an Hout buffer different from Hin (tiles read their neighbours' halo rows)");
    SS_CHECK_ARG((!a.bias_r || (((uintptr_t)a.bias_r) & 15) == 0) && (!a.next_bias || (((uintptr_t)a.next_bias) & 15) == 0) &&
                     (!a.cur_bias || (((uintptr_t)a.cur_bias) & 15) == 0), "ss_layer512: bias vectors must be 16-byte aligned");
  }
  const int tpi = ss_cdiv(a.T, BM);
  const int n_tiles = tpi * a.B;
  int grid = n_tiles < ss_n_cu() ? n_tiles : ss_n_cu();
#ifdef SS_L512_TRACE
  if (const char* e = getenv("SS_L512_GRID")) grid = atoi(e) > 0 && atoi(e) < grid ? atoi(e) : grid;   // trace builds: fewer CUs (is a phase memory-starved? it is not:
                                                                                                       // profiles/r06_trace_layer512_v2_experiments.log)
#endif
  // the last round as half tiles when it would keep at most half of the workgroups busy (see the schedule in the kernel); knob "layer512_tail"
  const int rem = n_tiles % grid;
  const int split_tail = (g_ss_tuning.layer512_tail != 0 && n_tiles >= grid && rem > 0 && 2 * rem <= grid) ? g_ss_tuning.layer512_tail : 0;
  const size_t lds = (size_t)2 * REGION;
  auto go = [&](auto kern) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      ss_set_error("ss_layer512: hipFuncSetAttribute(%d bytes of LDS): %s", (int)lds, hipGetErrorString(e));
      return SS_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, (hipStream_t)stream, a, tpi, n_tiles, split_tail, g_ss_tuning.clock_probe);
    return SS_OK;
  };
  SS_CHECK_ARG(a.n_products == 0 || a.n_products == 1 || a.n_products == 2, "ss_layer512: n_products = 1 | 2 (0 = 2)");
  SS_CHECK_ARG(!a.e_f16 || a.n_products == 1, "ss_layer512: the fp16 addend sets (e_f16) exist in the one-product form only");
  if (a.n_products == 1 && a.e_f16) SS_PROPAGATE(fuse ? go(&layer512_kernel<true, 1, true>) : go(&layer512_kernel<false, 1, true>));
  else if (a.n_products == 1) SS_PROPAGATE(fuse ? go(&layer512_kernel<true, 1>) : go(&layer512_kernel<false, 1>));
  else SS_PROPAGATE(fuse ? go(&layer512_kernel<true, 2>) : go(&layer512_kernel<false, 2>));
  SS_CHECK_LAUNCH("ss_layer512");
  return SS_OK;
}
