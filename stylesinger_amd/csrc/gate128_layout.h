// Index math of gate128_kernel (gemm_bf16_gate128.hip): which global element every LDS-DMA lane fetches, where it lands, and where every
// fragment / epilogue access reads. ONE definition, used by the kernel AND by the host-side checker (tools/layout_check_gate128.cpp, run by
// tests/test_host_cpu.py): the checker replays every DMA piece into a tagged LDS image and verifies that every read the kernel issues finds the
// element the arithmetic expects, that the images are covered exactly once, and that no ds_read_b128 has a bank conflict - the class of bug a
// kernel written without a GPU at hand is most likely to have.
#pragma once

#if defined(__HIPCC__)
#define G128_HD __host__ __device__ __forceinline__
#else
#define G128_HD inline
#endif

namespace g128 {

constexpr int BM = 256;      // rows (frames) per workgroup
constexpr int BN = 128;      // packed columns per workgroup = 64 output channels, both gate operands
constexpr int WAVES = 4;     // wave (wm, wn) = (wave >> 1, wave & 1) owns rows 128 wm .. +128, packed columns 64 wn .. +64
constexpr int HALO = 8;      // rows staged before / after the tile (dilations up to 8)
constexpr int AROWS = 320;   // staged A rows: tile row r lives in LDS row HALO + r; rows >= BM + 2 HALO never fetched
constexpr int A_ROWB = 64;   // bytes per A row in LDS: the HI plane of a 32-channel chunk = 4 slots of 16 B (8 fp16)
constexpr int B_ROWB = 128;  // bytes per B row in LDS: 32 channels x (hi | lo) = 8 slots of 16 B
constexpr int A_PIECES = AROWS * A_ROWB / 1024;   // 20 DMA instructions of 64 lanes x 16 B; wave w issues pieces w + 4 j, j < 5
constexpr int B_PIECES = BN * B_ROWB / 1024;      // 16; wave w issues pieces w + 4 j, j < 4
constexpr int E_ROWB = BN * 4;                    // 512 B of fp32 addend per tile row
constexpr int E_PIECES = 64 * E_ROWB / 1024;      // 32 per quarter of 64 rows; wave w issues pieces w + 4 j, j < 8
constexpr int OUT_ROWB = 256;                     // staged output row: 64 channels in the pair layout (32 hi | 32 lo) x 2

// ---- A: compact image. 16-byte slot s of LDS row r sits at physical slot s ^ ((r >> 2) & 3): a 16-lane group of a ds_read_b128 (16
// consecutive rows, one logical slot) then touches 16 distinct 16-byte units of the 256-byte bank window
G128_HD int a_swz(int row) { return (row >> 2) & 3; }
// DMA piece p, lane i: lands at byte p * 1024 + i * 16 = (row 16 p + (i >> 2), physical slot i & 3) and therefore fetches ...
G128_HD int a_dma_row(int piece, int lane) { return 16 * piece + (lane >> 2); }
G128_HD int a_dma_slot(int piece, int lane) { return (lane & 3) ^ a_swz(a_dma_row(piece, lane)); }   // ... this logical slot of that row
G128_HD int a_dma_lds(int piece, int lane) { return piece * 1024 + lane * 16; }
// fragment read of acc block m, tap row shift sh = (tap - 1) * d, k-step ks: lane (l31, lh) reads 8 channels 16 ks + 8 lh of tile row
// 128 wm + 32 m + l31 + sh
G128_HD int a_frag_row(int wm, int m, int l31, int sh) { return HALO + sh + 128 * wm + 32 * m + l31; }
G128_HD int a_frag_lds(int row, int ks, int lh) { return row * A_ROWB + (((2 * ks + lh) ^ a_swz(row)) << 4); }

// ---- B: pair image, the layout of gate256_kernel. slot s of row r at physical slot s ^ ((r >> 1) & 7)
G128_HD int b_swz(int row) { return (row >> 1) & 7; }
G128_HD int b_dma_row(int piece, int lane) { return 8 * piece + (lane >> 3); }
G128_HD int b_dma_slot(int piece, int lane) { return (lane & 7) ^ b_swz(b_dma_row(piece, lane)); }
G128_HD int b_dma_lds(int piece, int lane) { return piece * 1024 + lane * 16; }
// fragment read: packed column (weight row) 64 wn + 32 n + l31; plane 0 = hi (slots 0-3), 1 = lo (slots 4-7)
G128_HD int b_frag_row(int wn, int n, int l31) { return 64 * wn + 32 * n + l31; }
G128_HD int b_frag_lds(int row, int plane, int ks, int lh) { return row * B_ROWB + (((4 * plane + 2 * ks + lh) ^ b_swz(row)) << 4); }

// ---- E (fp32 conditioner addend): quarter q = tile rows 128 h + 32 q + (0..31), h = 0, 1 -> 64 LDS rows k = 32 h + (0..31) of 512 B.
// piece p (two rows), lane i: LDS row 2 p + (i >> 5), bytes 16 (i & 31) ..
G128_HD int e_dma_k(int piece, int lane) { return 2 * piece + (lane >> 5); }
G128_HD int e_dma_tile_row(int piece, int lane, int q) {
  const int k = e_dma_k(piece, lane);
  return 128 * (k >> 5) + 32 * q + (k & 31);
}
G128_HD int e_dma_col_byte(int lane) { return (lane & 31) * 16; }
G128_HD int e_dma_lds(int piece, int lane) { return piece * 1024 + lane * 16; }
// accumulator element r of block q of wave (wm, wn), lane (l31, lh): tile row 128 wm + 32 q + 4 lh + rr(r), packed column 64 wn + l31 (+ 32)
G128_HD int acc_rr(int r) { return (r & 3) + 8 * (r >> 2); }
G128_HD int e_read_lds(int wm, int wn, int l31, int lh, int r, int second) {
  return (32 * wm + 4 * lh + acc_rr(r)) * E_ROWB + (64 * wn + l31 + 32 * second) * 4;
}

// ---- OUT staging (64 rows x 256 B) and the 16-byte stores that drain it
G128_HD int out_write_lds(int wm, int wn, int l31, int lh, int r) { return (32 * wm + 4 * lh + acc_rr(r)) * OUT_ROWB + wn * 128 + l31 * 2; }
// store piece p = tid + 256 j (j < 4): staged row p >> 4, 16 bytes (p & 15) of it; pieces with (p & 4) hold the second plane: not stored
G128_HD int out_store_k(int p) { return p >> 4; }
G128_HD int out_store_c16(int p) { return p & 15; }
G128_HD int out_store_tile_row(int p, int q) {
  const int k = out_store_k(p);
  return 128 * (k >> 5) + 32 * q + (k & 31);
}

}  // namespace g128

// ------------------------------------------------------------------------------------------------------------------------------------------
// gate128q_kernel (gemm_bf16_gate128q.hip): gate128_kernel with its SECOND product (activation x weight-lo) on the block-scaled fp4 matrix
// instruction. Steps S = 3 cc + tap are paired in issue order: pair p = steps (2 p, 2 p + 1). Over a pair, lane (row, h) of the fp16 product
// holds in its four A fragments the 32 values element e = 16 (S & 1) + 8 ks + t  <->  K index of step S, channel 16 ks + 8 h + t (t < 8); it
// converts them to fp4 in registers. v_mfma_scale_f32_32x32x64_f8f6f4 pairs lane (i, h) of A with lane (j, h) of B element by element
// (tools/ubench/mfma_mx_layout.hip), so the weights' lo plane is packed ONCE in that element order: for packed column j, pair p, lane half h
// the 32 nibbles sit in logical slot 4 + h of the weight line of the pair's ODD step (element e in nibble e & 1 of byte e >> 1) and their E8M0
// block scale in byte h of logical slot 6 of the same line. Lines of even steps carry nothing in slots 4-7.
namespace g128q {

constexpr int CCS = 8;          // chunks of 32 channels per tap (K = 256)
constexpr int STEPS = 3 * CCS;  // 24
constexpr int PAIRS = STEPS / 2;

G128_HD int step_tap(int S) { return S % 3; }
G128_HD int step_chunk(int S) { return S / 3; }
// weight line (128 bytes) of step S inside a packed row: the rows are tap-major
G128_HD int step_line(int S) { return step_tap(S) * CCS + step_chunk(S); }
// K index (tap * K + channel) of element e of lane half h in pair p
G128_HD int q_kindex(int p, int h, int e, int K) {
  const int S = 2 * p + (e >> 4), ks = (e >> 3) & 1, t = e & 7;
  return step_tap(S) * K + 32 * step_chunk(S) + 16 * ks + 8 * h + t;
}
// where the operand register r (0..3) of the A-side conversion comes from: step parity r >> 1, k-step r & 1
G128_HD int q_reg_parity(int r) { return r >> 1; }
G128_HD int q_reg_ks(int r) { return r & 1; }

}  // namespace g128q

// tile256q_store_kernel (gemm_bf16_tile256q.hip): the same pairing for a 1-tap GEMM - pair p = chunks (2 p, 2 p + 1); element e of lane half h:
// chunk 2 p + (e >> 4), k-step (e >> 3) & 1, channel 16 ks + 8 h + (e & 7) of it. The fp4 terms sit in the weight line of chunk 2 p + 1.
G128_HD int t128q_kindex(int p, int h, int e) { return 32 * (2 * p + (e >> 4)) + 16 * ((e >> 3) & 1) + 8 * h + (e & 7); }
