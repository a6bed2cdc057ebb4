// Host-side check of gate128_kernel's index math (stylesinger_amd/csrc/gate128_layout.h - the SAME header the kernel compiles): every LDS-DMA
// piece is replayed into a tagged LDS image, then every access the kernel issues is looked up in it.
//   g++ -std=c++17 -I stylesinger_amd/csrc tools/layout_check_gate128.cpp -o /tmp/layout_check_gate128 && /tmp/layout_check_gate128
// Checked: (1) the per-wave "piece w + 4 j = piece w shifted by a constant number of rows, same swizzle" identities the kernel's single
// per-lane offsets rely on; (2) each image (A, B, addend quarter, output staging) is covered exactly once; (3) every fragment / addend read
// finds the element the MFMA lane layout needs; (4) no ds_read_b128 has a bank conflict (64 banks x 4 B, 16 lanes per pass); (5) every staged
// output lands on the channel / row it belongs to and only hi halves are stored. Run by tests/test_host_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "gate128_layout.h"

using namespace g128;

static int fails = 0;
#define CHECK(cond, ...)                        \
  do {                                          \
    if (!(cond)) {                              \
      if (fails < 20) {                         \
        std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
        std::printf(__VA_ARGS__);               \
        std::printf("\n");                      \
      }                                         \
      ++fails;                                  \
    }                                           \
  } while (0)

struct Tag { int row = -1, slot = -1; };

// one ds_read_b128 of 64 lanes: the hardware serves 16 lanes per pass; inside a pass the sixteen 16-byte units must sit in distinct
// quarters-of-bank-window, i.e. (byte >> 4) & 15 all different
static void check_b128_conflicts(const int (&byte)[64], const char* what) {
  for (int g = 0; g < 4; ++g) {
    std::set<int> seen;
    for (int i = 16 * g; i < 16 * g + 16; ++i) seen.insert((byte[i] >> 4) & 15);
    CHECK(seen.size() == 16, "%s: bank conflict in lanes %d..%d (%zu distinct units)", what, 16 * g, 16 * g + 15, seen.size());
  }
}

int main() {
  // ------------------------------------------------------------------ A
  {
    std::vector<Tag> img(AROWS * A_ROWB / 16);
    for (int w = 0; w < WAVES; ++w)
      for (int j = 0; j < 5; ++j)
        for (int lane = 0; lane < 64; ++lane) {
          const int p = w + 4 * j;
          CHECK(a_dma_row(p, lane) == a_dma_row(w, lane) + 64 * j, "A piece identity (row) w=%d j=%d lane=%d", w, j, lane);
          CHECK(a_dma_slot(p, lane) == a_dma_slot(w, lane), "A piece identity (slot) w=%d j=%d lane=%d", w, j, lane);
          const int row = a_dma_row(p, lane), slot = a_dma_slot(p, lane), byte = a_dma_lds(p, lane);
          CHECK(slot >= 0 && slot < 4 && row >= 0 && row < AROWS, "A dma range");
          CHECK(byte == row * A_ROWB + ((slot ^ a_swz(row)) << 4), "A dma lands off its swizzled slot: p=%d lane=%d", p, lane);
          Tag& t = img[byte >> 4];
          CHECK(t.row < 0, "A image unit written twice");
          t.row = row;
          t.slot = slot;
        }
    for (const Tag& t : img) CHECK(t.row >= 0, "A image unit never written");
    CHECK(A_PIECES == 20, "A_PIECES");
    for (int d : {1, 2, 4, 8})
      for (int tap = 0; tap < 3; ++tap)
        for (int wm = 0; wm < 2; ++wm)
          for (int m = 0; m < 4; ++m)
            for (int ks = 0; ks < 2; ++ks) {
              int byte[64];
              for (int lane = 0; lane < 64; ++lane) {
                const int l31 = lane & 31, lh = lane >> 5;
                // the kernel's expression: a_base[tap] + ((2 ks ^ a_sw[tap]) << 4) + m * 32 * A_ROWB, a_base / a_sw from the m = 0 row
                const int row0 = a_frag_row(wm, 0, l31, (tap - 1) * d);
                const int a_base = row0 * A_ROWB, a_sw = a_swz(row0) ^ lh;
                byte[lane] = a_base + (((2 * ks) ^ a_sw) << 4) + m * 32 * A_ROWB;
                const int want_row = HALO + (tap - 1) * d + 128 * wm + 32 * m + l31;
                CHECK(want_row >= 0 && want_row < BM + 2 * HALO, "A fragment row %d outside the staged rows", want_row);
                CHECK(byte[lane] == a_frag_lds(want_row, ks, lh), "A fragment expression != a_frag_lds");
                const Tag& t = img[byte[lane] >> 4];
                CHECK(t.row == want_row && t.slot == 2 * ks + lh, "A fragment d=%d tap=%d wm=%d m=%d ks=%d lane=%d reads (row %d, slot %d), wants (%d, %d)", d, tap,
                      wm, m, ks, lane, t.row, t.slot, want_row, 2 * ks + lh);
              }
              check_b128_conflicts(byte, "A fragment");
            }
    // rows the tail pieces must still cover: the highest row any fragment reads is HALO + 8 + 255 = 271 = piece 16 (wave 0, j = 4)
    CHECK(a_dma_row(16, 63) == 271, "A tail piece");
  }
  // ------------------------------------------------------------------ B
  {
    std::vector<Tag> img(BN * B_ROWB / 16);
    for (int w = 0; w < WAVES; ++w)
      for (int j = 0; j < 4; ++j)
        for (int lane = 0; lane < 64; ++lane) {
          const int p = w + 4 * j;
          CHECK(b_dma_row(p, lane) == b_dma_row(w, lane) + 32 * j, "B piece identity (row)");
          CHECK(b_dma_slot(p, lane) == b_dma_slot(w, lane), "B piece identity (slot)");
          const int row = b_dma_row(p, lane), slot = b_dma_slot(p, lane), byte = b_dma_lds(p, lane);
          CHECK(byte == row * B_ROWB + ((slot ^ b_swz(row)) << 4), "B dma lands off its swizzled slot");
          Tag& t = img[byte >> 4];
          CHECK(t.row < 0, "B image unit written twice");
          t.row = row;
          t.slot = slot;
        }
    for (const Tag& t : img) CHECK(t.row >= 0, "B image unit never written");
    CHECK(B_PIECES == 16, "B_PIECES");
    for (int wn = 0; wn < 2; ++wn)
      for (int n = 0; n < 2; ++n)
        for (int slot4 : {0, 2, 4, 6}) {   // the kernel's rd_b(slot): 4 plane + 2 ks
          int byte[64];
          for (int lane = 0; lane < 64; ++lane) {
            const int l31 = lane & 31, lh = lane >> 5;
            const int row0 = b_frag_row(wn, 0, l31);
            const int b_base = row0 * B_ROWB, b_sw = b_swz(row0) ^ lh;
            byte[lane] = b_base + ((slot4 ^ b_sw) << 4) + n * 32 * B_ROWB;
            const int want_row = 64 * wn + 32 * n + l31;
            CHECK(byte[lane] == b_frag_lds(want_row, slot4 >> 2, (slot4 >> 1) & 1, lh), "B fragment expression != b_frag_lds");
            const Tag& t = img[byte[lane] >> 4];
            CHECK(t.row == want_row && t.slot == slot4 + lh, "B fragment wn=%d n=%d slot=%d lane=%d reads (row %d, slot %d)", wn, n, slot4, lane, t.row, t.slot);
          }
          check_b128_conflicts(byte, "B fragment");
        }
  }
  // ------------------------------------------------------------------ E (one quarter; tags = tile row, byte column)
  for (int q = 0; q < 4; ++q) {
    std::vector<Tag> img(64 * E_ROWB / 16);
    for (int w = 0; w < WAVES; ++w)
      for (int j = 0; j < 8; ++j)
        for (int lane = 0; lane < 64; ++lane) {
          const int p = w + 4 * j;
          // the kernel: per-lane row 2 w + (lane >> 5), SGPR row offset q * 32 + (j < 4 ? 8 j : 128 + 8 (j - 4))
          const int krow = 2 * w + (lane >> 5) + q * 32 + (j < 4 ? 8 * j : 128 + 8 * (j - 4));
          CHECK(krow == e_dma_tile_row(p, lane, q), "E piece identity: w=%d j=%d lane=%d q=%d: %d vs %d", w, j, lane, q, krow, e_dma_tile_row(p, lane, q));
          const int byte = e_dma_lds(p, lane);
          CHECK(byte == e_dma_k(p, lane) * E_ROWB + e_dma_col_byte(lane), "E dma lands off its row");
          Tag& t = img[byte >> 4];
          CHECK(t.row < 0, "E image unit written twice");
          t.row = krow;
          t.slot = e_dma_col_byte(lane);
        }
    for (const Tag& t : img) CHECK(t.row >= 0, "E image unit never written");
    CHECK(E_PIECES == 32, "E_PIECES");
    for (int wave = 0; wave < WAVES; ++wave)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r)
          for (int second = 0; second < 2; ++second) {
            const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
            // the kernel: e_rd (r = 0, first operand) + rr * E_ROWB (+ 128)
            const int byte = e_read_lds(wm, wn, l31, lh, 0, 0) + acc_rr(r) * E_ROWB + 128 * second;
            CHECK(byte == e_read_lds(wm, wn, l31, lh, r, second), "E read expression");
            const Tag& t = img[byte >> 4];
            const int want_row = 128 * wm + 32 * q + 4 * lh + acc_rr(r), want_col = (64 * wn + l31 + 32 * second) * 4;
            CHECK(t.row == want_row && t.slot + (byte & 15) == want_col, "E read wave=%d lane=%d r=%d: (row %d, col %d) wants (%d, %d)", wave, lane, r, t.row,
                  t.slot + (byte & 15), want_row, want_col);
          }
  }
  // ------------------------------------------------------------------ OUT staging -> stores
  for (int q = 0; q < 4; ++q) {
    struct W { int row = -1, ch = -1; };
    std::vector<W> img(64 * OUT_ROWB / 2);   // 2-byte elements
    for (int wave = 0; wave < WAVES; ++wave)
      for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 16; ++r) {
          const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lh = lane >> 5;
          const int byte = out_write_lds(wm, wn, l31, lh, 0) + acc_rr(r) * OUT_ROWB;   // the kernel: o_wr + rr * OUT_ROWB
          CHECK(byte == out_write_lds(wm, wn, l31, lh, r), "OUT write expression");
          CHECK(byte >= 0 && byte + 2 <= 64 * OUT_ROWB, "OUT write range");
          W& e = img[byte >> 1];
          CHECK(e.row < 0, "OUT element written twice");
          e.row = 128 * wm + 32 * q + 4 * lh + acc_rr(r);
          e.ch = 32 * wn + l31;   // output channel inside the tile
        }
    int stored = 0;
    for (int p = 0; p < 1024; ++p) {   // tid + 256 j
      const int c16 = out_store_c16(p), trow = out_store_tile_row(p, q);
      const bool hi = !(c16 & 4);
      for (int e = 0; e < 8; ++e) {
        const W& w = img[(p * 16 >> 1) + e];
        if (!hi) {
          CHECK(w.row < 0, "a value was staged into a second-plane slot (piece %d)", p);
          continue;
        }
        // global position of this element: row t0 + trow, physical element n0 + c16 * 8 + e; the pair layout wants channel c of the tile
        // (n0 / 2 + c) at physical element n0 + (c >> 5) * 64 + (c & 31)
        CHECK(w.row == trow, "OUT store piece %d e=%d: row %d wants %d", p, e, w.row, trow);
        CHECK(w.ch >= 0 && (w.ch >> 5) * 64 + (w.ch & 31) == c16 * 8 + e, "OUT store piece %d e=%d: channel %d at physical element %d", p, e, w.ch, c16 * 8 + e);
        ++stored;
      }
    }
    CHECK(stored == 64 * 64, "OUT: %d of %d values stored", stored, 64 * 64);   // 64 rows x 64 output channels per pass
  }
  // ================================================================== gate128q_kernel: element order of the block-scaled second product
  {
    using namespace g128q;
    std::vector<int> seen(3 * 256, 0);
    for (int p = 0; p < PAIRS; ++p)
      for (int h = 0; h < 2; ++h)
        for (int e = 0; e < 32; ++e) {
          const int k = q_kindex(p, h, e, 256);
          CHECK(k >= 0 && k < 768, "g128q K index range");
          ++seen[k];
          // element e sits in operand register e >> 3 (8 fp4 per register); the kernel fills register 2 * (S & 1) + ks from the fragment of step S,
          // k-step ks, whose lane half h holds channels 16 ks + 8 h + t (the A fragment check above: logical slot 2 ks + lh = channels 8 (2 ks + lh) ..)
          const int r = e >> 3, t = e & 7, S = 2 * p + q_reg_parity(r), ks = q_reg_ks(r);
          CHECK(k == step_tap(S) * 256 + 32 * step_chunk(S) + 8 * (2 * ks + h) + t, "g128q element (p=%d h=%d e=%d) != fragment (S=%d ks=%d t=%d)", p, h, e, S, ks, t);
        }
    for (int k = 0; k < 768; ++k) CHECK(seen[k] == 1, "g128q: K index %d covered %d times", k, seen[k]);
    for (int S = 0; S < STEPS; ++S) CHECK(step_line(S) == (S % 3) * CCS + S / 3 && step_line(S) < 24, "g128q weight line of step %d", S);
    // tile256q_store_kernel: pairs of consecutive chunks of a 1-tap GEMM (K = 5120: 80 pairs)
    std::vector<int> seen1(5120, 0);
    for (int p = 0; p < 80; ++p)
      for (int h = 0; h < 2; ++h)
        for (int e = 0; e < 32; ++e) {
          const int k = t128q_kindex(p, h, e), r = e >> 3, t = e & 7, c = 2 * p + (r >> 1), ks = r & 1;
          CHECK(k == 32 * c + 8 * (2 * ks + h) + t && k >= 0 && k < 5120, "t128q element (p=%d h=%d e=%d)", p, h, e);
          ++seen1[k];
        }
    for (int k = 0; k < 5120; ++k) CHECK(seen1[k] == 1, "t128q: K index %d covered %d times", k, seen1[k]);
  }
  if (fails) {
    std::printf("layout_check_gate128: %d check(s) FAILED\n", fails);
    return 1;
  }
  std::printf("layout_check_gate128: all checks passed (A 20 pieces / 96 fragment reads x 4 dilations, B 16 / 8, addend 32 x 4 quarters, 4096 x 4 staged outputs)\n");
  return 0;
}
