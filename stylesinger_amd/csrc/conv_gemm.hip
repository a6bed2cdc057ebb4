// Host dispatcher of the generic fp32-MFMA implicit-GEMM conv (kernel: conv_gemm_kernel.h).
#include "common.h"
#include "../../include/stylesinger_hip.h"

int ss_conv_gemm_launch_store(int tile, const ss_conv_gemm_args& a, hipStream_t stream);
int ss_conv_gemm_launch_gate(int tile, const ss_conv_gemm_args& a, hipStream_t stream);
int ss_conv_gemm_launch_resskip(int tile, const ss_conv_gemm_args& a, hipStream_t stream);
int ss_conv_gemm_launch_ddpm(int tile, const ss_conv_gemm_args& a, hipStream_t stream);
static constexpr int BK = 32;

extern "C" int ss_conv_gemm(const ss_conv_gemm_args* args, void* stream_) {
  SS_CHECK_ARG(args != nullptr, "ss_conv_gemm: null args");
  const ss_conv_gemm_args& a = *args;
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(a.A && a.W && a.C, "ss_conv_gemm: null A/W/C");
  SS_CHECK_ARG(a.B > 0 && a.T > 0 && a.N > 0, "ss_conv_gemm: bad dims B=%d T=%d N=%d", a.B, a.T, a.N);
  SS_CHECK_ARG(a.ntaps >= 1 && a.ntaps <= SS_MAX_TAPS, "ss_conv_gemm: ntaps=%d out of range", a.ntaps);
  SS_CHECK_ARG((a.lda & 3) == 0 && (a.Cin & 3) == 0, "ss_conv_gemm: lda=%d/Cin=%d must be multiples of 4", a.lda, a.Cin);
  SS_CHECK_ARG((a.Kp % BK) == 0 && a.Kp >= a.Cin, "ss_conv_gemm: Kp=%d must be a multiple of 32 and >= Cin=%d", a.Kp, a.Cin);
  SS_CHECK_ARG((a.Np & 31) == 0, "ss_conv_gemm: Np=%d must be a multiple of 32", a.Np);
  SS_CHECK_ARG(a.epi == SS_EPI_GATE || a.Np >= a.N, "ss_conv_gemm: Np=%d must be >= N=%d", a.Np, a.N);
  SS_CHECK_ARG((((uintptr_t)a.A) & 15) == 0 && (((uintptr_t)a.W) & 15) == 0, "ss_conv_gemm: A/W must be 16-byte aligned");
  SS_CHECK_ARG((a.a_batch_stride & 3) == 0, "ss_conv_gemm: a_batch_stride must be a multiple of 4");
  if (a.epi == SS_EPI_GATE) SS_CHECK_ARG((a.Np & 63) == 0, "ss_conv_gemm: GATE needs Np multiple of 64 (got %d)", a.Np);
  if (a.epi == SS_EPI_RESSKIP) SS_CHECK_ARG(a.R && a.C2 && (a.Nh & 31) == 0, "ss_conv_gemm: RESSKIP needs R, C2 and Nh%%32==0");
  SS_CHECK_ARG(a.epi >= SS_EPI_STORE && a.epi <= SS_EPI_DDPM, "ss_conv_gemm: bad epilogue %d", a.epi);

  const int n_cols = (a.epi == SS_EPI_GATE) ? a.Np : a.N;
  const bool gate = a.epi == SS_EPI_GATE;
  auto blocks = [&](int bm, int bn) { return (long)ss_cdiv(a.T, bm) * a.B * ss_cdiv(n_cols, bn); };
  int tile = a.tile;
  if (tile == 0) {
    if (n_cols <= 32) tile = SS_TILE_128x32;
    else if (n_cols <= 64) tile = (gate || blocks(128, 64) >= 512) ? SS_TILE_128x64 : SS_TILE_64x64;
    // 65 .. 96 columns over many rounds with a short K (the mel denoiser's output projection at BASELINE configs[3]: 180 000 rows, K = 256, N = 80, fused
    // with the DDPM update): three 32-column tiles waste nothing of a 128-column tile's matrix work and epilogue, and the A tile's three readers run side by
    // side out of L2 - 138 us against 327 us (tools/kbench_final.py, profiles/r06_kbench_final_projection.log)
    else if (!gate && n_cols <= 96 && a.Cin * a.ntaps <= 256 && blocks(128, 32) >= 3072) tile = SS_TILE_128x32;
    else if (blocks(128, 128) >= 768) tile = SS_TILE_128x128;
    else if (gate || blocks(64, 128) >= 384) tile = SS_TILE_64x128;
    else tile = SS_TILE_64x64;
  }
  if (gate && (tile == SS_TILE_128x32 || tile == SS_TILE_64x64)) tile = SS_TILE_64x128;
  switch (a.epi) {
    case SS_EPI_STORE: return ss_conv_gemm_launch_store(tile, a, stream);
    case SS_EPI_GATE: return ss_conv_gemm_launch_gate(tile, a, stream);
    case SS_EPI_RESSKIP: return ss_conv_gemm_launch_resskip(tile, a, stream);
    default: return ss_conv_gemm_launch_ddpm(tile, a, stream);
  }
}
