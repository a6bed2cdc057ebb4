// conv_gemm instantiations for the SS_EPI_GATE epilogue (split per epilogue so hipcc builds them in parallel)
#include "conv_gemm_kernel.h"
int ss_conv_gemm_launch_gate(int tile, const ss_conv_gemm_args& a, hipStream_t stream) {
  return launch_tile<SS_EPI_GATE>(tile, a, stream);
}
