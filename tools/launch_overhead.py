"""Fixed cost of a dependent launch in the diffusion loops (lead for the next round).

Times, graph-replayed back to back: (a) the residual-half output projection at the C2 shape (1.57 GFLOP, MFMA work ~11 us)
and (b) the same launch with K cut to one 32-chunk, i.e. almost pure launch + prologue + epilogue + drain.
    python tools/launch_overhead.py
"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stylesinger_amd import lib as L  # noqa: E402
from kbench import timeit  # noqa: E402

d = torch.device("cuda:0")
B, T, C = 8, 1500, 256
lens = torch.full((B,), T, device=d, dtype=torch.int32)
G = torch.randn(B, T, C, device=d)
X = torch.randn(B, T, C, device=d)
S = torch.zeros(B, T, C, device=d)
for K in (256, 32):
    wo = torch.randn(C, K, 1, device=d) / math.sqrt(K)
    Wo = L.pack_conv_weight(wo)
    bo = L.pack_bias(torch.randn(C, device=d) * 0.1)

    def f():
        L.conv_gemm(G, Wo, X, B=B, T=T, Cin=K, N=C, Np=Wo.shape[0], Kp=Wo.shape[1], lda=C, lens=lens, epi=L.EPI_RESSKIP, bias=bo, Nh=C,
                    R=X, ldr=C, ldc=C, post_scale=0.7071, C2=S, ldc2=C, c2_bs=T * C, tile=3)
    s = timeit(f, 100)
    fl = 2.0 * B * T * K * C
    print(f"res-half projection K={K:3d}: {s * 1e6:6.1f} us per dependent launch ({fl / 157.3e12 * 1e6:5.1f} us of MFMA work at the fp32 peak)")
