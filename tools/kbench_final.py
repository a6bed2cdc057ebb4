"""The output projection of a mel-denoiser evaluation fused with the DDPM update (ss_conv_gemm, SS_EPI_DDPM: K = 256 -> N = 80 on the skip GEMM's fp32
output) at BASELINE configs[3]'s shape, per tile choice and with / without the in-kernel Philox noise: where do its 371 us go?
    python tools/kbench_final.py [--B 32] [--T 5625]"""
import argparse
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=5625)
    ap.add_argument("--iters", type=int, default=100)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    B, T, Cc, M = a.B, a.T, 256, 80
    lens = torch.full((B,), T, device=d, dtype=torch.int32)
    G = torch.relu(torch.randn(B, T, Cc, device=d))
    w = torch.randn(M, Cc, 1, device=d) / math.sqrt(Cc)
    Wf = L.pack_conv_weight(w)
    bf = L.pack_bias(torch.randn(M, device=d))
    x = torch.randn(B, T, M, device=d)
    names = {0: "auto", 1: "128x128", 2: "64x128", 3: "64x64", 4: "128x64", 5: "128x32"}
    for sigma in (0.1, 0.0):
        for tile in (0, 1, 2, 3, 4, 5):
            args = L._fill_args(G, Wf, x, B=B, T=T, Cin=Cc, N=M, Np=Wf.shape[0], Kp=Wf.shape[1], epi=L.EPI_DDPM, bias=bf, ldc=M, tile=tile, lens=lens)
            args.ddpm_recip, args.ddpm_recipm1, args.ddpm_c1, args.ddpm_c2, args.ddpm_sigma = 1.05, 0.3, 0.4, 0.6, sigma
            args.seed, args.step = 1234, 7

            def run():
                L.check(L.load().ss_conv_gemm(C.byref(args), L.stream_ptr()), "ss_conv_gemm")
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            by = B * T * (Cc * 4 + 2 * M * 4)
            print(f"output projection + DDPM update, tile {names[tile]:8s} sigma {sigma}: {us:8.1f} us   {by / us / 1e6:5.2f} TB/s algorithmic   {2.0 * B * T * Cc * M / us / 1e6:6.1f} TF/s")


if __name__ == "__main__":
    main()
