"""ss_layer512 (one launch per residual layer of the fp16x2 mel denoiser) against the launch pair it replaces, at the BASELINE configs[3] shape.
    python tools/kbench_layer512.py [--B 32] [--T 5625] [--iters 40]
Every variant cycles through four addend slabs and alternates the two stream buffers, as the denoiser loop does."""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L  # noqa: E402
from tools.kbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=5625)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--which", default="all")
    ap.add_argument("--lo-bits", type=int, default=0, help="experiment: keep only this many mantissa bits of the weights' lo terms (0 = all 10): does the "
                    "matrix pipe draw less power - and the chip clock higher - when one operand's low mantissa bits are zero?")
    ap.add_argument("--zero-lo", action="store_true", help="experiment: lo terms = 0 (the power a second product of zeros draws)")
    ap.add_argument("--e16", action="store_true", help="with --one: the addend as fp16 sets (ss_layer512_args.e_f16), one set per launch - the fp16sd loop's form")
    ap.add_argument("--one", action="store_true", help="n_products = 1: ONE fp16 weight term (the fp16sd mode's launch; the pair column still shows fp16x2's two launches)")
    a = ap.parse_args()
    d = torch.device("cuda:0")
    B, T, C, NS = a.B, a.T, 256, 4
    lens = torch.full((B,), T, device=d, dtype=torch.int32)
    X0 = torch.randn(B, T, C, device=d)
    Y = [L.split_f16(X0)]
    H0, P = L.layer512_entry(X0, None, B=B, T=T, lens=lens)
    H = [H0, torch.empty_like(H0)]
    E = torch.randn(B, T, NS * 2 * C, device=d)
    E512 = [(L.layer512_tile_addend_f16(E[..., s * 2 * C:], 1, B=B, T=T, lde=NS * 2 * C)[0] if a.e16 else L.layer512_tile_addend(E[..., s * 2 * C:], B=B, T=T, lde=NS * 2 * C))
            for s in range(NS)]
    GA = torch.empty(B, T, NS * 2 * C, device=d, dtype=torch.float16)
    w = torch.randn(2 * C, C, 3, device=d) / math.sqrt(3 * C)
    Ws = L.split_f16(L.pack_conv_weight(w, interleave_half=C), scale=256.0)
    wo = torch.randn(2 * C, C, 1, device=d) / math.sqrt(C)
    Wos = L.split_f16(L.pack_conv_weight(wo), scale=256.0)
    def coarse(Wp):   # the (hi | lo) pack, pairs interleaved by 32: lo terms at [.., 32:64] of every 64
        v = Wp.view(Wp.shape[0], -1, 64)
        lo = v[:, :, 32:].float()
        if a.zero_lo:
            lo = torch.zeros_like(lo)
        elif a.lo_bits:
            m, e = torch.frexp(lo)
            lo = torch.ldexp(torch.round(m * 2.0 ** (a.lo_bits + 1)) / 2.0 ** (a.lo_bits + 1), e)
        v[:, :, 32:] = lo.to(torch.float16)
        return Wp
    if a.lo_bits or a.zero_lo:
        Ws, Wos = coarse(Ws), coarse(Wos)
    NP = 1 if a.one else 2
    Wg, Wr = L.layer512_pack_gate(Ws, NP), L.layer512_pack_res(Wos, NP)
    cb, nb, bo = (torch.randn(C, device=d) for _ in range(3))
    bop = L.pack_bias(bo)
    k = [0]
    res = []
    fl_gate = 2.0 * 2.0 * B * T * 3 * C * 2 * C
    fl_res = 2.0 * 2.0 * B * T * C * C
    fl1 = 0.5 if a.one else 1.0   # executed flops of the layer512 rows with one product

    def pair():
        k[0] += 1
        s = k[0] % NS
        L.gemm_bf16(Y[0], Ws, B=B, T=T, K=C, taps=(-2, 0, 2), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens, E=E[..., s * 2 * C:], lde=NS * 2 * C,
                    out=GA[..., s * 2 * C:], ldc=NS * 2 * C, c_bs=T * NS * 2 * C, lda=2 * C, split=2, out_scale=1.0 / 256.0)
        L.gemm_bf16(GA[..., s * 2 * C:], Wos, B=B, T=T, K=C, taps=(0,), N=C, Np=Wos.shape[0], epi=L.HEPI_RESX, lens=lens, bias=bop, X=None, post_scale=0.7071,
                    next_bias=nb, Y=Y[0], lda=NS * 2 * C, a_bs=T * NS * 2 * C, split=2, out_scale=1.0 / 256.0, cur_bias=cb)

    def gate_only_old():
        k[0] += 1
        s = k[0] % NS
        L.gemm_bf16(Y[0], Ws, B=B, T=T, K=C, taps=(-2, 0, 2), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens, E=E[..., s * 2 * C:], lde=NS * 2 * C,
                    out=GA[..., s * 2 * C:], ldc=NS * 2 * C, c_bs=T * NS * 2 * C, lda=2 * C, split=2, out_scale=1.0 / 256.0)

    def fused():
        k[0] += 1
        s = k[0] % NS
        L.layer512(H[k[0] & 1], Wg, E512[s], GA[..., s * 2 * C:], B=B, T=T, d=2, lens=lens, Hout=H[(k[0] & 1) ^ 1], P=P, Wr=Wr, bias_r=bo, next_bias=nb, cur_bias=nb,
                   ldg=NS * 2 * C, g_bs=T * NS * 2 * C, n_products=NP, e_f16=a.e16)

    def gate_only_new():
        k[0] += 1
        s = k[0] % NS
        L.layer512(H[0], Wg, E512[s], GA[..., s * 2 * C:], B=B, T=T, d=2, lens=lens, ldg=NS * 2 * C, g_bs=T * NS * 2 * C, n_products=NP, e_f16=a.e16)

    for name, fn, fl in (("gate128 + tile256 RESX (the launch pair)", pair, fl_gate + fl_res), ("gate128 alone", gate_only_old, fl_gate),
                         ("layer512 fused (gate + residual projection)" + (", ONE product" if a.one else ""), fused, (fl_gate + fl_res) * fl1),
                         ("layer512 gate only" + (", ONE product" if a.one else ""), gate_only_new, fl_gate * fl1),
                         ("layer512 entry (fp32 stream -> H rows + pairs)", lambda: L.layer512_entry(X0, cb, B=B, T=T, lens=lens), 0.0)):
        if a.which != "all" and a.which not in name:
            continue
        s = timeit(fn, a.iters)
        print(f"{name:46s} {s * 1e6:9.1f} us  {fl / s / 1e12:7.1f} TF/s executed ({fl / s / 2.5e15 * 100:4.1f}% of the fp16 peak)")


if __name__ == "__main__":
    main()
