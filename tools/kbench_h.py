"""Micro-benchmark of the bf16-in-HBM GEMM launches of the denoiser (ss_gemm_bf16: gate, residual projection) at a given size.
    python tools/kbench_h.py [--B 32] [--T 5625]        (C4 size by default)
"""
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
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--which", default="all")
    ap.add_argument("--split", action="store_true", help="bf16x2: (hi, mid) operand pairs, three products (ss_gemm_bf16_args.split)")
    ap.add_argument("--f16", action="store_true", help="fp16x2: fp16 terms, weights-only split, two products (ss_gemm_bf16_args.split = 2); implies --split")
    ap.add_argument("--q4", action="store_true", help="with --f16: the gate on ss_gemm_bf16_gate128q (fp16q4: second product on the block-scaled fp4 instruction; NOT yet validated)")
    ap.add_argument("--gate128", action="store_true", help="with --f16: the gate on ss_gemm_bf16_gate128 (256 x 128 tiles, two workgroups per CU)")
    ap.add_argument("--pair-only", action="store_true", help="res with --split: the pair-only residual stream (X = NULL, Y read + rewritten in place)")
    ap.add_argument("--no-e", action="store_true", help="gate without the conditioner addend (what-if: how much of the launch is the addend?)")
    ap.add_argument("--compact", type=int, default=0, help="with --f16 --which skip: 1 = compact A (hi terms only), 2 = + compact one-term weights (one_product = 2: the fp16sd "
                    "skip GEMM; knob SS_SKIP_DENSE=0 keeps its 32-channel steps)")
    ap.add_argument("--e-layout", default="row", help="'row' = [B][T][L*2C], 'layer' = [L][B][T][2C]")
    a = ap.parse_args()
    a.split = a.split or a.f16
    d = torch.device("cuda:0")
    B, T, C, Lyr = a.B, a.T, 256, 20
    lens = torch.full((B,), T, device=d, dtype=torch.int32)
    sp = 2 if a.split else 1
    to_h = L.split_f16 if a.f16 else L.split_bf16 if a.split else L.to_bf16
    to_w = (lambda w_: L.split_f16(w_, scale=256.0)) if a.f16 else to_h
    hdt = torch.float16 if a.f16 else torch.bfloat16
    skw = dict(split=2, out_scale=1.0 / 256.0) if a.f16 else dict(split=int(a.split))
    nprod = 2.0 if a.f16 else 3.0 if a.split else 1.0
    skw_gate = skw
    Xh = to_h(torch.randn(B, T, C, device=d))
    E = torch.randn(B, T, 4 * 2 * C, device=d)   # 4 layer slabs are enough to defeat the L2
    El = torch.randn(4, B, T, 2 * C, device=d)
    Gh = torch.empty(B, T, C * sp, device=d, dtype=hdt)
    w = torch.randn(2 * C, C, 3, device=d) / math.sqrt(3 * C)
    Wh = to_w(L.pack_conv_weight(w, interleave_half=C))
    if a.q4:
        Wh = L.pack_gate_q4(L.pack_conv_weight(w, interleave_half=C))[0]
        skw_gate = dict(split=3, out_scale=1.0 / 256.0, q_scale=4.0)
    wo = torch.randn(C, C, 1, device=d) / math.sqrt(C)
    Woh = to_w(L.pack_conv_weight(wo))
    X = torch.randn(B, T, C, device=d)
    Yh = torch.empty(B, T, C * sp, device=d, dtype=hdt)
    nb = torch.randn(C, device=d)
    layer = [0]
    res = []
    if a.which in ("gate", "all"):
        def fg():
            layer[0] = (layer[0] + 1) % 4
            if a.e_layout == "layer":
                L.gemm_bf16(Xh, Wh, B=B, T=T, K=C, taps=(-2, 0, 2), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens,
                            E=El[layer[0]], lde=2 * C, e_bs=T * 2 * C, out=Gh, **skw)
                return
            if a.no_e:
                L.gemm_bf16(Xh, Wh, B=B, T=T, K=C, taps=(-2, 0, 2), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens, out=Gh, **skw)
                return
            L.gemm_bf16(Xh, Wh, B=B, T=T, K=C, taps=(-2, 0, 2), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens,
                        E=E[:, :, layer[0] * 2 * C:], lde=4 * 2 * C, e_bs=T * 4 * 2 * C, out=Gh, gate256=128 if (a.gate128 or a.q4) else False, **skw_gate)
        s = timeit(fg, a.iters)
        res.append(("gate K=768 N=512" + (" fp16q4 (fp16 + block-scaled fp4 product) gate128q" if a.q4 else " fp16x2 (2 products)" + (" gate128" if a.gate128 else "") if a.f16 else " split x3" if a.split else " bf16"), s, nprod * 2.0 * B * T * 3 * C * 2 * C,
                    B * T * (sp * 2.0 * C + 4.0 * 2 * C + sp * 2.0 * C)))
    if a.which in ("res", "all"):
        po = a.split and a.pair_only
        def fr():
            L.gemm_bf16(Gh, Woh, B=B, T=T, K=C, taps=(0,), N=C, Np=Woh.shape[0], epi=L.HEPI_RESX, lens=lens, X=None if po else X, post_scale=0.7071,
                        next_bias=nb, Y=Yh, cur_bias=nb if po else None, **skw)
        s = timeit(fr, a.iters)
        res.append(("residual projection K=256 N=256" + (" fp16x2" if a.f16 else " split x3" if a.split else " bf16") + (" pair-only stream" if po else ""), s,
                    nprod * 2.0 * B * T * C * C, B * T * (sp * 2.0 * C + (sp * 2.0 * C if po else 4.0 * C + 4.0 * C) + sp * 2.0 * C)))
    if a.which == "skip":   # the K = L*C skip GEMM (STORE + ReLU) on the layer-slot operand of all 20 layers
        GA = to_h(torch.randn(B, T, Lyr * C, device=d))
        wsk = torch.randn(C, Lyr * C, 1, device=d) / math.sqrt(Lyr * C)
        Wsk = to_w(L.pack_conv_weight(wsk))
        S = torch.empty(B, T, C, device=d)
        bsk = torch.randn(C, device=d)
        if a.q4:   # fp16q4: the second product on the block-scaled fp4 instruction (ss_gemm_bf16_tile256q)
            Wq = L.pack_skip_q4(L.pack_conv_weight(wsk))[0]

            def fs():
                L.gemm_bf16(GA, Wq, B=B, T=T, K=Lyr * C, taps=(0,), N=C, Np=L.round_up(C, 32), epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S,
                            bias=bsk, split=3, out_scale=1.0 / 256.0, q_scale=0.25, gate256=True)
        elif a.compact:
            GAc = L.split_planes(GA)[0].to(torch.float16).contiguous()
            Wc = L.split_planes(Wsk)[0].to(torch.float16).contiguous() if a.compact == 2 else Wsk
            nprod = 1.0 if a.compact == 2 else nprod

            def fs():
                L.gemm_bf16(GAc, Wc, B=B, T=T, K=Lyr * C, taps=(0,), N=C, Np=Wc.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S, bias=bsk, a_compact=True,
                            one_product=2 if a.compact == 2 else 0, **skw)
        else:
            def fs():
                L.gemm_bf16(GA, Wsk, B=B, T=T, K=Lyr * C, taps=(0,), N=C, Np=Wsk.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S, bias=bsk, **skw)
        s = timeit(fs, a.iters)
        res.append(("skip GEMM K=5120 N=256" + (" fp16q4" if a.q4 else " fp16x2" if a.f16 else " split x3" if a.split else " bf16"), s, nprod * 2.0 * B * T * Lyr * C * C,
                    B * T * ((1 if a.f16 else sp) * 2.0 * Lyr * C + 4.0 * C)))
    for name, s, fl, by in res:
        print(f"{name:40s} {s * 1e6:9.1f} us  {fl / s / 1e12:7.1f} TF/s ({fl / s / 2.5e15 * 100:4.1f}% of bf16 peak)  {by / s / 1e12:5.2f} TB/s algorithmic HBM")


if __name__ == "__main__":
    main()
