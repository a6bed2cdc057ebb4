"""Numerics study (test infrastructure, CPU): fp32 GEMM operands as sums of bf16 terms on the bf16 matrix cores.

The fp32 matrix pipe of gfx950 is 16x slower than the bf16 one. Writing each operand as a = a_hi + a_mid + a_lo (three bf16 terms, 24 of the
24 significand bits) and keeping the partial products whose weight is >= 2^-16 of the leading one -
    a.b ~= hi.hi + (hi.mid + mid.hi) + (hi.lo + lo.hi + mid.mid)          (6 of the 9 products, fp32 accumulation)
- would run the denoisers' hidden GEMMs at 6/16 of their fp32 matrix time (lead 2). This script measures what that arithmetic
does to the path BEFORE any kernel exists: it swaps it into the oracle's denoiser GEMMs (the same call sites the bf16 mode rounds,
oracle/restatement.py `rounded=True`) and compares the mel with the REAL reference's golden fixtures, next to plain fp32 and to the cheaper
3-product form (hi.hi + hi.mid + mid.hi).

    python -m oracle.bf16x3_numerics [--long]        (--long adds the 1000-step chain: a few minutes)
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, restatement as R  # noqa: E402
from stylesinger_amd import synth  # noqa: E402


def split3(x):
    hi = x.bfloat16().float()
    r1 = x - hi
    mid = r1.bfloat16().float()
    lo = (r1 - mid).bfloat16().float()
    return hi, mid, lo


def make_conv(products):
    def conv1d_cl(x, w, b, dilation=1, rounded=False):
        k = w.shape[-1]
        pad = (k - 1) // 2 * dilation
        xt = x.transpose(1, 2)
        if not rounded or products == 0:
            return F.conv1d(xt, w, b, padding=pad, dilation=dilation).transpose(1, 2)
        xs, ws = split3(xt), split3(w)
        pairs = [(0, 0), (0, 1), (1, 0)] + ([(0, 2), (2, 0), (1, 1)] if products == 6 else [])
        # smallest terms first, as a kernel would order the MFMAs of one accumulator
        y = None
        for (i, j) in reversed(pairs):
            t = F.conv1d(xs[i], ws[j], None, padding=pad, dilation=dilation)
            y = t if y is None else y + t
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    return conv1d_cl


def run_case(name, products):
    case = harness.load_case(name)
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    orig = R.conv1d_cl
    R.conv1d_cl = make_conv(products)
    try:
        with torch.no_grad():
            ret = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(meta["tape_seed"]), mel2ph=batch.get("mel2ph"))
    finally:
        R.conv1d_cl = orig
    d = (ret["mel_out"] - gold["mel_out"]).abs()
    uv = ((ret["pitch_pred"][..., 1] > 0) != (gold["pitch_pred"][..., 1] > 0)).float().mean().item()
    return d.mean().item(), d.max().item(), uv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--long", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    x = torch.randn(4096, 256)
    w = torch.randn(256, 512) / 16
    ref = x.double() @ w.double()
    xs, ws = split3(x), split3(w)
    e32 = (x @ w - ref).abs().max().item()
    p3 = xs[0] @ ws[0] + (xs[0] @ ws[1] + xs[1] @ ws[0])
    p6 = p3 + (xs[0] @ ws[2] + xs[2] @ ws[0] + xs[1] @ ws[1])
    print(f"GEMM 4096x256x512 vs float64: fp32 {e32:.2e}, 3 products {(p3 - ref).abs().max().item():.2e}, 6 products {(p6 - ref).abs().max().item():.2e}")
    cases = ["acoustic_t64_s100"] + (["acoustic_t32_mel1000"] if a.long else [])
    for name in cases:
        for products, label in ((0, "fp32"), (6, "3 x bf16 terms, 6 products"), (3, "3 products")):
            l1, mx, uv = run_case(name, products)
            print(f"{name}: {label:28s} mel L1 vs the reference {l1:.3e}  max {mx:.3e}  voicing decisions differing {uv:.4f}")


if __name__ == "__main__":
    main()
