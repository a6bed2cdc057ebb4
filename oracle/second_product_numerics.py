"""Numerics study (test infrastructure, CPU) for the NEXT step after "fp16x2": how few bits the second matrix product (activation x weight-lo) needs.

fp16x2 runs every hidden GEMM of the mel denoiser as a * hi + a * lo with fp16 terms - two products at the fp16 MFMA rate, and the
many-round kernels sit at the chip's power-limited matrix rate, so only fewer matrix flops make BASELINE configs[3] faster. lo = w - fp16(w) is a
correction of relative size 2^-12: it needs a few significant bits, not eleven. gfx950's MX-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) runs
fp8 operands at 2x and fp4 operands at 4x the fp16 rate, so the second product would cost 0.5 / 0.25 of a product instead of 1.
Every variant swaps its arithmetic into the oracle's denoiser GEMMs and compares the mel with the REAL reference's goldens (bar: mel L1 <= 1e-4):

    python -m oracle.second_product_numerics            (~15 min on 8 cores; the MXFP4 emulation is slow)

Measured (round 4, this container), mel L1 vs the reference, 1000-step golden `acoustic_t32_mel1000` / 100-step golden `acoustic_t64_s100`:
    plain fp16, one product ......................................... 1.94e-4 / 2.00e-4    (fails)
    + mean-field compensation (lo applied to the per-channel MEAN of a) 1.16e-4 / 1.23e-4    (the coherent part is NOT just the mean: fails)
    second product fp16 x fp16 (= fp16x2 as built) .................. 1.89e-5 / 3.28e-5
    second product fp16 a x fp8 (e4m3) lo ........................... 1.96e-5 / 3.31e-5
    second product fp8 a x fp8 lo ................................... 1.97e-5 / 3.30e-5    -> 1.5 products: free numerically
    second product fp8 a x MXFP4 lo (e2m1, block scale per 32) ...... 3.18e-5 / 4.15e-5
    second product MXFP4 a x MXFP4 lo ............................... 3.48e-5 / 4.37e-5    -> 1.25 products, 2.3x margin
    ... with a FIXED activation scale per site (gate outputs 2^-2, stream 2^1: no block max in the producing epilogue) 3.37e-5 / 4.16e-5
(the conditioner projection exact, the f0 denoisers untouched, as in fp16x2)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, restatement as R  # noqa: E402
from stylesinger_amd import synth  # noqa: E402

F8 = torch.float8_e4m3fn
GRID = torch.tensor([0, 0.5, 1, 1.5, 2, 3, 4, 6.0])   # e2m1 magnitudes


def q8(x, scale=1.0):
    return (x * scale).clamp(-448, 448).to(F8).float() / scale


def q4_blocks(x, dim):
    """MXFP4: blocks of 32 along `dim` (the K dimension) share a power-of-two scale (E8M0), elements are e2m1 (nearest grid point)"""
    xs = x.movedim(dim, -1)
    shp = xs.shape
    K = shp[-1]
    pad = (-K) % 32
    if pad:
        xs = F.pad(xs, (0, pad))
    b = xs.reshape(*xs.shape[:-1], -1, 32)
    amax = b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    s = torch.exp2(torch.floor(torch.log2(amax)) - 2)   # the largest element lands in [4, 8): the grid's top is 6
    v = (b / s).clamp(-6, 6)
    idx = (v.abs().unsqueeze(-1) - GRID).abs().argmin(dim=-1)
    q = (GRID[idx] * v.sign() * s).reshape(*xs.shape)[..., :K].reshape(shp)
    return q.movedim(-1, dim)


def q4_fixed(x, scale):
    idx = (x.abs().div(scale).clamp(max=6.0).unsqueeze(-1) - GRID).abs().argmin(dim=-1)
    return GRID[idx] * x.sign() * scale


def make_conv(mode, names):
    def conv1d_cl(x, w, b, dilation=1, rounded=False):
        k = w.shape[-1]
        pad = (k - 1) // 2 * dilation
        xt = x.transpose(1, 2)
        key = names.get(id(w), "")
        hidden = any(t in key for t in ("dilated", "residual_layers", "skip_projection")) and "conditioner" not in key
        if not rounded or not hidden or not key.startswith("postdiff"):
            return F.conv1d(xt, w, b, padding=pad, dilation=dilation).transpose(1, 2)
        xh = xt.half().float()
        wh = w.half().float()
        wl = w - wh
        y = F.conv1d(xh, wh, None, padding=pad, dilation=dilation)
        second = {"plain": None,
                  "meanfield": lambda: (xh.mean(dim=(0, 2), keepdim=True).expand_as(xh), wl),
                  "f16 x f16": lambda: (xh, wl.half().float()),
                  "f16 x f8": lambda: (xh, q8(wl, 2.0 ** 16)),
                  "f8 x f8": lambda: (q8(xt, 16.0), q8(wl, 2.0 ** 16)),
                  "f8 x mxfp4": lambda: (q8(xt, 16.0), q4_blocks(wl, 1)),
                  "mxfp4 x mxfp4": lambda: (q4_blocks(xt, 1), q4_blocks(wl, 1)),
                  # activations on a FIXED power-of-two scale per site (z = gate outputs in (-1, 1): 2^-2; the stream x + dstep: 2^1): a plain
                  # elementwise conversion in the producing epilogue, the instruction's scale byte a constant
                  "fixed4 x mxfp4": lambda: (q4_fixed(xt, 2.0 if "dilated" in key else 0.25), q4_blocks(wl, 1))}[mode]
        if second is not None:
            a2, w2 = second()
            y = y + F.conv1d(a2, w2, None, padding=pad, dilation=dilation)
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    return conv1d_cl


def run(name, mode):
    case = harness.load_case(name)
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    names = {id(v): k for k, v in sd.items()}
    orig = R.conv1d_cl
    R.conv1d_cl = make_conv(mode, names)
    try:
        with torch.no_grad():
            ret = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(meta["tape_seed"]), mel2ph=batch.get("mel2ph"))
    finally:
        R.conv1d_cl = orig
    d = (ret["mel_out"] - gold["mel_out"]).abs()
    return d.mean().item(), d.max().item()


if __name__ == "__main__":
    for case in ("acoustic_t32_mel1000", "acoustic_t64_s100"):
        for mode in ("plain", "meanfield", "f16 x f16", "f16 x f8", "f8 x f8", "f8 x mxfp4", "mxfp4 x mxfp4", "fixed4 x mxfp4"):
            t0 = time.time()
            l1, mx = run(case, mode)
            print(f"{case:22s} second product {mode:14s} mel L1 {l1:.3e}  max {mx:.3e}  ({time.time() - t0:.0f} s)", flush=True)
