"""Numerics study (test infrastructure, CPU): ONE fp16 product per hidden GEMM of the mel denoiser with the weight rounding NOISE-SHAPED across the
diffusion steps, on BASELINE configs[3]'s 1000-step chain.

Background (oracle/bf16x2_numerics.py): with plain fp16 operands - one product - the chain ends 1.94e-4 from the real reference (bar 1e-4), and
the study there shows WHY: the weight rounding is the coherent part (the same perturbation in all 1000 steps), the activation rounding averages out.
"fp16x2" removes the weight rounding with a second product (a * lo): 1.9e-5, at twice the matrix work. Question here: can the weight rounding be made
to average out too? Evaluation j of the loop uses the weight set W_(j mod N), where the N sets are a first-order sigma-delta sequence of fp16
roundings of the same fp32 weight: r_0 = 0, W_k = RNE16(w + r_k), r_(k+1) = r_k + (w - W_k) - so that sum_k W_k = N w - r_N with |r_N| <= ulp / 2:
the MEAN weight over any N consecutive evaluations is exact to 1 / N of an fp16 rounding, while every single evaluation is a plain one-product
fp16 GEMM (half the matrix work, half the weight bytes of fp16x2).

    python -m oracle.dither_numerics          (~40 s per variant on 8 cores; golden acoustic_t32_mel1000 of the REAL reference)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, restatement as R  # noqa: E402
from stylesinger_amd import synth  # noqa: E402

SHIFT = 8.0   # as fp16x2: weights are rounded as w * 2^8 (keeps the small ones out of the fp16 subnormals)


def weight_sets(w, n, sites_scale=2.0 ** SHIFT):
    ws = w * sites_scale
    sets, r = [], torch.zeros_like(ws)
    for _ in range(n):
        wk = (ws + r).half().float()
        r = r + (ws - wk)
        sets.append(wk)
    return sets


def make_conv(n_sets, sites, names, order="cyclic", e_sets=0):
    """e_sets > 0: the step-invariant conditioner addend of every layer is ALSO stored in fp16, as e_sets sigma-delta sets cycled over the evaluations
    (what a 2-byte addend slab of ss_layer512 would hold; 0 = exact fp32, the product's form)"""
    calls, cache = {}, {}

    def conv1d_cl(x, w, b, dilation=1, rounded=False):
        k = w.shape[-1]
        pad = (k - 1) // 2 * dilation
        xt = x.transpose(1, 2)
        key = names.get(id(w), "")
        site = ("cond" if "conditioner" in key else "dil" if "dilated" in key else "out" if "residual_layers" in key else
                "skip" if "skip_projection" in key else None)
        if site == "cond" and e_sets > 0 and key.startswith("postdiff"):
            j = calls.get(id(w), 0)
            calls[id(w)] = j + 1
            if id(w) not in cache:
                cache[id(w)] = weight_sets(F.conv1d(xt, w, b, padding=pad, dilation=dilation), e_sets, sites_scale=1.0)
            return cache[id(w)][j % e_sets].transpose(1, 2)
        if not rounded or site not in sites or not key.startswith("postdiff"):
            return F.conv1d(xt, w, b, padding=pad, dilation=dilation).transpose(1, 2)
        j = calls.get(id(w), 0)
        calls[id(w)] = j + 1
        if id(w) not in cache:
            cache[id(w)] = weight_sets(w, n_sets)
        wk = cache[id(w)][j % n_sets]
        y = F.conv1d(xt.half().float(), wk, None, padding=pad, dilation=dilation) * (2.0 ** -SHIFT)
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    return conv1d_cl


def run(name, n_sets, sites=("dil", "out", "skip"), e_sets=0):
    case = harness.load_case(name)
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    names = {id(v): k for k, v in sd.items()}
    orig = R.conv1d_cl
    R.conv1d_cl = make_conv(n_sets, sites, names, e_sets=e_sets)
    try:
        with torch.no_grad():
            ret = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(meta["tape_seed"]), mel2ph=batch.get("mel2ph"))
    finally:
        R.conv1d_cl = orig
    d = (ret["mel_out"] - gold["mel_out"]).abs()
    uv = int(((ret["pitch_pred"][..., 1] > 0) != (gold["pitch_pred"][..., 1] > 0)).sum())
    return d.mean().item(), d.max().item(), uv


if __name__ == "__main__":
    golden = "acoustic_t32_mel1000"
    for a in sys.argv[1:]:
        if a.startswith("--golden="):
            golden = a.split("=", 1)[1]
    es = [int(a.split("=", 1)[1]) for a in sys.argv[1:] if a.startswith("--e-sets=")] or [0]
    ns = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8, 16]
    for n in ns:
        for e in es:
            t0 = time.time()
            l1, mx, uv = run(golden, n, e_sets=e)
            print(f"one fp16 product, {n:2d} noise-shaped weight set(s) cycled over the evaluations" + (f", conditioner addend in fp16 as {e} set(s)" if e else "") +
                  f": mel L1 {l1:.3e}  max {mx:.3e}  voicing flips {uv}  ({time.time() - t0:.0f} s)", flush=True)
