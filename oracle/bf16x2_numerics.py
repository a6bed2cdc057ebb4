"""Numerics study (test infrastructure, CPU) behind the "bf16x2" precision mode: which operands of which hidden GEMMs of the mel denoiser may
be plain bf16 on BASELINE configs[3]'s 1000-step chain, and which must be split into (hi, mid) bf16 pairs.

Every variant swaps its arithmetic into the oracle's denoiser GEMMs (the `rounded=` call sites of oracle/restatement.py: conditioner projection
`cond`, dilated conv `dil`, output projection `out`, skip projection `skip`) and compares the mel with the REAL reference's golden
`acoustic_t32_mel1000` (north_star: mel L1 <= 1e-4). Products per site: 1 = hi*hi (plain bf16), "A" = hi*hi + mid*hi (activations split),
"W" = hi*hi + hi*mid (weights split), 3 = hi*hi + hi*mid + mid*hi, None = exact fp32.

    python -m oracle.bf16x2_numerics            (~40 s per variant on 8 cores)

Measured (round 4, this container), mel L1 vs the reference:
    all sites plain bf16 ........................... 2.50e-3      all sites A-split 2.10e-3, all sites W-split 1.28e-3
    only cond plain (rest fp32) .................... 1.88e-3      (a FIXED rounding error of E enters all 1000 steps -> hoisted projection in fp32)
    only dil / out / skip plain (rest fp32) ........ 5.4e-4 / 1.14e-3 / 1.22e-3
    dil W-split / A-split (rest fp32) .............. 5.6e-5 / 5.4e-4   (the weight rounding is the coherent part, the activation rounding averages)
    out W-split / A-split (rest fp32) .............. 1.03e-4 / 1.14e-3
    cond fp32; dil, out, skip W-split (2 products) . 1.55e-4      -> two products are NOT enough
    cond fp32; dil W-split; out, skip 3 products ... 5.7e-5       (inside the bar, margin 1.75x: not adopted)
    cond fp32; dil, out, skip 3 products ........... 2.4e-6       = the mode as built (oracle.set_matmul_rounding("bf16x2"); on the GPU 2.2e-6)

    python -m oracle.bf16x2_numerics --fp16     the same sites with FP16 terms (11 significand bits: the activation rounding is 8x smaller)
    cond fp32; dil, out, skip plain fp16 (1 product) 1.94e-4      all sites plain incl. cond 2.9e-4
    cond fp32; dil, out, skip W-split (2 products) . 1.89e-5      = the "fp16x2" mode (oracle.set_matmul_rounding("fp16x2"); on the GPU 1.5e-5)
    cond W-split too ............................... 1.60e-4      (the hoisted projection must stay exact here as well)
    cond fp32; dil plain, out / skip W-split ....... 6.8e-5       (1.5 products on average: inside the bar, margin 1.5x - not adopted)

    python -m oracle.bf16x2_numerics --eslab    round 5: fp16x2 with the exact fp32 conditioner slab E STORED in fewer bits (see the results in DESIGN.md 3.1j)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, restatement as R  # noqa: E402
from stylesinger_amd import synth  # noqa: E402

P3, P1, PA, PW = [(0, 0), (0, 1), (1, 0)], [(0, 0)], [(0, 0), (1, 0)], [(0, 0), (0, 1)]


TERM = torch.bfloat16   # --fp16: torch.float16


def split2(x):
    hi = x.to(TERM).float()
    return hi, (x - hi).to(TERM).float()


def make_conv(site_pairs, names):
    def conv1d_cl(x, w, b, dilation=1, rounded=False):
        k = w.shape[-1]
        pad = (k - 1) // 2 * dilation
        xt = x.transpose(1, 2)
        key = names.get(id(w), "")
        site = ("cond" if "conditioner" in key else "dil" if "dilated" in key else "out" if "residual_layers" in key else
                "skip" if "skip_projection" in key else None)
        if rounded and key.startswith("postdiff") and isinstance(site_pairs.get(site), str):
            # round-5 question (C4: half of the gate launch's bytes are the fp32 conditioner slab E): the EXACT fp32 projection, STORED in fewer bits
            y = F.conv1d(xt, w, b, padding=pad, dilation=dilation)
            hi = y.to(TERM).float()
            if site_pairs[site] == "store16":            # one 16-bit term per element (2 of 4 bytes)
                y = hi
            elif site_pairs[site] == "store16+8":        # 16-bit term + an e4m3 correction of (y - hi) * 2^11 (3 of 4 bytes)
                y = hi + ((y - hi) * 2048.0).to(torch.float8_e4m3fn).float() / 2048.0
            return y.transpose(1, 2)
        if not rounded or site_pairs.get(site) is None or not key.startswith("postdiff"):
            return F.conv1d(xt, w, b, padding=pad, dilation=dilation).transpose(1, 2)
        xs, ws = split2(xt), split2(w)
        y = None
        for (i, j) in reversed(site_pairs[site]):
            t = F.conv1d(xs[i], ws[j], None, padding=pad, dilation=dilation)
            y = t if y is None else y + t
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    return conv1d_cl


def run(name, site_pairs):
    case = harness.load_case(name)
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    names = {id(v): k for k, v in sd.items()}
    orig = R.conv1d_cl
    R.conv1d_cl = make_conv(site_pairs, names)
    try:
        with torch.no_grad():
            ret = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(meta["tape_seed"]), mel2ph=batch.get("mel2ph"))
    finally:
        R.conv1d_cl = orig
    d = (ret["mel_out"] - gold["mel_out"]).abs()
    return d.mean().item(), d.max().item()


VARIANTS = {
    "all sites plain bf16": dict(cond=P1, dil=P1, out=P1, skip=P1),
    "only cond plain": dict(cond=P1, dil=None, out=None, skip=None),
    "only dil plain": dict(cond=None, dil=P1, out=None, skip=None),
    "only out plain": dict(cond=None, dil=None, out=P1, skip=None),
    "only skip plain": dict(cond=None, dil=None, out=None, skip=P1),
    "dil W-split, rest fp32": dict(cond=None, dil=PW, out=None, skip=None),
    "dil A-split, rest fp32": dict(cond=None, dil=PA, out=None, skip=None),
    "out W-split, rest fp32": dict(cond=None, dil=None, out=PW, skip=None),
    "out A-split, rest fp32": dict(cond=None, dil=None, out=PA, skip=None),
    "cond fp32; dil, out, skip W-split": dict(cond=None, dil=PW, out=PW, skip=PW),
    "cond fp32; dil W-split; out, skip 3 products": dict(cond=None, dil=PW, out=P3, skip=P3),
    "cond fp32; dil, out, skip 3 products (= bf16x2)": dict(cond=None, dil=P3, out=P3, skip=P3),
}


VARIANTS_FP16 = {
    "fp16: cond fp32; dil, out, skip plain (1 product)": dict(cond=None, dil=P1, out=P1, skip=P1),
    "fp16: all sites plain incl. cond": dict(cond=P1, dil=P1, out=P1, skip=P1),
    "fp16: cond fp32; dil, out, skip W-split (= fp16x2)": dict(cond=None, dil=PW, out=PW, skip=PW),
    "fp16: all sites W-split incl. cond": dict(cond=PW, dil=PW, out=PW, skip=PW),
    "fp16: cond fp32; dil plain; out, skip W-split": dict(cond=None, dil=P1, out=PW, skip=PW),
}


VARIANTS_ESLAB = {   # --eslab (fp16 terms): fp16x2 with the hoisted conditioner projection computed in fp32 but STORED narrower
    "fp16x2, E stored as one fp16 term (2 B)": dict(cond="store16", dil=PW, out=PW, skip=PW),
    "fp16x2, E stored as fp16 + e4m3 correction (3 B)": dict(cond="store16+8", dil=PW, out=PW, skip=PW),
}


if __name__ == "__main__":
    if "--eslab" in sys.argv[1:]:
        TERM = torch.float16
        VARIANTS = VARIANTS_ESLAB
    if "--fp16" in sys.argv[1:]:
        TERM = torch.float16
        VARIANTS = VARIANTS_FP16
    for label, sp in VARIANTS.items():
        t0 = time.time()
        l1, mx = run("acoustic_t32_mel1000", sp)
        print(f"{label:52s} mel L1 {l1:.3e}  max {mx:.3e}  ({time.time() - t0:.0f} s)", flush=True)
