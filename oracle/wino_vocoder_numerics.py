"""Numerics study (test infrastructure, CPU): HiFi-GAN ResBlock convs as grouped Winograd F(4,3) in fp32.

A k-tap dilated conv (k = 3 / 7 / 11) is split into ceil(k/3) groups of three taps; every group is an F(4,3) product (6 multiplies per 4
outputs instead of 12) whose six transformed products accumulate over the groups before ONE output transform - the arithmetic
stylesinger_amd/csrc/wino43_conv.hip performs on the matrix cores. This script swaps that arithmetic (emulated with torch fp32 on the
CPU, the device's operation order for the transforms) into oracle.restatement.hifigan_forward and measures the waveform against the
golden fixtures of the REAL reference (tests/golden/vocoder_*.pt), next to the direct fp32 restatement.

    python -m oracle.wino_vocoder_numerics
"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import harness, restatement as R  # noqa: E402
from stylesinger_amd import synth  # noqa: E402


def wino43_weights(w3):
    """[Co, Ci, 3] -> [6, Co, Ci] (the G matrix of F(4,3), fp32, the device kernel's expressions)."""
    w0, w1, w2 = w3[..., 0], w3[..., 1], w3[..., 2]
    return torch.stack([w0 * 0.25, -(w0 + w1 + w2) / 6.0, -(w0 - w1 + w2) / 6.0, w0 / 24.0 + w1 / 12.0 + w2 / 6.0,
                        w0 / 24.0 - w1 / 12.0 + w2 / 6.0, w2])


def conv1d_wino43(x, w, bias, dilation):
    """x [B, Ci, T], w [Co, Ci, k] (k odd), 'same' padding: grouped F(4,3) in fp32."""
    B, Ci, T = x.shape
    Co, _, k = w.shape
    d = dilation
    G = (k + 2) // 3
    wp = torch.zeros(Co, Ci, 3 * G, dtype=w.dtype)
    wp[..., :k] = w
    Tq = -(-T // (4 * d)) * 4 * d
    lo = (k - 1) // 2 * d                       # frames read before t
    hi = (3 * G - 1) * d - lo + 3 * d + (Tq - T)  # and after
    xp = F.pad(x, (lo, hi))
    # quad base frames
    t = torch.arange(Tq).view(-1, 4, d)[:, 0, :].reshape(-1)       # [Q]
    acc = [torch.zeros(B, Co, t.numel(), dtype=torch.float32) for _ in range(6)]
    for g in range(G):
        gw = wino43_weights(wp[..., 3 * g:3 * g + 3].float())      # [6, Co, Ci]
        r = [xp[:, :, t + lo + (3 * g - (k - 1) // 2) * d + i * d] for i in range(6)]   # rows y[t + s_g + i d]
        A = torch.addcmul(r[4], r[2], torch.tensor(-4.0))          # r4 - 4 r2 (fma)
        Bt = torch.addcmul(r[3], r[1], torch.tensor(-4.0))
        Cc = r[4] - r[2]
        D = r[3] - r[1]
        c = [None] * 6
        c[1] = A + Bt
        c[2] = A - Bt
        c[3] = torch.addcmul(Cc, D, torch.tensor(2.0))
        c[4] = torch.addcmul(Cc, D, torch.tensor(-2.0))
        c[0] = torch.addcmul(torch.addcmul(r[4], r[0], torch.tensor(4.0)), r[2], torch.tensor(-5.0))
        c[5] = torch.addcmul(torch.addcmul(r[5], r[1], torch.tensor(4.0)), r[3], torch.tensor(-5.0))
        for j in range(6):
            acc[j] = acc[j] + torch.einsum("oc,bcq->boq", gw[j], c[j])
    s12, d12 = acc[1] + acc[2], acc[1] - acc[2]
    s34, d34 = acc[3] + acc[4], acc[3] - acc[4]
    z0 = acc[0] + s12 + s34
    z1 = torch.addcmul(d12, d34, torch.tensor(2.0))
    z2 = torch.addcmul(s12, s34, torch.tensor(4.0))
    z3 = torch.addcmul(d12, d34, torch.tensor(8.0)) + acc[5]
    out = torch.empty(B, Co, Tq, dtype=torch.float32)
    for o, z in enumerate((z0, z1, z2, z3)):
        out[:, :, t + o * d] = z
    out = out[:, :, :T]
    return out + bias.view(1, -1, 1) if bias is not None else out


class WinoConvs:
    """Context manager: F.conv1d calls with a 3/7/11-tap square weight and 'same' padding go through conv1d_wino43."""

    def __enter__(self):
        self.orig = F.conv1d
        orig = self.orig

        def conv1d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
            k = w.shape[-1]
            if w.shape[0] == w.shape[1] and k in (3, 7, 11) and stride == 1 and groups == 1 and padding == (k - 1) // 2 * dilation:
                return conv1d_wino43(x, w, b, dilation)
            return orig(x, w, b, stride, padding, dilation, groups)
        F.conv1d = conv1d
        return self

    def __exit__(self, *exc):
        F.conv1d = self.orig


def main():
    torch.manual_seed(0)
    # unit check of the emulation itself against the direct conv in float64
    for (C, k, d, T) in ((32, 3, 1, 50), (16, 7, 3, 77), (8, 11, 5, 131), (8, 11, 1, 40)):
        x = torch.randn(2, C, T)
        w = torch.randn(C, C, k) / math.sqrt(C * k)
        b = torch.randn(C)
        ref = F.conv1d(x.double(), w.double(), b.double(), padding=(k - 1) // 2 * d, dilation=d)
        e_w = (conv1d_wino43(x, w, b, d).double() - ref).abs().max().item()
        e_d = (F.conv1d(x, w, b, padding=(k - 1) // 2 * d, dilation=d).double() - ref).abs().max().item()
        print(f"conv C={C} k={k} d={d}: grouped F(4,3) fp32 max err {e_w:.2e}, direct fp32 {e_d:.2e} (vs float64)")
    for name in ("vocoder_t12", "vocoder_b2_t9"):
        case = harness.load_case(name)
        meta = case["meta"]
        cfg, vsd = harness.vocoder_case_setup(meta)
        with torch.no_grad():
            wav_d, _ = R.hifigan_forward(vsd, cfg, case["inp"]["mel"], case["inp"]["f0"], synth.NoiseTape(meta["tape_seed"]))
            with WinoConvs():
                wav_w, _ = R.hifigan_forward(vsd, cfg, case["inp"]["mel"], case["inp"]["f0"], synth.NoiseTape(meta["tape_seed"]))
        gold = case["out"]["wav"]
        print(f"{name}: wav max err vs the reference: direct fp32 {(wav_d - gold).abs().max().item():.2e}, "
              f"grouped F(4,3) {(wav_w - gold).abs().max().item():.2e}  (mean {(wav_w - gold).abs().mean().item():.2e}; |wav| max {gold.abs().max().item():.2f})")


if __name__ == "__main__":
    main()
