"""Numerics of Winograd F(2,3) / F(4,3) for the dilated conv of the mel denoiser (modules/diff/net.py:66-73), CPU only: the
transforms are emulated in torch fp32 inside the oracle's 100-step mel diffusion (oracle/restatement.py::mel_diffusion) and compared
with the direct conv in fp32 and with the conv evaluated in fp64. Run before wino43_gate.hip was written; output recorded in
profiles/r02_wino43_numerics.md.   python tools/wino43_numerics.py"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import restatement as R
from stylesinger_amd import config, synth

torch.manual_seed(0)
torch.set_num_threads(16)
hp = config.make_hparams({})
sd = synth.synth_acoustic_state_dict(hp, 1234)
B, T = 1, 256
cond = torch.randn(B, T, 256) * 0.5
coarse = torch.randn(B, T, 80) * 0.8 - 3.0

BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def wino_conv(x, w, b, d, BT, G, AT):
    """x [B,T,C] fp32, w [N,C,3]; output tile m = AT.shape[0]; everything in fp32 like the kernel would."""
    m, a = AT.shape
    Bn, T, C = x.shape
    N = w.shape[0]
    wt = torch.einsum('ak,nck->anc', G.float(), w)          # [a][N][C] fp32 (transform in fp32)
    grp = m * d
    Tp = (T + grp - 1) // grp * grp
    xp = F.pad(x, (0, 0, d, Tp - T + (a - 2) * d))           # frame t -> index t + d; rows up to t + (a-2)d
    out = torch.zeros(Bn, Tp, N)
    # quads: base frames t0 = g*grp + r, r in [0,d)
    t0 = (torch.arange(Tp // grp)[:, None] * grp + torch.arange(d)[None, :]).reshape(-1)   # [Q]
    rows = torch.stack([xp[:, t0 + i * d] for i in range(a)], 0)      # rows[i] = x[t0 + (i-1) d]  [a][B][Q][C]
    comp = torch.einsum('ji,ibqc->jbqc', BT.float(), rows)           # input transform in fp32
    mm = torch.einsum('jbqc,jnc->jbqn', comp, wt)                    # a GEMMs (fp32 accumulate)
    y = torch.einsum('oj,jbqn->obqn', AT.float(), mm)                # [m][B][Q][N]
    for o in range(m):
        out[:, t0 + o * d] = y[o]
    return out[:, :T] + b


mode = {"m": "direct"}
orig = R.conv1d_cl


def patched(x, w, b, dilation=1, rounded=False):
    if w.shape[-1] == 3 and mode["m"] != "direct" and w.shape[0] == 512:
        if mode["m"] == "f23":
            return wino_conv(x, w, b, dilation, BT2, G2, AT2)
        if mode["m"] == "f43":
            return wino_conv(x, w, b, dilation, BT4, G4, AT4)
        if mode["m"] == "f64":
            return orig(x.double(), w.double(), b.double(), dilation=dilation).float()
    return orig(x, w, b, dilation=dilation, rounded=rounded)


R.conv1d_cl = patched
outs = {}
for m in ("direct", "f64", "f23", "f43"):
    mode["m"] = m
    with torch.no_grad():
        outs[m] = R.mel_diffusion(sd, hp, coarse, cond, synth.NoiseTape(7))
    print(m, "done", flush=True)
for m in ("f64", "f23", "f43"):
    print(f"mel L1 {m} vs direct fp32: {(outs[m] - outs['direct']).abs().mean().item():.3e}   max {(outs[m] - outs['direct']).abs().max().item():.3e}")
for m in ("direct", "f23", "f43"):
    print(f"mel L1 {m} vs conv-in-fp64: {(outs[m] - outs['f64']).abs().mean().item():.3e}")
# single conv error
x = torch.randn(1, 256, 256); w = torch.randn(512, 256, 3) / math.sqrt(768); b = torch.zeros(512)
ref = orig(x.double(), w.double(), b.double(), dilation=2)
for nm, (BT, G, AT) in {"f23": (BT2, G2, AT2), "f43": (BT4, G4, AT4)}.items():
    e = (wino_conv(x, w, b, 2, BT, G, AT).double() - ref).abs().mean().item()
    print(nm, "single conv mean abs err vs fp64", f"{e:.3e}")
print("direct fp32 single conv err", f"{(orig(x, w, b, dilation=2).double() - ref).abs().mean().item():.3e}")
