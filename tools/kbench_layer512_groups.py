"""Round-6 experiment: ss_layer512's fp16sd launch in a TWO-GROUP form (tools/experiments/layer512_groups.hip, its own shared object: the eight waves of a
workgroup as two groups of four that work half a period apart on the two 64-row halves of a tile - one multiplies while the other runs its epilogues)
against the product's layer512_kernel. Prints: bit-identity of G, H and the stream remainder on six shapes (one tile, a few, ragged lengths, whole-tile
tail, split tail, gate-only form), then us per launch of both at BASELINE configs[3]'s shape.
    python tools/kbench_layer512_groups.py [--iters 400]
Measured (profiles/r06_kbench_layer512_groups.log): bit-identical everywhere; 267-272 us against 226 us - NOT adopted. One conv wave per SIMD has
to carry the matrix pipe alone, and on this chip a wave's VALU work is not hidden under another wave's MFMAs."""
import argparse
import ctypes
import math
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylesinger_amd import lib as L  # noqa: E402

EXP_SRC = os.path.join(ROOT, "tools", "experiments", "layer512_groups.hip")
EXP_SO = os.path.join(ROOT, "tools", "experiments", "liblayer512_groups.so")
C = 256


def load_experiment():
    deps = [EXP_SRC, os.path.join(ROOT, "stylesinger_amd", "csrc", "common.h"), os.path.join(ROOT, "include", "stylesinger_hip.h")]
    if not os.path.exists(EXP_SO) or os.path.getmtime(EXP_SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", EXP_SRC, "-o", EXP_SO], check=True)
    x = ctypes.CDLL(EXP_SO)
    x.ssx_layer512_groups_last_error.restype = ctypes.c_char_p
    x.ssx_layer512_groups.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return x


def launch(x, use_groups, Hin, Wg, E512, G, *, B, T, d, lens, Hout=None, P=None, Wr=None, bias_r=None, next_bias=None, cur_bias=None, ldg=None, g_bs=None,
           g_compact=False):
    if not use_groups:
        return L.layer512(Hin, Wg, E512, G, B=B, T=T, d=d, lens=lens, Hout=Hout, P=P, Wr=Wr, bias_r=bias_r, next_bias=next_bias, cur_bias=cur_bias, ldg=ldg, g_bs=g_bs,
                          n_products=1, e_f16=True, g_compact=g_compact)
    a = L.Layer512Args()
    a.Hin = L.ptr(Hin); a.d = d; a.n_products = 1; a.g_compact = int(g_compact); a.e_f16 = 1
    a.Hout = L.ptr(Hout); a.P = L.ptr(P)
    a.lens = L.ptr(lens); a.B = B; a.T = T; a.Wg = L.ptr(Wg); a.Wr = L.ptr(Wr); a.E512 = L.ptr(E512)
    a.G = L.ptr(G); a.ldg = ldg if ldg is not None else G.shape[-1]; a.g_batch_stride = g_bs if g_bs is not None else T * a.ldg
    a.mask_rows = 1; a.bias_r = L.ptr(bias_r); a.next_bias = L.ptr(next_bias); a.cur_bias = L.ptr(cur_bias)
    a.out_scale = 1.0 / 256.0; a.post_scale = 0.70710678118654752440
    rc = x.ssx_layer512_groups(ctypes.byref(a), L.stream_ptr())
    if rc != 0:
        raise RuntimeError(f"ssx_layer512_groups failed ({rc}): {x.ssx_layer512_groups_last_error().decode(errors='replace')}")


def case(B, T, lens_list, seed, dev):
    g = torch.Generator().manual_seed(seed)
    lens = torch.tensor(lens_list, dtype=torch.int32, device=dev)
    xs = (torch.randn(B, T, C, generator=g) * 2.0).to(dev)
    cb, nb, bo = (torch.randn(C, generator=g).to(dev) for _ in range(3))
    H, P = L.layer512_entry(xs, cb, B=B, T=T, lens=lens)
    w = (torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C)).to(dev)
    wo = (torch.randn(2 * C, C, 1, generator=g) / math.sqrt(C)).to(dev)
    Wg = L.layer512_pack_gate(L.split_f16(L.pack_conv_weight(w, interleave_half=C), scale=256.0), 1)
    Wr = L.layer512_pack_res(L.split_f16(L.pack_conv_weight(wo), scale=256.0), 1)
    E = (torch.randn(B, T, 2 * C, generator=g) * 0.5).to(dev)
    sets = L.layer512_tile_addend_f16(E, 2, B=B, T=T)
    return dict(lens=lens, cb=cb, nb=nb, bo=bo, H=H, P=P, Wg=Wg, Wr=Wr, sets=sets)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    x = load_experiment()
    Lyr = 3
    for B, T, lens, d in ((1, 128, [128], 2), (2, 600, [600, 411], 1), (3, 517, [517, 480, 5], 8), (6, 128 * 70 + 37, None, 2), (5, 7701, None, 4), (7, 128 * 73, None, 4)):
        if lens is None:
            lens = [T - 37 * b for b in range(B)]
            lens[-1] = 77
        c = case(B, T, lens, B * 1000 + d, dev)
        res = []
        for groups in (False, True):
            GA = torch.full((B, T, Lyr * C), 7.0, device=dev, dtype=torch.float16)
            Hout = torch.full_like(c["H"], 5.0)
            P = c["P"].clone()
            launch(x, groups, c["H"], c["Wg"], c["sets"][1], GA[..., C:], B=B, T=T, d=d, lens=c["lens"], Hout=Hout, P=P, Wr=c["Wr"], bias_r=c["bo"], next_bias=c["nb"],
                   cur_bias=c["cb"], ldg=Lyr * C, g_bs=T * Lyr * C, g_compact=True)
            G2 = torch.full((B, T, 2 * C), 7.0, device=dev, dtype=torch.float16)
            launch(x, groups, c["H"], c["Wg"], c["sets"][0], G2, B=B, T=T, d=d, lens=c["lens"])   # gate only, pair layout
            torch.cuda.synchronize()
            res.append((GA, Hout, P, G2))
        for name, u, v in zip(("G", "Hout", "R", "G of the gate-only form"), res[0], res[1]):
            if not torch.equal(u.view(torch.uint8), v.view(torch.uint8)):
                bad = (u.view(torch.int16) != v.view(torch.int16)).nonzero()
                raise AssertionError(f"{B} x {T}: {name}: {bad.shape[0]} of {u.numel()} elements differ; first {bad[:4].tolist()}")
        print(f"two-group form {B} x {T} ({B * ((T + 127) // 128)} tiles), d = {d}: bit-identical to layer512_kernel (G, Hout, stream remainder, gate-only G)")
    # ---- timing at BASELINE configs[3]'s shape
    B, T = 32, 5625
    c = case(B, T, [T] * B, 3, dev)
    NS = 4
    E = torch.randn(B, T, NS * 2 * C, device=dev)
    E512 = [L.layer512_tile_addend_f16(E[..., s * 2 * C:], 1, B=B, T=T, lde=NS * 2 * C)[0] for s in range(NS)]
    GA = torch.empty(B, T, NS * C, device=dev, dtype=torch.float16)
    H = [c["H"], torch.empty_like(c["H"])]
    for groups in (False, True, False, True):
        k = [0]

        def run():
            k[0] += 1
            s = k[0] % NS
            launch(x, groups, H[k[0] & 1], c["Wg"], E512[s], GA[..., s * C:], B=B, T=T, d=2, lens=c["lens"], Hout=H[(k[0] & 1) ^ 1], P=c["P"], Wr=c["Wr"], bias_r=c["bo"],
                   next_bias=c["nb"], cur_bias=c["nb"], ldg=NS * C, g_bs=T * NS * C, g_compact=True)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"{'two-group form (experiment)' if groups else 'layer512_kernel (product)  '}: {e0.elapsed_time(e1) / a.iters * 1e3:8.1f} us per launch (32 x 5625, one product, fp16 addend set, compact G)")


if __name__ == "__main__":
    main()
