"""Round-5 experiment: one layer's F(4,3) gate + residual projection of the mel denoiser as ONE dataflow launch (tools/experiments/fused_gate_res.hip, its own shared object: per-row-tile
counters, agent-scope release / acquire) against the two dependent launches the loop uses, at BASELINE configs[1]'s shape (B = 8 x T = 1500).
Prints: bit-identity of the 20-layer chain's outputs, us per layer of (gate, projection) as 40 launches vs 20 fused launches (+ one memset of
the counters), both replayed from a hipGraph.
    python tools/kbench_fused.py [--B 8] [--T 1500] [--iters 30]
"""
import argparse
import math
import os
import sys

import torch

import ctypes
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylesinger_amd import lib as L  # noqa: E402

EXP_SRC = os.path.join(ROOT, "tools", "experiments", "fused_gate_res.hip")
EXP_SO = os.path.join(ROOT, "tools", "experiments", "libfused_gate_res.so")


def load_experiment():
    """The experiment is NOT in libstylesinger_hip.so: build its own shared object (hipcc, ~20 s) unless a current one travelled with the tree."""
    deps = [EXP_SRC, os.path.join(ROOT, "stylesinger_amd", "csrc", "wino43_gate16.hip"), os.path.join(ROOT, "stylesinger_amd", "csrc", "gemm16.hip")]
    if not os.path.exists(EXP_SO) or os.path.getmtime(EXP_SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", EXP_SRC, "-o", EXP_SO], check=True)
    x = ctypes.CDLL(EXP_SO)
    x.ssx_fused_last_error.restype = ctypes.c_char_p
    x.ssx_fused_gate_res_counters.argtypes = [ctypes.c_int] * 3
    x.ssx_fused_gate_res.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int, ctypes.c_void_p]
    return x


def fused_gate_res(x, gate_kw, res_kw, *, dilation, counters, error, write_through):
    g, r = dict(gate_kw), dict(res_kw)
    g.setdefault("epi", L.EPI_GATE)
    ga = L._fill_args(g.pop("A"), g.pop("W"), g.pop("out"), **{k: v for k, v in g.items() if k != "W16"})
    ra = L._fill_args(r.pop("A"), r.pop("W"), r.pop("out"), **{k: v for k, v in r.items() if k != "W16"})
    rc = x.ssx_fused_gate_res(ctypes.byref(ga), L.ptr(gate_kw["W16"]), int(dilation), ctypes.byref(ra), L.ptr(res_kw["W16"]), L.ptr(counters), L.ptr(error),
                              int(bool(write_through)), L.stream_ptr())
    if rc != 0:
        raise RuntimeError(f"ssx_fused_gate_res failed ({rc}): {x.ssx_fused_last_error().decode(errors='replace')}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, T, C, Lyr = a.B, a.T, 256, 20
    g = torch.Generator().manual_seed(7)
    lens = torch.full((B,), T, device=dev, dtype=torch.int32)
    lens[-1] = T - 37
    X0 = torch.randn(B, T, C, generator=g).to(dev)
    for b in range(B):
        X0[b, int(lens[b]):] = 0
    E = (torch.randn(B, T, Lyr * 2 * C, generator=g) * 0.5).to(dev)
    dstep = (torch.randn(Lyr, C, generator=g) * 0.1).to(dev)
    lib = L.load()
    exp = load_experiment()
    packs = []
    for l in range(Lyr):
        d = 1 << (l % 4)
        w = (torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C)).to(dev)
        wt = L.pack_conv_weight(L.wino43_weight(w), interleave_half=C)
        wo = (torch.randn(2 * C, C, 1, generator=g) / math.sqrt(C)).to(dev)
        wop = L.pack_conv_weight(wo)
        bo = L.pack_bias((torch.randn(2 * C, generator=g) * 0.1).to(dev))
        packs.append(dict(d=d, wt=wt, wt16=L.pack_gate16_weights(wt, C), wo=wop, wo16=L.pack_gemm16_weights(wop[:C].contiguous(), wop.shape[1]), bo=bo,
                          e16=L.gate16_tile_addend(E[:, :, l * 2 * C:], B=B, T=T, Np=2 * C, lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, dilation=d, mt=2)))
    ncnt = [exp.ssx_fused_gate_res_counters(B, T, p["d"]) for p in packs]
    counters = torch.zeros(sum(ncnt), device=dev, dtype=torch.int32)
    coff = [sum(ncnt[:l]) for l in range(Lyr)]
    err = torch.zeros(1, device=dev, dtype=torch.int32)

    def kws(l, X, GA):
        p = packs[l]
        gk = dict(A=X, W=p["wt"], out=GA[:, :, l * C:], W16=p["wt16"], B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=lens, a_bias=dstep[l], epi=L.EPI_GATE,
                  E=p["e16"], e_tiled=True, lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=Lyr * C, c_bs=T * Lyr * C, mask_rows=True)
        rk = dict(A=GA[:, :, l * C:], W=p["wo"], out=X, W16=p["wo16"], R=X, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lda=Lyr * C, a_bs=T * Lyr * C, lens=lens,
                  bias=p["bo"], ldr=C, ldc=C, post_scale=0.70710678)
        return gk, rk

    def two_launch(X, GA):
        for l in range(Lyr):
            gk, rk = kws(l, X, GA)
            g2 = {k: v for k, v in gk.items() if k not in ("A", "W", "out", "W16")}
            L.wino43_gate16(gk["A"], gk["W"], gk["out"], dilation=packs[l]["d"], mt=2, W16=gk["W16"], **g2)
            r2 = {k: v for k, v in rk.items() if k not in ("A", "W", "out", "W16")}
            L.gemm16_res(rk["A"], rk["W"], rk["out"], mt=6, W16=rk["W16"], **r2)

    def make_fused(wt):
        def fused(X, GA):
            counters.zero_()
            for l in range(Lyr):
                gk, rk = kws(l, X, GA)
                fused_gate_res(exp, gk, rk, dilation=packs[l]["d"], counters=counters[coff[l]:], error=err, write_through=wt)
        return fused
    fused, fused_wt = make_fused(False), make_fused(True)

    outs = []
    for fn in (two_launch, fused, fused_wt):
        X = X0.clone()
        GA = torch.zeros(B, T, Lyr * C, device=dev)
        fn(X, GA)
        torch.cuda.synchronize()
        outs.append((X, GA))
    for name, o in (("release / acquire fences", outs[1]), ("write-through stores + sc1 loads", outs[2])):
        same = torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1])
        print(f"dataflow launch ({name}) vs two launches after {Lyr} layers: bit-identical {same}; max |dX| {(outs[0][0] - o[0]).abs().max().item():.3e}, "
              f"max |dG| {(outs[0][1] - o[1]).abs().max().item():.3e}; wait gave up: {int(err.item())}")
    # the write-through form has no fence to fall back on: repeat it under load and compare every time (staleness would show as a mismatch)
    bad = 0
    for rep in range(10):
        X = X0.clone()
        GA = torch.zeros(B, T, Lyr * C, device=dev)
        fused_wt(X, GA)
        bad += int(not (torch.equal(outs[0][0], X) and torch.equal(outs[0][1], GA)))
    print(f"write-through form repeated 10x: {bad} mismatching runs")
    assert torch.isfinite(outs[0][0]).all()

    def timed(fn):
        X = X0.clone()
        GA = torch.zeros(B, T, Lyr * C, device=dev)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            fn(X, GA)
            st.synchronize()
            with torch.cuda.graph(gr, stream=st):
                fn(X, GA)
            gr.replay()
            st.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(a.iters):
                gr.replay()
            e1.record(st)
            st.synchronize()
        torch.cuda.current_stream().wait_stream(st)
        return e0.elapsed_time(e1) * 1e3 / (a.iters * Lyr)
    for rep in range(2):
        t2 = timed(two_launch)
        tf = timed(fused)
        tw = timed(fused_wt)
        print(f"B={B} T={T}: per layer, two launches (gate, projection) {t2:.2f} us; one dataflow launch with fences {tf:.2f} us ({(tf / t2 - 1) * 100:+.1f} %), "
              f"with write-through stores / sc1 loads {tw:.2f} us ({(tw / t2 - 1) * 100:+.1f} %); wait gave up: {int(err.item())}")


if __name__ == "__main__":
    main()
