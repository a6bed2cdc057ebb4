"""Micro-benchmark of the hot conv_gemm launches (mel/f0 gate + res/skip, vocoder convs) with HIP events.
    python tools/kbench.py [--tile N] [--iters K] [--which gate|resskip|voc|all]
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L  # noqa: E402


def timeit(fn, iters):
    """Back-to-back launches: `fn` captured 10x into a hipGraph (a Python loop of ctypes calls would time the host for
    kernels of ~100 us), replayed between two events on the capture stream."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    reps = 10
    graph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            for _ in range(reps):
                fn()
        graph.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = max(1, iters // reps)
        e0.record(st)
        for _ in range(n):
            graph.replay()
        e1.record(st)
        st.synchronize()
    torch.cuda.current_stream().wait_stream(st)
    return e0.elapsed_time(e1) * 1e-3 / (n * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--sweep", default="", help="comma separated tile ids to sweep (overrides --tile)")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--which", default="all")
    ap.add_argument("--net", default="both", help="mel|f0|both")
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--e16", action="store_true", help="wino43_16: conditioner addend in the kernel's fetch order (ss_gate16_tile_addend)")
    ap.add_argument("--prio", type=int, default=-1, help="ss_set_tuning('wave_prio', N)")
    ap.add_argument("--mt", default="-1,3,2", help="wino43_16: comma list of tilings (-1 = the 32x32x2 kernel, 0 = library pick, 2, 3)")
    ap.add_argument("--w16", type=int, default=1, help="wino43_16: 1 = weights in the kernel's fetch order (ss_wino43_gate16w), 0 = packed rows")
    ap.add_argument("--x3", action="store_true", help="wino43_16: the bf16x3 form (ss_wino43_gate16x: split operands on the bf16 matrix cores)")
    ap.add_argument("--pair", type=int, default=1, help="wino43_16: time the f0 launch with 2B items (both nets), as the loop launches it")
    ap.add_argument("--e-layout", default="row", help="conditioner addend: 'row' = [B][T][L*2C] (one row holds all layers), 'layer' = [L][B][T][2C]")
    a = ap.parse_args()
    if a.prio >= 0:
        L.check(L.load().ss_set_tuning(b"wave_prio", a.prio), "ss_set_tuning")
    if a.sweep:
        import subprocess
        for t in a.sweep.split(","):
            print(f"--- tile {t}")
            sys.stdout.flush()
            subprocess.run([sys.executable, __file__, "--tile", t, "--which", a.which, "--iters", str(a.iters), "--B", str(a.B), "--T", str(a.T)])
        return
    d = torch.device("cuda:0")
    B, T = a.B, a.T
    lens = torch.full((B,), T, device=d, dtype=torch.int32)
    res = []
    for (name, C, Lyr) in (("mel", 256, 20), ("f0", 192, 10)):
        if a.net not in ("both", name):
            continue
        X = torch.randn(B, T, C, device=d)
        G = torch.randn(B, T, C, device=d)
        S = torch.zeros(B, T, C, device=d)
        E = torch.randn(B, T, Lyr * 2 * C, device=d)
        if a.e_layout == "layer":
            El = torch.randn(Lyr, B, T, 2 * C, device=d)
        layer = [0]
        w = torch.randn(2 * C, C, 3, device=d) / math.sqrt(3 * C)
        wo = torch.randn(2 * C, C, 1, device=d) / math.sqrt(C)
        bo = torch.randn(2 * C, device=d) * 0.1
        ab = torch.randn(C, device=d)
        W = L.pack_conv_weight(w, interleave_half=C)
        Wo = L.pack_conv_weight(wo)
        bop = L.pack_bias(bo)
        if a.which in ("gate", "all"):
            def f():
                layer[0] = (layer[0] + 1) % Lyr  # walk the conditioner slab like the real loop (E is 491 MB: HBM, not L2)
                L.conv_gemm(X, W, G, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, taps=(-2, 0, 2), lens=lens, a_bias=ab,
                            epi=L.EPI_GATE, E=E[:, :, layer[0] * 2 * C:], lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C, tile=a.tile)
            s = timeit(f, a.iters)
            fl = 2.0 * B * T * 3 * C * 2 * C
            res.append((f"{name} gate  K={3 * C} N={2 * C}", s, fl))
        if a.which in ("wino", "all"):
            Wt = L.pack_conv_weight(L.wino_weight(w), interleave_half=C)
            def fw():
                layer[0] = (layer[0] + 1) % Lyr
                L.wino_gate(X, Wt, G, dilation=2, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=lens, a_bias=ab,
                            E=E[:, :, layer[0] * 2 * C:], lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C, tile=a.tile)
            s = timeit(fw, a.iters)
            fl = 2.0 * B * T * 3 * C * 2 * C
            res.append((f"{name} WINO gate K={3 * C} N={2 * C} (algorithmic flops)", s, fl))
        if a.which in ("wino43", "all"):
            Wt4 = L.pack_conv_weight(L.wino43_weight(w), interleave_half=C)
            def fw4():
                layer[0] = (layer[0] + 1) % Lyr
                if a.e_layout == "layer":
                    L.wino43_gate(X, Wt4, G, dilation=2, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=lens, a_bias=ab,
                                  E=El[layer[0]], lde=2 * C, e_bs=T * 2 * C, ldc=C)
                    return
                L.wino43_gate(X, Wt4, G, dilation=2, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=lens, a_bias=ab,
                              E=E[:, :, layer[0] * 2 * C:], lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C)
            s = timeit(fw4, a.iters)
            fl = 2.0 * B * T * 3 * C * 2 * C
            res.append((f"{name} WINO F(4,3) gate K={3 * C} N={2 * C} (algorithmic flops)", s, fl))
        if a.which in ("wino43_16", "all"):
            Wt4 = L.pack_conv_weight(L.wino43_weight(w), interleave_half=C)
            Wx4 = L.split3_weights(Wt4, C)
            W16 = L.pack_gate16_weights(Wt4, C)
            Bn = B * (2 if name == "f0" and a.pair else 1)   # the f0 launch of the real loop carries both nets: 2B items
            Xn = torch.randn(Bn, T, C, device=d)
            Gn = torch.empty(Bn, T, C, device=d)
            En = torch.randn(Bn, T, Lyr * 2 * C, device=d)
            ln = torch.full((Bn,), T, device=d, dtype=torch.int32)
            for mt in [int(v) for v in a.mt.split(",")]:
                # --e16: the addend of every layer slab in the kernel's fetch order (what the loops launch since round 4)
                E16s = [L.gate16_tile_addend(En[:, :, l * 2 * C:], B=Bn, T=T, Np=2 * C, lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, dilation=2, mt=mt)
                        for l in range(Lyr)] if (a.e16 and mt > 0 and not a.x3) else None
                def fw16():
                    layer[0] = (layer[0] + 1) % Lyr
                    kw = dict(dilation=2, B=Bn, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=ln, a_bias=ab,
                              E=En[:, :, layer[0] * 2 * C:], lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C)
                    kwx = kw
                    if E16s is not None:
                        kw = dict(kw, E=E16s[layer[0]], e_tiled=True)
                    if mt < 0:
                        L.wino43_gate(Xn, Wt4, Gn, **kw)
                    elif a.x3:
                        L.wino43_gate16x(Xn, Wx4, Gn, mt=mt, **kwx)
                    else:
                        L.wino43_gate16(Xn, Wt4, Gn, mt=mt, W16=W16 if a.w16 else None, **kw)
                s = timeit(fw16, a.iters)
                fl = 2.0 * Bn * T * 3 * C * 2 * C
                res.append((f"{name} F(4,3) gate {'32x32x2 tiles' if mt < 0 else ('bf16x3 mt=%d' if a.x3 else '16x16x4 mt=%d') % mt} rows={Bn * T}", s, fl))
        if a.which in ("res16", "all"):
            Bn = B * (2 if name == "f0" and a.pair else 1)
            GAn = torch.randn(Bn, T, Lyr * C, device=d)
            Xn = torch.randn(Bn, T, C, device=d)
            Sn = torch.zeros(Bn, T, C, device=d)
            ln = torch.full((Bn,), T, device=d, dtype=torch.int32)
            kwr = dict(B=Bn, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lda=Lyr * C, a_bs=T * Lyr * C, lens=ln, bias=bop, ldr=C, ldc=C, post_scale=0.7071)
            Wo16 = L.pack_gemm16_weights(Wo[:C].contiguous(), C)
            for mt in [int(v) for v in a.mt.split(",")]:
                def fr():
                    layer[0] = (layer[0] + 1) % Lyr
                    Al = GAn[:, :, layer[0] * C:]
                    if mt < 0:
                        L.conv_gemm(Al, Wo, Xn, epi=L.EPI_RESSKIP, R=Xn, Nh=C, C2=Sn, ldc2=C, c2_bs=T * C, tile=3, **kwr)
                    else:
                        L.gemm16_res(Al, Wo, Xn, mt=mt, R=Xn, W16=Wo16 if a.w16 else None, **kwr)
                s = timeit(fr, a.iters)
                fl = 2.0 * Bn * T * C * C
                res.append((f"{name} residual projection {'conv_gemm 64x64' if mt < 0 else 'gemm16 mt=%d' % mt} rows={Bn * T} K=N={C}", s, fl))
        if a.which in ("resskip", "all"):
            f = lambda: L.conv_gemm(G, Wo, X, B=B, T=T, Cin=C, N=2 * C, Np=2 * C, Kp=C, lens=lens, epi=L.EPI_RESSKIP, bias=bop, Nh=C,
                                    R=X, ldr=C, ldc=C, post_scale=0.7071, C2=S, ldc2=C, c2_bs=T * C, accumulate=True, tile=a.tile)
            s = timeit(f, a.iters)
            fl = 2.0 * B * T * C * 2 * C
            res.append((f"{name} resskip K={C} N={2 * C}", s, fl))
    if a.which in ("voc", "all"):
        for (C, R, k) in ((256, 8, 7), (128, 64, 7), (64, 128, 11), (32, 256, 11), (32, 256, 3)):
            rows = T * R
            X = torch.randn(B, rows, C, device=d)
            Y = torch.empty(B, rows, C, device=d)
            w = torch.randn(C, C, k, device=d) / math.sqrt(k * C)
            W = L.pack_conv_weight(w)
            bias = L.pack_bias(torch.randn(C, device=d))
            ln = torch.full((B,), rows, device=d, dtype=torch.int32)
            f = lambda: L.conv_gemm(X, W, Y, B=B, T=rows, Cin=C, N=C, Np=W.shape[0], Kp=W.shape[1] // k,
                                    taps=[(j - k // 2) * 3 for j in range(k)], lens=ln, a_lrelu=0.1, bias=bias, R=X, ldr=C, tile=a.tile)
            s = timeit(f, max(3, a.iters // 5))
            res.append((f"voc C={C} rows={rows} k={k}", s, 2.0 * B * rows * C * C * k))
            if L.load().ss_wino43_conv_ok(C, k, 3):   # the same conv as grouped Winograd F(4,3) (algorithmic flops of the direct form)
                Ww = L.pack_conv_weight(L.wino43_group_weight(w))
                fw = lambda: L.wino43_conv(X, Ww, Y, k=k, dilation=3, B=B, T=rows, Cin=C, N=C, Np=C, Kp=C, lens=ln, a_lrelu=0.1, bias=bias,
                                           R=X, ldr=C)
                s = timeit(fw, max(3, a.iters // 5))
                res.append((f"voc C={C} rows={rows} k={k} grouped F(4,3)", s, 2.0 * B * rows * C * C * k))
    for name, s, fl in res:
        print(f"{name:52s} {s * 1e6:9.1f} us  {fl / s / 1e12:7.2f} TF/s  ({fl / s / 157.3e12 * 100:5.1f}% of fp32 MFMA peak)")


if __name__ == "__main__":
    main()
