"""What-if for the deferred-skip restructuring: res-only 1x1 GEMM per layer + one K = L*C skip GEMM per step."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit

if len(sys.argv) > 1:
    L.check(L.load().ss_set_tuning(b"wave_prio", int(sys.argv[1])), "ss_set_tuning")
d = torch.device("cuda:0")
for (name, B, T, C, Lyr) in (("mel", 8, 1500, 256, 20), ("f0", 16, 1500, 192, 10)):
    lens = torch.full((B,), T, device=d, dtype=torch.int32)
    Gall = torch.randn(B, T, Lyr * C, device=d)
    X = torch.randn(B, T, C, device=d)
    S = torch.zeros(B, T, C, device=d)
    wo = torch.randn(C, C, 1, device=d) / math.sqrt(C)
    Wo = L.pack_conv_weight(wo)
    bo = L.pack_bias(torch.randn(C, device=d) * 0.1)
    lay = [0]
    for tile in (2, 3):
        def f():
            lay[0] = (lay[0] + 1) % Lyr
            L.conv_gemm(Gall[:, :, lay[0] * C:], Wo, X, B=B, T=T, Cin=C, N=C, Np=Wo.shape[0], Kp=Wo.shape[1], lda=Lyr * C, a_bs=T * Lyr * C, lens=lens,
                        epi=L.EPI_RESSKIP, bias=bo, Nh=C, R=X, ldr=C, ldc=C, post_scale=0.7071, C2=S, ldc2=C, c2_bs=T * C, tile=tile)
        s = timeit(f, 60)
        print(f"{name} res-only K={C} N={C} tile {tile}: {s * 1e6:7.1f} us")
    ws = torch.randn(C, Lyr * C, 1, device=d) / math.sqrt(Lyr * C)
    Ws = L.pack_conv_weight(ws)
    for tile in (2, 3, 1):
        def g():
            L.conv_gemm(Gall, Ws, S, B=B, T=T, Cin=Lyr * C, N=C, Np=Ws.shape[0], Kp=Ws.shape[1], lens=lens, bias=bo, tile=tile)
        s = timeit(g, 20)
        fl = 2.0 * B * T * Lyr * C * C
        print(f"{name} skip GEMM K={Lyr * C} N={C} tile {tile}: {s * 1e6:7.1f} us = {s * 1e6 / Lyr:6.1f} us/layer  {fl / s / 1e12:6.1f} TF/s")
    for mt in (6, 4, 8):
        def g16():
            L.gemm16_store(Gall, Ws, S, mt=mt, B=B, T=T, Cin=Lyr * C, N=C, Np=Ws.shape[0], Kp=Ws.shape[1], lens=lens, bias=bo, act=L.ACT_RELU)
        s = timeit(g16, 20)
        fl = 2.0 * B * T * Lyr * C * C
        print(f"{name} skip GEMM K={Lyr * C} N={C} gemm16_store mt={mt}: {s * 1e6:7.1f} us  {fl / s / 1e12:6.1f} TF/s ({fl / s / 157.3e12 * 100:.1f} % of peak)")
    Wsx = L.split3_gemm16_weights(Ws, Ws.shape[1])
    def g16x():
        L.gemm16x_store(Gall, Ws, Wsx, S, mt=6, B=B, T=T, Cin=Lyr * C, N=C, Np=Ws.shape[0], Kp=Ws.shape[1], lens=lens, bias=bo, act=L.ACT_RELU)
    s = timeit(g16x, 20)
    print(f"{name} skip GEMM K={Lyr * C} N={C} gemm16x_store (bf16x3) mt=6: {s * 1e6:7.1f} us  {fl / s / 1e12:6.1f} TF/s algorithmic")
