import math, sys, os, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stylesinger_amd import lib as L
dv = torch.device("cuda:0")
K, N, B, T, ksplit = 5120, 256, 1, 750, 5
g = torch.Generator().manual_seed(1)
A = torch.randn(B, T, K, generator=g).to(dv)
w = (torch.randn(N, K, 1, generator=g) / math.sqrt(K)).to(dv)
Wp = L.pack_conv_weight(w)
out = torch.zeros(B, T, N, device=dv)
a = L._fill_args(A, Wp, out, B=B, T=T, Cin=K, N=N, Np=Wp.shape[0], Kp=Wp.shape[1], mask_rows=False)
part = torch.full((ksplit, B, T, N), 77.0, device=dv)
L.check(L.load().ss_gemm16_store_splitk(C.byref(a), 0, ksplit, L.ptr(part), L.stream_ptr()), "x")
torch.cuda.synchronize()
kc = K // 32
for s in range(ksplit):
    k0, k1 = s * kc // ksplit * 32, (s + 1) * kc // ksplit * 32
    ref = A[0, :, k0:k1].double() @ w[:, k0:k1, 0].double().t()
    d = (part[s, 0].double() - ref).abs()
    print(s, k0, k1, "max err", d.max().item(), "untouched", (part[s] == 77.0).sum().item(), "rows bad", (d.max(1).values > 1e-3).sum().item(), "cols bad", (d.max(0).values > 1e-3).sum().item())
ref = A[0].double() @ w[:, :, 0].double().t()
print("total", (out[0].double() - ref).abs().max().item(), (part.sum(0)[0].double() - ref).abs().max().item())
