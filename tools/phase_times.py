"""Debug: per-phase cycle breakdown of the gate kernel (needs SS_DBG=16)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d = torch.device("cuda:0")
B, T, C, Lyr = 8, 1500, 256, 20
lens = torch.full((B,), T, device=d, dtype=torch.int32)
X = torch.randn(B, T, C, device=d); G = torch.empty(B, T, C, device=d)
E = torch.randn(B, T, Lyr * 2 * C, device=d)
W = L.pack_conv_weight(torch.randn(2 * C, C, 3, device=d) / math.sqrt(3 * C), interleave_half=C)
ab = torch.randn(C, device=d)
dbg = torch.zeros(4096 * 4 * 8, device=d, dtype=torch.int64)
def run():
    L.conv_gemm(X, W, G, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, taps=(-2, 0, 2), lens=lens, a_bias=ab, epi=L.EPI_GATE, E=E,
                lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C, tile=tile, C2=dbg)
for _ in range(3): run()
torch.cuda.synchronize()
r = dbg.cpu().view(-1, 8)
r = r[r[:, 4] > 0].double()
names = ["load issue", "ds_read+MFMA", "vmcnt+ds_write", "barrier", "loop total", "t_begin", "epilogue", "t_end"]
print(f"tile {tile}: {r.shape[0]} waves")
for i in (0, 1, 2, 3, 4, 6):
    print(f"  {names[i]:16s} mean {r[:, i].mean():10.0f}  min {r[:, i].min():10.0f}  max {r[:, i].max():10.0f}  (x100MHz-counter?)")
span = (r[:, 7].max() - r[:, 5].min())
print("  kernel span (first begin -> last end):", span, " mean wave start offset:", (r[:, 5] - r[:, 5].min()).mean())
