"""Debug: per-wave timestamps of the gate kernel: entry / loop start / loop end / exit.
Needs a library built with -DSS_KERNEL_TIMESTAMPS (SS_EXTRA_HIPCC_FLAGS=-DSS_KERNEL_TIMESTAMPS python -m stylesinger_amd.build)
and SS_DBG=16 at run time; the shipped library carries no instrumentation."""
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
dbg = torch.zeros(8192 * 8, device=d, dtype=torch.int64)
def run():
    L.conv_gemm(X, W, G, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, taps=(-2, 0, 2), lens=lens, a_bias=ab, epi=L.EPI_GATE, E=E,
                lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C, tile=tile, C2=dbg)
for _ in range(3): run()
torch.cuda.synchronize()
dbg.zero_(); run(); torch.cuda.synchronize()
r = dbg.cpu().view(-1, 8)
r = r[r[:, 0] > 0]
hw = r[:, 4]; xcc = r[:, 5] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
import collections
percu = collections.defaultdict(list)
for i in range(r.shape[0]):
    percu[int(key[i])].append((int(r[i, 0]), int(r[i, 3]), int(simd[i]), int(r[i, 6])))
print(f"tile {tile}: {r.shape[0]} waves on {len(percu)} distinct CUs")
nb = collections.Counter(len(set(b for _, _, _, b in v)) for v in percu.values())
print("  blocks per CU histogram:", sorted(nb.items()))
# max concurrency per CU (waves alive at the same time, same clock domain)
conc = collections.Counter()
for v in percu.values():
    ev = sorted([(s, 1) for s, e, _, _ in v] + [(e, -1) for s, e, _, _ in v])
    c = m = 0
    for _, d_ in ev:
        c += d_; m = max(m, c)
    conc[m] += 1
print("  max concurrent waves per CU histogram:", sorted(conc.items()))
rd = r.double()
print(f"  prologue mean {(rd[:,1]-rd[:,0]).mean():.0f}  loop mean {(rd[:,2]-rd[:,1]).mean():.0f} (min {(rd[:,2]-rd[:,1]).min():.0f} max {(rd[:,2]-rd[:,1]).max():.0f})  epilogue mean {(rd[:,3]-rd[:,2]).mean():.0f}")
span = [max(e for _, e, _, _ in v) - min(s for s, _, _, _ in v) for v in percu.values()]
print(f"  per-CU busy span: mean {sum(span)/len(span):.0f} max {max(span)} min {min(span)} cycles")
