"""Per-wave phase timing of the Winograd gate kernels (debug build with -DSS_TRACE; see csrc/wino_gate.hip, csrc/wino43_gate.hip).

Build in the container:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSS_TRACE stylesinger_amd/csrc/*.hip -o stylesinger_amd/_abl/libss_trace.so
Run on the GPU box:
    SS_LIB_PATH=stylesinger_amd/_abl/libss_trace.so python tools/wave_trace.py [--kernel wino|wino43] [--B 8] [--T 1500]
(`--kernel wino43`, the F(4,3) kernel, was instrumented after the round's GPU budget was spent: its product build is bit-identical to
the un-instrumented one, the traced build has not run yet.)

Every wave sums the shader-clock time of five phases over its K chunks:
    reads  : barrier release -> first two LDS fragment pairs arrived
    mfma1  : -> first 8 MFMAs issued and the other two fragment pairs arrived
    vmwait : -> global fetches of the next chunk arrived (s_waitcnt vmcnt(0))
    stage  : -> transform + LDS stores + fetch issue + last 8 MFMAs issued, LDS stores landed
    barrier: -> barrier released
"""
import argparse
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--C", type=int, default=256)
    ap.add_argument("--dil", type=int, default=2)
    ap.add_argument("--kernel", default="wino", help="wino = F(2,3) wino_gate_kernel_v2, wino43 = F(4,3) wino43_gate_kernel")
    a = ap.parse_args()
    lib = L.load()
    f43 = a.kernel == "wino43"
    fn = lib.ss_debug_set_wino43_trace if f43 else lib.ss_debug_set_wino_trace  # only in -DSS_TRACE builds
    d = torch.device("cuda:0")
    B, T, C, Lyr = a.B, a.T, a.C, 20
    lens = torch.full((B,), T, device=d, dtype=torch.int32)
    X = torch.randn(B, T, C, device=d)
    G = torch.randn(B, T, C, device=d)
    E = torch.randn(B, T, Lyr * 2 * C, device=d)
    w = torch.randn(2 * C, C, 3, device=d) / math.sqrt(3 * C)
    ab = torch.randn(C, device=d)
    Wt = L.pack_conv_weight(L.wino43_weight(w) if f43 else L.wino_weight(w), interleave_half=C)
    nblk = 8192
    tr = torch.zeros(nblk * 4 * 16, device=d, dtype=torch.int64)
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_int
    assert fn(ctypes.c_void_p(tr.data_ptr())) == 0

    def run(layer):
        (L.wino43_gate if f43 else L.wino_gate)(X, Wt, G, dilation=a.dil, B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=lens, a_bias=ab,
                                                E=E[:, :, layer * 2 * C:], lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C, tile=0)
    for i in range(5):
        run(i)
    torch.cuda.synchronize()
    tr.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(7)
    e1.record()
    torch.cuda.synchronize()
    print(f"launch: {e0.elapsed_time(e1) * 1e3:.1f} us (events around one launch, includes launch latency)")
    t = tr.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 5] > 0]
    n = t.shape[0]
    chunks = t[:, 5].astype(np.float64)
    names = ["reads", "mfma1", "vmwait", "stage", "barrier"]
    per = t[:, :5].astype(np.float64) / chunks[:, None]
    print(f"{n} waves traced, {int(chunks[0])} chunks each")
    tot = per.sum(1)
    print(f"cycles per chunk (mean over waves): total {tot.mean():.0f}  (min {tot.min():.0f} max {tot.max():.0f}); a wave's 16 MFMAs need 1024")
    for k, nm in enumerate(names):
        print(f"  {nm:8s} {per[:, k].mean():8.0f}  ({100 * per[:, k].mean() / tot.mean():5.1f} %)   p10 {np.percentile(per[:, k], 10):7.0f}  p90 {np.percentile(per[:, k], 90):7.0f}")
    pro = (t[:, 7] - t[:, 6]).astype(np.float64)
    loop = t[:, :5].sum(1).astype(np.float64)
    whole = (t[:, 8] - t[:, 6]).astype(np.float64)
    print(f"per wave, cycles: whole {whole.mean():.0f}; instrumented loop {loop.mean():.0f}; entry->loop end {pro.mean():.0f}; epilogue+last chunk {(whole - pro).mean():.0f}")
    hw = t[:, 9]
    xcc = t[:, 10] & 0xF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    simd = (hw >> 4) & 0x3
    key = xcc * 1000 + se * 100 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print(f"distinct (xcc, se, cu): {len(uniq)}; waves per CU min {cnt.min()} max {cnt.max()}; SIMD histogram {np.bincount(simd.astype(int))}")
    # s_memrealtime: constant 100 MHz, the same counter on every XCD
    r0, r1 = t[:, 11].astype(np.float64), t[:, 12].astype(np.float64)
    life_us = (r1 - r0) / 100.0
    ghz = whole / np.maximum(life_us, 1e-9) / 1e3
    print(f"wave lifetime {life_us.mean():.1f} us (min {life_us.min():.1f} max {life_us.max():.1f}); shader clock from s_memtime / s_memrealtime: {np.median(ghz):.3f} GHz")
    print(f"kernel span, first wave entry -> last wave exit: {(r1.max() - r0.min()) / 100.0:.1f} us; entries spread over {(r0.max() - r0.min()) / 100.0:.1f} us; "
          f"exits spread over {(r1.max() - r1.min()) / 100.0:.1f} us")
    order = np.argsort(r0)
    q = [(r0[order[int(f * (n - 1))]] - r0.min()) / 100.0 for f in (0.25, 0.5, 0.75, 0.9, 1.0)]
    print("entry time of the 25/50/75/90/100 % wave after the first: " + " / ".join(f"{x:.1f}" for x in q) + " us")
    # per CU: how many of its waves are alive together (entry of the last-starting block vs exit of the first-finishing)
    ov = []
    for k in uniq:
        m = key == k
        ov.append((r1[m].min() - r0[m].max()) / 100.0)
    print(f"per CU: (first exit - last entry) mean {np.mean(ov):.1f} us (all its workgroups co-resident for that long)")


if __name__ == "__main__":
    main()
