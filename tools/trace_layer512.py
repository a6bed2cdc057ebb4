"""Per-phase shader-clock timing of layer512_kernel (debug build with -DSS_L512_TRACE).
Build in the container (only layer512.hip needs the flag; the other objects are the product's):
    mkdir -p stylesinger_amd/_abl && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSS_L512_TRACE -c stylesinger_amd/csrc/layer512.hip -o /tmp/l512t.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC -o stylesinger_amd/_abl/libss_l512trace.so /tmp/l512t.o $(ls stylesinger_amd/_obj/*.o | grep -v layer512)
Run on the GPU box:
    SS_LIB_PATH=stylesinger_amd/_abl/libss_l512trace.so python tools/trace_layer512.py [--B 32] [--T 5625] [--gate-only]
Stamps per tile: 0 after [B1], 1 conv loop done, 2 after [B2], 3 gate epilogue done, 4 after [B3], 5 G pass issued, 6 projection MFMAs done, 7 tile end."""
import argparse
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=5625)
    ap.add_argument("--gate-only", action="store_true")
    ap.add_argument("--e16", action="store_true", help="with --one: the addend as fp16 sets (ss_layer512_args.e_f16), one set per launch - the fp16sd loop's form")
    ap.add_argument("--one", action="store_true", help="n_products = 1 (the fp16sd launch)")
    ap.add_argument("--warm", type=int, default=800, help="launches before the stamped one: the chip ramps its clock up over ~0.1 s of load, and a phase that waits "
                    "for memory costs more CYCLES at a higher clock - a cold trace (1.3 GHz) understates them")
    a = ap.parse_args()
    d = torch.device("cuda:0")
    B, T, C, NS = a.B, a.T, 256, 4
    lens = torch.full((B,), T, device=d, dtype=torch.int32)
    X0 = torch.randn(B, T, C, device=d)
    H0, P = L.layer512_entry(X0, None, B=B, T=T, lens=lens)
    H = [H0, torch.empty_like(H0)]
    E = torch.randn(B, T, NS * 2 * C, device=d)
    E512 = [(L.layer512_tile_addend_f16(E[..., s * 2 * C:], 1, B=B, T=T, lde=NS * 2 * C)[0] if a.e16 else L.layer512_tile_addend(E[..., s * 2 * C:], B=B, T=T, lde=NS * 2 * C))
            for s in range(NS)]
    GA = torch.empty(B, T, NS * 2 * C, device=d, dtype=torch.float16)
    w = torch.randn(2 * C, C, 3, device=d) / math.sqrt(3 * C)
    NP = 1 if a.one else 2
    Wg = L.layer512_pack_gate(L.split_f16(L.pack_conv_weight(w, interleave_half=C), scale=256.0), NP)
    wo = torch.randn(2 * C, C, 1, device=d) / math.sqrt(C)
    Wr = L.layer512_pack_res(L.split_f16(L.pack_conv_weight(wo), scale=256.0), NP)
    cb, nb, bo = (torch.randn(C, device=d) for _ in range(3))
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    if os.environ.get("SS_L512_GRID"):
        ncu = min(ncu, int(os.environ["SS_L512_GRID"]))
    tr = torch.zeros(ncu * 8 * 8 * 8 + 4 * ncu, device=d, dtype=torch.int64)

    def run(k):
        s = k % NS
        if a.gate_only:
            L.layer512(H[0], Wg, E512[s], GA[..., s * 2 * C:], B=B, T=T, d=2, lens=lens, ldg=NS * 2 * C, g_bs=T * NS * 2 * C, n_products=NP, e_f16=a.e16)
        else:
            L.layer512(H[k & 1], Wg, E512[s], GA[..., s * 2 * C:], B=B, T=T, d=2, lens=lens, Hout=H[(k & 1) ^ 1], P=P, Wr=Wr, bias_r=bo, cur_bias=nb,
                       next_bias=nb, ldg=NS * 2 * C, g_bs=T * NS * 2 * C, n_products=NP, e_f16=a.e16)
    for k in range(6):
        run(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(6, 12):
        run(k)
    e1.record()
    torch.cuda.synchronize()
    print(f"launch time (trace build, no stamps written): {e0.elapsed_time(e1) / 6 * 1e3:.1f} us")
    L.check(L.load().ss_set_clock_probe(ctypes.c_void_p(tr.data_ptr())), "ss_set_clock_probe")
    L.load().ss_set_clock_probe(None)
    for k in range(a.warm):
        run(k)
    L.check(L.load().ss_set_clock_probe(ctypes.c_void_p(tr.data_ptr())), "ss_set_clock_probe")
    e0.record()
    run(a.warm)
    e1.record()
    torch.cuda.synchronize()
    print(f"the stamped launch: {e0.elapsed_time(e1) * 1e3:.1f} us between its events")
    L.load().ss_set_clock_probe(None)
    life = tr[ncu * 512:].view(ncu, 4).cpu().double()   # per workgroup: start (100 MHz ticks), start (shader cycles), end, end
    tr = tr[:ncu * 512]
    t = tr.view(ncu, 8, 8, 8).cpu().double()     # workgroup, wave, tile slot, stamp
    # ---- the launch as a whole. The shader-cycle counters differ between XCDs, so: (1) every workgroup's start / end on the constant 100 MHz
    # counter, relative to the earliest start; (2) a workgroup's items on its OWN cycle axis, from its own start
    s0 = life[:, 0].min()
    st_us, en_us = (life[:, 0] - s0) / 100.0, (life[:, 2] - s0) / 100.0
    ghz = ((life[:, 3] - life[:, 1]) / (life[:, 2] - life[:, 0]) / 10.0)
    print(f"workgroup life on the 100 MHz counter: start {st_us.mean().item():.1f} us after the first one (max {st_us.max().item():.1f}), end {en_us.mean().item():.1f} us "
          f"(min {en_us.min().item():.1f}, max {en_us.max().item():.1f}); shader clock over a workgroup's life {ghz.mean().item():.3f} GHz ({ghz.min().item():.3f} .. {ghz.max().item():.3f})")
    ran = t[..., 0] > 0
    print("a workgroup's items on its own cycle axis (from its first instruction): [B1] and tile end, mean (min .. max) over workgroups")
    for i in range(8):
        sel = ran[:, 0, i]
        if not sel.any():
            continue
        b1 = t[sel][:, :, i, 0].mean(dim=1) - life[sel][:, 1]
        en = t[sel][:, :, i, 7].amax(dim=1) - life[sel][:, 1]
        print(f"  slot {i}: {int(sel.sum()):4d} workgroups   [B1] {b1.mean().item():9.0f} ({b1.min().item():9.0f} .. {b1.max().item():9.0f})   end {en.mean().item():9.0f} ({en.min().item():9.0f} .. {en.max().item():9.0f})")
    tot = life[:, 3] - life[:, 1]
    print(f"  workgroup life {tot.mean().item():9.0f} cycles ({tot.min().item():9.0f} .. {tot.max().item():9.0f})")
    n_tiles = B * ((T + 127) // 128)
    full = [i for i in range(8) if (i + 1) * ncu <= n_tiles]          # tile slots every workgroup ran
    t = t[:, :, full]
    names = ["conv loop (48 k-steps)", "wait [B2]", "gate epilogue", "wait [B3]", "G pass"] + ([] if a.gate_only else ["projection MFMAs", "stream epilogue"])
    print(f"layer512 trace, {B} x {T}, {n_tiles} tiles on {ncu} workgroups, slots {full}; shader cycles per tile, mean over workgroups and waves (min .. max of the per-wave means)")
    for k, nm in enumerate(names):
        dlt = t[..., k + 1] - t[..., k]
        pw = dlt.mean(dim=(0, 2))
        print(f"  {nm:26s} {dlt.mean().item():9.0f}   ({pw.min().item():8.0f} .. {pw.max().item():8.0f})")
    gap = t[:, :, 1:, 0] - t[:, :, :-1, 7]
    print(f"  {'tile end -> next [B1]':26s} {gap.mean().item():9.0f}")
    whole = t[:, :, 1:, 0] - t[:, :, :-1, 0]
    print(f"  {'tile period':26s} {whole.mean().item():9.0f}   matrix time of a tile at 32 cycles per MFMA and two waves per SIMD: {(768 + (0 if a.gate_only else 128)) * 2 * 32 // (2 if a.one else 1)}")


if __name__ == "__main__":
    main()
