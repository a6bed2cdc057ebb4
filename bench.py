#!/usr/bin/env python
"""Benchmark of the StyleSinger inference hot path on MI355X (contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path (phoneme encoder -> RSA -> two f0 diffusions -> FFT decoder ->
100-step shallow mel diffusion -> [RCCL all_gather of mels when N>1] -> HiFi-GAN-NSF) over one batch of
synthetic utterances per GPU, inputs resident in HBM, device Philox noise.  Workload = BASELINE.json
configs[1]: batch 8 x 8 s (T=1500 frames, 48 kHz / hop 256), 100 diffusion steps, fp32; N>1 is
configs[2] (8 utterances per GPU, weak scaling).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MEL_FLOP_PER_FRAME_STEP = 26.43e6   # SURVEY.md §8(d), algorithmic (cond-proj counted)
F0_FLOP_PER_FRAME_STEP = 7.94e6     # per network
VOC_FLOP_PER_FRAME = 614.6e6
REST_FLOP_PER_FRAME = 45e6
PEAK_FP32_MFMA = 157.3e12           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1500, help="mel frames per utterance (1500 = 8 s)")
    ap.add_argument("--diff-steps", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", type=int, default=int(os.environ.get("SS_BENCH_PIPELINE", "0")),
                    help="1 = vocode batch i on a second stream while the diffusion loops of batch i+1 run (all K batches still "
                         "finish inside the timed region)")
    ap.add_argument("--cpu-frames", type=int, default=3000, help="frames of the single utterance the CPU oracle is timed on (~20 s of CPU work)")
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="threads for the CPU oracle; 16 is the fastest setting on the 2x64-core EPYC GPU-box host "
                         "(tools/cpu_threads.py: 8:0.38s 16:0.35s 32:0.78s 64:2.0s 128:5.0s per 10 steps)")
    return ap.parse_args()


def kernel_roofline(infer, B, T, iters=20):
    """Dominant kernel = dilated-conv+gate GEMM of the mel denoiser (2000 launches per step).
    Timed live with events on the launch stream; algorithmic flops = 2*frames*(3*256)*512 per launch."""
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

    wino = infer.model.use_wino

    def launch(l):
        d = 1 << (l % 4)
        kw = dict(B=B, T=T, Cin=C, N=C, Np=2 * C, Kp=C, lens=lens, a_bias=dstep[0, l], epi=L.EPI_GATE, E=E[:, :, l * 2 * C:],
                  lde=Lyr * 2 * C, e_bs=T * Lyr * 2 * C, ldc=C, mask_rows=True)
        if wino:  # what the loop driver launches: Winograd F(2,3) form of the same layer
            L.wino_gate(X, packs[f"w_dil_wino.{l}"], G, dilation=d, **kw)
        else:
            L.conv_gemm(X, packs[f"w_dil.{l}"], G, taps=(-d, 0, d), **kw)
    for l in range(Lyr):
        launch(l)
    torch.cuda.synchronize()
    # back-to-back launches: one pass over the 20 layers captured in a hipGraph (the Python ctypes call costs about as much
    # host time as the kernel runs, so a plain Python loop would time the host), replayed `iters` times between two events
    # on the capture stream.
    graph = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(graph, stream=st):
            for l in range(Lyr):
                launch(l)
        graph.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(iters):
            graph.replay()
        e1.record(st)
        st.synchronize()
    torch.cuda.current_stream().wait_stream(st)
    sec = e0.elapsed_time(e1) * 1e-3 / (iters * Lyr)
    flops = 2.0 * B * T * (3 * C) * (2 * C)
    # traffic: FETCH_SIZE + WRITE_SIZE of this launch from separate rocprofv3 --pmc passes (profiles/r01_pmc_mel_gate.md):
    # 27.9 MB fetched (uncorrected; wide reads are tallied at 1/2 on gfx950) + 12.3 MB written; algorithmic bytes 50.7 MB.
    traffic = 40.2e6 if (B * T == 12000) else None
    name = "wino_gate_kernel (Winograd F(2,3)" if wino else "conv_gemm_kernel<64,128,2,2,GATE> (direct"
    return dict(bound="mfma", kernel=name + " mel dilated conv k=3, 256->512, + gate)",
                achieved=flops / sec / 1e12, peak=PEAK_FP32_MFMA / 1e12, unit="TFLOP/s", frac=flops / sec / PEAK_FP32_MFMA,
                traffic=traffic, us_per_launch=sec * 1e6, flops_per_launch=flops, launches_per_step=2000,
                algorithmic_bytes_per_launch=50.7e6)


def cpu_baseline(hp, frames, threads):
    """The oracle (CPU restatement of the reference) timed on this box's host cores: a reported baseline only."""
    from oracle import restatement as R
    from stylesinger_amd import config, synth
    torch.manual_seed(0)
    torch.set_num_threads(max(1, min(threads, os.cpu_count() or threads)))
    sd = synth.synth_acoustic_state_dict(hp, 1234)
    cfg = config.make_vocoder_config()
    vsd = synth.synth_vocoder_state_dict(cfg, 1234)
    Tp = max(2, frames * 28 // 1500)
    batch = synth.synth_batch(1, frames, Tp, frames, hp, 1234)
    tape = synth.NoiseTape(1)
    with torch.no_grad():
        t0 = time.time()
        ret = R.acoustic_forward(sd, hp, batch, tape, mel2ph=batch["mel2ph"])
        mel = ret["mel_out"].clamp(hp["mel_vmin"], hp["mel_vmax"])
        R.hifigan_forward(vsd, cfg, mel, ret["f0_denorm"], tape)
        dt = time.time() - t0
    return dict(value=frames / dt, unit="mel-frames/s", cores=torch.get_num_threads(), host_logical_cpus=os.cpu_count(), kind="port",
                sample=f"B=1, T={frames} frames, {hp['K_step']}+2x{hp['f0_timesteps']} diffusion steps + HiFi-GAN-NSF, fp32, "
                       f"oracle/restatement.py (torch CPU, {torch.get_num_threads()} threads), {dt:.1f} s")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("SS_BENCH_ONE_DEVICE"):  # orchestration smoke test on a 1-GPU box: all ranks on cuda:0 over gloo
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from stylesinger_amd import config, dist as ssd, synth
    from stylesinger_amd.infer import StyleSingerInfer
    hp = config.make_hparams(dict(timesteps=args.diff_steps, K_step=args.diff_steps, f0_timesteps=args.diff_steps))
    B, T = args.batch, args.frames
    Tp, Tr = max(2, T * 28 // 1500), T
    sd = synth.synth_acoustic_state_dict(hp, 1234)
    vsd = synth.synth_vocoder_state_dict(None, 1234)
    infer = StyleSingerInfer(hp, device=dev, model_state=sd, vocoder_state=vsd)
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 1234, first_index=rank * B)
    batch = {k: v.to(dev) for k, v in batch.items()}

    voc_stream = torch.cuda.Stream(device=dev) if args.pipeline else None

    def step(i):
        res = infer.infer_batch(batch, seed=1234 + 7919 * i + rank, vocode=False)
        mel, f0, lens = ssd.gather_mels(res["mel"], res["f0"], res["lens"])
        own = slice(rank * B, (rank + 1) * B) if world > 1 else slice(None)
        if voc_stream is None:
            return infer.vocode(mel[own], f0[own], lens[own], seed=4321 + i), lens
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(voc_stream):
            voc_stream.wait_event(ready)
            for t in (mel, f0, lens):
                t.record_stream(voc_stream)
            wav = infer.vocode(mel[own], f0[own], lens[own], seed=4321 + i)
        return wav, lens

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(-1 - i)
    sync()
    t0 = time.perf_counter()
    frames_local = 0
    for i in range(args.steps):
        wav, lens = step(i)
        frames_local += B * T
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    assert torch.isfinite(wav).all(), "non-finite waveform"
    total_frames = frames_local * world
    value = total_frames / dt
    flop_per_frame = (MEL_FLOP_PER_FRAME_STEP + 2 * F0_FLOP_PER_FRAME_STEP) * args.diff_steps + VOC_FLOP_PER_FRAME + REST_FLOP_PER_FRAME
    # what the kernels actually execute: the step-invariant conditioner projections are hoisted out of the loops
    # (mel 5.24 -> once, f0 1.97 -> once per net; SURVEY.md §8d "report both")
    exec_flop_per_frame = ((MEL_FLOP_PER_FRAME_STEP - 5.24e6) + 2 * (F0_FLOP_PER_FRAME_STEP - 1.97e6)) * args.diff_steps \
        + 5.24e6 + 2 * 1.97e6 + VOC_FLOP_PER_FRAME + REST_FLOP_PER_FRAME

    if rank == 0:
        out = {
            "metric": "mel-frames/sec (end-to-end infer incl. vocoder)", "value": value, "unit": "mel-frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if infer.model.bf16 else "f32", "data": "synthetic (seeded random weights + inputs; no checkpoint ships)",
            "config": {"workload": f"batch={B}x{T / 187.5:.1f}s utterances per GPU (T={T} frames, Tp={Tp}, Tr={Tr}), "
                                   f"{args.diff_steps} mel + 2x{args.diff_steps} f0 diffusion steps + HiFi-GAN-NSF, fp32",
                       "global_batch": B * world, "frames_per_utterance": T, "parallelism": f"dp{world}",
                       "diffusion_loops": "hipGraph replay" if infer.model._want_graphs(B, T) else "eager launches",
                       "mfma_precision": "bf16" if infer.model.bf16 else "fp32",
                       "step_overlap": "vocoder(i) on a 2nd stream under acoustic(i+1)" if args.pipeline else "none (one stream)",
                       "algorithmic_gflop_per_frame": flop_per_frame / 1e9,
                       "executed_gflop_per_frame": exec_flop_per_frame / 1e9,
                       "e2e_fraction_of_fp32_mfma_peak_algorithmic": value / world * flop_per_frame / PEAK_FP32_MFMA,
                       "e2e_fraction_of_fp32_mfma_peak_executed": value / world * exec_flop_per_frame / PEAK_FP32_MFMA},
        }
        out["roofline"] = kernel_roofline(infer, B, T)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hp, args.cpu_frames, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
