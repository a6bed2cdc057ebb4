"""Round 6: what the round-5 review asked to see run.

* `bench.py --gpus 8` end to end on ONE device (SS_BENCH_ONE_DEVICE=1: eight ranks over gloo, all on cuda:0): the torchrun spawn, the port
  choice, the rank-0-only legs and the final barrier at the world size of BASELINE configs[2]; the gathered mel batch must equal the
  single-process run of the same 8 x B utterances. (No 8-GPU node has been available to any round's driver: the 1 -> 8 CURVE stays unmeasured.)
* The `fp16q4` range guard: a checkpoint whose residual stream leaves the fixed fp4 activation scale is refused on its first forward.
* The unconditional non-finite flag of the fp16 modes.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from conftest import record_measurement  # noqa: E402
from stylesinger_amd import config, synth  # noqa: E402
from stylesinger_amd import lib as L  # noqa: E402
from stylesinger_amd.model import StyleSingerHIP  # noqa: E402


def _run_bench(args, env_extra=None, timeout=1500):
    env = dict(os.environ)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_eight_ranks_on_one_device_matches_single_process():
    small = ["--steps", "1", "--warmup", "0", "--batch", "1", "--frames", "128", "--diff-steps", "2", "--no-cpu-baseline", "--no-roofline", "--no-secondary",
             "--checksum"]
    eight = _run_bench(["--gpus", "8"] + small, {"SS_BENCH_ONE_DEVICE": "1"})
    assert eight["n_gpus"] == 8 and eight["config"]["global_batch"] == 8 and eight["dist"]["ranks"] == 8 and eight["scaling"] == "weak"
    assert eight["metric"].startswith("mel-frames/sec") and eight["value"] > 0 and eight["steps"] == 1
    one = _run_bench(["--gpus", "1", "--emulate-ranks", "8"] + small)
    assert one["n_gpus"] == 1 and len(one["checksum"]["mel_items"]) == 8
    assert eight["checksum"]["mel_items"] == one["checksum"]["mel_items"], (eight["checksum"], one["checksum"])


def _fwd(model, b, **kw):
    return model(b["txt_tokens"], mel2ph=b.get("mel2ph"), spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                 ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], **kw)


def _model(hp, sd):
    m = StyleSingerHIP(None, hparams=hp)
    m.load_state_dict(sd)
    m.eval().to("cuda:0")
    return m


def test_fp16q4_range_guard_refuses_a_stream_beyond_the_fp4_scale():
    """`fp16q4` converts the stream x + dstep to fp4 on the FIXED scale q_scale_gate = 2 (saturation at |a| > 12) and the gate outputs on 0.25. With
    the synthetic weights the stream stays inside (the forward succeeds, the guard's reduction reads < 1); with the mel denoiser's input projection
    scaled so that the stream exceeds 12, the FIRST forward of the plan raises and names fp16x2."""
    hp = config.make_hparams(dict(timesteps=3, K_step=3, f0_timesteps=3, mfma_precision="fp16q4"))
    sd = synth.synth_acoustic_state_dict(hp, 21)
    batch = {k: v.cuda() for k, v in synth.synth_batch(1, 256, 12, 40, hp, 21).items()}
    L.check(L.load().ss_set_tuning(b"q4_force", 1), "q4_force")
    try:
        out = _fwd(_model(hp, sd), batch)
        assert torch.isfinite(out["mel_out"]).all()
        bad = dict(sd)
        for k in ("postdiff.denoise_fn.input_projection.weight", "postdiff.denoise_fn.input_projection.bias"):
            bad[k] = sd[k] * 400.0
        with pytest.raises(L.StyleSingerHipError, match="fp16q4.*fp16x2"):
            _fwd(_model(hp, bad), batch)
        # the guard is armed for the first forward of a plan only, and never left armed
        g = torch.zeros(2, device="cuda:0", dtype=torch.int32)
        x = L.split_f16(torch.randn(1, 300, 256, device="cuda:0") * 10.0)
        assert int(g.sum()) == 0
    finally:
        L.check(L.load().ss_set_tuning(b"q4_force", 0), "q4_force")
        L.check(L.load().ss_set_q4_guard(None), "ss_set_q4_guard")


def test_nonfinite_flag_is_raised_on_every_forward_not_only_the_first():
    """ss_mel_denorm flags a non-finite valid frame in a device word on EVERY forward (ret['nonfinite']); `check_finite` raises from it. The first
    forward of a plan checks by itself; later forwards leave the check to the caller's next synchronisation point (infer.py does it)."""
    hp = config.make_hparams(dict(timesteps=2, K_step=2, f0_timesteps=2, mfma_precision="fp16x2"))
    sd = synth.synth_acoustic_state_dict(hp, 22)
    m = _model(hp, sd)
    batch = {k: v.cuda() for k, v in synth.synth_batch(1, 128, 8, 24, hp, 22).items()}
    ok = _fwd(m, batch)
    assert int(ok["nonfinite"].item()) == 0
    m.check_finite(ok)
    lib = L.load()
    x = torch.full((1, 128, 80), float("nan"), device="cuda:0")
    mel = torch.empty_like(x)
    flag = torch.zeros(1, device="cuda:0", dtype=torch.int32)
    lens = torch.tensor([100], device="cuda:0", dtype=torch.int32)
    smin, smax = torch.full((80,), -6.0, device="cuda:0"), torch.zeros(80, device="cuda:0")
    L.check(lib.ss_mel_denorm(L.ptr(x), L.ptr(smin), L.ptr(smax), L.ptr(mel), 1, 128, 80, L.ptr(lens), L.ptr(flag), L.stream_ptr()), "denorm")
    assert int(flag.item()) == 1
    with pytest.raises(L.StyleSingerHipError, match="non-finite"):
        m.check_finite(dict(nonfinite=flag))
    x[:, :100] = 0.5                                      # NaN only in the masked tail: not an error
    flag.zero_()
    L.check(lib.ss_mel_denorm(L.ptr(x), L.ptr(smin), L.ptr(smax), L.ptr(mel), 1, 128, 80, L.ptr(lens), L.ptr(flag), L.stream_ptr()), "denorm")
    assert int(flag.item()) == 0 and torch.isfinite(mel).all()


def test_f0_tracker_in_item_groups_equals_one_launch():
    """the tracker's float64 autocorrelation workspace (40 MB per 30 s item) is bounded by tracking a batch in groups of items
    (`f0track.WS_CAP_BYTES`); items are independent, so any grouping gives the same contours bit for bit."""
    import numpy as np
    from stylesinger_amd import f0track
    sr, hop = 48000, 256
    rng = np.random.default_rng(3)
    lens = [hop * 90, hop * 61, hop * 75, hop * 33]
    wav = np.zeros((4, max(lens)), dtype=np.float32)
    for b, n in enumerate(lens):
        t = np.arange(n) / sr
        f = 150.0 + 40.0 * b + 30.0 * np.sin(2 * np.pi * 1.5 * t)
        wav[b, :n] = (0.4 * np.sin(2 * np.pi * np.cumsum(f) / sr) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    x = torch.from_numpy(wav).cuda()
    n_out = max(lens) // hop + 1
    whole = f0track.track_f0_device(x, lens, n_out, sr=sr, hop_size=hop)
    cap = f0track.WS_CAP_BYTES
    try:
        f0track.WS_CAP_BYTES = 1          # one item per launch group
        single = f0track.track_f0_device(x, lens, n_out, sr=sr, hop_size=hop)
    finally:
        f0track.WS_CAP_BYTES = cap
    assert torch.equal(whole, single) and (whole > 0).sum().item() > 100


def test_skip_gemm_with_one_weight_term_equals_two_products_with_zero_lo_terms():
    """"fp16sd": the K = L C skip GEMM on `tile256s_kernel<STORE, true, ONE>` - one fp16 weight term, the lo plane of the weights neither fetched nor
    multiplied - against the two-product kernel on the same pack (whose lo terms are zero): the non-zero products enter every accumulator in the same
    order, so the outputs are equal bit for bit; and both match float64 of the one product."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(77)
    B, T, C, Lyr = 2, 1500, 256, 4
    K = Lyr * C
    lens = torch.tensor([T, T - 211], dtype=torch.int32, device=dev)
    x = torch.randn(B, T, K, generator=g).to(dev) * 0.5
    for b in range(B):
        x[b, lens[b]:] = 0
    A = L.split_f16(x)
    w = (torch.randn(C, K, 1, generator=g) / K ** 0.5).to(dev)
    Wp = L.pack_conv_weight(w)
    wk = (Wp * 256.0).to(torch.float16).float()
    W1 = L.split_f16(wk, scale=1.0)                 # hi = the fp16 term, lo = 0
    assert float(L.split_planes(W1)[1].abs().max()) == 0.0
    bias = torch.randn(C, generator=g).to(dev)
    outs = []
    for one in (True, False):
        S = torch.empty(B, T, C, device=dev)
        L.gemm_bf16(A, W1, B=B, T=T, K=K, taps=(0,), N=C, Np=W1.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S, bias=L.pack_bias(bias), split=2,
                    out_scale=1.0 / 256.0, gate256=True, one_product=one)
        outs.append(S)
    assert torch.equal(outs[0], outs[1])
    xh = x.to(torch.float16).double()
    ref = torch.relu(xh @ (wk[:C, :K].double().t() / 256.0) + bias.double()).float()
    for b in range(B):
        ref[b, lens[b]:] = 0
    err = (outs[0] - ref).abs().max().item()
    print(f"skip GEMM, one weight term: vs float64 {err:.2e}; bit-identical to the two-product kernel with zero lo terms")
    assert err <= 2e-5


@pytest.mark.parametrize("one", [True, False])
def test_skip_gemm_compact_operand_equals_the_pair_layout(one):
    """`a_compact`: the skip GEMM's A operand as [B, T, K] fp16 hi terms only (what ss_layer512 writes with g_compact) against the pair layout whose
    second plane the fp16 kernels never read: the same fragments reach the matrix cores, so the outputs are equal bit for bit. Rows past an item's
    length and the last partial tile go through the buffer bounds of the narrower rows."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(78)
    B, T, C, Lyr = 3, 1337, 256, 5
    K = Lyr * C
    lens = torch.tensor([T, T - 211, 3], dtype=torch.int32, device=dev)
    x = torch.randn(B, T, K, generator=g).to(dev) * 0.5
    for b in range(B):
        x[b, lens[b]:] = 0
    A = L.split_f16(x)
    Ah = L.split_planes(A)[0].to(torch.float16).contiguous()       # the compact operand
    A.view(B, T, -1, 64)[..., 32:] = 77.0                          # poison the second plane: nobody reads it
    w = (torch.randn(C, K, 1, generator=g) / K ** 0.5).to(dev)
    Wp = L.pack_conv_weight(w)
    Ws = L.split_f16(L.split_planes(L.split_f16(Wp, scale=256.0))[0], scale=1.0) if one else L.split_f16(Wp, scale=256.0)
    bias = torch.randn(C, generator=g).to(dev)
    outs = []
    for compact in (False, True):
        S = torch.full((B, T, C), 3.0, device=dev)
        L.gemm_bf16(Ah if compact else A, Ws, B=B, T=T, K=K, taps=(0,), N=C, Np=Ws.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S, bias=L.pack_bias(bias),
                    split=2, out_scale=1.0 / 256.0, gate256=True, one_product=one, a_compact=compact)
        outs.append(S)
    assert torch.equal(outs[0], outs[1])
    if one:   # one_product = 2: the weights without their zero plane too (fp16 [Np][K]) - again the same fragments
        Wc = L.split_planes(Ws)[0].to(torch.float16).contiguous()
        S = torch.full((B, T, C), 3.0, device=dev)
        L.gemm_bf16(Ah, Wc, B=B, T=T, K=K, taps=(0,), N=C, Np=Wc.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S, bias=L.pack_bias(bias),
                    split=2, out_scale=1.0 / 256.0, gate256=True, one_product=2, a_compact=True)
        assert torch.equal(S, outs[0]), f"compact one-term weights (64 channels per step: tile256s_kernel<.., DENSE>): max diff {(S - outs[0]).abs().max().item():.3e}"
        L.check(L.load().ss_set_tuning(b"skip_dense", 0), "skip_dense")
        try:
            S0 = torch.full((B, T, C), 3.0, device=dev)
            L.gemm_bf16(Ah, Wc, B=B, T=T, K=K, taps=(0,), N=C, Np=Wc.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S0, bias=L.pack_bias(bias),
                        split=2, out_scale=1.0 / 256.0, gate256=True, one_product=2, a_compact=True)
        finally:
            L.check(L.load().ss_set_tuning(b"skip_dense", 1), "skip_dense")
        assert torch.equal(S0, outs[0]), "compact one-term weights, 32-channel steps"
        S2 = torch.full((B, T, C), 3.0, device=dev)
        L.gemm_bf16(A, Wc, B=B, T=T, K=K, taps=(0,), N=C, Np=Wc.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S2, bias=L.pack_bias(bias),
                    split=2, out_scale=1.0 / 256.0, gate256=True, one_product=2)
        assert torch.equal(S2, outs[0]), "compact one-term weights with the pair-layout A operand"
    wh, wl = (t[:C, :K].double() for t in L.split_planes(Ws))
    ref = torch.relu(Ah.double() @ ((wh + wl).t() / 256.0) + bias.double()).float()
    for b in range(B):
        ref[b, lens[b]:] = 0
    err = (outs[1] - ref).abs().max().item()
    print(f"skip GEMM, compact A operand (one={one}): vs float64 {err:.2e}; bit-identical to the pair layout")
    assert err <= 2e-5
    # the generic entry refuses the flag for a launch its other kernels would take (too few tiles for the many-round kernel)
    with pytest.raises(L.StyleSingerHipError):
        L.gemm_bf16(Ah[:1, :300].contiguous(), Ws, B=1, T=300, K=K, taps=(0,), N=C, Np=Ws.shape[0], epi=L.HEPI_STORE, out=torch.empty(1, 300, C, device=dev),
                    bias=L.pack_bias(bias), split=2, out_scale=1.0 / 256.0, a_compact=True)


def test_fp16sd_fused_path_under_the_ddim_sampler():
    """The weight sets and the fp16 addend sets of "fp16sd" are indexed by the evaluation number of the sampling loop, which every sampler has to hand
    to the stack (`g_wset_eval`): here the strided sampler. (1) eta = 1 over all 100 steps IS the reference's ancestral chain (tests/test_gpu_round4.py):
    the fused one-product path (knob layer512 = 2 forces the one item onto it) against the REAL reference's 100-step golden; (2) eta = 0 over 10 of the
    steps - only 10 of the 32 sets take part, the least favourable case for the noise shaping - against the fp32 path's own 10-step result."""
    from oracle import harness
    case = harness.load_case("acoustic_t64_s100")
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    K = hp["K_step"]
    noise = synth.draw_acoustic_noise(synth.NoiseTape(meta["tape_seed"]), meta["B"], meta["T"], meta["steps_f0"], meta["steps_mel"])
    b = {k: v.cuda() for k, v in batch.items()}
    exact = _model(hp, sd)
    ref10 = _fwd(exact, b, noise=noise, sampler="ddim", ddim_steps=10)["mel_out"]
    L.check(L.load().ss_set_tuning(b"layer512", 2), "layer512")
    try:
        m = _model(dict(hp, mfma_precision="fp16sd"), sd)
        assert m.sd and m.sd_e_sets == 8
        got = _fwd(m, b, noise=noise, sampler="ddim", ddim_steps=K, eta=1.0)
        got10 = _fwd(m, b, noise=noise, sampler="ddim", ddim_steps=10)["mel_out"]
        torch.cuda.synchronize()
    finally:
        L.check(L.load().ss_set_tuning(b"layer512", 1), "layer512")
    d = (got["mel_out"].cpu() - gold["mel_out"]).abs()
    uv = int(((got["pitch_pred"][..., 1].cpu() > 0) != (gold["pitch_pred"][..., 1] > 0)).sum())
    d10 = (got10 - ref10).abs()
    print(f"fp16sd on ss_layer512 under the strided sampler: eta = 1, 100 steps vs the real reference {d.mean().item():.3e} (max {d.max().item():.3e}), voicing flips {uv}; "
          f"eta = 0, 10 steps vs the fp32 path {d10.mean().item():.3e} (max {d10.max().item():.3e})")
    record_measurement("fp16sd_ddim_eta1_100steps_vs_reference_golden_t64", mel_l1=d.mean().item(), mel_max=d.max().item(), voicing_flips=uv,
                       ddim10_vs_fp32_l1=d10.mean().item())
    assert uv == 0 and d.mean().item() <= 6e-5, d.mean().item()
    assert d10.mean().item() <= 3e-4, d10.mean().item()
