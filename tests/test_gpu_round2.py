"""Round-2 GPU tests: BASELINE-config-sized parity against the oracle run on this box, the hipGraph/plan cache, weight reloads,
seeding, the emotion encoder, and the multi-process bench step."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import harness  # noqa: E402
from oracle import restatement as R  # noqa: E402
from stylesinger_amd import config, synth  # noqa: E402
from stylesinger_amd.model import StyleSingerHIP  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# mel L1 of the bf16-operand 1000-step chain on `acoustic_t32_mel1000` vs the real fp32 reference, measured on MI355X (profiles/r03_parity.json)
BF16_CHAIN_L1_MEASURED = float(os.environ.get("SS_BF16_CHAIN_L1", "2.5e-3"))


class ListTape:
    """Replays a fixed list of noise tensors in order (the oracle draws in the reference's order)."""

    def __init__(self, items):
        self.items = list(items)

    def randn(self, *shape):
        t = self.items.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), shape)
        return t.clone()

    rand = randn


def item_tape(noise, i, steps_f0, steps_mel):
    """The draws item i of a batched tape would see in a B=1 run, in the reference's order (synth.draw_acoustic_noise)."""
    out = []
    for net in ("f0_a", "f0_b"):
        n = noise[net]
        out += [n["u_init"][i:i + 1], n["z0"][i:i + 1]]
        for s in reversed(range(steps_f0)):
            out += [n["z_steps"][s, i:i + 1], n["u_steps"][s, i:i + 1]]
    out.append(noise["mel"]["z_q"][i:i + 1])
    out += [noise["mel"]["z_steps"][s, i:i + 1] for s in reversed(range(steps_mel))]
    return ListTape(out)


def _model(hp, seed, dev="cuda:0"):
    m = StyleSingerHIP(None, hparams=hp)
    m.load_state_dict(synth.synth_acoustic_state_dict(hp, seed))
    m.eval().to(dev)
    return m


def _fwd(model, b, **kw):
    return model(b["txt_tokens"], mel2ph=b.get("mel2ph"), spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                 ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], **kw)


def test_c2_batch_item_matches_oracle_at_full_size_and_100_steps():
    """BASELINE configs[1] as specified: the B=8 x T=1500 (Tp=28, Tr=1500) batch with the FULL 100 + 2x100 step chains, shared
    noise tape; item 3 of the HIP batch vs the oracle's B=1 run of the same utterance on this box (the reference only runs
    B=1). Integers exact; mel L1 <= 1e-5 (north_star: 1e-4)."""
    S = 100
    hp = config.make_hparams(dict(timesteps=S, K_step=S, f0_timesteps=S))
    B, T, Tp, Tr = 8, 1500, 28, 1500
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 1234)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(77), B, T, S, S)
    model = _model(hp, 1234)
    full = _fwd(model, {k: v.cuda() for k, v in batch.items()}, noise=noise)
    torch.cuda.synchronize()
    i = 3
    sd = synth.synth_acoustic_state_dict(hp, 1234)
    one = {k: v[i:i + 1] for k, v in batch.items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.acoustic_forward(sd, hp, one, item_tape(noise, i, S, S), mel2ph=one["mel2ph"])
    assert torch.equal(full["rq_codes"][i].cpu(), ref["rq_codes"][0])
    flips = (full["uv_a"][i].cpu().long() != ref["uv_a"][0]).sum().item() + (full["uv_b"][i].cpu().long() != ref["uv_b"][0]).sum().item()
    cflips = (full["pitch_coarse"][i].cpu() != ref["pitch_coarse"][0]).sum().item()
    l1 = (full["mel_out"][i].cpu() - ref["mel_out"][0]).abs().mean().item()
    mx = (full["mel_out"][i].cpu() - ref["mel_out"][0]).abs().max().item()
    f0e = (full["f0_denorm"][i].cpu() - ref["f0_denorm"][0]).abs().max().item()
    print(f"C2 item {i} (T=1500, 100+100+100 steps): mel L1 {l1:.3e} max {mx:.3e}; voicing flips {flips}; coarse-pitch flips {cflips}/1500; "
          f"f0 max err {f0e:.3e} Hz")
    assert flips == 0
    assert cflips <= 1          # a 1e-3 Hz difference can move one frame across a coarse-pitch bin edge
    # always bound the mel: a flipped frame selects another pitch-embedding row, which the decoder's conv-FFN spreads over
    # +-4 frames per layer -> compare away from +-20 frames around a flip (as test_long_form_30s_sequence_matches_oracle does)
    keep = torch.ones(T, dtype=torch.bool)
    for tt in (full["pitch_coarse"][i].cpu() != ref["pitch_coarse"][0]).nonzero().flatten().tolist():
        keep[max(0, tt - 20):tt + 21] = False
    dm = (full["mel_out"][i].cpu() - ref["mel_out"][0]).abs()[keep]
    l1k, mxk = dm.mean().item(), dm.max().item()
    from conftest import record_measurement
    record_measurement("c2_item3_t1500_100steps_vs_oracle", mel_l1=l1, mel_max=mx, mel_l1_away_from_flips=l1k, voicing_flips=flips,
                       coarse_flips=cflips, f0_max_err_hz=f0e)
    assert l1k <= 1e-5 and mxk <= 1e-3, (l1k, mxk, cflips)
    # the same item in the split-operand precision modes against the same oracle run (round-4 hardening: these modes were only checked on the
    # two small goldens): "bf16x3" = F(4,3) gates from 3 bf16 terms per operand, 6 products; "bf16x2" = all hidden GEMMs from (hi, mid) pairs
    for mode in ("bf16x3", "bf16x2"):
        m2 = StyleSingerHIP(None, hparams=dict(hp, mfma_precision=mode))
        m2.load_state_dict(sd)
        m2.eval().to("cuda:0")
        got = _fwd(m2, {k: v[i:i + 1].cuda() for k, v in batch.items()}, noise={k: ({kk: vv[:, i:i + 1] if kk in ("z_steps", "u_steps") else vv[i:i + 1] for kk, vv in v.items()}) for k, v in noise.items()})
        fl = (got["uv_a"][0].cpu().long() != ref["uv_a"][0]).sum().item() + (got["uv_b"][0].cpu().long() != ref["uv_b"][0]).sum().item()
        d2 = (got["mel_out"][0].cpu() - ref["mel_out"][0]).abs()
        print(f"C2 item {i} in {mode} mode: mel L1 {d2.mean().item():.3e} max {d2.max().item():.3e}; voicing flips {fl}")
        record_measurement(f"c2_item3_t1500_100steps_{mode}_vs_oracle", mel_l1=d2.mean().item(), mel_max=d2.max().item(), voicing_flips=fl)
        assert fl == 0 and d2.mean().item() <= 2e-5, (mode, d2.mean().item())
        del m2


def test_winograd_f43_f23_and_direct_forms_agree_through_the_whole_path(monkeypatch):
    """The dilated conv of both denoisers as Winograd F(4,3) (default), F(2,3) (SS_WINO_M=2) and direct (SS_WINO=0): same noise
    tape, 30 + 2x30 steps, ragged batch. Integer outputs equal; mel within 1e-5 L1 of each other."""
    S = 30
    hp = config.make_hparams(dict(timesteps=S, K_step=S, f0_timesteps=S))
    B, T = 3, 200
    batch = {k: v.cuda() for k, v in synth.synth_batch(B, T, 8, 120, hp, 4321).items()}
    noise = synth.draw_acoustic_noise(synth.NoiseTape(5), B, T, S, S)
    outs = {}
    for name, env in (("f43", {"SS_WINO_M": "4"}), ("f23", {"SS_WINO_M": "2"}), ("direct", {"SS_WINO": "0"})):
        for k in ("SS_WINO_M", "SS_WINO"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m = _model(hp, 99)
        assert m.use_wino == (name != "direct") and (name == "direct" or m.wino_m == int(env["SS_WINO_M"]))
        outs[name] = _fwd(m, batch, noise=noise)
        torch.cuda.synchronize()
    for name in ("f23", "direct"):
        assert torch.equal(outs["f43"]["rq_codes"], outs[name]["rq_codes"])
        assert torch.equal(outs["f43"]["uv_a"], outs[name]["uv_a"]) and torch.equal(outs["f43"]["uv_b"], outs[name]["uv_b"])
        l1 = (outs["f43"]["mel_out"] - outs[name]["mel_out"]).abs().mean().item()
        print(f"mel L1 F(4,3) vs {name}: {l1:.3e}")
        assert l1 <= 1e-5, (name, l1)


def test_bucketed_graph_cache_20_random_lengths():
    """20 utterance lengths through forward(): frames are padded to the 64-frame bucket, so only 4 plans exist; a shape is
    captured on its SECOND use (<= 4 captures per loop); every result equals the un-bucketed, eager run bit for bit."""
    hp = config.make_hparams(dict(timesteps=3, K_step=3, f0_timesteps=3))
    model = _model(hp, 7)
    plain = _model(hp, 7)
    plain.t_bucket, plain.use_graphs = 1, "off"
    g = torch.Generator().manual_seed(5)
    lengths = [int(x) for x in torch.randint(130, 380, (20,), generator=g)]
    assert len({model.bucket_frames(t) for t in lengths}) <= 4
    for n, T in enumerate(lengths):
        b = {k: v.cuda() for k, v in synth.synth_batch(2, T, 6, 64, hp, 100 + n).items()}
        a = _fwd(model, b, seed=500 + n)
        e = _fwd(plain, b, seed=500 + n)
        assert a["mel_out"].shape == (2, T, 80) and a["f0_denorm"].shape == (2, T)
        assert torch.equal(a["mel_out"], e["mel_out"]), (n, T)
        assert torch.equal(a["uv_a"], e["uv_a"]) and torch.equal(a["pitch_coarse"], e["pitch_coarse"])
    assert len(model._plans) <= 4
    assert model.n_captures <= 2 * 4      # f0 loop + mel loop per bucket
    assert plain.n_captures == 0


def test_reload_after_capture_uses_the_new_weights():
    """captured hipGraphs hold packed-weight pointers; load_state_dict() must drop them."""
    hp = config.make_hparams(dict(timesteps=3, K_step=3, f0_timesteps=3))
    model = _model(hp, 3)
    model.use_graphs = "on"
    b = {k: v.cuda() for k, v in synth.synth_batch(2, 96, 6, 70, hp, 3).items()}
    first = _fwd(model, b, seed=9)["mel_out"].clone()
    again = _fwd(model, b, seed=9)["mel_out"].clone()
    assert torch.equal(first, again)
    model.load_state_dict(synth.synth_acoustic_state_dict(hp, 4))
    new = _fwd(model, b, seed=9)["mel_out"].clone()
    fresh = _model(hp, 4)
    fresh.use_graphs = "off"
    want = _fwd(fresh, b, seed=9)["mel_out"]
    assert torch.equal(new, want)
    assert (new - first).abs().max().item() > 1e-3


def test_graph_and_eager_agree_for_any_seed_history():
    """the noise for `seed` must not depend on which seed a graph was captured with."""
    hp = config.make_hparams(dict(timesteps=4, K_step=4, f0_timesteps=4))
    b = {k: v.cuda() for k, v in synth.synth_batch(2, 128, 6, 70, hp, 3).items()}
    gm, em = _model(hp, 3), _model(hp, 3)
    gm.use_graphs, em.use_graphs = "on", "off"
    _fwd(gm, b, seed=77)                       # captures with seed 77
    for s in (78, 5, 77):
        assert torch.equal(_fwd(gm, b, seed=s)["mel_out"], _fwd(em, b, seed=s)["mel_out"]), s
    ddim_g = _fwd(gm, b, seed=11, sampler="ddim", ddim_steps=2)["mel_out"]
    ddim_e = _fwd(em, b, seed=11, sampler="ddim", ddim_steps=2)["mel_out"]
    assert torch.equal(ddim_g, ddim_e)


def test_emotion_encoder_matches_reference_golden():
    """3 x LSTM-256 + mean/L2 (data_gen/tts/emotion/model.py:11-78, inference.py:139-151) vs the real reference's outputs."""
    from stylesinger_amd.emotion import EmotionEncoderHIP
    case = harness.load_case("emotion_p5")
    esd = synth.synth_emotion_state_dict(case["meta"]["seed"])
    enc = EmotionEncoderHIP(esd, device="cuda:0")
    frames = synth.synth_emotion_frames(case["meta"]["n_partials"], seed=case["meta"]["seed"])
    embed, partial = enc.embed_partials(frames)
    fwd = enc(frames)
    e_p = (partial.cpu() - case["out"]["partial_embeds"]).abs().max().item()
    e_e = (embed.cpu() - case["out"]["embed"]).abs().max().item()
    e_f = (fwd.cpu() - case["out"]["forward_embeds"]).abs().max().item()
    print(f"emotion encoder: partial embeds max err {e_p:.3e}, utterance embed {e_e:.3e}, forward() embeds {e_f:.3e}")
    assert e_p <= 1e-5 and e_e <= 1e-5 and e_f <= 1e-5
    # a longer utterance through the slicing path equals the batch of its partials
    mel = synth.synth_emotion_frames(1, n_frames=800, seed=9)[0]
    from stylesinger_amd import emotion
    _, sl = emotion.compute_partial_slices(128000)
    want = enc.embed_partials(torch.stack([mel[s] for s in sl]))[0]
    got = enc.embed_utterance_frames(mel, n_samples=128000)
    assert torch.equal(want, got)


def _run_bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_on_one_device_matches_single_process():
    """`python bench.py --gpus 2` must really run 2 ranks (here both on cuda:0 over gloo, SS_BENCH_ONE_DEVICE=1): the REAL step
    (acoustic model -> all_gather of mels -> own shard -> vocoder) per rank; the gathered mel batch must equal what a single
    process computes for the same 2 x B utterances (per-utterance mel checksums travel in the JSON line)."""
    small = ["--steps", "1", "--warmup", "0", "--batch", "2", "--frames", "192", "--diff-steps", "4", "--no-cpu-baseline", "--no-roofline",
             "--checksum"]
    two = _run_bench(["--gpus", "2"] + small, {"SS_BENCH_ONE_DEVICE": "1"})
    assert two["n_gpus"] == 2 and two["config"]["global_batch"] == 4 and two["dist"]["ranks"] == 2
    one = _run_bench(["--gpus", "1", "--emulate-ranks", "2"] + small)
    assert one["n_gpus"] == 1 and len(one["checksum"]["mel_items"]) == 4
    assert two["checksum"]["mel_items"] == one["checksum"]["mel_items"], (two["checksum"], one["checksum"])
    # three batches in flight on three HIP streams, four steps: the collectives of in-flight batches are issued from ONE
    # communication stream (dist.comm_stream), so both ranks see them in the same order and the gathered batch is unchanged
    multi = ["--steps", "4", "--warmup", "0", "--streams", "3", "--batch", "2", "--frames", "192", "--diff-steps", "4", "--no-cpu-baseline",
             "--no-roofline", "--checksum"]
    two3 = _run_bench(["--gpus", "2"] + multi, {"SS_BENCH_ONE_DEVICE": "1"})
    one3 = _run_bench(["--gpus", "1", "--emulate-ranks", "2"] + multi)
    assert two3["n_gpus"] == 2 and "3 HIP streams" in two3["config"]["step_overlap"]
    assert two3["checksum"]["mel_items"] == one3["checksum"]["mel_items"], (two3["checksum"], one3["checksum"])


def _gemm_bf16(A, Wp, **kw):
    from stylesinger_amd import lib as L
    L.gemm_bf16(A, Wp, **kw)


@pytest.mark.parametrize("T,K", [(200, 256), (333, 192), (5600, 256)])   # the last shape takes the 128x128 tile (C4-size path)
def test_gemm_bf16_kernel_matches_torch_on_bf16_operands(T, K):
    """ss_gemm_bf16 (bf16 operands in HBM, fp32 accumulate) vs torch fp32 math on the SAME bf16-rounded operands: STORE with a
    3-tap dilated conv, GATE with the fp32 addend, RESX with the next layer's bf16 operand. Products of bf16 numbers are exact in
    fp32, so only the summation order differs (tolerance 2e-5 relative to sqrt(K)-sized sums)."""
    from stylesinger_amd import lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(T + K)
    B, C = 3, K
    lens = torch.tensor([T, T - 37, 5], dtype=torch.int32, device=dev)
    x = torch.randn(B, T, C, generator=g).to(dev)
    for b in range(B):
        x[b, lens[b]:] = 0
    xh = L.to_bf16(x)                                    # the operand as it would sit in HBM
    xr = xh.float()
    d = 2
    w = (torch.randn(2 * C, C, 3, generator=g) / (3 * C) ** 0.5).to(dev)
    Wp = L.pack_conv_weight(w, interleave_half=C)        # [2C][3*Kp] gate-interleaved, fp32
    Wh = L.to_bf16(Wp)
    wr = w.to(torch.bfloat16).float()
    # reference conv on rounded operands (zero padding outside [0, len))
    y_ref = torch.nn.functional.conv1d(xr.transpose(1, 2), wr, padding=d, dilation=d).transpose(1, 2)   # [B,T,2C]
    for b in range(B):   # frames near the end of a shorter item must see zeros beyond len: xr is already zero there
        pass
    E = torch.randn(B, T, 2 * C, generator=g).to(dev) * 0.5
    # E in packed column order: block 2p = sigmoid half rows [32p, 32p+32), block 2p+1 = tanh half
    Ep = torch.empty_like(E)
    for p in range(C // 32):
        Ep[..., 64 * p:64 * p + 32] = E[..., 32 * p:32 * p + 32]
        Ep[..., 64 * p + 32:64 * p + 64] = E[..., C + 32 * p:C + 32 * p + 32]
    G = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
    _gemm_bf16(xh, Wh, B=B, T=T, K=C, taps=(-d, 0, d), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens, E=Ep, lde=2 * C, out=G)
    z = y_ref + E
    g_ref = torch.sigmoid(z[..., :C]) * torch.tanh(z[..., C:])
    for b in range(B):
        g_ref[b, lens[b]:] = 0
    assert (G.float() - g_ref).abs().max().item() <= 4e-3 + 1e-6            # one bf16 ulp of values in (-1, 1)
    assert (G.float() - g_ref.to(torch.bfloat16).float()).abs().mean().item() <= 2e-4   # almost always the same bf16
    # RESX: x <- (x + G . Wo^T + b) / sqrt(2); Y = bf16(x + next_bias)
    wo = (torch.randn(C, C, 1, generator=g) / C ** 0.5).to(dev)
    Woh = L.to_bf16(L.pack_conv_weight(wo))
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    nb = torch.randn(C, generator=g).to(dev)
    X = torch.randn(B, T, C, generator=g).to(dev)
    X0 = X.clone()
    Y = torch.empty(B, T, C, device=dev, dtype=torch.bfloat16)
    _gemm_bf16(G, Woh, B=B, T=T, K=C, taps=(0,), N=C, Np=Woh.shape[0], epi=L.HEPI_RESX, lens=lens, bias=L.pack_bias(bo), X=X,
               post_scale=0.5 ** 0.5, next_bias=nb, Y=Y)
    x_ref = (X0 + (G.float() @ wo[:, :, 0].to(torch.bfloat16).float().t() + bo)) * (0.5 ** 0.5)
    for b in range(B):
        x_ref[b, lens[b]:] = 0
    assert (X - x_ref).abs().max().item() <= 2e-5 * (C ** 0.5)
    y_ref2 = (x_ref + nb)
    for b in range(B):
        y_ref2[b, lens[b]:] = 0
    assert (Y.float() - y_ref2).abs().max().item() <= 0.04 and (Y.float() - y_ref2.to(torch.bfloat16).float()).abs().mean().item() <= 1e-3
    # STORE with ReLU on a K = 2*C, 1-tap GEMM
    A2 = L.to_bf16(torch.randn(B, T, 2 * C, generator=g).to(dev))
    w2 = (torch.randn(C, 2 * C, 1, generator=g) / (2 * C) ** 0.5).to(dev)
    S = torch.empty(B, T, C, device=dev)
    if (2 * C) % 64 == 0:
        _gemm_bf16(A2, L.to_bf16(L.pack_conv_weight(w2)), B=B, T=T, K=2 * C, taps=(0,), N=C, Np=L.round_up(C, 32), epi=L.HEPI_STORE, lens=lens,
                   bias=L.pack_bias(bo), act=L.ACT_RELU, out=S)
        s_ref = torch.relu(A2.float() @ w2[:, :, 0].to(torch.bfloat16).float().t() + bo)
        for b in range(B):
            s_ref[b, lens[b]:] = 0
        assert (S - s_ref).abs().max().item() <= 2e-5 * ((2 * C) ** 0.5)


def test_batches_in_flight_equal_sequential_runs():
    """StyleSingerInfer.infer_batches (3 batches in flight on 3 streams, separate plan slots) == infer_batch one after the other."""
    from stylesinger_amd.infer import StyleSingerInfer
    hp = config.make_hparams(dict(timesteps=4, K_step=4, f0_timesteps=4))
    inf = StyleSingerInfer(hp, device="cuda:0", model_state=synth.synth_acoustic_state_dict(hp, 9), vocoder_state=synth.synth_vocoder_state_dict(None, 9))
    batches = [{k: v.cuda() for k, v in synth.synth_batch(2, 100 + 30 * (i % 2), 6, 64, hp, 300 + i).items()} for i in range(5)]
    got = [r for r in inf.infer_batches(batches, in_flight=3, seed=40)]
    torch.cuda.synchronize()
    assert len(got) == 5
    for i, b in enumerate(batches):
        want = inf.infer_batch(b, seed=40 + i)
        assert torch.equal(got[i]["mel"], want["mel"]) and torch.equal(got[i]["wav"], want["wav"]), i
    # a LAZY producer (round 4): the generator is consumed one batch at a time - at most in_flight batches have been pulled
    # when the first result comes back - and the results are the same bits
    pulled = []

    def producer():
        for i, b in enumerate(batches):
            pulled.append(i)
            yield b
    it = inf.infer_batches(producer(), in_flight=2, seed=40)
    first = next(it)
    assert len(pulled) <= 3, pulled          # two in flight + the one whose arrival released the first result
    rest = [first] + list(it)
    torch.cuda.synchronize()
    assert len(rest) == 5 and all(torch.equal(rest[i]["mel"], got[i]["mel"]) for i in range(5))


def test_bf16_mode_on_the_1000_step_golden_reports_its_distance_to_the_fp32_reference():
    """BASELINE config 4 as specified (1000 mel steps) in bf16-operand mode, on the real-reference golden `acoustic_t32_mel1000`.
    No reference arithmetic exists for this mode (parity unpinned by construction), so distances to the fp32 reference are REPORTED:
    (a) the mel chain alone - the bf16 model's 1000-step sampler on the fp32 path's own coarse mel / condition and the same tape -
    against the real reference's mel (loosely bounded: a broken chain, coefficients up to 3e6, would be off by O(1));
    (b) the whole path in bf16 mode (the f0 samplers run bf16 too, so discrete voicing decisions may differ: informational)."""
    case = harness.load_case("acoustic_t32_mel1000")
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    b = {k: v.cuda() for k, v in batch.items()}
    noise = synth.draw_acoustic_noise(synth.NoiseTape(meta["tape_seed"]), meta["B"], meta["T"], meta["steps_f0"], meta["steps_mel"])
    f32 = StyleSingerHIP(None, hparams=hp)
    f32.load_state_dict(sd)
    f32.eval().to("cuda:0")
    ref = _fwd(f32, b, noise=noise)
    h = StyleSingerHIP(None, hparams=dict(hp, mfma_precision="bf16"))
    h.load_state_dict(sd)
    h.eval().to("cuda:0")
    assert h.bf16 and h.bf16_hbm
    mel_chain = h.mel_stage(ref["fs2_mel"], ref["diff_cond"], z_q=noise["mel"]["z_q"], z_steps=noise["mel"]["z_steps"])
    l1 = (mel_chain.cpu() - gold["mel_out"]).abs().mean().item()
    mx = (mel_chain.cpu() - gold["mel_out"]).abs().max().item()
    full = _fwd(h, b, noise=noise)
    l1_full = (full["mel_out"].cpu() - gold["mel_out"]).abs().mean().item()
    flips = (full["uv_a"] != ref["uv_a"]).sum().item() + (full["uv_b"] != ref["uv_b"]).sum().item()
    print(f"bf16 mode, 1000-step golden: mel chain alone L1 vs the fp32 reference {l1:.3e} (max {mx:.3e}); whole path {l1_full:.3e} "
          f"({flips} voicing decisions differ from the fp32 run)")
    assert torch.isfinite(full["mel_out"]).all() and torch.isfinite(mel_chain).all()
    from conftest import record_measurement
    record_measurement("c4_bf16_t32_1000steps_vs_fp32_reference", mel_chain_l1=l1, mel_chain_max=mx, whole_path_l1=l1_full,
                       voicing_decisions_differing=flips, pinned=False)
    # round-3 measurement on MI355X: see BF16_CHAIN_L1_MEASURED below; the bound is 3x that (a regression guard, NOT parity: this
    # mode does not meet north_star's 1e-4 against the fp32 reference and cannot by construction)
    assert l1 <= 3 * BF16_CHAIN_L1_MEASURED, (l1, BF16_CHAIN_L1_MEASURED)


def test_from_clean_build_on_this_box_runs(tmp_path):
    """The shipped .so is built incrementally in the container; this rebuilds EVERY source from scratch on the GPU box (hipcc,
    gfx950), loads the result through the same ctypes path and runs a kernel from it."""
    import ctypes
    from stylesinger_amd import build as B
    from stylesinger_amd import lib as L
    so = B.build_clean(str(tmp_path))
    fresh = ctypes.CDLL(so)
    assert fresh.ss_abi_version() == L.ABI_VERSION
    for name in L.declared_symbols():
        assert hasattr(fresh, name), name
    x = torch.linspace(-3, 3, 1000, device="cuda")
    y = torch.empty_like(x)
    fresh.ss_clip.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    assert fresh.ss_clip(x.data_ptr(), y.data_ptr(), 1000, -1.0, 1.5, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, x.clamp(-1.0, 1.5))
