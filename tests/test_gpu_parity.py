"""GPU parity of the whole hot path: HIP (through the C-ABI) vs the golden outputs of the REAL reference and
vs the CPU restatement, with a shared noise tape.  Tolerances are stated next to each check."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import harness  # noqa: E402
from oracle import restatement as R  # noqa: E402
from stylesinger_amd import config, synth  # noqa: E402
from stylesinger_amd.model import StyleSingerHIP  # noqa: E402
from stylesinger_amd.vocoder import HifiGAN  # noqa: E402

MEL_L1_TOL = 1e-5      # north_star asks mel L1 <= 1e-4 vs the reference (fp32); measured 3e-7..8e-7 on MI355X -> 10x margin only
WAV_TOL = 1e-5         # waveform max-abs vs the reference (measured 2e-7)
STAGE_TOL = 5e-5       # max-abs on intermediate activations of O(1) magnitude
# per-case overrides of the mel L1 bound (none needed: the 1000-step chain measures 1.0e-6 on MI355X)
CASE_MEL_L1_TOL = {}


def _run_hip(meta):
    hp, sd, batch = harness.case_setup(meta)
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd, strict=True)
    model.eval().to(dev)
    tape = synth.NoiseTape(meta["tape_seed"])
    T = meta["T"]
    gold_T = None
    noise = None
    if meta["give_mel2ph"]:
        noise = synth.draw_acoustic_noise(tape, meta["B"], T, meta["steps_f0"], meta["steps_mel"])
    b = {k: v.to(dev) for k, v in batch.items()}
    kw = dict(spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"], ref_f0=b["ref_f0"], global_steps=320000,
              infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"])
    if meta["give_mel2ph"]:
        ret = model(b["txt_tokens"], mel2ph=b["mel2ph"], noise=noise, **kw)
    else:
        # durations are predicted: T is only known after the duration stage, so draw the tape for that T
        pre = model(b["txt_tokens"], mel2ph=None, skip_decoder=True, **kw)
        T = pre["mel2ph"].shape[1]
        noise = synth.draw_acoustic_noise(tape, meta["B"], T, meta["steps_f0"], meta["steps_mel"])
        ret = model(b["txt_tokens"], mel2ph=None, noise=noise, **kw)
    torch.cuda.synchronize()
    return ret, tape


@pytest.mark.parametrize("name", ["acoustic_tiny_s4", "acoustic_b2_s3", "acoustic_dur_s2", "acoustic_t64_s100",
                                  "acoustic_t300_s100",        # round 4: 300 frames x (100 + 2 x 100) steps of the REAL reference
                                  "acoustic_t32_mel1000",      # BASELINE config 4's schedule: 1000 mel steps, coefficients up to ~3e6
                                  "prodiff_t40_vpsde", "prodiff_b2_t32_linear"])   # hparams['decoder'] = 'prodiff' (8 teacher steps)
def test_acoustic_hip_matches_reference_golden(name):
    case = harness.load_case(name)
    meta, gold = case["meta"], case["out"]
    ret, tape = _run_hip(meta)
    assert tape.log == meta["tape_log"]
    assert torch.equal(ret["mel2ph"].cpu(), gold["mel2ph"])           # integer: bit-exact
    if "dur_choice" in gold:
        assert torch.equal(ret["dur_choice"].cpu(), gold["dur_choice"])
    for k_hip, k_gold in [("encoder_out_text", "encoder_out"), ("style_pre_rq", "style_pre_rq"), ("style_rq", "style_rq"),
                          ("style", "style"), ("decoder_inp", "decoder_inp"), ("decoder_out", "decoder_out"),
                          ("diff_cond", "diff_cond"), ("pitch_pred", "pitch_pred")]:
        if k_gold in gold:
            err = (ret[k_hip].cpu() - gold[k_gold]).abs().max().item()
            print(f"{name}: {k_hip} max abs err {err:.3e}")
            assert err <= STAGE_TOL, f"{name}:{k_hip} max abs err {err:.3e}"
    uv_flip = ((ret["pitch_pred"][..., 1].cpu() > 0) != (gold["pitch_pred"][..., 1] > 0)).float().mean().item()
    assert uv_flip == 0.0, f"voicing flips {uv_flip}"
    assert torch.allclose(ret["f0_denorm"].cpu(), gold["f0_denorm"], rtol=2e-4, atol=1e-2)
    l1 = (ret["mel_out"].cpu() - gold["mel_out"]).abs().mean().item()
    mx = (ret["mel_out"].cpu() - gold["mel_out"]).abs().max().item()
    f0e = (ret["f0_denorm"].cpu() - gold["f0_denorm"]).abs().max().item()
    print(f"{name}: mel L1 {l1:.3e} max {mx:.3e}; f0_denorm max abs err {f0e:.3e} Hz")
    assert l1 <= CASE_MEL_L1_TOL.get(name, MEL_L1_TOL), f"{name}: mel L1 {l1:.3e}"


@pytest.mark.parametrize("name", ["vocoder_t12", "vocoder_b2_t9", "vocoder_t200"])   # t200: 51 200 samples of the REAL reference (round 4)
def test_vocoder_hip_matches_reference_golden(name):
    case = harness.load_case(name)
    meta = case["meta"]
    cfg, vsd = harness.vocoder_case_setup(meta)
    voc = HifiGAN(cfg, vsd, device="cuda:0")
    tape = synth.NoiseTape(meta["tape_seed"])
    B, T = meta["B"], meta["T"]
    noise = synth.draw_vocoder_noise(tape, B, T * 256)
    assert tape.log == meta["tape_log"]
    wav, har = voc.model(case["inp"]["mel"].cuda(), case["inp"]["f0"].cuda(), noise=noise, return_source=True)
    e_h = (har.cpu() - case["out"]["har"]).abs().max().item()
    e_w = (wav.cpu() - case["out"]["wav"]).abs().max().item()
    print(f"{name}: har max err {e_h:.3e} wav max err {e_w:.3e}")
    assert e_h <= 2e-6     # harmonic source (measured 3e-8)
    assert e_w <= WAV_TOL  # waveform in [-1,1] after 4 upsampling stages
    if B == 1:
        w1 = voc.spec2wav(case["inp"]["mel"][0].numpy(), f0=case["inp"]["f0"][0].numpy(), noise=noise)
        assert abs(w1 - case["out"]["wav"][0].numpy()).max() <= WAV_TOL


def test_ragged_batch_equals_per_item_runs():
    """Per-item lengths: a padded batch must reproduce each item run alone (the reference is B=1 only)."""
    hp = config.make_hparams(dict(timesteps=3, K_step=3, f0_timesteps=3))
    sd = synth.synth_acoustic_state_dict(hp, 5)
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)
    items = [synth.synth_utterance(0, 70, 7, 50, hp, 5), synth.synth_utterance(1, 45, 5, 38, hp, 5)]
    T, Tp, Tr = 70, 7, 50
    def pad(t, n):
        out = torch.zeros((n,) + tuple(t.shape[1:]), dtype=t.dtype)
        out[:t.shape[0]] = t
        return out
    batch = {k: torch.stack([pad(it[k], {"txt_tokens": Tp, "note": Tp, "note_type": Tp, "note_dur": Tp, "mel2ph": T,
                                         "ref_mels": Tr, "ref_f0": Tr}.get(k, it[k].shape[0])) for it in items]) for k in items[0]}
    tape = synth.NoiseTape(9)
    noise = synth.draw_acoustic_noise(tape, 2, T, 3, 3)
    def run(b, nz):
        b = {k: v.to(dev) for k, v in b.items()}
        return model(b["txt_tokens"], mel2ph=b["mel2ph"], spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                     ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], noise=nz)
    full = run(batch, noise)
    for i, (Ti, it) in enumerate(zip((70, 45), items)):
        nz = {k: {kk: (vv[:, i:i + 1, ..., :Ti] if kk in ("z_steps", "u_steps") else vv[i:i + 1, ..., :Ti]) for kk, vv in v.items()} for k, v in noise.items()}
        one = run({k: v[None] for k, v in it.items()}, nz)
        d = (full["mel_out"][i, :Ti] - one["mel_out"][0]).abs().max().item()
        assert d <= 1e-5, (i, d)
        assert torch.equal(full["uv_a"][i, :Ti], one["uv_a"][0])
    assert full["mel_out"][1, 45:].abs().max().item() == 0.0


def test_oracle_vs_hip_medium_size():
    """Beyond the committed fixtures: a mid-size case checked against the CPU restatement run on this box."""
    hp = config.make_hparams(dict(timesteps=8, K_step=8, f0_timesteps=8))
    sd = synth.synth_acoustic_state_dict(hp, 21)
    B, T, Tp, Tr = 2, 300, 10, 260
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 21)
    tape = synth.NoiseTape(22)
    with torch.no_grad():
        ref = R.acoustic_forward(sd, hp, batch, tape, mel2ph=batch["mel2ph"])
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(22), B, T, 8, 8)
    b = {k: v.to(dev) for k, v in batch.items()}
    ret = model(b["txt_tokens"], mel2ph=b["mel2ph"], spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], noise=noise)
    assert torch.equal(ret["rq_codes"].cpu(), ref["rq_codes"])
    flips = (ret["uv_a"].cpu().long() != ref["uv_a"]).float().mean().item() + (ret["uv_b"].cpu().long() != ref["uv_b"]).float().mean().item()
    assert flips == 0.0
    l1 = (ret["mel_out"].cpu() - ref["mel_out"]).abs().mean().item()
    print(f"medium: mel L1 {l1:.3e}")
    assert l1 <= MEL_L1_TOL


def test_hipgraph_replay_matches_eager_and_reseeds():
    """The diffusion loops captured as hipGraphs must reproduce the eager launches bit for bit (same Philox
    seeds), and a different seed must give different noise on replay (device seed word)."""
    hp = config.make_hparams(dict(timesteps=5, K_step=5, f0_timesteps=5))
    sd = synth.synth_acoustic_state_dict(hp, 3)
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)
    batch = {k: v.to(dev) for k, v in synth.synth_batch(2, 90, 6, 70, hp, 3).items()}
    def run(seed):
        return model(batch["txt_tokens"], mel2ph=batch["mel2ph"], spk_embed=batch["spk_embed"], emo_embed=batch["emo_embed"],
                     ref_mels=batch["ref_mels"], ref_f0=batch["ref_f0"], global_steps=320000, infer=True, note=batch["note"],
                     note_dur=batch["note_dur"], note_type=batch["note_type"], seed=seed)
    model.use_graphs = "off"
    eager = run(77)["mel_out"].clone()
    model.use_graphs = "on"
    g1 = run(77)["mel_out"].clone()   # captures
    g2 = run(77)["mel_out"].clone()   # replays
    g3 = run(78)["mel_out"].clone()
    assert torch.equal(eager, g1) and torch.equal(g1, g2)
    assert (g3 - g2).abs().max().item() > 1e-3
    assert torch.isfinite(g3).all()


def test_long_form_30s_sequence_matches_oracle():
    """BASELINE config 4 shape (30 s = 5625 frames, beyond max_frames=3000: sinusoidal tables auto-extend,
    attention over 5625 keys is flash-tiled) with a short 2-step chain so the CPU oracle finishes in seconds."""
    hp = config.make_hparams(dict(timesteps=2, K_step=2, f0_timesteps=2))
    sd = synth.synth_acoustic_state_dict(hp, 31)
    B, T, Tp, Tr = 1, 5625, 105, 1500
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 31)
    with torch.no_grad():
        ref = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(32), mel2ph=batch["mel2ph"])
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(32), B, T, 2, 2)
    b = {k: v.to(dev) for k, v in batch.items()}
    ret = model(b["txt_tokens"], mel2ph=b["mel2ph"], spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], noise=noise)
    # f0 -> coarse pitch bin is a rounding decision inside the float pipeline (utils/pitch_utils.py:22-31): a 1e-3 Hz
    # difference can move a frame to the neighbouring bin (different embedding row). Report the flip rate, bound it,
    # and compare the decoder away from flipped frames (the conv-FFN spreads a flip over +-4 frames per layer).
    flips = (ret["pitch_coarse"].cpu() != ref["pitch_coarse"])
    rate = flips.float().mean().item()
    keep = torch.ones(B, T, dtype=torch.bool)
    for bb, tt in flips.nonzero().tolist():
        keep[bb, max(0, tt - 20):tt + 21] = False
    e_dec = ((ret["decoder_out"].cpu() - ref["decoder_out"]).abs().max(-1).values * keep).max().item()
    e_sty = (ret["style"].cpu() - ref["style"]).abs().max().item()
    l1 = (ret["mel_out"].cpu() - ref["mel_out"]).abs().mean().item()
    print(f"long-form: coarse-pitch flips {int(flips.sum())}/{flips.numel()} decoder_out max err {e_dec:.3e} style max err {e_sty:.3e} mel L1 {l1:.3e}")
    assert rate <= 1e-3
    assert e_dec <= STAGE_TOL and e_sty <= STAGE_TOL
    assert l1 <= MEL_L1_TOL


def test_single_utterance_entrypoint_matches_batched_path(golden_dir):
    """StyleSingerInfer.forward_model (the reference's B=1 numpy surface, inference/StyleSinger.py:41-63) must agree
    with the batched device path on the same utterance and seed. `inp['f0']` is a tracker contour in Hz as at
    inference/StyleSinger.py:125-136; `input_to_batch` runs it through norm_interp_f0 like :152 - checked against the output of
    the REAL utils/pitch_utils.py function (tests/golden/norm_interp_f0.pt)."""
    from stylesinger_amd.infer import StyleSingerInfer
    hp = config.make_hparams(dict(timesteps=4, K_step=4, f0_timesteps=4))
    sd = synth.synth_acoustic_state_dict(hp, 11)
    vsd = synth.synth_vocoder_state_dict(None, 11)
    inf = StyleSingerInfer(hp, device="cuda:0", model_state=sd, vocoder_state=vsd)
    pc = torch.load(os.path.join(golden_dir, "norm_interp_f0.pt"), weights_only=False)["cases"]["t300_f64"]
    Tr = pc["hz"].numel()
    it = synth.synth_utterance(0, 40, 5, Tr, hp, 11)
    inp = dict(ph_token=it["txt_tokens"].numpy(), mel=it["ref_mels"].numpy(), spk_embed=it["spk_embed"].numpy(),
               emo_embed=it["emo_embed"].numpy(), note=it["note"].numpy(), note_dur=it["note_dur"].numpy(),
               note_type=it["note_type"].numpy(), f0=pc["hz"].numpy(), mel2ph=it["mel2ph"].numpy())
    sample = inf.input_to_batch(inp)
    assert torch.equal(sample["ref_f0"][0].cpu(), pc["f0"])      # what the reference's input_to_batch would hand the model
    tape = synth.NoiseTape(5)
    noise = synth.draw_acoustic_noise(tape, 1, 40, 4, 4)
    vnoise = synth.draw_vocoder_noise(tape, 1, 40 * 256)
    wav1 = inf.forward_model(inp, noise=noise, vocoder_noise=vnoise)
    it["ref_f0"] = pc["f0"]
    batch = {k: v[None].cuda() for k, v in it.items()}
    res = inf.infer_batch(batch, noise=noise, vocoder_noise=vnoise)
    assert wav1.shape == (40 * 256,)
    assert abs(wav1 - res["wav"][0].cpu().numpy()).max() <= 1e-5


def test_ddim_sampler_matches_oracle():
    """BASELINE config 5 sampler (50-step-style DDIM over the same denoiser; here 6 of 24 steps). No reference sampler
    exists: parity is against oracle.restatement.mel_ddim on the oracle's own coarse mel / condition."""
    hp = config.make_hparams(dict(timesteps=24, K_step=24, f0_timesteps=3))
    sd = synth.synth_acoustic_state_dict(hp, 41)
    B, T = 2, 120
    batch = synth.synth_batch(B, T, 6, 90, hp, 41)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(42), B, T, 3, 24)
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)
    b = {k: v.to(dev) for k, v in batch.items()}
    ret = model(b["txt_tokens"], mel2ph=b["mel2ph"], spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"],
                noise=noise, sampler="ddim", ddim_steps=6)
    ts = model.ddim_timesteps(6)
    assert ts[0] == 23 and ts[-1] == 0 and len(ts) == 6

    class OneDraw:
        def randn(self, *shape):
            return noise["mel"]["z_q"].clone()
    with torch.no_grad():
        ref = R.mel_ddim(sd, hp, ret["fs2_mel"].cpu(), ret["diff_cond"].cpu(), OneDraw(), ts)
    l1 = (ret["mel_out"].cpu() - ref).abs().mean().item()
    print(f"ddim: mel L1 {l1:.3e}")
    assert l1 <= MEL_L1_TOL


def test_pcm16_writer_matches_numpy_cast(tmp_path):
    """ss_wav_to_pcm16 == numpy `(wav * 32767).astype(np.int16)` (utils/audio.py:12-17), bit exact; per-item crop."""
    import numpy as np
    from scipy.io import wavfile
    from stylesinger_amd.writer import WavWriter, wav_to_pcm16
    g = torch.Generator().manual_seed(3)
    wav = (torch.rand(3, 5 * 256, generator=g) * 2 - 1) * 0.999
    lens = torch.tensor([5, 3, 4], dtype=torch.int32)
    pcm = wav_to_pcm16(wav.cuda(), lens.cuda(), 256)
    ref = (wav.numpy().copy() * 32767).astype(np.int16)
    for b in range(3):
        n = int(lens[b]) * 256
        assert np.array_equal(pcm[b, :n].cpu().numpy(), ref[b, :n])
        assert int(pcm[b, n:].abs().sum()) == 0
    pn = wav_to_pcm16(wav.cuda(), None, 256, norm=True).cpu().numpy()
    w = wav.numpy().copy()
    refn = np.stack([((w[b] / np.abs(w[b]).max()) * 32767).astype(np.int16) for b in range(3)])
    assert np.abs(pn.astype(np.int32) - refn.astype(np.int32)).max() <= 1   # x/peak*32767 vs x*(32767/peak): <= 1 LSB
    wr = WavWriter(str(tmp_path), 48000)
    wr.submit_batch(["u0", "u1", "u2"], pcm, lens.cuda(), 256)
    wr.close()
    sr, back = wavfile.read(str(tmp_path / "u1.wav"))
    assert sr == 48000 and np.array_equal(back, ref[1, :3 * 256])


def test_style_transfer_sweep_equals_uncached_batches():
    """BASELINE config 5 driver: cached per-reference style encodings + DDIM + (optionally) hipGraph replay must give
    exactly what an uncached forward of the same (target x references) batch gives; every pair is produced once."""
    from stylesinger_amd.infer import StyleSingerInfer
    from stylesinger_amd.sweep import style_transfer_sweep
    hp = config.make_hparams(dict(timesteps=8, K_step=8, f0_timesteps=3))
    sd = synth.synth_acoustic_state_dict(hp, 5)
    vsd = synth.synth_vocoder_state_dict(None, 5)
    inf = StyleSingerInfer(hp, device="cuda:0", model_state=sd, vocoder_state=vsd)
    refs, targets = [], []
    for i, Tr in enumerate((40, 52, 40)):
        it = synth.synth_utterance(100 + i, 16, 4, Tr, hp, 5)
        refs.append({k: it[k] for k in ("ref_mels", "ref_f0", "spk_embed", "emo_embed")})
    for j, (T, Tp) in enumerate(((48, 5), (64, 6))):
        it = synth.synth_utterance(200 + j, T, Tp, 8, hp, 5)
        targets.append({k: it[k] for k in ("txt_tokens", "note", "note_dur", "note_type", "mel2ph")})
    got = {}
    stats = {}
    n_pairs, n_frames = style_transfer_sweep(inf, refs, targets, batch=2, ddim_steps=4, seed=77, stats=stats,
                                             emit=lambda r, t, mel, f0, wav: got.__setitem__((r, t), (mel.clone(), f0.clone(), wav.clone())))
    assert n_pairs == 6 and n_frames == 3 * (48 + 64) and len(got) == 6
    # per-reference style cache: 3 references encoded once each, served from the cache for the second target
    assert stats["style_encodes"] == 3 and stats["style_cache_hits"] == 3 and stats["pairs"] == 6 and stats["refs_on_rank"] == 3
    dev = inf.device
    for t, tgt in enumerate(targets):
        for ridx in ([0, 1], [2]):
            nb = len(ridx)
            Tr = max(refs[i]["ref_mels"].shape[0] for i in ridx)
            rm = torch.stack([torch.nn.functional.pad(refs[i]["ref_mels"], (0, 0, 0, Tr - refs[i]["ref_mels"].shape[0])) for i in ridx])
            rf = torch.stack([torch.nn.functional.pad(refs[i]["ref_f0"], (0, Tr - refs[i]["ref_f0"].shape[0])) for i in ridx])
            rep = lambda x: x[None].expand(nb, *x.shape).contiguous().to(dev)
            out = inf.model(rep(tgt["txt_tokens"]), mel2ph=rep(tgt["mel2ph"]), spk_embed=torch.stack([refs[i]["spk_embed"] for i in ridx]).to(dev),
                            emo_embed=torch.stack([refs[i]["emo_embed"] for i in ridx]).to(dev), ref_mels=rm.to(dev), ref_f0=rf.to(dev),
                            global_steps=320000, infer=True, note=rep(tgt["note"]), note_dur=rep(tgt["note_dur"]), note_type=rep(tgt["note_type"]),
                            sampler="ddim", ddim_steps=4, seed=77)
            for k, r in enumerate(ridx):
                mel, f0, wav = got[(r, t)]
                assert (mel - out["mel_out"][k]).abs().max().item() <= 1e-5, (r, t)
                assert (f0 - out["f0_denorm"][k]).abs().max().item() <= 1e-3
                assert wav.shape[0] == mel.shape[0] * 256 and torch.isfinite(wav).all()


class _ListTape:
    """Replays a fixed list of noise tensors (for checking one stage of the path in isolation)."""

    def __init__(self, items):
        self.items = list(items)

    def randn(self, *shape):
        t = self.items.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.clone()

    rand = randn


BF16_MEL_L1_TOL = 1e-3   # bf16-operand mode vs the oracle with the same operand rounding (only rounding-boundary flips and
#                          accumulation order differ); the fp32 result is ~1e-2 away, reported below for information


def test_bf16_mfma_mode_matches_rounded_oracle():
    """BASELINE config 4 ("bf16 MFMA"): hidden GEMMs of the denoisers and the vocoder convs on v_mfma_f32_32x32x16_bf16.
    The mel denoiser and the vocoder are checked stage by stage against oracle.restatement with set_matmul_rounding("bf16")
    on the HIP path's own stage inputs, so discrete pitch decisions upstream cannot mask a kernel error."""
    from stylesinger_amd.infer import StyleSingerInfer
    K, S = 12, 4
    hp = config.make_hparams(dict(timesteps=K, K_step=K, f0_timesteps=S, mfma_precision="bf16"))
    sd = synth.synth_acoustic_state_dict(hp, 51)
    vsd = synth.synth_vocoder_state_dict(None, 51)
    B, T = 2, 96
    batch = synth.synth_batch(B, T, 6, 80, hp, 51)
    tape = synth.NoiseTape(52)
    noise = synth.draw_acoustic_noise(tape, B, T, S, K)
    vnoise = synth.draw_vocoder_noise(tape, B, T * 256)
    inf = StyleSingerInfer(hp, device="cuda:0", model_state=sd, vocoder_state=vsd)
    assert inf.model.bf16 and not inf.model.use_wino
    b = {k: v.cuda() for k, v in batch.items()}
    res = inf.infer_batch(b, noise=noise, vocoder_noise=vnoise)
    ret = res["model_out"]
    nz = noise["mel"]
    draws = [nz["z_q"].reshape(B, 1, 80, T)] + [nz["z_steps"][K - 1 - i].reshape(B, 1, 80, T) for i in range(K)]
    try:
        R.set_matmul_rounding("bf16")
        with torch.no_grad():
            mel_ref = R.mel_diffusion(sd, hp, ret["fs2_mel"].cpu(), ret["diff_cond"].cpu(), _ListTape(draws))
            mel_c = ret["mel_out"].cpu().clamp(hp["mel_vmin"], hp["mel_vmax"])
            wav_ref, _ = R.hifigan_forward(vsd, inf.vocoder.config, mel_c, ret["f0_denorm"].cpu(),
                                           _ListTape([vnoise["rand_ini"], vnoise["sine_noise"], torch.zeros(B, T * 256, 1)]))
    finally:
        R.set_matmul_rounding(None)
    with torch.no_grad():
        mel_f32 = R.mel_diffusion(sd, hp, ret["fs2_mel"].cpu(), ret["diff_cond"].cpu(), _ListTape(draws))
    l1 = (ret["mel_out"].cpu() - mel_ref).abs().mean().item()
    l1_f32 = (ret["mel_out"].cpu() - mel_f32).abs().mean().item()
    ew = (res["wav"].cpu() - wav_ref).abs().max().item()
    print(f"bf16 mode: mel L1 vs rounded oracle {l1:.3e} (vs fp32 oracle {l1_f32:.3e}); wav max err vs rounded oracle {ew:.3e}")
    assert l1 <= BF16_MEL_L1_TOL
    assert l1_f32 > l1          # the mode really is bf16
    assert ew <= 5e-3           # |wav| <= 1


@pytest.mark.parametrize("name", ["plms_t40_k20_i3", "plms_t24_k12_i4", "plms_t32_k12of20_i3"])  # last: K_step 12 < timesteps 20
def test_plms_sampler_matches_reference_golden(name):
    """ss_meldiff_sample_plms vs the REAL reference's p_sample_plms loop (fixture from oracle/gen_golden.py); also batched
    (B = 3 copies with different lengths), which the reference's implementation cannot run."""
    case = harness.load_case(name)
    meta = case["meta"]
    hp = config.make_hparams(dict(timesteps=meta["steps_mel"], K_step=meta.get("k_step", meta["steps_mel"]), f0_timesteps=2))
    sd = synth.synth_acoustic_state_dict(hp, meta["seed"])
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)
    T = meta["T"]
    zq = synth.NoiseTape(meta["tape_seed"]).randn(1, 1, 80, T)
    mel = model.mel_stage(case["inp"]["coarse_mel"].to(dev), case["inp"]["cond"].to(dev), z_q=zq, sampler="plms",
                          plms_interval=meta["interval"])
    l1 = (mel.cpu() - case["out"]["mel_out"]).abs().mean().item()
    print(f"{name}: mel L1 {l1:.3e}")
    assert l1 <= MEL_L1_TOL
    lens = torch.tensor([T, T - 5, T - 11], dtype=torch.int32, device=dev)
    mel3 = model.mel_stage(case["inp"]["coarse_mel"].expand(3, -1, -1).contiguous().to(dev), case["inp"]["cond"].expand(3, -1, -1).contiguous().to(dev),
                           lens=lens, z_q=zq.expand(3, -1, -1, -1).contiguous(), sampler="plms", plms_interval=meta["interval"])
    assert (mel3[0] - mel[0]).abs().max().item() <= 1e-5
    # shorter items see zero padding instead of the reference's frames -> only frames far from the cut can be compared
    assert mel3[2, T - 11:].abs().max().item() == 0.0


def test_mel_frontend_matches_oracle():
    """Reference-audio front end (utils/audios/__init__.py:36-84) as two fp32-MFMA GEMMs vs oracle/frontend.py (numpy,
    float64 FFT). Tolerance: 1e-4 max / 1e-5 mean abs error in log10-mel (measured 7.6e-6 / 2.6e-7 on MI355X)."""
    import numpy as np
    from oracle import frontend as F
    from stylesinger_amd.frontend import MelFrontendHIP
    rng = np.random.default_rng(5)
    Ls = 256 * 57 + 131
    t = np.arange(Ls) / 48000.0
    wavs = []
    for b, f0 in enumerate((196.0, 311.1)):
        w = sum(0.3 / (h + 1) * np.sin(2 * np.pi * f0 * (h + 1) * t * (1 + 0.01 * np.sin(2 * np.pi * 5 * t))) for h in range(12))
        w = w * np.linspace(0.05, 1.0, Ls) + 0.003 * rng.standard_normal(Ls)
        wavs.append(w.astype(np.float32))
    lens = [Ls, 256 * 31 + 7]
    fe = MelFrontendHIP(None, device="cuda:0")
    mel, frames = fe.wav2mel(torch.from_numpy(np.stack(wavs)).cuda(), lens=torch.tensor(lens))
    assert frames.tolist() == [n // 256 + 1 for n in lens]
    for b in range(2):
        ref = F.wav2mel(wavs[b][:lens[b]])
        got = mel[b, :ref.shape[0]].cpu().numpy()
        err = np.abs(got - ref)
        print(f"frontend item {b}: frames {ref.shape[0]} max err {err.max():.3e} mean err {err.mean():.3e} (mel range {ref.min():.2f}..{ref.max():.2f})")
        assert err.max() <= 1e-4 and err.mean() <= 1e-5
        assert mel[b, ref.shape[0]:].abs().max().item() == 0.0 if ref.shape[0] < mel.shape[1] else True


def test_c2_full_size_batch_items_equal_single_runs():
    """BASELINE configs[1] shape (B=8 x T=1500 frames, Tp=28, Tr=1500) - too large for the CPU oracle in a test, so the
    size-independent property is used: every utterance of the batch must equal its own B=1 run on the same noise tape
    (the reference only ever runs B=1). 3+3+3 diffusion steps keep it to seconds; integers exact, mel to 1e-5."""
    hp = config.make_hparams(dict(timesteps=3, K_step=3, f0_timesteps=3))
    sd = synth.synth_acoustic_state_dict(hp, 61)
    B, T, Tp, Tr = 8, 1500, 28, 1500
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 61)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(62), B, T, 3, 3)
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)

    def run(b, nz):
        bb = {k: v.to(dev) for k, v in b.items()}
        return model(bb["txt_tokens"], mel2ph=bb["mel2ph"], spk_embed=bb["spk_embed"], emo_embed=bb["emo_embed"], ref_mels=bb["ref_mels"],
                     ref_f0=bb["ref_f0"], global_steps=320000, infer=True, note=bb["note"], note_dur=bb["note_dur"], note_type=bb["note_type"], noise=nz)
    full = run(batch, noise)
    assert torch.isfinite(full["mel_out"]).all()
    for i in (0, 5, 7):
        one_b = {k: v[i:i + 1] for k, v in batch.items()}
        nz = {net: {k: (v[:, i:i + 1] if k in ("z_steps", "u_steps") else v[i:i + 1]) for k, v in noise[net].items()} for net in ("f0_a", "f0_b")}
        nz["mel"] = dict(z_q=noise["mel"]["z_q"][i:i + 1], z_steps=noise["mel"]["z_steps"][:, i:i + 1])
        one = run(one_b, nz)
        assert torch.equal(one["rq_codes"][0], full["rq_codes"][i])
        assert torch.equal(one["uv_a"][0], full["uv_a"][i]) and torch.equal(one["uv_b"][0], full["uv_b"][i])
        assert torch.equal(one["pitch_coarse"][0], full["pitch_coarse"][i])
        e = (one["mel_out"][0] - full["mel_out"][i]).abs().max().item()
        assert e <= 1e-5, (i, e)


def test_batch_with_an_empty_item_leaves_the_others_untouched():
    """Edge case of the batched path (the reference is B=1 and never sees it): an item that is all padding (no phonemes, no
    frames) must produce zeros / finite values and must not disturb its neighbours."""
    hp = config.make_hparams(dict(timesteps=3, K_step=3, f0_timesteps=3))
    sd = synth.synth_acoustic_state_dict(hp, 71)
    B, T = 3, 50
    batch = synth.synth_batch(B, T, 6, 40, hp, 71)
    for k in ("txt_tokens", "note", "note_type", "mel2ph"):
        batch[k][1] = 0
    batch["note_dur"][1] = 0.0
    noise = synth.draw_acoustic_noise(synth.NoiseTape(72), B, T, 3, 3)
    dev = torch.device("cuda:0")
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd)
    model.eval().to(dev)

    def run(b, nz):
        bb = {k: v.to(dev) for k, v in b.items()}
        return model(bb["txt_tokens"], mel2ph=bb["mel2ph"], spk_embed=bb["spk_embed"], emo_embed=bb["emo_embed"], ref_mels=bb["ref_mels"],
                     ref_f0=bb["ref_f0"], global_steps=320000, infer=True, note=bb["note"], note_dur=bb["note_dur"], note_type=bb["note_type"], noise=nz)
    full = run(batch, noise)
    assert full["lens"].tolist() == [T, 0, T]
    assert torch.isfinite(full["mel_out"]).all() and torch.isfinite(full["f0_denorm"]).all()
    assert full["mel_out"][1].abs().max().item() == 0.0
    for i in (0, 2):
        nz = {net: {k: (v[:, i:i + 1] if k in ("z_steps", "u_steps") else v[i:i + 1]) for k, v in noise[net].items()} for net in ("f0_a", "f0_b")}
        nz["mel"] = dict(z_q=noise["mel"]["z_q"][i:i + 1], z_steps=noise["mel"]["z_steps"][:, i:i + 1])
        one = run({k: v[i:i + 1] for k, v in batch.items()}, nz)
        assert (one["mel_out"][0] - full["mel_out"][i]).abs().max().item() <= 1e-5
        assert torch.equal(one["uv_a"][0], full["uv_a"][i])
