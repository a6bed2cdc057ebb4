"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the ctypes mirror
matches the C structs, the parameter/hparams contract matches the reference dump, synthetic data is
deterministic, and the product refuses to run without a GPU instead of falling back."""
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from stylesinger_amd import config, lib, spec, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
    l = lib.load()
    names = lib.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(l, n), n
    assert l.ss_abi_version() == lib.ABI_VERSION == 19
    assert l.ss_last_error() is not None


def test_ctypes_struct_mirror_matches_c():
    import ctypes
    l = lib.load()
    sizes = (ctypes.c_int64 * 3)()
    assert l.ss_struct_sizes(sizes, 3) == 0
    assert tuple(sizes) == (ctypes.sizeof(lib.ConvGemmArgs), ctypes.sizeof(lib.WaveNet), ctypes.sizeof(lib.HifiGan))


def test_argument_errors_are_reported_not_crashed():
    l = lib.load()
    assert l.ss_conv_gemm(None, None) != 0
    assert b"null args" in l.ss_last_error()
    assert l.ss_layernorm(None, None, None, None, 1, 1, 8, 8, 8, 8, 8, 1e-5, None, 0, None) != 0
    # round-5 entry points: the input producers refuse bad arguments before touching a device
    assert l.ss_f0track(None, 0, None, None, None, 1, 1, None, None, None, None, 1, 0, None, 0, None) != 0 and b"ss_f0track" in l.ss_last_error()
    assert l.ss_f0track_workspace_bytes(2, 143, 899) > 2 * 143 * 900 * 8 and l.ss_f0track_workspace_bytes(0, 1, 1) == 0
    assert l.ss_vad_trim(None, 0, None, None, 0, 1, 1, 480, 8, 6, None, 0, None, None, None) != 0 and b"ss_vad_trim" in l.ss_last_error()
    assert l.ss_normalize_volume(None, None, None, 1, 1, -30.0, None) != 0
    assert l.ss_round_f16_rows(None, 0, 0, None, None, None, 0, 1, None) != 0
    assert not hasattr(l, "ss_fused_gate_res")   # the dataflow-launch experiment is built from tools/experiments/, not shipped in the library
    assert l.ss_set_tuning(b"q4_force", 1) == 0 and l.ss_get_tuning(b"q4_force") == 1 and l.ss_set_tuning(b"q4_force", 0) == 0
    assert l.ss_set_tuning(b"tile128", 1) != 0 and l.ss_set_tuning(b"skip_deep", 1) != 0      # removed in round 5
    # round-6 entry points
    import ctypes
    assert l.ss_layer512(None, None) != 0 and b"ss_layer512" in l.ss_last_error()
    a = lib.Layer512Args()
    assert l.ss_layer512(ctypes.byref(a), None) != 0 and b"null Hin" in l.ss_last_error()
    assert l.ss_layer512_entry(None, 256, 0, None, None, None, None, 1, 1, None) != 0 and l.ss_layer512_tile_addend(None, 512, 0, None, 1, 1, None) != 0
    assert l.ss_layer512_pack_gate(None, None, 2, None) != 0 and l.ss_layer512_pack_res(None, None, 2, None) != 0
    assert l.ss_debug_null_launch(0, 64, None) != 0 and l.ss_set_q4_guard(None) == 0


def test_param_spec_matches_reference_dump(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "param_spec.json")))
    mine = spec.acoustic_spec(config.make_hparams())
    assert [[k, list(s)] for k, s in mine] == ref["acoustic"]
    minev = spec.vocoder_spec(config.make_vocoder_config())
    assert [[k, list(s)] for k, s in minev] == ref["vocoder"]


def test_prodiff_and_emotion_specs_match_reference_dump(golden_dir):
    """hparams['decoder'] = 'prodiff' (stylesinger.py:111-117) and the emotion encoder (data_gen/tts/emotion/model.py)."""
    ref = json.load(open(os.path.join(golden_dir, "param_spec.json")))
    hp = config.make_hparams(dict(decoder="prodiff", timesteps=8, K_step=8, f0_timesteps=2, schedule_type="vpsde"))
    assert [[k, list(s)] for k, s in spec.acoustic_spec(hp)] == ref["acoustic_prodiff"]
    assert [[k, list(s)] for k, s in spec.emotion_spec()] == ref["emotion"]
    sd = synth.synth_acoustic_state_dict(hp, 3)
    assert set(sd) == {k for k, _ in ref["acoustic_prodiff"]} and sd["diff_decoder.betas"].shape == (9,)


def test_emotion_partial_slices_match_reference(golden_dir):
    """emotion.compute_partial_slices vs the reference's (inference.py:56-108) for several utterance lengths."""
    from stylesinger_amd import emotion
    case = torch.load(os.path.join(golden_dir, "emotion_p5.pt"), weights_only=False)
    assert case["meta"]["partial_slices"]
    for n, (wav_ref, mel_ref) in case["meta"]["partial_slices"].items():
        wav, mel = emotion.compute_partial_slices(n)
        assert [(s.start, s.stop) for s in wav] == [tuple(x) for x in wav_ref], n
        assert [(s.start, s.stop) for s in mel] == [tuple(x) for x in mel_ref], n


def test_plan_cache_is_lru_bounded_by_bytes():
    """The plan / hipGraph cache of StyleSingerHIP (host logic only): frames are bucketed, plans evicted least-recently-used."""
    from stylesinger_amd.model import StyleSingerHIP, _pad_frames
    hp = config.make_hparams(dict(timesteps=2, K_step=2, f0_timesteps=2))
    m = StyleSingerHIP(None, hparams=hp)
    assert m.t_bucket == 64 and [m.bucket_frames(t) for t in (1, 64, 65, 1500)] == [64, 64, 128, 1536]
    x = torch.arange(6.0).reshape(1, 2, 3)
    assert _pad_frames(x, 5).shape == (1, 2, 5) and _pad_frames(x, 4, dim=1).shape == (1, 4, 3)
    assert torch.equal(_pad_frames(x, 5)[..., :3], x) and _pad_frames(x, 5)[..., 3:].abs().sum() == 0

    class FakePlan:
        def __init__(self, b):
            self.bytes, self.uses = b, 0
    import stylesinger_amd.model as M
    orig = M._DiffPlan
    M._DiffPlan = lambda model, B, T, dev: FakePlan(B * T)
    try:
        m.plan_bytes = 1000
        dev = torch.device("cuda", 0)
        a = m._plan(1, 400, dev); b = m._plan(1, 500, dev)
        assert list(m._plans) == [(1, 400, 0), (1, 500, 0)]
        assert m._plan(1, 400, dev) is a                       # hit: moves to the recent end
        m._plan(1, 300, dev)                                   # 400 + 500 + 300 > 1000 -> evicts the LRU one (500)
        assert list(m._plans) == [(1, 400, 0), (1, 300, 0)]
        m.use_graphs = "auto"
        a.uses = 1
        assert not m._want_graphs(a)
        a.uses = 2
        assert m._want_graphs(a)                               # captured on the second use of a shape
    finally:
        M._DiffPlan = orig


def test_hparams_match_reference_dump(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "hparams.json")))
    hp = config.make_hparams()
    for k, v in ref.items():
        assert hp[k] == v, k


def test_synthetic_weights_and_inputs_are_deterministic():
    hp = config.make_hparams(dict(timesteps=4, K_step=4, f0_timesteps=4))
    a = synth.synth_acoustic_state_dict(hp, 7)
    b = synth.synth_acoustic_state_dict(hp, 7)
    assert set(a) == {n for n, _ in spec.acoustic_spec(hp)}
    for k in ("mel_out.weight", "postdiff.denoise_fn.residual_layers.3.dilated_conv.weight", "postdiff.betas"):
        assert torch.equal(a[k], b[k])
    assert a["gm_diffnet.mlp.0.weight"] is a["f0_gen._denoise_fn.mlp.0.weight"]
    x = synth.synth_batch(2, 20, 4, 16, hp, 3)
    y = synth.synth_batch(2, 20, 4, 16, hp, 3)
    assert all(torch.equal(x[k], y[k]) for k in x)
    assert (x["mel2ph"] > 0).all() and x["mel2ph"].max() == 4


def test_model_state_dict_contract_and_no_cpu_fallback():
    from stylesinger_amd.model import StyleSingerHIP
    hp = config.make_hparams(dict(timesteps=2, K_step=2, f0_timesteps=2))
    m = StyleSingerHIP(None, hparams=hp)
    sd = synth.synth_acoustic_state_dict(hp, 1)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: tuple(v.shape) for k, v in sd.items()}
    m.load_state_dict(sd, strict=True)
    assert torch.equal(m.state_dict()["ln_proj.weight"], sd["ln_proj.weight"])
    with pytest.raises(RuntimeError):
        m.load_state_dict({"bogus": torch.zeros(1)}, strict=True)
    m.eval()
    with pytest.raises(RuntimeError):
        m.train()
    if not torch.cuda.is_available():
        b = synth.synth_batch(1, 8, 2, 8, hp, 1)
        with pytest.raises(lib.StyleSingerHipError):
            m(b["txt_tokens"], mel2ph=b["mel2ph"], spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
              ref_f0=b["ref_f0"], infer=True, global_steps=320000, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"])


def test_vocoder_registry_surface():
    from stylesinger_amd import vocoder
    assert vocoder.get_vocoder_cls({"vocoder": "HifiGAN_NSF"}) is vocoder.HifiGAN
    g = vocoder.HifiGanGeneratorHIP()
    vsd = synth.synth_vocoder_state_dict()
    g.load_state_dict(vsd, strict=True)
    assert g.hop == 256


def test_sweep_plan_covers_every_pair_exactly_once():
    """BASELINE config 5 sharding: references rank::world (tasks/tts/tts_base.py:132), one target x `batch` refs per batch."""
    from stylesinger_amd.sweep import bucket_frames, sweep_plan
    n_refs, n_tgt, world, batch = 13, 5, 4, 3
    seen = set()
    for rank in range(world):
        for t, refs in sweep_plan(n_refs, n_tgt, rank, world, batch):
            assert 1 <= len(refs) <= batch
            for r in refs:
                assert r % world == rank
                assert (r, t) not in seen
                seen.add((r, t))
    assert len(seen) == n_refs * n_tgt
    # the BASELINE sizes: 256 x 256 over 8 GPUs -> 8192 pairs per GPU
    assert sum(len(r) for _, r in sweep_plan(256, 256, 3, 8, 8)) == 8192
    assert bucket_frames(1500, 64) == 1536 and bucket_frames(1536, 64) == 1536 and bucket_frames(7, 1) == 7


def test_wav_writer_emits_pcm16_riff(tmp_path):
    """write_wav_pcm16 must be readable as what scipy.io.wavfile.write produces for int16 (utils/audio.py:12-17)."""
    import numpy as np
    from scipy.io import wavfile
    from stylesinger_amd.writer import write_wav_pcm16
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=4801, dtype=np.int16)
    p = tmp_path / "a.wav"
    write_wav_pcm16(str(p), pcm, 48000)
    sr, back = wavfile.read(str(p))
    assert sr == 48000 and back.dtype == np.int16 and np.array_equal(back, pcm)
    q = tmp_path / "b.wav"
    wavfile.write(str(q), 48000, pcm)
    assert p.read_bytes() == q.read_bytes()


def test_checkpoint_intake_reads_both_reference_layouts(tmp_path):
    """ckpt.load_ckpt follows utils/commons/ckpt_utils.py:26-67: newest step wins, flat and nested state_dict layouts,
    strict=False drops shape mismatches; the vocoder loader follows hifigan_nsf.py:24-61 (yaml and json flavours)."""
    import yaml
    from stylesinger_amd import ckpt
    from stylesinger_amd.model import StyleSingerHIP
    from stylesinger_amd.vocoder import HifiGanGeneratorHIP
    hp = config.make_hparams(dict(timesteps=4, K_step=4, f0_timesteps=4))
    sd = synth.synth_acoustic_state_dict(hp, 3)
    exp = tmp_path / "exp"
    exp.mkdir()
    old = {k: torch.zeros_like(v) for k, v in sd.items()}
    torch.save({"state_dict": {"model": old}, "optimizer_states": [1, 2, 3]}, exp / "model_ckpt_steps_1000.ckpt")
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}, "global_step": 2000}, exp / "model_ckpt_steps_2000.ckpt")
    assert [os.path.basename(p) for p in ckpt.get_all_ckpts(str(exp))] == ["model_ckpt_steps_2000.ckpt", "model_ckpt_steps_1000.ckpt"]
    m = StyleSingerHIP(None, hparams=hp)
    path = ckpt.load_ckpt(m, str(exp), "model", strict=True)
    assert path.endswith("2000.ckpt")
    got = m.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # nested layout + explicit file + strict=False with one mismatched shape
    bad = dict(sd)
    bad["mel_out.bias"] = torch.zeros(7)
    torch.save({"state_dict": {"model": bad}}, exp / "other.ckpt")
    m2 = StyleSingerHIP(None, hparams=hp)
    ckpt.load_ckpt(m2, str(exp / "other.ckpt"), "model", strict=False)
    assert torch.equal(m2.state_dict()["encoder.embed_tokens.weight"], sd["encoder.embed_tokens.weight"])
    assert ckpt.load_ckpt(m2, str(tmp_path / "nothing"), force=False) is None
    with pytest.raises(FileNotFoundError):
        ckpt.load_ckpt(m2, str(tmp_path / "nothing"))
    n = ckpt.strip(str(exp), str(tmp_path / "infer.pt"))
    assert n == len(sd) and set(torch.load(tmp_path / "infer.pt")["state_dict"]["model"]) == set(sd)
    # vocoder: yaml flavour and json flavour
    cfg = config.make_vocoder_config()
    vsd = synth.synth_vocoder_state_dict(cfg, 3)
    vy = tmp_path / "voc_yaml"
    vy.mkdir()
    yaml.safe_dump(cfg, open(vy / "config.yaml", "w"))
    torch.save({"state_dict": {"model_gen": vsd}}, vy / "model_ckpt_steps_5.ckpt")
    vj = tmp_path / "voc_json"
    vj.mkdir()
    json.dump(cfg, open(vj / "config.json", "w"))
    torch.save({"generator": vsd}, vj / "generator_v1")
    for d in (vy, vj):
        st, c = ckpt.load_vocoder_ckpt(str(d))
        g = HifiGanGeneratorHIP(c)
        g.load_state_dict(st, strict=True)
        assert c["upsample_rates"] == cfg["upsample_rates"]


def test_frontend_host_tables_match_the_oracle():
    """The host-built tables of the GPU mel front end (Slaney filterbank) against oracle/frontend.py, itself pinned by
    librosa's documented known answers; pure numpy, no GPU."""
    import numpy as np
    from oracle import frontend as F
    from stylesinger_amd.frontend import mel_filterbank
    for sr, n_fft, n_mels, fmin, fmax in ((48000, 1024, 80, 20, 24000), (22050, 2048, 128, 0.0, 11025.0), (22050, 1024, 80, 80, 7600)):
        a = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        b = F.mel_basis(sr, n_fft, n_mels, fmin, fmax)
        assert a.shape == b.shape == (n_mels, 1 + n_fft // 2)
        assert np.abs(a - b).max() <= 1e-7


def test_sampler_schedules_and_mode_flags():
    from stylesinger_amd.model import StyleSingerHIP
    m = StyleSingerHIP(None, hparams=config.make_hparams(dict(timesteps=100, K_step=100, f0_timesteps=100)))
    ts = m.ddim_timesteps(50)
    assert ts[0] == 99 and ts[-1] == 0 and len(ts) == 50 and all(a > b for a, b in zip(ts, ts[1:]))
    assert m.ddim_timesteps(1000) == list(range(99, -1, -1)) and m.ddim_timesteps(1) == [0]
    assert not m.bf16 and m.use_wino and m.wino_m == 4 and m.defer_skip and m.fold_skip
    mb = StyleSingerHIP(None, hparams=config.make_hparams(dict(mfma_precision="bf16")))
    assert mb.bf16 and not mb.use_wino and mb.defer_skip and not mb.fold_skip


def test_winograd_form_is_selected_by_the_environment(monkeypatch):
    from stylesinger_amd.model import StyleSingerHIP
    hp = config.make_hparams({})
    monkeypatch.setenv("SS_WINO_M", "2")
    assert StyleSingerHIP(None, hparams=hp).wino_m == 2
    monkeypatch.setenv("SS_WINO_M", "3")
    with pytest.raises(AssertionError):
        StyleSingerHIP(None, hparams=hp)
    monkeypatch.delenv("SS_WINO_M")
    monkeypatch.setenv("SS_WINO", "0")
    assert not StyleSingerHIP(None, hparams=hp).use_wino


def test_f43_transform_matrices_reproduce_the_three_tap_conv():
    """The F(4,3) constants written in csrc/wino43_gate.hip (input rows, weight transform, output combination), restated with numpy:
    four outputs of a 3-tap correlation from six products, for random inputs, to fp64 rounding."""
    rng = np.random.default_rng(0)
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float)
    for _ in range(20):
        r, w = rng.standard_normal(6), rng.standard_normal(3)
        m = (BT @ r) * (G @ w)
        z = np.array([m[0] + m[1] + m[2] + m[3] + m[4], (m[1] - m[2]) + 2 * (m[3] - m[4]), (m[1] + m[2]) + 4 * (m[3] + m[4]),
                      (m[1] - m[2]) + 8 * (m[3] - m[4]) + m[5]])
        ref = np.array([w @ r[o:o + 3] for o in range(4)])
        assert np.abs(z - ref).max() < 1e-12
    # dstep enters component j with the sum of its coefficients over the valid rows; for interior quads: (0, -6, 0, 0, 0, 0)
    assert np.allclose(BT.sum(1), [0, -6, 0, 0, 0, 0])


def test_norm_interp_f0_matches_the_reference_function(golden_dir):
    """pitch.norm_interp_f0 vs the REAL utils/pitch_utils.py:47-62 on float64 / float32 contours with leading, trailing and
    interior unvoiced runs, all-voiced, all-unvoiced and single-voiced-frame inputs: bit-exact (same numpy arithmetic)."""
    from stylesinger_amd import pitch
    g = torch.load(os.path.join(golden_dir, "norm_interp_f0.pt"), weights_only=False)
    hp = config.make_hparams()
    assert hp["pitch_norm"] == g["meta"]["pitch_norm"] and hp["use_uv"] == g["meta"]["use_uv"]
    for key, c in g["cases"].items():
        f0, uv = pitch.norm_interp_f0(c["hz"].numpy(), hp)
        assert f0.dtype == torch.float32 and uv.dtype == torch.float32
        assert torch.equal(uv, c["uv"]), key
        assert torch.equal(f0, c["f0"]), (key, (f0 - c["f0"]).abs().max().item())
        f0t, uvt = pitch.norm_interp_f0(c["hz"], hp)   # torch input, as the reference also accepts
        assert torch.equal(f0t, c["f0"]) and torch.equal(uvt, c["uv"]), key


def test_lds_swizzles_are_conflict_free_in_the_bank_model():
    """tools/lds_sim.py (lane groups / bank functions of MI355X_MICROARCH.md): the fragment reads and staging writes of the 16x16 kernels
    (128-byte fp32 rows with swz16, 64-byte bf16 rows with swz64) cost the conflict-free number of LDS cycles; the unswizzled reads do not
    (the model is not vacuous). On the GPU the fp32 form reads SQ_LDS_BANK_CONFLICT = 0 (profiles/r03_pmc_gate.json)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_sim", os.path.join(root, "tools", "lds_sim.py"))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    assert sim.main() == 0
    c, ideal = sim.cycles("ds_read_b128", lambda lane: (lane & 15) * 128 + ((2 * (lane >> 4)) << 4))
    assert c > ideal


def test_forward_signature_equals_the_reference_operator_surface():
    """The drop-in boundary: StyleSingerHIP.forward has the parameters of the REAL StyleSinger.forward
    (/root/reference/modules/StyleSinger/stylesinger.py:119-121) - same names, order, defaults, **kwargs - and the vocoder plugin /
    entry-point surface keeps the reference's method names. Imports the unmodified reference (skipped where it is not mounted)."""
    import inspect
    from oracle import refimport
    if not refimport.available():
        pytest.skip("/root/reference is not mounted on this box")
    from stylesinger_amd.model import StyleSingerHIP
    from stylesinger_amd.vocoder import HifiGAN, REGISTERED_VOCODERS
    ref = refimport.load()["StyleSinger"]
    want, got = inspect.signature(ref.forward), inspect.signature(StyleSingerHIP.forward)
    assert [(p.name, p.kind, p.default) for p in want.parameters.values()] == [(p.name, p.kind, p.default) for p in got.parameters.values()]
    assert "HifiGAN_NSF" in REGISTERED_VOCODERS and REGISTERED_VOCODERS["HifiGAN_NSF"] is HifiGAN
    assert list(inspect.signature(HifiGAN.spec2wav).parameters)[:2] == ["self", "mel"]
    # every output the goldens recorded from the reference's returned dict is a key the HIP forward documents it returns (the GPU tests compare
    # the values: tests/test_gpu_parity.py::test_acoustic_hip_matches_reference_golden)
    import torch
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "acoustic_tiny_s4.pt"), weights_only=False)["out"]
    src = inspect.getsource(StyleSingerHIP.forward) + inspect.getsource(StyleSingerHIP)
    ALIAS = {"encoder_out": "encoder_out_text"}   # the reference overwrites ret['encoder_out'] later in forward; the stage value is kept under this name
    for k in gold:
        assert f'ret["{ALIAS.get(k, k)}"]' in src or f"'{ALIAS.get(k, k)}'" in src, k


def test_launch_planning_functions_of_the_library_run_without_a_gpu():
    """The host-side launch planners of the C-ABI (no kernel launch): tiling picks, split-K slice count, size of the addend re-layout, the tuning
    knob table. Without a device the CU count falls back to MI355X's 256, so the BASELINE shapes give their documented answers."""
    l = lib.load()
    # fp32 F(4,3) gate tiling: C2 mel / f0 pair, one 4 s utterance (16-quad tiles), the C4 shape (32x32x2 kernel)
    assert l.ss_wino43_gate16_pick(8, 1500, 512, 2) == 2 and l.ss_wino43_gate16_pick(16, 1500, 384, 1) == 3
    assert l.ss_wino43_gate16_pick(1, 750, 512, 4) == 1 and l.ss_wino43_gate16_pick(32, 5625, 512, 4) == 0
    # addend in fetch order: [q tiles][n tiles][MT * 4096 floats]; C2 mel at d = 2: 375 quads -> 376 (groups of d) -> 12 tiles of 32 quads per item
    assert l.ss_gate16_tiled_floats(8, 1500, 512, 2, 2) == 8 * 12 * 8 * 2 * 4096
    assert l.ss_gate16_tiled_floats(8, 1500, 512, 2, 0) == -1 and l.ss_gate16_tiled_floats(8, 1500, 500, 2, 2) == -1
    # split-K of the skip GEMM: only for launches that leave CUs idle
    assert l.ss_gemm16_ksplit_pick(1, 750, 256, 5120) == 5 and l.ss_gemm16_ksplit_pick(8, 1500, 256, 5120) == 1 and l.ss_gemm16_ksplit_pick(1, 750, 256, 256) == 1
    # residual-projection row tile: 96 rows at C2 (16 row tiles per 8 s item), 32 rows for one short utterance
    assert l.ss_gemm16_pick(8, 1500, 256) == 6 and l.ss_gemm16_pick(1, 750, 256) == 2
    # knob table: every documented key round-trips, bad values are refused with a message
    for key, val in ((b"e16", 0), (b"mel_tail", 0), (b"gate16", 3), (b"htile", 128), (b"voc_wino_max_mb", 7), (b"wino_tn", 2)):
        before = l.ss_get_tuning(key)
        assert before >= 0
        assert l.ss_set_tuning(key, val) == 0 and l.ss_get_tuning(key) == val
        assert l.ss_set_tuning(key, before) == 0
    assert l.ss_set_tuning(b"htile", 96) != 0 and b"htile" in l.ss_last_error()
    assert l.ss_get_tuning(b"nope") < 0
    # round 6: the fused residual layer of the fp16x2 mel stack. Size rule = four rounds of 128-row tiles per CU (256 CUs without a device): the
    # BASELINE configs[3] shape qualifies (32 x 44 = 1408 tiles), C2's does not; the knob value 2 lifts the size rule (parity tests), never the shape rules
    assert l.ss_get_tuning(b"layer512") == 1 and l.ss_get_tuning(b"layer512_tail") == 1
    assert l.ss_layer512_ok(32, 5625, 256, 8, 20 * 256 * 2) == 1 and l.ss_layer512_ok(8, 1500, 256, 8, 20 * 256 * 2) == 0
    assert l.ss_layer512_ok(32, 5625, 192, 8, 20 * 192 * 2) == 0 and l.ss_layer512_ok(32, 5625, 256, 16, 20 * 256 * 2) == 0
    assert l.ss_set_tuning(b"layer512", 2) == 0 and l.ss_layer512_ok(1, 100, 256, 8, 512) == 1 and l.ss_layer512_ok(1, 100, 192, 8, 512) == 0
    assert l.ss_set_tuning(b"layer512", 1) == 0 and l.ss_set_tuning(b"layer512", 3) != 0
    # buffer sizes: 44 tiles per 30 s item; addend 128 x 512 floats, stream 128 x 256 fp32, H 128 x 256 fp16 per tile
    assert l.ss_layer512_addend_floats(32, 5625) == 1408 * 128 * 512 and l.ss_layer512_stream_bytes(32, 5625) == 1408 * 128 * 256 * 2
    assert l.ss_layer512_h_elems(32, 5625) == 1408 * 128 * 256 and l.ss_layer512_h_elems(1, 1) == 128 * 256


def test_gate128_index_math_against_a_tagged_lds_image(tmp_path):
    """gate128_kernel (the fp16x2 gate on 256 x 128 tiles, two workgroups per CU) takes all of its DMA / fragment / epilogue addresses from
    stylesinger_amd/csrc/gate128_layout.h; tools/layout_check_gate128.cpp compiles the SAME header on the host, replays every LDS-DMA piece
    into a tagged image and looks every access of the kernel up in it (coverage, right element, no bank conflicts, right output channel)."""
    exe = str(tmp_path / "layout_check_gate128")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "stylesinger_amd", "csrc"), os.path.join(ROOT, "tools", "layout_check_gate128.cpp"),
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-2000:]


def test_fp16q4_weight_pack_round_trips_through_the_kernel_side_decoder():
    """lib.pack_gate_q4 (the operand of ss_gemm_bf16_gate128q: hi plane fp16 of w * 2^8, lo plane as block-scaled fp4 in the kernel's lane
    order) against lib.unpack_gate_q4, which reads the pack the way the kernel addresses it (weight line of the pair's odd step, slots 4 + h
    and 6): every term lands where the kernel looks for it, the fp4 image is within half a grid step of lo, even lines carry nothing."""
    g = torch.Generator().manual_seed(5)
    Wp = torch.randn(64, 768, generator=g) * 0.05
    Wp[3] = 0.0                                   # an all-zero row: scale byte must not produce NaN / inf
    pack, lo_q = lib.pack_gate_q4(Wp, shift=8)
    assert pack.dtype == torch.float16 and pack.shape == (64, 1536)
    hi, lo_dec = lib.unpack_gate_q4(pack)
    ws = Wp * 256.0
    assert torch.equal(hi, ws.half().float())
    assert torch.equal(lo_dec, lo_q), "the decoder (kernel's view) and the packer's own dequantised values agree exactly"
    lo = ws - ws.half().float()
    blocks = lib.gate128q_kindex()
    for p in (0, 5, 11):
        for h in (0, 1):
            idx = blocks[p, h]
            amax = lo[:, idx].abs().amax(dim=1)
            err = (lo_q[:, idx] - lo[:, idx]).abs().amax(dim=1)
            assert bool((err <= 0.26 * amax + 1e-30).all())   # half the grid's largest step (4 -> 6) relative to a block maximum in [4, 8)
    wb = pack.view(torch.uint8).view(64, 24, 128)
    even_lines = [(S % 3) * 8 + S // 3 for S in range(0, 24, 2)]
    assert int(wb[:, even_lines, 64:].max()) == 0
    assert torch.isfinite(lo_dec).all() and float(lo_dec[3].abs().max()) == 0.0


def test_fp16q4_skip_weight_pack_round_trips_through_the_kernel_side_decoder():
    """lib.pack_skip_q4 (operand of ss_gemm_bf16_tile256q: pairs of consecutive 32-channel chunks) against lib.unpack_skip_q4, the kernel's view."""
    g = torch.Generator().manual_seed(6)
    Wp = torch.randn(32, 512, generator=g) * 0.03
    pack, lo_q = lib.pack_skip_q4(Wp, shift=8)
    hi, lo_dec = lib.unpack_skip_q4(pack)
    ws = Wp * 256.0
    assert torch.equal(hi, ws.half().float()) and torch.equal(lo_dec, lo_q)
    lo = ws - ws.half().float()
    tab = lib.tile256q_kindex(8)
    assert sorted(tab.reshape(-1).tolist()) == list(range(512))
    for p in (0, 7):
        for h in (0, 1):
            idx = tab[p, h]
            assert bool(((lo_q[:, idx] - lo[:, idx]).abs().amax(dim=1) <= 0.26 * lo[:, idx].abs().amax(dim=1) + 1e-30).all())
    wb = pack.view(torch.uint8).view(32, 16, 128)
    assert int(wb[:, 0::2, 64:].max()) == 0     # even chunks carry nothing in their second half


def test_speaker_encoder_partial_slicing_known_answers():
    """stylesinger_amd.speaker.compute_partial_slices = resemblyzer.VoiceEncoder.compute_partial_slices (un-vendored dependency of the reference,
    inference/StyleSinger.py:100-104): rate 1.3 partials per second -> one every round(16000 / 1.3 / 160) = 77 frames; hand-computed cases."""
    from stylesinger_amd import speaker
    w, m = speaker.compute_partial_slices(48000)                 # 3 s: n_frames 301, steps 219 -> starts 0, 77, 154; last coverage 0.9125
    assert [(s.start, s.stop) for s in m] == [(0, 160), (77, 237), (154, 314)]
    assert (w[-1].start, w[-1].stop) == (24640, 50240)
    _, m = speaker.compute_partial_slices(16000)                 # 1 s: a single (padded) partial
    assert [(s.start, s.stop) for s in m] == [(0, 160)]
    _, m = speaker.compute_partial_slices(16000 * 2 + 1000)      # n_frames 207, steps 125 -> starts 0, 77; last coverage (33000 - 12320) / 25600 = 0.81
    assert [(s.start, s.stop) for s in m] == [(0, 160), (77, 237)]
    _, m = speaker.compute_partial_slices(16000 * 2 - 2000)      # n_frames 188, steps 106 -> starts 0, 77; last coverage (30000 - 12320) / 25600 = 0.69 < 0.75: dropped
    assert [(s.start, s.stop) for s in m] == [(0, 160)]
    _, m = speaker.compute_partial_slices(16000 * 2 - 2000, min_coverage=0.5)
    assert len(m) == 2


def test_token_text_encoder_equals_the_reference_class(golden_dir, tmp_path):
    """`build_token_encoder(phone_set.json)` / `TokenTextEncoder` (utils/text/text_encoder.py:107-147,257-259; used at inference/StyleSinger.py:28,96
    and as the model's dictionary) - bit-exact against what the REAL class returned in the build container (tests/golden/token_encoder.json,
    `python -m oracle.gen_golden --round6`): example_run's phoneme list with ZH_checkpoint_phone_set.json, out-of-vocabulary and reserved
    tokens, whitespace, decode with padding / EOS stripping, the reversed form. Where /root/reference is mounted the two classes are also run
    side by side on random token strings."""
    import json
    import random
    from stylesinger_amd.text_encoder import TokenTextEncoder, build_token_encoder
    gold = json.load(open(os.path.join(golden_dir, "token_encoder.json")))
    pf = tmp_path / "phone_set.json"
    pf.write_text(json.dumps(gold["phone_set"]))
    enc = build_token_encoder(str(pf))
    assert len(enc) == enc.vocab_size == gold["vocab_size"] == 61
    assert (enc.pad(), enc.eos(), enc.unk(), enc.seg()) == (gold["pad"], gold["eos"], gold["unk"], gold["seg"])
    assert [enc.id_to_token[i] for i in range(len(enc))] == gold["id_to_token"]
    for case in gold["encode"]:
        assert enc.encode(case["s"]) == case["ids"], case["s"]
    assert enc.encode(" ".join(gold["example"]["ph"])) == gold["encode"][0]["ids"]       # what preprocess_input computes (inference/StyleSinger.py:96)
    d = gold["decode"]
    assert enc.decode(d[0]["ids"]) == d[0]["s"] and enc.decode(d[1]["ids"], strip_padding=True) == d[1]["s"]
    assert enc.decode(d[2]["ids"], strip_eos=True) == d[2]["s"] and enc.decode(d[3]["ids"]) == d[3]["s"]
    assert enc.sil_phonemes() == gold["sil"]
    rev = TokenTextEncoder(None, vocab_list=gold["phone_set"] + ["|"], replace_oov=None, reverse=True)
    r = gold["reverse"]
    assert (len(rev), rev.seg(), rev.encode("zh i uan"), rev.decode(rev.encode("zh i uan"))) == (r["vocab_size"], r["seg"], r["ids"], r["s"])
    with pytest.raises(KeyError):
        rev.encode("notaphone")
    vf = tmp_path / "vocab.txt"
    enc.store_to_file(str(vf))
    again = TokenTextEncoder(str(vf))
    assert again.token_to_id == enc.token_to_id
    from oracle import refimport
    if refimport.available():
        refimport.load()
        from utils.text.text_encoder import build_token_encoder as ref_build
        ref = ref_build(os.path.join(refimport.REF, "ZH_checkpoint_phone_set.json"))
        rng = random.Random(7)
        pool = gold["phone_set"] + ["<pad>", "<EOS>", "<UNK>", "|", "xx", "a1"]
        for _ in range(200):
            s_ = " ".join(rng.choice(pool) for _ in range(rng.randrange(0, 40)))
            ids = ref.encode(s_)
            assert enc.encode(s_) == ids
            assert enc.decode(ids) == ref.decode(ids)


def test_entry_point_host_logic_without_a_gpu(tmp_path, golden_dir):
    """Host-side contracts of the reference-shaped entry point (inference/StyleSinger.py:94-137,175-331) that need no device: the explicit opt-out
    of `trim_long_silences` (the reference always trims, audio.py:36-38), `loud_norm` refused instead of ignored (utils/audios/__init__.py:55-59
    is pyloudnorm), bytes paths, the example score shipped with the package, `save_wav` = utils/audio.py:12-17."""
    import json
    import wave
    import warnings
    import numpy as np
    from stylesinger_amd import vadtrim, writer
    from stylesinger_amd.infer import StyleSingerInfer
    inf = StyleSingerInfer.__new__(StyleSingerInfer)       # no device: only the host helpers are exercised
    if not vadtrim.have_webrtcvad():
        with pytest.raises(ImportError, match="vad_flags=False"):
            inf._resolve_vad(None)
    else:
        assert inf._resolve_vad(None) == "webrtc"
    StyleSingerInfer._warned_untrimmed = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert inf._resolve_vad(False) is None and inf._resolve_vad(False) is None
    assert len([x for x in w if "trim_long_silences skipped" in str(x.message)]) == 1, "warns once"
    f = inf._resolve_vad([1, 0, 1])
    assert f.shape == (1, 3)
    with pytest.raises(NotImplementedError, match="loud_norm"):
        StyleSingerInfer(dict(loud_norm=True), device="cuda")      # refused before anything touches a device
    # a bytes path reaches wave.open as a path, not as "b'...'"
    p = tmp_path / "a.wav"
    pcm = (np.arange(-500, 500) * 30).astype("<i2")
    with wave.open(str(p), "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(48000)
        wf.writeframes(pcm.tobytes())
    a = StyleSingerInfer._load_wav(str(p), 48000)
    b = StyleSingerInfer._load_wav(os.fsencode(str(p)), 48000)
    assert np.array_equal(a, b) and np.array_equal(a, pcm.astype(np.float32) / 32768.0)
    with pytest.raises(ValueError, match="16-bit PCM at 44100"):
        StyleSingerInfer._load_wav(str(p), 44100)
    # the example score = the reference's example_run input (fixture generated from the reference by oracle/gen_golden.py --round6)
    gold = json.load(open(os.path.join(golden_dir, "token_encoder.json")))
    ex = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stylesinger_amd", "example_input.json")))
    for k in ("name", "ph", "note", "note_dur", "note_type"):
        assert ex[k] == gold["example"][k], k
    # save_wav: wav * 32767 truncated toward zero, as numpy's astype(int16) does in the reference
    x = np.array([0.0, 0.5, -0.5, 0.99999, -1.0, 1e-5], dtype=np.float32)
    writer.save_wav(x, str(tmp_path / "o.wav"), 48000)
    with wave.open(str(tmp_path / "o.wav"), "rb") as wf:
        got = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2")
    assert np.array_equal(got, (x * 32767).astype(np.int16)) and x[1] == 0.5, "input not modified"
