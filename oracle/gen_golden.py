"""Generate tests/golden/*.pt by running the REAL reference (AaronZ345/StyleSinger at /root/reference).

TEST INFRASTRUCTURE ONLY; runs in the build container (the reference does not exist on the GPU box).
    python -m oracle.gen_golden            # regenerate every fixture
The reference modules are imported unmodified (oracle/refimport.py); weights come from
stylesinger_amd.synth (loaded through the reference's own load_state_dict(strict=True)); every
torch.rand*/randn* draw is served from a seeded NoiseTape so the HIP path and the CPU restatement can
consume the identical noise.  One reference instance per step-count (the schedule buffers are sized at
construction, modules/StyleSinger/stylesinger.py:69-73,101-113).
"""
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport  # noqa: E402
from stylesinger_amd import config, synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@contextlib.contextmanager
def tape_rng(tape):
    """Route the reference's global-RNG draws through the tape (order-preserving)."""
    o = (torch.randn, torch.rand, torch.randn_like, torch.rand_like)

    def _shape(args):
        return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else tuple(args)

    torch.randn = lambda *a, **k: tape.randn(*_shape(a))
    torch.rand = lambda *a, **k: tape.rand(*_shape(a))
    torch.randn_like = lambda x, **k: tape.randn(*x.shape)
    torch.rand_like = lambda x, **k: tape.rand(*x.shape)
    try:
        yield
    finally:
        torch.randn, torch.rand, torch.randn_like, torch.rand_like = o


def build_reference(steps_mel, steps_f0, seed=1234, hp_over=None):
    """`hp_over`: further hparams overrides (K_step < timesteps, decoder='prodiff', schedule_type ...). The reference reads
    the GLOBAL hparams dict at construction, so the keys a previous case may have changed are always reset here."""
    over = dict(timesteps=steps_mel, K_step=steps_mel, f0_timesteps=steps_f0, decoder="diffsinger", schedule_type="linear",
                timescale=1)
    over.update(hp_over or {})
    R = refimport.load(over)
    hp = config.make_hparams(over)
    model = R["StyleSinger"](refimport.FakeDict(hp["vocab_size"]))
    sd = synth.synth_acoustic_state_dict(hp, seed)
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model, hp, sd


def run_acoustic_case(name, B, T, Tp, Tr, steps_mel, steps_f0, give_mel2ph=True, seed=1234, keep_stages=True, hp_over=None,
                      keep_keys=None):
    """`keep_keys`: store only these outputs (the full-size C4 case keeps the fixture at ~2 MB)."""
    model, hp, sd = build_reference(steps_mel, steps_f0, seed, hp_over)
    batch = synth.synth_batch(B, T, Tp, Tr, hp, seed)
    tape = synth.NoiseTape(seed + 1)
    stages = {}
    hooks = []
    if keep_stages:
        def grab(key, fn=lambda o: o):
            def h(mod, inp, out):
                stages[key] = fn(out).detach().clone()
            return h
        hooks.append(model.encoder.register_forward_hook(grab("encoder_out")))
        hooks.append(model.style_extractor.encoder.register_forward_hook(grab("style_pre_rq")))
        hooks.append(model.style_extractor.register_forward_hook(grab("style_rq", lambda o: o[0])))
        if hasattr(model, "ln_proj"):
            hooks.append(model.ln_proj.register_forward_hook(grab("diff_cond")))
        hooks.append(model.decoder.register_forward_hook(grab("decoder_out")))
    with torch.no_grad(), tape_rng(tape):
        ret = model(batch["txt_tokens"], mel2ph=batch["mel2ph"] if give_mel2ph else None, spk_embed=batch["spk_embed"],
                    emo_embed=batch["emo_embed"], ref_mels=batch["ref_mels"], ref_f0=batch["ref_f0"], global_steps=320000,
                    infer=True, note=batch["note"], note_dur=batch["note_dur"], note_type=batch["note_type"])
    for h in hooks:
        h.remove()
    keys = keep_keys or ["mel2ph", "style", "pitch_pred", "f0_denorm", "decoder_inp", "mel_out", "dur"]
    out = {k: ret[k].detach().clone() for k in keys if k in ret and torch.is_tensor(ret[k])}
    if "dur_choice" in ret:
        out["dur_choice"] = ret["dur_choice"].clone()
    out.update(stages)
    meta = dict(B=B, T=T, Tp=Tp, Tr=Tr, steps_mel=steps_mel, steps_f0=steps_f0, give_mel2ph=give_mel2ph, seed=seed,
                tape_seed=seed + 1, tape_log=tape.log, hp_over=dict(hp_over or {}))
    torch.save(dict(meta=meta, out=out), os.path.join(GOLD, name + ".pt"))
    print(f"[gen_golden] {name}: T_out={out['mel_out'].shape[1]} draws={len(tape.log)} keys={sorted(out)}")


def run_vocoder_case(name, B, T, seed=1234):
    R = refimport.load()
    cfg = config.make_vocoder_config()
    gen = R["HifiGanGenerator"](cfg)
    vsd = synth.synth_vocoder_state_dict(cfg, seed)
    gen.load_state_dict(vsd, strict=True)
    gen.remove_weight_norm()
    gen.eval()
    g = torch.Generator().manual_seed(seed + 7)
    mel = (torch.randn(B, T, 80, generator=g) * 0.8 - 3.0).clamp(-6, 1.5)
    f0 = 220.0 + 80.0 * torch.sin(torch.arange(T)[None, :] / 5.0 + torch.arange(B)[:, None])
    f0[:, T // 3: T // 3 + max(T // 6, 1)] = 0.0  # an unvoiced run
    tape = synth.NoiseTape(seed + 2)
    grabbed = {}
    hook = gen.m_source.register_forward_hook(lambda m, i, o: grabbed.__setitem__("har", o[0].detach().clone()))
    with torch.no_grad(), tape_rng(tape):
        wav = gen(mel.transpose(1, 2), f0)
    hook.remove()
    torch.save(dict(meta=dict(B=B, T=T, seed=seed, tape_seed=seed + 2, tape_log=tape.log),
                    inp=dict(mel=mel, f0=f0), out=dict(wav=wav[:, 0].clone(), har=grabbed["har"][:, :, 0].clone())),
               os.path.join(GOLD, name + ".pt"))
    print(f"[gen_golden] {name}: wav {tuple(wav.shape)} draws={len(tape.log)}")


def run_plms_case(name, T, steps_mel, interval, seed=1234, k_step=None):
    """The reference's PLMS sampler (GaussianDiffusion.p_sample_plms, shallow_diffusion_tts.py:165-197) driven exactly as
    GaussianDiffusion.forward drives it under hparams['pndm_speedup'] (:239-260), on the StyleSinger model's own `postdiff`
    (DiffusionDecoder inherits the method). B = 1: the reference's `max(t - interval, 0)` only works for one utterance."""
    from collections import deque
    model, hp, sd = build_reference(steps_mel, 2, seed, dict(K_step=k_step) if k_step else None)
    pd = model.postdiff
    g = torch.Generator().manual_seed(seed + 11)
    coarse = (torch.randn(1, T, 80, generator=g) * 0.8 - 3.0).clamp(-6, 0.5)
    cond = torch.randn(1, T, 256, generator=g) * 0.5
    tape = synth.NoiseTape(seed + 3)
    with torch.no_grad(), tape_rng(tape):
        t = pd.K_step
        fs2 = pd.norm_spec(coarse).transpose(1, 2)[:, None, :, :]
        x = pd.q_sample(x_start=fs2, t=torch.tensor([t - 1]).long())
        pd.noise_list = deque(maxlen=4)
        for i in reversed(range(0, t, interval)):
            x = pd.p_sample_plms(x, torch.full((1,), i, dtype=torch.long), interval, cond.transpose(1, 2))
        mel = pd.denorm_spec(x[:, 0].transpose(1, 2))
    torch.save(dict(meta=dict(T=T, steps_mel=steps_mel, interval=interval, seed=seed, tape_seed=seed + 3, tape_log=tape.log,
                              k_step=k_step or steps_mel),
                    inp=dict(coarse_mel=coarse, cond=cond), out=dict(mel_out=mel.clone())), os.path.join(GOLD, name + ".pt"))
    print(f"[gen_golden] {name}: mel {tuple(mel.shape)} draws={len(tape.log)}")


def run_emotion_case(name, n_partials, seed=1234):
    """The reference emotion encoder (data_gen/tts/emotion/model.py:11-78) on synthetic 40-mel partials, driven as
    inference.py:39-53,139-151 drives it: `inference()` on the [P,160,40] batch, mean over partials, L2 normalise."""
    refimport.load()
    import numpy as np
    from data_gen.tts.emotion.model import EmotionEncoder
    enc = EmotionEncoder(torch.device("cpu"), torch.device("cpu"))
    sd = synth.synth_emotion_state_dict(seed)
    enc.load_state_dict(sd, strict=True)
    enc.eval()
    frames = synth.synth_emotion_frames(n_partials, seed=seed)
    with torch.no_grad():
        partial = enc.inference(frames)
        full = enc(frames)
    raw = np.mean(partial.numpy(), axis=0)
    embed = raw / np.linalg.norm(raw, 2)
    slices = {}
    try:  # the partial-slicing rule (inference.py:56-108), pinned for a few utterance lengths
        from data_gen.tts.emotion import inference as einf
        for n in (1000, 25600, 40000, 100000, 128000, 131111):
            w, m = einf.compute_partial_slices(n)
            slices[n] = ([(int(x.start), int(x.stop)) for x in w], [(int(x.start), int(x.stop)) for x in m])
    except Exception as e:  # noqa: BLE001
        print("[gen_golden] compute_partial_slices not importable:", e)
    torch.save(dict(meta=dict(n_partials=n_partials, seed=seed, keys=[[k, list(v.shape)] for k, v in enc.state_dict().items()],
                              partial_slices=slices),
                    out=dict(partial_embeds=partial.clone(), embed=torch.from_numpy(embed), forward_embeds=full.clone())),
               os.path.join(GOLD, name + ".pt"))
    print(f"[gen_golden] {name}: partial {tuple(partial.shape)}")


def run_pitch_case(name="norm_interp_f0", seed=1234):
    """The reference's own `norm_interp_f0` (utils/pitch_utils.py:47-62, called by inference/StyleSinger.py:152) on tracker-like
    contours: float64 (parselmouth) and float32 inputs, leading / trailing / interior unvoiced runs, all voiced, all unvoiced."""
    R = refimport.load()
    from utils.pitch_utils import norm_interp_f0
    hp = dict(R["hparams"])
    cases = {}
    def add(key, hz):
        f0, uv = norm_interp_f0(hz.numpy().copy(), hp)
        cases[key] = dict(hz=hz.clone(), f0=f0.clone(), uv=uv.clone())
    add("t300_f64", synth.synth_f0_hz(0, 300, seed))
    add("t1500_f64", synth.synth_f0_hz(1, 1500, seed))
    add("t1500_f32", synth.synth_f0_hz(1, 1500, seed).float())
    h = synth.synth_f0_hz(2, 700, seed).float()
    h[:37] = 0.0
    h[-51:] = 0.0
    add("t700_edges_f32", h)
    add("t64_all_voiced_f32", synth.synth_f0_hz(3, 64, seed, unvoiced=0.0).float().clamp_min(100.0))
    add("t40_all_unvoiced_f32", torch.zeros(40))
    z = torch.zeros(50)
    z[17] = 220.0
    add("t50_one_voiced_f32", z)
    torch.save(dict(meta=dict(seed=seed, pitch_norm=hp["pitch_norm"], use_uv=hp["use_uv"]), cases=cases), os.path.join(GOLD, name + ".pt"))
    print(f"[gen_golden] {name}: {sorted(cases)}")


def dump_extra_param_specs():
    """Pin the ProDiff-decoder and emotion-encoder state_dict contracts (names + shapes) next to the main ones."""
    import json
    model, hp, sd = build_reference(8, 2, 1234, dict(decoder="prodiff", schedule_type="vpsde"))
    from data_gen.tts.emotion.model import EmotionEncoder
    enc = EmotionEncoder(torch.device("cpu"), torch.device("cpu"))
    path = os.path.join(GOLD, "param_spec.json")
    d = json.load(open(path))
    d["acoustic_prodiff"] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    d["emotion"] = [[k, list(v.shape)] for k, v in enc.state_dict().items()]
    json.dump(d, open(path, "w"))
    print(f"[gen_golden] param_spec.json: +acoustic_prodiff ({len(d['acoustic_prodiff'])}) +emotion ({len(d['emotion'])})")


def round2_cases():
    # BASELINE config 4's schedule: 1000 mel steps (sqrt_recip coefficients reach ~3e6, shallow_diffusion_tts.py:99-119)
    run_acoustic_case("acoustic_t32_mel1000", B=1, T=32, Tp=4, Tr=32, steps_mel=1000, steps_f0=4, keep_stages=False)
    # PLMS with a shallow depth K_step < timesteps
    run_plms_case("plms_t32_k12of20_i3", T=32, steps_mel=20, interval=3, k_step=12)
    # ProDiff teacher decoder (modules/diff/prodiff.py:59-221): 8 steps, x0-prediction, both schedules
    run_acoustic_case("prodiff_t40_vpsde", B=1, T=40, Tp=5, Tr=36, steps_mel=8, steps_f0=3, keep_stages=False,
                      hp_over=dict(decoder="prodiff", schedule_type="vpsde"))
    run_acoustic_case("prodiff_b2_t32_linear", B=2, T=32, Tp=4, Tr=30, steps_mel=8, steps_f0=3, keep_stages=False,
                      hp_over=dict(decoder="prodiff", schedule_type="linear"))
    run_emotion_case("emotion_p5", n_partials=5)
    dump_extra_param_specs()


def round3_cases():
    run_pitch_case("norm_interp_f0")


def run_vad_trim_case(name="vad_trim", seed=1234):
    """The reference's own `trim_long_silences` (data_gen/tts/emotion/audio.py:58-100) with the third-party decision INJECTED: `webrtcvad` is stubbed
    by a Vad whose `is_speech` replays preset flags (and checks the PCM window it is handed), so everything around the decision - window cut,
    moving average + np.round, binary_dilation, compaction - is the real code. Cases: random flags at several densities, all voiced, all unvoiced,
    isolated voiced windows, lengths that are not a multiple of the 480-sample window."""
    import numpy as np
    refimport.load()
    import webrtcvad

    class Vad:
        flags, seen = [], []

        def __init__(self, mode=3):
            assert mode == 3
            self.i = 0

        def is_speech(self, buf, sample_rate):
            assert sample_rate == 16000 and len(buf) == 960
            v = Vad.flags[self.i]
            self.i += 1
            return bool(v)
    webrtcvad.Vad = Vad
    from data_gen.tts.emotion import audio as A
    rng = np.random.default_rng(seed)
    cases = {}

    def add(key, n, flags_fn):
        wav = (rng.standard_normal(n) * 0.1).astype(np.float32)
        nw = n // 480
        Vad.flags = [int(v) for v in flags_fn(nw)]
        out = A.trim_long_silences(wav)
        cases[key] = dict(wav=torch.from_numpy(wav), flags=torch.tensor(Vad.flags, dtype=torch.uint8), out=torch.from_numpy(np.ascontiguousarray(out)))
    add("dense", 16000 * 2 + 123, lambda nw: rng.random(nw) > 0.3)
    add("half", 16000 * 3, lambda nw: rng.random(nw) > 0.5)
    add("sparse", 16000 * 3 + 479, lambda nw: rng.random(nw) > 0.85)
    add("runs", 16000 * 4, lambda nw: (np.arange(nw) // 9) % 3 == 0)
    add("all_voiced", 480 * 20 + 7, lambda nw: np.ones(nw))
    add("all_unvoiced", 480 * 20, lambda nw: np.zeros(nw))
    add("one_voiced_run", 480 * 40, lambda nw: (np.arange(nw) >= 15) & (np.arange(nw) < 21))
    add("short", 480 * 3 + 100, lambda nw: np.ones(nw))
    torch.save(dict(meta=dict(seed=seed, window=480, avg_width=8, max_silence=6), cases=cases), os.path.join(GOLD, name + ".pt"))
    print(f"[gen_golden] {name}: " + ", ".join(f"{k} {len(v['wav'])}->{len(v['out'])}" for k, v in cases.items()))


def round5_cases():
    # BASELINE configs[3] AS SPECIFIED, one item: T = 5625 (30 s) AND 1000 mel steps (+ 2 x 100 f0 steps), the REAL reference in fp32.
    # ~1.5e14 flop on the CPU (tens of minutes); only the outputs the C4 parity test compares are kept.
    run_acoustic_case("acoustic_t5625_mel1000", B=1, T=5625, Tp=105, Tr=1500, steps_mel=1000, steps_f0=100, keep_stages=False,
                      keep_keys=["mel_out", "pitch_pred", "f0_denorm"])


def run_token_encoder_case(name="token_encoder"):
    """The REAL `TokenTextEncoder` / `build_token_encoder` (utils/text/text_encoder.py:107-147,257-259) on the phone set the checkpoint ships with
    (ZH_checkpoint_phone_set.json) and on `example_run`'s own input (inference/StyleSinger.py:181-321; the dict literal is read with `ast`, the
    file itself cannot be imported here: it needs resemblyzer / parselmouth / librosa). -> tests/golden/token_encoder.json (data: the phone list,
    the example score, and the ids / strings the reference class returns) + stylesinger_amd/example_input.json (the example score alone)."""
    import ast
    import json
    refimport.load()
    from utils.text.text_encoder import TokenTextEncoder, build_token_encoder
    phone_file = os.path.join(refimport.REF, "ZH_checkpoint_phone_set.json")
    phones = json.load(open(phone_file))
    enc = build_token_encoder(phone_file)
    tree = ast.parse(open(os.path.join(refimport.REF, "inference", "StyleSinger.py")).read())
    example = None
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == "example_run":
            for st in ast.walk(node):
                if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", None) == "inp":
                    example = ast.literal_eval(st.value)
    assert example and len(example["ph"]) == len(example["note"]) == len(example["note_dur"]) == len(example["note_type"])
    probes = [" ".join(example["ph"]), "zh i  breathe   _NONE", "a b notaphone c <pad> <EOS> | ", "", "  uang\tvn\nve  "]
    out = dict(phone_set=phones, example=example, vocab_size=len(enc), pad=enc.pad(), eos=enc.eos(), unk=enc.unk(), seg=enc.seg(),
               id_to_token=[enc.id_to_token[i] for i in range(len(enc))],
               encode=[dict(s=s_, ids=enc.encode(s_)) for s_ in probes], sil=enc.sil_phonemes())
    ids = enc.encode(probes[0])
    out["decode"] = [dict(ids=ids, s=enc.decode(ids)), dict(ids=ids[:5] + [0, 7, 1, 9], s=enc.decode(ids[:5] + [0, 7, 1, 9], strip_padding=True)),
                     dict(ids=ids[:5] + [1, 7, 0, 9], s=enc.decode(ids[:5] + [1, 7, 0, 9], strip_eos=True), strip_eos=True), dict(ids=[3, 999, 4], s=enc.decode([3, 999, 4]))]
    rev = TokenTextEncoder(None, vocab_list=phones + ["|"], replace_oov=None, reverse=True)
    out["reverse"] = dict(vocab_size=len(rev), seg=rev.seg(), ids=rev.encode("zh i uan"), s=rev.decode(rev.encode("zh i uan")))
    json.dump(out, open(os.path.join(GOLD, name + ".json"), "w"), indent=0)
    pkg = os.path.join(os.path.dirname(GOLD), "..", "stylesinger_amd", "example_input.json")
    json.dump(dict(source="inference/StyleSinger.py:186-321 (example_run's input dict)", **example), open(os.path.normpath(pkg), "w"), indent=0)
    print("wrote", name + ".json", "and stylesinger_amd/example_input.json:", len(phones), "phones,", len(example["ph"]), "example phonemes")


def main():
    os.makedirs(GOLD, exist_ok=True)
    if "--round6" in sys.argv:
        run_token_encoder_case()
        return
    if "--vad" in sys.argv:
        run_vad_trim_case()
        return
    if "--round5" in sys.argv:
        round5_cases()
        run_vad_trim_case()
        return
    if "--only-plms" in sys.argv:
        run_plms_case("plms_t40_k20_i3", T=40, steps_mel=20, interval=3)
        run_plms_case("plms_t24_k12_i4", T=24, steps_mel=12, interval=4)
        return
    if "--round2" in sys.argv:
        round2_cases()
        return
    if "--round3" in sys.argv:
        round3_cases()
        return
    if "--emotion" in sys.argv:
        run_emotion_case("emotion_p5", n_partials=5)
        dump_extra_param_specs()
        return
    run_acoustic_case("acoustic_tiny_s4", B=1, T=48, Tp=6, Tr=40, steps_mel=4, steps_f0=4)
    run_acoustic_case("acoustic_b2_s3", B=2, T=40, Tp=5, Tr=36, steps_mel=3, steps_f0=3)
    run_acoustic_case("acoustic_dur_s2", B=1, T=0, Tp=7, Tr=32, steps_mel=2, steps_f0=2, give_mel2ph=False)
    run_acoustic_case("acoustic_t64_s100", B=1, T=64, Tp=8, Tr=48, steps_mel=100, steps_f0=100, keep_stages=False)
    # round 4: 300 frames (1.6 s), full 100 + 2 x 100 step chains - several frame tiles of every kernel, directly against the real reference
    run_acoustic_case("acoustic_t300_s100", B=1, T=300, Tp=12, Tr=200, steps_mel=100, steps_f0=100, keep_stages=False)
    run_vocoder_case("vocoder_t12", B=1, T=12)
    run_vocoder_case("vocoder_b2_t9", B=2, T=9)
    run_vocoder_case("vocoder_t200", B=1, T=200)   # 51 200 samples: the NSF phase integration and the 4-stage generator at a length where tile interiors exist
    run_plms_case("plms_t40_k20_i3", T=40, steps_mel=20, interval=3)
    run_plms_case("plms_t24_k12_i4", T=24, steps_mel=12, interval=4)
    round2_cases()
    round3_cases()


if __name__ == "__main__":
    main()
