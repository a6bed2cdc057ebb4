"""Deterministic synthetic weights, inputs and noise tapes.

No checkpoint ships with the reference (README.md:34-41), so parity and benchmarks run on seeded random
weights shared through the reference's own `state_dict` contract.  Everything here is plain CPU torch
with explicit generators so the build container (where the real reference runs) and the GPU box produce
bit-identical tensors.
"""
import math
import zlib

import numpy as np
import torch

_torch_randn, _torch_rand = torch.randn, torch.rand  # originals: gen_golden patches the module attributes

from . import spec as _spec
from .config import make_hparams, make_vocoder_config


def _gen(seed, name):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


# ------------------------------------------------------------------------------------------------
# schedules: the arithmetic of the reference's constructors, float64 numpy then cast to fp32
# (modules/diff/shallow_diffusion_tts.py:41-46,83-119; gaussian_multinomial_diffusion.py:237-284)
# ------------------------------------------------------------------------------------------------
def gaussian_schedule(timesteps, max_beta):
    betas = np.linspace(1e-4, max_beta, timesteps)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    d = {
        "betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac), "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac), "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1), "posterior_variance": pv,
        "posterior_log_variance_clipped": np.log(np.maximum(pv, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }
    return {k: torch.tensor(v, dtype=torch.float32) for k, v in d.items()}


def prodiff_betas(schedule_mode, timesteps, min_beta=0.1, max_beta=40.0, s=0.008):
    """get_noise_schedule_list (modules/diff/prodiff.py:28-49) as ProDiffusion.__init__ calls it (:69-75: timesteps + 1
    entries, min_beta = 0.1, max_beta = 40)."""
    if schedule_mode == "linear":
        return np.linspace(0.000001, 0.01, timesteps)
    if schedule_mode == "cosine":
        steps = timesteps + 1
        x = np.linspace(0, steps, steps)
        ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        return np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    if schedule_mode == "vpsde":
        t = np.arange(1, timesteps + 1)
        return 1.0 - np.exp(-min_beta / timesteps - 0.5 * (max_beta - min_beta) * (2 * t - 1) / (timesteps ** 2))
    raise NotImplementedError(schedule_mode)


def prodiff_schedule(hp):
    """Buffers of ProDiffusion (modules/diff/prodiff.py:77-113), float64 numpy then fp32 like the reference."""
    betas = prodiff_betas(hp["schedule_type"], hp["timesteps"] + 1)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    d = {
        "betas": betas, "alphas_cumprod": ac, "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac), "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac), "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1), "posterior_variance": pv,
        "posterior_log_variance_clipped": np.log(np.maximum(pv, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }
    out = {k: torch.tensor(v, dtype=torch.float32) for k, v in d.items()}
    out["timesteps"] = torch.tensor(int(hp["timesteps"]), dtype=torch.float32)
    out["timescale"] = torch.tensor(hp.get("timescale", 1), dtype=torch.float32)
    return out


def multinomial_schedule(timesteps, max_beta):
    betas = np.linspace(1e-4, max_beta, timesteps)
    alphas = torch.tensor((1.0 - betas).astype("float64"))
    log_alpha = np.log(alphas)
    log_cumprod_alpha = np.cumsum(log_alpha)
    l1m = lambda a: torch.log(1 - a.exp() + 1e-40)
    d = {"log_alpha": log_alpha, "log_1_min_alpha": l1m(log_alpha), "log_cumprod_alpha": log_cumprod_alpha,
         "log_1_min_cumprod_alpha": l1m(log_cumprod_alpha)}
    out = {k: torch.as_tensor(v).to(torch.float32) for k, v in d.items()}
    out["Lt_history"] = torch.zeros(timesteps)
    out["Lt_count"] = torch.zeros(timesteps)
    return out


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
_EMB_PAD0 = ("encoder_embed_tokens.weight", "encoder.embed_tokens.weight", "pitch_embed.weight",
             "note_encoder.emb.weight", "note_encoder.type_emb.weight")


def _init_tensor(name, shape, seed):
    g = _gen(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "_float_tensor":
        return torch.zeros(shape)
    if leaf == "pos_embed_alpha":
        return torch.full(shape, 0.9)
    if leaf in ("cluster_size_ema",):
        return torch.ones(shape)
    if leaf == "weight_g":
        return 0.5 + 0.5 * torch.rand(shape, generator=g)
    if name.endswith("uv_embed.weight") or name in _EMB_PAD0:
        w = torch.randn(shape, generator=g) * (shape[1] ** -0.5)
        if name in _EMB_PAD0:
            w[0] = 0
        return w
    if "rqvae.codebooks" in name:
        d = int(name.split("codebooks.")[1].split(".")[0])
        w = torch.randn(shape, generator=g) * (0.8 * 0.6 ** d)
        if leaf == "weight":
            w[-1] = 0  # padding row (RQ.py:14 padding_idx=n_embed)
        return w
    if name == "dur_predictor.linear.bias":
        return torch.full(shape, 1.7)  # exp(1.7)-1 ~ 4.5 frames per phoneme under random weights
    if leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf in ("weight", "weight_v", "in_proj_weight"):
        if len(shape) == 1:  # LayerNorm gamma
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        fan_in = int(np.prod(shape[1:]))
        return torch.randn(shape, generator=g) / math.sqrt(fan_in)
    if leaf == "in_proj_bias":
        return 0.05 * torch.randn(shape, generator=g)
    raise KeyError(f"no init rule for {name} {shape}")


def synth_acoustic_state_dict(hp=None, seed=1234):
    hp = hp or make_hparams()
    sd = {}
    sched_f0 = {**multinomial_schedule(hp["f0_timesteps"], hp["f0_max_beta"]),
                **gaussian_schedule(hp["f0_timesteps"], hp["f0_max_beta"])}
    prodiff = hp.get("decoder", "diffsinger") == "prodiff"
    sched_mel = prodiff_schedule(hp) if prodiff else gaussian_schedule(hp["timesteps"], hp["max_beta"])
    for name, shape in _spec.acoustic_spec(hp):
        head = name.split(".")[0]
        if head == "diff_decoder":
            leaf = name.split(".", 1)[1]
            if leaf in sched_mel:
                sd[name] = sched_mel[leaf].clone()
                continue
            if leaf in ("spec_min", "spec_max"):
                sd[name] = torch.tensor(hp[leaf], dtype=torch.float32)[None, None, :hp["keep_bins"]]
                continue
        if head in ("f0_gen", "f0_gen_inpainte"):
            rest = name[len(head) + 1:]
            if rest.startswith("_denoise_fn."):
                alias = ("gm_diffnet." if head == "f0_gen" else "gm_diffnet_inpainte.") + rest[len("_denoise_fn."):]
                sd[name] = sd[alias]  # the reference registers the same module twice (stylesinger.py:69-73)
            else:
                sd[name] = sched_f0[rest].clone()
            continue
        if head == "postdiff" and name.split(".")[1] in sched_mel:
            sd[name] = sched_mel[name.split(".")[1]].clone()
            continue
        if name == "postdiff.spec_min":
            sd[name] = torch.tensor(hp["spec_min"], dtype=torch.float32)[None, None, :hp["keep_bins"]]
            continue
        if name == "postdiff.spec_max":
            sd[name] = torch.tensor(hp["spec_max"], dtype=torch.float32)[None, None, :hp["keep_bins"]]
            continue
        if name == "encoder.embed_tokens.weight":
            sd[name] = sd["encoder_embed_tokens.weight"]  # same nn.Embedding (fs2.py:28-29)
            continue
        if name.endswith("embed_ema"):
            sd[name] = sd[name.replace("embed_ema", "weight")][:-1].clone()
            continue
        sd[name] = _init_tensor(name, tuple(shape), seed).contiguous()
    return sd


def synth_emotion_state_dict(seed=1234):
    """Seeded weights of the emotion encoder (data_gen/tts/emotion/model.py:11-31): U(-1/sqrt(H), 1/sqrt(H)) like nn.LSTM."""
    sd = {}
    for name, shape in _spec.emotion_spec():
        if name.startswith("similarity_"):
            sd[name] = torch.tensor([10.0 if name.endswith("weight") else -5.0])
            continue
        g = _gen(seed, "emotion." + name)
        sd[name] = ((torch.rand(shape, generator=g) * 2 - 1) / 16.0).contiguous()
    return sd


def synth_emotion_frames(n_partials, n_frames=160, n_mel=40, seed=1234):
    """Stand-in for audio.wav_to_mel_spectrogram partials (librosa power mel of 16 kHz speech, un-vendored): positive,
    heavy-tailed values of the same order (1e-5 .. 1)."""
    g = _gen(seed, "emotion.frames")
    return torch.exp(torch.randn(n_partials, n_frames, n_mel, generator=g) * 1.5 - 6.0)


def synth_vocoder_state_dict(cfg=None, seed=1234):
    cfg = cfg or make_vocoder_config()
    sd = {}
    for name, shape in _spec.vocoder_spec(cfg):
        t = _init_tensor("vocoder." + name, tuple(shape), seed)
        if name.endswith("weight_v") and (".convs" in name):
            t = t * 0.7  # keep the 9-deep residual stacks O(1) under random weights
        sd[name] = t.contiguous()
    return sd


# ------------------------------------------------------------------------------------------------
# inputs (one generator per utterance index so shards are rank-independent)
# ------------------------------------------------------------------------------------------------
def synth_utterance(idx, T, Tp, Tr, hp=None, seed=1234):
    """One synthetic utterance: phoneme/note inputs, explicit mel2ph, reference mel/f0, spk/emo embeddings."""
    hp = hp or make_hparams()
    g = _gen(seed, f"utt{idx}")
    V = hp["vocab_size"]
    txt = torch.randint(3, V, (Tp,), generator=g)
    rest = torch.rand(Tp, generator=g) < 0.1
    note = torch.randint(48, 77, (Tp,), generator=g)
    note[rest] = 0
    note_type = torch.full((Tp,), 2, dtype=torch.long)
    note_type[rest] = 1
    note_dur = 0.2 + 0.65 * torch.rand(Tp, generator=g)
    # phoneme i repeated ~T/Tp frames
    bounds = torch.linspace(0, T, Tp + 1).round().long()
    mel2ph = torch.zeros(T, dtype=torch.long)
    for i in range(Tp):
        mel2ph[bounds[i]:bounds[i + 1]] = i + 1
    ref_mels = (torch.randn(Tr, 80, generator=g) * 0.8 - 3.0).clamp(-6.0, 0.6)
    ref_mels[:, 0] = ref_mels[:, 0].clamp(max=-0.05)  # never exactly 0 in bin 0 (padding probe, lse.py:109)
    # log2-domain f0 contour, 100-500 Hz, interpolated through unvoiced runs (utils/pitch_utils.py:47-62)
    tt = torch.arange(Tr, dtype=torch.float32)
    hz = 250.0 + 120.0 * torch.sin(tt * (2 * math.pi / 180.0) + torch.rand(1, generator=g) * 6.28) \
        + 60.0 * torch.sin(tt * (2 * math.pi / 37.0))
    ref_f0 = torch.log2(hz.clamp(100.0, 500.0))
    spk = torch.relu(torch.randn(256, generator=g))
    spk = spk / spk.norm().clamp_min(1e-6)
    emo = torch.randn(256, generator=g)
    emo = emo / emo.norm()
    return dict(txt_tokens=txt, note=note, note_type=note_type, note_dur=note_dur, mel2ph=mel2ph, ref_mels=ref_mels,
                ref_f0=ref_f0, spk_embed=spk, emo_embed=emo)


def synth_f0_hz(idx, Tr, seed=1234, unvoiced=0.2, dtype=torch.float64):
    """A tracker-like reference contour in Hz: 100-500 Hz with ~`unvoiced` of the frames in unvoiced runs
    (value 0), float64 like parselmouth's output - the input `norm_interp_f0` (utils/pitch_utils.py:47-62) expects."""
    g = _gen(seed, f"f0hz{idx}")
    tt = torch.arange(Tr, dtype=torch.float64)
    hz = 250.0 + 120.0 * torch.sin(tt * (2 * math.pi / 180.0) + torch.rand(1, generator=g, dtype=torch.float64) * 6.28) \
        + 60.0 * torch.sin(tt * (2 * math.pi / 37.0))
    hz = hz.clamp(100.0, 500.0)
    t = 0
    while t < Tr:  # alternate voiced / unvoiced runs
        run = int(torch.randint(8, 60, (1,), generator=g))
        if float(torch.rand(1, generator=g)) < unvoiced * 2.0 and run < Tr:
            hz[t:t + run // 2] = 0.0
        t += run
    return hz.to(dtype)


def synth_batch(B, T, Tp, Tr, hp=None, seed=1234, first_index=0, indices=None):
    """`indices`: explicit utterance indices (e.g. a rank's shard rank, rank + W, ...); default first_index .. first_index + B - 1."""
    idx = list(indices) if indices is not None else [first_index + i for i in range(B)]
    assert len(idx) == B
    items = [synth_utterance(i, T, Tp, Tr, hp, seed) for i in idx]
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}


# ------------------------------------------------------------------------------------------------
# noise tape: every random draw of the path, in the reference's order
# ------------------------------------------------------------------------------------------------
class NoiseTape:
    """Sequential source of the path's random numbers.

    The reference draws from torch's global generator in a fixed order (SURVEY.md §7 "RNG parity"):
    per f0 sampler: rand_like[B,1,T], randn[B,1,T], then per step randn[B,1,T], rand_like[B,2,T];
    mel: randn_like[B,1,80,T], per step randn[B,1,80,T]; vocoder: rand[B,9], randn_like[B,L,9], randn_like[B,L,1].
    `oracle/gen_golden.py` patches torch.rand*/randn* to pull from this tape; the HIP path receives the
    same tensors as explicit noise arguments.
    """

    def __init__(self, seed=1234):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(int(seed))
        self.log = []

    def randn(self, *shape):
        t = _torch_randn(*shape, generator=self.g, dtype=torch.float32)
        self.log.append(("randn", tuple(t.shape)))
        return t

    def rand(self, *shape):
        t = _torch_rand(*shape, generator=self.g, dtype=torch.float32)
        self.log.append(("rand", tuple(t.shape)))
        return t


def draw_acoustic_noise(tape, B, T, steps_f0, steps_mel, M=80):
    """Pre-draw the acoustic model's noise in reference order, keyed for the HIP path."""
    out = {}
    for net in ("f0_a", "f0_b"):
        u_init = tape.rand(B, 1, T)
        z0 = tape.randn(B, 1, T)
        zs = torch.empty(steps_f0, B, 1, T)
        us = torch.empty(steps_f0, B, 2, T)
        for i in reversed(range(steps_f0)):
            zs[i] = tape.randn(B, 1, T)
            us[i] = tape.rand(B, 2, T)
        out[net] = dict(u_init=u_init, z0=z0, z_steps=zs, u_steps=us)
    zq = tape.randn(B, 1, M, T)
    zm = torch.empty(steps_mel, B, 1, M, T)
    for i in reversed(range(steps_mel)):
        zm[i] = tape.randn(B, 1, M, T)
    out["mel"] = dict(z_q=zq, z_steps=zm)
    return out


def draw_vocoder_noise(tape, B, L, dim=9):
    rand_ini = tape.rand(B, dim)
    sine_noise = tape.randn(B, L, dim)
    _ = tape.randn(B, L, 1)  # SourceModuleHnNSF noise branch: drawn, unused by the generator (source.py:529)
    return dict(rand_ini=rand_ini, sine_noise=sine_noise)
