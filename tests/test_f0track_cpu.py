"""The f0 tracker's CPU restatement (oracle/praat_pitch.py) against ANALYTIC known answers, and the host-side geometry of the device
tracker (stylesinger_amd/f0track.py) against it. parselmouth / Praat are un-vendored (inference/StyleSinger.py:125-127): parity with
the package itself is UNPINNED; these known answers are what the published algorithm guarantees (Boersma 1993: a stationary periodic
signal comes back at its fundamental, also with the fundamental missing; silence and low-level noise are unvoiced)."""
import numpy as np
import pytest

from oracle import praat_pitch as P
from stylesinger_amd import f0track as FT

SR = 48000


def _t(sec):
    return np.arange(int(sec * SR)) / SR


@pytest.mark.parametrize("f0", [110.0, 220.0, 523.25])
def test_stationary_sine_comes_back_at_its_frequency(f0):
    f = P.to_pitch_ac(0.3 * np.sin(2 * np.pi * f0 * _t(0.4)), SR)
    assert len(f) > 50 and (f > 0).all()
    assert np.abs(f - f0).max() <= 0.02, np.abs(f - f0).max()


def test_missing_fundamental_and_harmonic_complex():
    t = _t(0.4)
    w = sum(0.2 / h * np.sin(2 * np.pi * 150.0 * h * t) for h in range(2, 8))   # harmonics 2..7 of 150 Hz, no fundamental
    f = P.to_pitch_ac(w, SR)
    assert (f > 0).all() and np.abs(f - 150.0).max() <= 0.02


def test_silence_noise_and_voicing_boundaries():
    rng = np.random.default_rng(0)
    seg = 12000
    w = np.concatenate([np.zeros(seg), 0.3 * np.sin(2 * np.pi * 200.0 * _t(0.25)), 0.001 * rng.standard_normal(seg)])
    f = P.to_pitch_ac(w, SR)
    centre = (1024 + 256 * np.arange(len(f)))          # frame centres in samples for N a multiple of the hop
    inside = (centre > seg + 1500) & (centre < 2 * seg - 1500)
    outside = (centre < seg - 1500) | (centre > 2 * seg + 1500)
    assert (np.abs(f[inside] - 200.0) <= 0.05).all()
    assert (f[outside] == 0).all()
    assert (P.to_pitch_ac(np.zeros(20000), SR) == 0).all()


def test_vibrato_is_tracked_without_octave_jumps():
    t = _t(0.6)
    inst = 330.0 * (1 + 0.03 * np.sin(2 * np.pi * 5.5 * t))
    ph = 2 * np.pi * np.cumsum(inst) / SR
    w = sum(0.3 / h * np.sin(h * ph) for h in range(1, 6))
    f = P.to_pitch_ac(w, SR)
    c = 1024 + 256 * np.arange(len(f))
    assert (f > 0).all() and np.abs(f - inst[c]).max() <= 1.5     # the 37.5 ms window smooths the 5.5 Hz modulation slightly


@pytest.mark.parametrize("n_mel", [40, 93, 750, 1500])
def test_frame_grid_follows_the_manual_and_the_reference_alignment(n_mel):
    """N = n_mel * hop samples (what process_audio returns): n_mel - 7 frames, the first centred 1024 samples in, so that the reference's
    padding (inference/StyleSinger.py:128-135: 4 frames left, the rest right) needs no length fix."""
    n = n_mel * 256
    g = P.geometry(n, SR, 256 / 48000 * 1000 / 1000, 80.0, 800.0)
    assert g["n_frames"] == n_mel - 7 and abs(g["t1"] * SR - 1024.0) < 1e-6
    assert (g["nsamp_window"], g["halfnsamp_window"], g["nsamp_period"], g["halfnsamp_period"], g["maximum_lag"], g["brent_ixmax"]) == (1798, 899, 600, 301, 601, 899)
    d = FT.geometry(SR, 256 / 48000 * 1000 / 1000, 80.0, 800.0)
    nf, left = FT.frame_grid(d, n)
    assert nf == g["n_frames"] and left + 1 - d["halfnsamp_window"] == P.frame_start(g, 0)[0]
    for k in ("nsamp_window", "halfnsamp_window", "nsamp_period", "halfnsamp_period", "maximum_lag"):
        assert d[k] == g[k]
    assert d["nlag"] == g["brent_ixmax"] and d["hop"] == 256
    w = 0.2 * np.sin(2 * np.pi * 180.0 * np.arange(n) / SR) if n_mel <= 93 else None
    if w is not None:
        f0 = P.reference_f0(w, n_mel)
        assert len(f0) == n_mel and (f0[:4] == 0).all() and (f0[-3:] == 0).all() and (f0[4:-3] > 0).all()


@pytest.mark.parametrize("n", [256 * 50 + 18, 256 * 50 + 254, 9000])   # even counts: centres between samples (odd counts: see f0track.frame_grid)
def test_frame_grid_for_lengths_that_are_not_a_multiple_of_the_hop(n):
    g = P.geometry(n, SR, 256 / 48000, 80.0, 800.0)
    d = FT.geometry(SR, 256 / 48000, 80.0, 800.0)
    nf, left = FT.frame_grid(d, n)
    assert nf == g["n_frames"]
    for i in (0, nf - 1):
        ws, ms, me = P.frame_start(g, i)
        assert ws == left + i * 256 + 1 - d["halfnsamp_window"] and ms == left + i * 256 + 1 - d["nsamp_period"]
        assert ws >= 0 and ws + d["nsamp_window"] <= n and ms >= 0 and me <= n


def test_vad_trim_host_mirror_equals_the_real_function_with_injected_flags(golden_dir):
    """`stylesinger_amd.vadtrim.window_mask` (the integer logic of the device kernel) against the REAL `trim_long_silences`
    (data_gen/tts/emotion/audio.py:58-100) run with its webrtcvad decision injected (tests/golden/vad_trim.pt, oracle/gen_golden.py --vad):
    the kept samples must be exactly the reference's."""
    import os
    import torch
    from stylesinger_amd import vadtrim
    g = torch.load(os.path.join(golden_dir, "vad_trim.pt"), weights_only=False)
    for key, c in g["cases"].items():
        wav, flags = c["wav"].numpy(), c["flags"].numpy()
        nw = len(wav) // 480
        mask = vadtrim.window_mask(flags[:nw])
        out = wav[:nw * 480].reshape(nw, 480)[mask].reshape(-1)
        assert out.shape == tuple(c["out"].shape) and (out == c["out"].numpy()).all(), key
