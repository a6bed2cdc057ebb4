"""CPU restatement of the reference-audio mel front end  —  TEST INFRASTRUCTURE ONLY.

Follows `utils/audios/__init__.py:36-84 librosa_wav2spec` (reference), whose arithmetic lives in **librosa==0.8.0**
(requirements.txt:2), a third-party dependency that is absent from /root/reference and from this image.  Its published
algorithm is restated here in numpy:
  * `librosa.stft(y, n_fft, hop_length, win_length, window="hann", center=True, pad_mode="constant")`: zero-pad n_fft//2 on
    both sides, frames of n_fft at stride hop, periodic Hann (`scipy.signal.get_window('hann', N, fftbins=True)`),
    `rfft` per frame, result cast to complex64; 1 + len(y)//hop frames.
  * `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)` with the 0.8.0 defaults htk=False, norm='slaney'.
Pinning: librosa cannot be executed here, so this file is pinned by (a) the known answers printed in librosa's own
docstrings (hz_to_mel / mel_to_hz / filters.mel examples, checked in tests/test_oracle_golden.py) and (b) an independent
STFT (`scipy.signal.stft`) - not by running the reference's call itself: **parity partially pinned**.
"""
import numpy as np


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    out = f / (200.0 / 3)
    log_region = f >= 1000.0
    out = np.where(log_region, 15.0 + np.log(np.where(log_region, f, 1000.0) / 1000.0) / (np.log(6.4) / 27.0), out)
    return out


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    out = (200.0 / 3) * m
    log_region = m >= 15.0
    return np.where(log_region, 1000.0 * np.exp((np.log(6.4) / 27.0) * (np.where(log_region, m, 15.0) - 15.0)), out)


def mel_basis(sr, n_fft, n_mels, fmin, fmax):
    n_bins = 1 + n_fft // 2
    freqs = np.arange(n_bins) * (sr / n_fft)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        lo, ce, hi = edges[i], edges[i + 1], edges[i + 2]
        up = (freqs - lo) / (ce - lo)
        down = (hi - freqs) / (hi - ce)
        w[i] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (hi - lo))
    return w.astype(np.float32)


def stft_mag(wav, n_fft, hop):
    wav = np.asarray(wav, dtype=np.float32)
    y = np.pad(wav, (n_fft // 2, n_fft // 2), mode="constant")
    T = 1 + len(wav) // hop
    n = np.arange(n_fft)
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)
    frames = np.stack([y[t * hop:t * hop + n_fft] for t in range(T)])
    spec = np.fft.rfft(frames * win[None, :], axis=1).astype(np.complex64)
    return np.abs(spec)  # [T, 1 + n_fft/2] float32


def wav2mel(wav, sr=48000, n_fft=1024, hop=256, n_mels=80, fmin=20, fmax=24000, eps=1e-6):
    """-> mel [T, n_mels] = log10(max(eps, mel_basis @ |STFT|)).T   (utils/audios/__init__.py:60-77)."""
    lin = stft_mag(wav, n_fft, hop)
    mel = mel_basis(sr, n_fft, n_mels, fmin, fmax) @ lin.T
    return np.log10(np.maximum(eps, mel)).T.astype(np.float32)
