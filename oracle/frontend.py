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


# ------------------------------------------------------------------------------------------------
# emotion encoder front end (data_gen/tts/emotion/audio.py:43-55,109-115; inference.py:111-151)
# ------------------------------------------------------------------------------------------------
def melspectrogram_power(wav, sr=16000, n_fft=400, hop=160, n_mels=40):
    """`librosa.feature.melspectrogram(y, sr, n_fft, hop_length, n_mels)` as audio.wav_to_mel_spectrogram calls it, librosa 0.8.0
    defaults: center=True with pad_mode="reflect", periodic Hann of n_fft, power=2.0 (|STFT|^2 on the complex64 STFT), Slaney mel
    basis fmin=0 .. fmax=sr/2 -> float32 [T, n_mels] (the reference's `.astype(np.float32).T`), T = 1 + len(y) // hop."""
    wav = np.asarray(wav, dtype=np.float32)
    y = np.pad(wav, n_fft // 2, mode="reflect")
    T = 1 + len(wav) // hop
    n = np.arange(n_fft)
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)
    frames = np.stack([y[t * hop:t * hop + n_fft] for t in range(T)])
    spec = np.abs(np.fft.rfft(frames * win[None, :], axis=1).astype(np.complex64)) ** 2.0
    mel = mel_basis(sr, n_fft, n_mels, 0.0, sr / 2.0) @ spec.T
    return mel.astype(np.float32).T


def normalize_volume(wav, target_dbfs=-30.0, increase_only=True):
    """audio.normalize_volume (:109-115) as preprocess_wav calls it (increase_only=True)."""
    wav = np.asarray(wav)
    change = target_dbfs - 10.0 * np.log10(np.mean(wav ** 2))
    if change < 0 and increase_only:
        return wav
    return wav * (10.0 ** (change / 20.0))


def partial_slices(n_samples, n_frames=160, min_pad_coverage=0.75, overlap=0.5, sr=16000, step_ms=10):
    """inference.compute_partial_slices (:56-108): (wav_slices, mel_slices) as (start, stop) pairs."""
    spf = int(sr * step_ms / 1000)
    total = int(np.ceil((n_samples + 1) / spf))
    step = max(int(np.round(n_frames * (1 - overlap))), 1)
    starts = list(range(0, max(1, total - n_frames + step + 1), step))
    mel = [(i, i + n_frames) for i in starts]
    wav = [(i * spf, (i + n_frames) * spf) for i in starts]
    cov = (n_samples - wav[-1][0]) / (wav[-1][1] - wav[-1][0])
    if cov < min_pad_coverage and len(mel) > 1:
        mel, wav = mel[:-1], wav[:-1]
    return wav, mel


def embed_utterance_frames(wav):
    """The data side of inference.embed_utterance (:128-140) with using_partials=True: zero-pad the waveform to the last partial's
    end, 40-mel power spectrogram, stack the partial windows -> float32 [P, 160, 40]."""
    wav = np.asarray(wav, dtype=np.float32)
    wav_slices, mel_slices = partial_slices(len(wav))
    need = wav_slices[-1][1]
    if need >= len(wav):
        wav = np.pad(wav, (0, need - len(wav)), "constant")
    frames = melspectrogram_power(wav)
    return np.stack([frames[a:b] for a, b in mel_slices])
