"""Reference-audio mel front end on the GPU (first item).

Mirror of `utils/audios/__init__.py:36-84 librosa_wav2spec` as the entrypoint uses it (`inference/StyleSinger.py:106-118`,
hparams of egs/stylesinger.yaml:29-37: 48 kHz, fft 1024, hop 256, win 1024, 80 mels, 20 Hz - 24 kHz, eps 1e-6, no
loudness norm): centred STFT with zero ("constant") padding and a periodic Hann window, magnitude, Slaney-normalised mel
filterbank (librosa 0.8.0 defaults: htk=False, norm='slaney'), log10(max(eps, .)).

MI355X mapping: with hop | fft the frame matrix never exists - the waveform is viewed channels-last as [L/hop][hop] and a
frame is 4 consecutive rows, i.e. the windowed real DFT is a 4-tap implicit-GEMM conv (K = 1024, N = 513 cos + 513 sin
columns) on the exact-fp32 MFMA path of `ss_conv_gemm`; the hardware range check of the buffer loads IS the centre padding.
The mel projection is a second GEMM (K = 513 -> 544, N = 80); |X| and log10 are two small element-wise kernels.
The speaker / emotion encoders and the f0 tracker of `preprocess_input` remain outside (un-vendored third-party models).
"""
import math

import numpy as np
import torch

from . import lib as L
from .config import make_hparams

FRONTEND_DEFAULTS = dict(fft_size=1024, win_size=1024, fmin=20, fmax=24000, mel_eps=1e-6)


def _hz_to_mel(f):
    """Slaney scale (librosa.core.convert.hz_to_mel, htk=False): linear below 1 kHz, log above."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its 0.8.0 defaults (htk=False, norm='slaney') -> [n_mels, 1+n_fft/2]."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


class MelFrontendHIP:
    def __init__(self, hparams=None, device="cuda"):
        hp = dict(FRONTEND_DEFAULTS)
        hp.update(make_hparams(hparams))
        if hparams:
            hp.update({k: hparams[k] for k in FRONTEND_DEFAULTS if k in hparams})
        self.sr, self.hop, self.n_fft, self.n_mels = hp["audio_sample_rate"], hp["hop_size"], hp["fft_size"], hp["audio_num_mel_bins"]
        self.eps = float(hp["mel_eps"])
        if hp["win_size"] != self.n_fft or self.n_fft % self.hop:
            raise NotImplementedError("MelFrontendHIP: needs win_size == fft_size and hop_size | fft_size (the reference's setting)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.StyleSingerHipError("MelFrontendHIP needs a GPU: there is no CPU path")
        self.taps = self.n_fft // self.hop
        self.nbins = self.n_fft // 2 + 1
        self.nb_pad = L.round_up(self.nbins, 32)
        k = np.arange(self.n_fft, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * k / self.n_fft)           # scipy.signal.get_window('hann', N, fftbins=True)
        ang = 2 * np.pi * np.outer(np.arange(self.nbins, dtype=np.float64), k) / self.n_fft
        basis = np.zeros((2 * self.nb_pad, self.n_fft), dtype=np.float64)
        basis[:self.nbins] = np.cos(ang) * win
        basis[self.nb_pad:self.nb_pad + self.nbins] = -np.sin(ang) * win
        # conv weight [Cout][Cin = hop][k = taps]: sample index inside the frame = tap * hop + ci
        w = torch.from_numpy(basis.astype(np.float32)).reshape(2 * self.nb_pad, self.taps, self.hop).permute(0, 2, 1).contiguous()
        self.W_dft = L.pack_conv_weight(w.to(self.device))
        fb = np.zeros((self.n_mels, self.nb_pad), dtype=np.float32)
        fb[:, :self.nbins] = mel_filterbank(self.sr, self.n_fft, self.n_mels, hp["fmin"], hp["fmax"])
        self.W_mel = L.pack_conv_weight(torch.from_numpy(fb)[:, :, None].contiguous().to(self.device))

    @torch.no_grad()
    def wav2mel(self, wav, lens=None):
        """wav fp32 [B, L] on the device (lens = valid samples per item, default L) -> (mel [B, T, 80], frames int32 [B]),
        T = 1 + L // hop frames as librosa.stft(center=True) yields; frames past an item's own count are 0."""
        wav = wav.to(self.device).float()
        B, Ls = wav.shape
        hop, taps = self.hop, self.taps
        R = (Ls + hop - 1) // hop
        if R * hop != Ls:
            wav = torch.nn.functional.pad(wav, (0, R * hop - Ls))
        wav = wav.contiguous()
        n = torch.full((B,), Ls, device=self.device, dtype=torch.int64) if lens is None else lens.to(self.device).to(torch.int64)
        lib = L.load()
        if lens is not None:  # samples past an item's length are not part of it: zeroed (the row-mask kernel on the [B][R * hop][1] view)
            n32 = n.to(torch.int32).contiguous()
            masked = torch.empty_like(wav)
            L.check(lib.ss_add_bcast_mask(L.ptr(wav), None, None, None, None, L.ptr(masked), B, R * hop, 1, L.ptr(n32), L.stream_ptr()), "wav mask")
            wav = masked
        rows = ((n + hop - 1) // hop).to(torch.int32)
        frames = (n // hop + 1).to(torch.int32)
        T = Ls // hop + 1
        S = torch.empty(B, T, 2 * self.nb_pad, device=self.device, dtype=torch.float32)
        L.conv_gemm(wav, self.W_dft, S, B=B, T=T, Cin=hop, N=2 * self.nb_pad, Np=self.W_dft.shape[0], Kp=self.W_dft.shape[1] // taps,
                    lda=hop, a_bs=R * hop, taps=tuple(j - taps // 2 for j in range(taps)), lens=rows, mask_rows=False)
        P = torch.empty(B, T, self.nb_pad, device=self.device, dtype=torch.float32)
        L.check(lib.ss_spec_magnitude(L.ptr(S), L.ptr(P), B * T, 2 * self.nb_pad, self.nb_pad, self.nbins, self.nb_pad, L.stream_ptr()), "mag")
        mel = torch.empty(B, T, self.n_mels, device=self.device, dtype=torch.float32)
        L.conv_gemm(P, self.W_mel, mel, B=B, T=T, Cin=self.nb_pad, N=self.n_mels, Np=self.W_mel.shape[0], Kp=self.W_mel.shape[1],
                    mask_rows=False)
        L.check(lib.ss_log10_floor(L.ptr(mel), L.ptr(mel), mel.numel(), self.eps, L.stream_ptr()), "log10")
        frames = frames.contiguous()
        L.check(lib.ss_add_bcast_mask(L.ptr(mel), None, None, None, None, L.ptr(mel), B, T, self.n_mels, L.ptr(frames), L.stream_ptr()), "mel mask")
        return mel, frames


class EmotionMelFrontendHIP:
    """The emotion encoder's own front end on the device: `data_gen/tts/emotion/audio.py:43-55 wav_to_mel_spectrogram` =
    librosa.feature.melspectrogram(y, 16000, n_fft=400, hop_length=160, n_mels=40), i.e. a centred STFT with REFLECT padding, a
    periodic Hann window of 400, |X|^2, a Slaney 40-mel basis 0-8 kHz, no log; plus `normalize_volume` (:109-115) as
    `preprocess_wav` applies it. (`trim_long_silences` needs webrtcvad - un-vendored - and stays with the caller.)

    MI355X mapping: hop (160) does not divide the frame (400), so the frame is padded to 3 rows of the waveform viewed as
    [L/160][160] (the window is zero beyond sample 400): the windowed real DFT is a 3-tap `ss_conv_gemm` (K = 480, N = 2 x 224
    cos|sin columns, exact-fp32 MFMA) over the reflect-padded waveform (`ss_reflect_pad`, per-item lengths), then
    `ss_spec_power` and the mel projection as a second GEMM (K = 201 -> 224, N = 40)."""
    SR, N_FFT, HOP, N_MELS = 16000, 400, 160, 40

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.StyleSingerHipError("EmotionMelFrontendHIP needs a GPU: there is no CPU path")
        n_fft, hop = self.N_FFT, self.HOP
        self.taps = (n_fft + hop - 1) // hop           # 3 rows of 160 samples cover the 400-sample frame
        self.nbins = n_fft // 2 + 1                    # 201
        self.nb_pad = L.round_up(self.nbins, 32)       # 224
        k = np.arange(n_fft, dtype=np.float64)
        win = 0.5 - 0.5 * np.cos(2 * np.pi * k / n_fft)
        ang = 2 * np.pi * np.outer(np.arange(self.nbins, dtype=np.float64), k) / n_fft
        basis = np.zeros((2 * self.nb_pad, self.taps * hop), dtype=np.float64)
        basis[:self.nbins, :n_fft] = np.cos(ang) * win
        basis[self.nb_pad:self.nb_pad + self.nbins, :n_fft] = -np.sin(ang) * win
        w = torch.from_numpy(basis.astype(np.float32)).reshape(2 * self.nb_pad, self.taps, hop).permute(0, 2, 1).contiguous()
        self.W_dft = L.pack_conv_weight(w.to(self.device))
        fb = np.zeros((self.N_MELS, self.nb_pad), dtype=np.float32)
        fb[:, :self.nbins] = mel_filterbank(self.SR, n_fft, self.N_MELS, 0.0, self.SR / 2.0)
        self.W_mel = L.pack_conv_weight(torch.from_numpy(fb)[:, :, None].contiguous().to(self.device))

    @torch.no_grad()
    def normalize_volume(self, wav, lens, target_dbfs=-30.0):
        """audio.normalize_volume(increase_only=True) per item of a zero-padded batch [B, L]: scale by 10^(change/20) when
        change = target - 10 log10(mean(wav^2)) >= 0 (mean over the item's own samples), else leave untouched."""
        wav = wav.to(self.device).float().contiguous()
        out = torch.empty_like(wav)
        n = torch.as_tensor(lens).to(device=self.device, dtype=torch.int32).contiguous()
        L.check(L.load().ss_normalize_volume(L.ptr(wav), L.ptr(n), L.ptr(out), wav.shape[0], wav.shape[1], float(target_dbfs), L.stream_ptr()),
                "ss_normalize_volume")
        n.record_stream(torch.cuda.current_stream(self.device))
        return out

    @torch.no_grad()
    def wav2mel(self, wav, lens=None):
        """wav fp32 [B, L] on the device, zero beyond lens[b] (host ints / CPU tensor, default L) -> (frames [B, T, 40] fp32 power
        mel, n_frames list): T = 1 + max(lens) // 160; rows past an item's own 1 + lens[b] // 160 frames are not meaningful."""
        wav = wav.to(self.device).float().contiguous()
        B, Lx = wav.shape
        hop, n_fft, taps = self.HOP, self.N_FFT, self.taps
        lens_h = [int(Lx)] * B if lens is None else [int(v) for v in lens]
        if min(lens_h) <= n_fft // 2:
            raise ValueError(f"EmotionMelFrontendHIP: reflect padding needs more than {n_fft // 2} samples per item")
        lens_d = torch.tensor(lens_h, device=self.device, dtype=torch.int32)
        T = 1 + max(lens_h) // hop
        rows = T + taps - 1
        Ly = rows * hop
        lib = L.load()
        yp = torch.empty(B, Ly, device=self.device, dtype=torch.float32)
        L.check(lib.ss_reflect_pad(L.ptr(wav), L.ptr(lens_d), L.ptr(yp), B, Lx, Ly, n_fft // 2, L.stream_ptr()), "ss_reflect_pad")
        S = torch.empty(B, T, 2 * self.nb_pad, device=self.device, dtype=torch.float32)
        L.conv_gemm(yp, self.W_dft, S, B=B, T=T, Cin=hop, N=2 * self.nb_pad, Np=self.W_dft.shape[0], Kp=self.W_dft.shape[1] // taps,
                    lda=hop, a_bs=Ly, taps=tuple(range(taps)), lens=torch.full((B,), rows, device=self.device, dtype=torch.int32),
                    mask_rows=False)
        P = torch.empty(B, T, self.nb_pad, device=self.device, dtype=torch.float32)
        L.check(lib.ss_spec_power(L.ptr(S), L.ptr(P), B * T, 2 * self.nb_pad, self.nb_pad, self.nbins, self.nb_pad, L.stream_ptr()), "ss_spec_power")
        mel = torch.empty(B, T, self.N_MELS, device=self.device, dtype=torch.float32)
        L.conv_gemm(P, self.W_mel, mel, B=B, T=T, Cin=self.nb_pad, N=self.N_MELS, Np=self.W_mel.shape[0], Kp=self.W_mel.shape[1],
                    mask_rows=False)
        return mel, [1 + n // hop for n in lens_h]
