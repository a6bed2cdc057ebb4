"""Emotion encoder on the GPU: mirror of `data_gen/tts/emotion/{model,inference}.py`.

The reference computes the 256-d emotion embedding of the reference audio with a 3-layer LSTM-256 over partial
utterances of 160 frames of a 40-channel (power) mel spectrogram at 16 kHz and averages the partial embeddings
(inference.py:111-151).  `EmotionEncoderHIP` loads the reference checkpoint's `model_state` unchanged and runs

  * the input projections of every layer as ONE fp32-MFMA GEMM over all partials x frames (ss_conv_gemm),
  * the recurrence as a persistent kernel per layer (ss_lstm_layer: one workgroup per partial),
  * the mean + L2 normalisation on the device (ss_mean_l2norm).

The 40-mel front end itself (`audio.wav_to_mel_spectrogram` = librosa.feature.melspectrogram, un-vendored) and the
VAD trimming (webrtcvad) stay outside: `frames` is the input, like `embed_frames_batch` in the reference.
"""
import numpy as np
import torch

from . import lib as L
from . import spec as _spec

# data_gen/tts/emotion/params_data.py
SAMPLING_RATE = 16000
MEL_WINDOW_STEP_MS = 10
PARTIALS_N_FRAMES = 160


def compute_partial_slices(n_samples, partial_utterance_n_frames=PARTIALS_N_FRAMES, min_pad_coverage=0.75, overlap=0.5):
    """Where to cut an utterance into partials (inference.py:56-108): returns (wav_slices, mel_slices)."""
    assert 0 <= overlap < 1 and 0 < min_pad_coverage <= 1
    spf = int(SAMPLING_RATE * MEL_WINDOW_STEP_MS / 1000)
    n_frames = int(np.ceil((n_samples + 1) / spf))
    step = max(int(np.round(partial_utterance_n_frames * (1 - overlap))), 1)
    starts = list(range(0, max(1, n_frames - partial_utterance_n_frames + step + 1), step))
    mel = [slice(i, i + partial_utterance_n_frames) for i in starts]
    wav = [slice(i * spf, (i + partial_utterance_n_frames) * spf) for i in starts]
    last = wav[-1]
    coverage = (n_samples - last.start) / (last.stop - last.start)
    if coverage < min_pad_coverage and len(mel) > 1:
        mel, wav = mel[:-1], wav[:-1]
    return wav, mel


class EmotionEncoderHIP:
    def __init__(self, state_dict=None, device="cuda", hidden=256, n_mel=40, layers=3, embed=256):
        if not torch.cuda.is_available():
            raise L.StyleSingerHipError("EmotionEncoderHIP needs a GPU: there is no CPU path")
        self.device = torch.device(device)
        self.H, self.n_mel, self.layers, self.E = hidden, n_mel, layers, embed
        self._pk = None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd, strict=True):
        names = dict(_spec.emotion_spec(self.H, self.n_mel, self.layers, self.E))
        missing = [k for k in names if k not in sd and not k.startswith("similarity_")]
        if strict and missing:
            raise RuntimeError(f"EmotionEncoderHIP.load_state_dict: missing {missing[:4]}")
        for k, shp in names.items():
            if k in sd and tuple(sd[k].shape) != tuple(shp):
                raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shp)}")
        H, dev = self.H, self.device
        pk = []
        for l in range(self.layers):
            w_ih = sd[f"lstm.weight_ih_l{l}"].to(dev).float()
            w_hh = sd[f"lstm.weight_hh_l{l}"].to(dev).float()
            b = (sd[f"lstm.bias_ih_l{l}"].to(dev).float() + sd[f"lstm.bias_hh_l{l}"].to(dev).float())
            cin = w_ih.shape[1]
            # gate-interleave: row 4*j + gate <- row gate*H + j  (pure re-indexing)
            w_ih_p = w_ih.view(4, H, cin).permute(1, 0, 2).reshape(4 * H, cin).contiguous()
            b_p = b.view(4, H).t().reshape(4 * H).contiguous()
            W = L.pack_conv_weight(w_ih_p)
            pk.append(dict(W=W, bias=L.pack_bias(b_p), Np=W.shape[0], Kp=W.shape[1], cin=cin,
                           whh=w_hh.view(4, H, H).permute(2, 1, 0).contiguous()))   # [k][j][gate]
        lw = sd["linear.weight"].to(dev).float()
        Wl = L.pack_conv_weight(lw)
        self._lin = dict(W=Wl, bias=L.pack_bias(sd["linear.bias"].to(dev).float()), Np=Wl.shape[0], Kp=Wl.shape[1])
        self._pk = pk
        torch.cuda.synchronize()

    @torch.no_grad()
    def embed_frames_batch(self, frames):
        """frames [P, n_frames, 40] (device or host) -> partial embeddings [P, 256] = final hidden state of the last layer
        (EmotionEncoder.inference, model.py:62-78, as inference.embed_frames_batch :39-53 calls it)."""
        lib = L.load()
        x = torch.as_tensor(frames, dtype=torch.float32).to(self.device).contiguous()
        P, n, _ = x.shape
        H = self.H
        xproj = torch.empty(P, n, 4 * H, device=self.device)
        h_last = torch.empty(P, H, device=self.device)
        for l, pk in enumerate(self._pk):
            L.conv_gemm(x, pk["W"], xproj, B=P, T=n, Cin=pk["cin"], N=4 * H, Np=pk["Np"], Kp=pk["Kp"], bias=pk["bias"], mask_rows=False)
            last = l + 1 == self.layers
            h_seq = None if last else torch.empty(P, n, H, device=self.device)
            L.check(lib.ss_lstm_layer(L.ptr(xproj), L.ptr(pk["whh"]), L.ptr(h_seq), L.ptr(h_last) if last else None, P, n, H,
                                      L.stream_ptr()), "ss_lstm_layer")
            x = h_seq
        return h_last

    @torch.no_grad()
    def embed_partials(self, frames):
        """-> (embed [256], partial_embeds [P,256]): embed_utterance's tail (inference.py:139-151)."""
        part = self.embed_frames_batch(frames)
        out = torch.empty(self.H, device=self.device)
        L.check(L.load().ss_mean_l2norm(L.ptr(part), L.ptr(out), part.shape[0], self.H, L.stream_ptr()), "ss_mean_l2norm")
        return out, part

    @torch.no_grad()
    def embed_utterance_frames(self, mel_frames, n_samples=None):
        """mel_frames [n_frames, 40] of one (padded) utterance -> embedding [256]: cuts the partials as embed_utterance does."""
        mel_frames = torch.as_tensor(mel_frames, dtype=torch.float32)
        if n_samples is None:
            n_samples = (mel_frames.shape[0] - 1) * int(SAMPLING_RATE * MEL_WINDOW_STEP_MS / 1000)
        _, mel_slices = compute_partial_slices(n_samples)
        need = mel_slices[-1].stop
        if mel_frames.shape[0] < need:
            raise ValueError(f"mel has {mel_frames.shape[0]} frames, the partial slicing needs {need}: pad the waveform first "
                             "(inference.py:128-131)")
        batch = torch.stack([mel_frames[s] for s in mel_slices])
        return self.embed_partials(batch)[0]

    @torch.no_grad()
    def forward(self, frames):
        """EmotionEncoder.forward (model.py:40-60): relu(linear(h_last)) L2-normalised per row."""
        lib = L.load()
        h = self.embed_frames_batch(frames)
        P = h.shape[0]
        e = torch.empty(P, self.E, device=self.device)
        L.conv_gemm(h, self._lin["W"], e, B=1, T=P, Cin=self.H, N=self.E, Np=self._lin["Np"], Kp=self._lin["Kp"], bias=self._lin["bias"],
                    act=L.ACT_RELU, mask_rows=False)
        out = torch.empty_like(e)
        L.check(lib.ss_l2norm_rows(L.ptr(e), L.ptr(out), P, self.E, L.stream_ptr()), "ss_l2norm_rows")
        return out

    __call__ = forward
