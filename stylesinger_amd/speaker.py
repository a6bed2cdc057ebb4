"""Speaker encoder on the GPU: `inference/StyleSinger.py:100,104` builds `spk_embed` with `resemblyzer.VoiceEncoder().embed_utterance(wav)`
(SURVEY.md §8f-1). resemblyzer is an UN-VENDORED dependency of the reference (requirements.txt: resemblyzer==0.1.1.dev0; its weights ship inside
the pip package, not in /root/reference), so this file restates the package's published algorithm - parity is UNPINNED: there is no golden of the
real package to check against here, only the oracle's restatement of the same text (oracle/restatement.py::speaker_embed):

  VoiceEncoder = LSTM(40 -> 256, 3 layers) + Linear(256, 256) + ReLU, L2-normalised per partial (the architecture the reference's own emotion
  encoder was derived from: data_gen/tts/emotion/model.py:11-60); embed_utterance(wav, rate=1.3, min_coverage=0.75) = partial utterances of 160
  frames of a 40-channel mel at 16 kHz / 10 ms, one every round(16000 / rate / 160) = 77 frames, the last one kept if it is covered to 75 %,
  embedding = L2-normalised mean of the partial embeddings.

Every launch is one the emotion encoder already makes (EmotionEncoderHIP.forward: per-layer input projections as one fp32-MFMA GEMM,
ss_lstm_layer recurrences, linear + ReLU epilogue, ss_l2norm_rows; ss_mean_l2norm) - only the slicing differs. The 40-mel front end is
stylesinger_amd.frontend.EmotionMelFrontendHIP (the same librosa.feature.melspectrogram parameters in both packages)."""
import numpy as np
import torch

from . import lib as L
from .emotion import PARTIALS_N_FRAMES, SAMPLING_RATE, MEL_WINDOW_STEP_MS, EmotionEncoderHIP


def compute_partial_slices(n_samples, rate=1.3, min_coverage=0.75):
    """resemblyzer.VoiceEncoder.compute_partial_slices (0.1.1.dev0): `rate` partial utterances per second instead of the emotion encoder's
    fixed overlap. Returns (wav_slices, mel_slices)."""
    assert 0 < min_coverage <= 1
    spf = int(SAMPLING_RATE * MEL_WINDOW_STEP_MS / 1000)
    n_frames = int(np.ceil((n_samples + 1) / spf))
    frame_step = int(np.round((SAMPLING_RATE / rate) / spf))
    assert 0 < frame_step, "the rate is too high"
    assert frame_step <= PARTIALS_N_FRAMES, f"the rate is too low, it should be {SAMPLING_RATE / (spf * PARTIALS_N_FRAMES)} at least"
    wav, mel = [], []
    steps = max(1, n_frames - PARTIALS_N_FRAMES + frame_step + 1)
    for i in range(0, steps, frame_step):
        mel.append(slice(i, i + PARTIALS_N_FRAMES))
        wav.append(slice(i * spf, (i + PARTIALS_N_FRAMES) * spf))
    last = wav[-1]
    coverage = (n_samples - last.start) / (last.stop - last.start)
    if coverage < min_coverage and len(mel) > 1:
        mel, wav = mel[:-1], wav[:-1]
    return wav, mel


class SpeakerEncoderHIP(EmotionEncoderHIP):
    """state dict = resemblyzer's `pretrained.pt["model_state"]` (lstm.* / linear.*: the names of the emotion encoder's checkpoint)"""

    @torch.no_grad()
    def embed_partials(self, frames):
        """frames [P, 160, 40] -> (embed [256], partial_embeds [P, 256]): VoiceEncoder.forward on every partial (ReLU(linear(h_last)), L2 norm),
        then the L2-normalised mean"""
        part = self.forward(frames)
        out = torch.empty(self.E, device=self.device)
        L.check(L.load().ss_mean_l2norm(L.ptr(part), L.ptr(out), part.shape[0], self.E, L.stream_ptr()), "ss_mean_l2norm")
        return out, part

    @torch.no_grad()
    def embed_utterance_frames(self, mel_frames, n_samples=None, rate=1.3, min_coverage=0.75):
        """mel_frames [n_frames, 40] of one utterance (padded as embed_utterance pads the waveform) -> embedding [256]"""
        mel_frames = torch.as_tensor(mel_frames, dtype=torch.float32)
        if n_samples is None:
            n_samples = (mel_frames.shape[0] - 1) * int(SAMPLING_RATE * MEL_WINDOW_STEP_MS / 1000)
        _, mel_slices = compute_partial_slices(n_samples, rate, min_coverage)
        need = mel_slices[-1].stop
        if mel_frames.shape[0] < need:
            raise ValueError(f"mel has {mel_frames.shape[0]} frames, the partial slicing needs {need}: pad the waveform first")
        return self.embed_partials(torch.stack([mel_frames[s] for s in mel_slices]))[0]
