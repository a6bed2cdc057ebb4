"""Output writer of the batched inference path.

The reference's test step handles one utterance at a time (`assert len(predictions) == 1`,
tasks/StyleSinger/stylesinger.py:199-202), crops the all-zero frames, vocodes and hands the waveform to
`utils/audio.py:12-17 save_wav` (`wav * 32767 -> int16`, optional peak normalisation) in a worker pool.  Here the whole
batch is vocoded at once on the device, quantised to PCM16 by `ss_wav_to_pcm16`, cropped per item by its frame count and
written as RIFF/WAVE files by a small thread pool (file IO only).
"""
import os
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import lib as L


def wav_to_pcm16(wav, lens=None, hop=1, norm=False):
    """wav fp32 [B, L] on the device -> int16 [B, L] on the device (samples past lens[b]*hop are 0).
    norm=True divides each item by its peak first (`out_wav_norm`)."""
    if wav.device.type != "cuda":
        raise L.StyleSingerHipError("wav_to_pcm16 needs the waveform on a GPU: there is no CPU path")
    wav = wav.contiguous().float()
    B, n = wav.shape
    pcm = torch.empty(B, n, device=wav.device, dtype=torch.int16)
    lib = L.load()
    if norm:
        peak = wav.abs().amax(dim=1).clamp_min(1e-20).cpu().tolist()
        for b in range(B):
            L.check(lib.ss_wav_to_pcm16(L.ptr(wav[b]), L.ptr(pcm[b]), n, float(np.float32(32767.0) / np.float32(peak[b])), L.stream_ptr()), "pcm16")
    else:
        L.check(lib.ss_wav_to_pcm16(L.ptr(wav), L.ptr(pcm), B * n, 32767.0, L.stream_ptr()), "pcm16")
    if lens is not None:
        idx = torch.arange(n, device=wav.device)[None, :]
        pcm.masked_fill_(idx >= (lens.to(torch.int64) * hop)[:, None], 0)
    return pcm


def write_wav_pcm16(path, pcm, sr):
    """Mono 16-bit RIFF/WAVE (what scipy.io.wavfile.write emits for an int16 vector)."""
    pcm = np.ascontiguousarray(np.asarray(pcm, dtype="<i2").reshape(-1))
    data = pcm.tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(data)) + data)


def save_wav(wav, path, sr, norm=False):
    """utils/audio.py:12-17 for ONE host waveform (the entry point's `example_run`): optional peak normalisation, `wav * 32767` truncated to int16
    (numpy's astype, as there), mono 16-bit RIFF/WAVE. The input array is not modified."""
    w = np.asarray(wav, dtype=np.float32)
    if norm:
        w = w / np.abs(w).max()
    write_wav_pcm16(path, (w * np.float32(32767)).astype(np.int16), sr)


class WavWriter:
    """Asynchronous per-item writer: the device->host copy of a batch happens once, the files are written off-thread."""

    def __init__(self, out_dir, sr, workers=4):
        self.out_dir, self.sr = out_dir, int(sr)
        os.makedirs(out_dir, exist_ok=True)
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.futures = []

    def submit_batch(self, names, pcm, lens, hop):
        host = pcm.cpu().numpy()
        ls = lens.cpu().tolist()
        for b, name in enumerate(names):
            path = os.path.join(self.out_dir, f"{name}.wav")
            self.futures.append(self.pool.submit(write_wav_pcm16, path, host[b, : ls[b] * hop].copy(), self.sr))

    def close(self):
        for f in self.futures:
            f.result()
        self.pool.shutdown()
