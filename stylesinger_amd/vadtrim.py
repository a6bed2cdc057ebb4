"""`trim_long_silences` of the emotion branch (data_gen/tts/emotion/audio.py:58-100, called by `preprocess_wav`, :38) around the caller's VAD flags.

The reference asks `webrtcvad.Vad(mode=3).is_speech` for one flag per 30 ms window of the volume-normalised 16-bit PCM. webrtcvad is an un-vendored
C library (a fixed-point GMM over sub-band energies; its model tables are not in the reference tree and there is no published text to restate), so the
DECISION stays the caller's: `flags[b, w] = vad.is_speech(pcm16(window w of item b), 16000)`. Everything around it - cutting the waveform to whole
windows, the moving average of width 8 rounded half to even, the dilation by 6 windows, the compaction of the kept windows - runs on the device
(`ss_vad_trim`) and is pinned bit-exactly against the REAL function run with injected flags (`tests/golden/vad_trim.pt`).
"""
import numpy as np
import torch

from . import lib as L

# data_gen/tts/emotion/params_data.py
VAD_WINDOW_MS = 30
VAD_MOVING_AVERAGE_WIDTH = 8
VAD_MAX_SILENCE_LENGTH = 6
SAMPLING_RATE = 16000


def window_mask(flags, avg_width=VAD_MOVING_AVERAGE_WIDTH, max_silence=VAD_MAX_SILENCE_LENGTH):
    """Host mirror (integer logic) of the reference's smoothing + dilation: flags [nW] (0/1) -> kept-window mask [nW] bool."""
    f = np.asarray(flags).astype(np.int64)
    n = len(f)
    lp, rp = (avg_width - 1) // 2, avg_width // 2
    pad = np.concatenate([np.zeros(lp, np.int64), f, np.zeros(rp, np.int64)])
    cnt = np.array([pad[i:i + avg_width].sum() for i in range(n)], dtype=np.int64)
    m1 = 2 * cnt > avg_width                      # np.round(cnt / width): exactly one half rounds to the even 0
    o = (max_silence + 1) // 2                    # binary_dilation(mask, ones(max_silence + 1)): origin at the centre
    out = np.zeros(n, dtype=bool)
    for i in range(n):
        lo, hi = i - max_silence + o, i + o       # j = i - k + o for k = 0 .. max_silence
        out[i] = m1[max(lo, 0):min(hi, n - 1) + 1].any() if hi >= 0 and lo <= n - 1 else False
    return out


def have_webrtcvad():
    try:
        import webrtcvad  # noqa: F401
        return True
    except Exception:
        return False


def webrtc_flags(wavs, lens, sr=SAMPLING_RATE):
    """The reference's own decision loop (data_gen/tts/emotion/audio.py:66-81) on the HOST with the webrtcvad package: per item the waveform cut
    to whole 30 ms windows, `round(wav * 32767)` as 16-bit PCM, `Vad(mode=3).is_speech(window, sample_rate)` per window. wavs [B, L] (device or
    host; the volume-normalised audio), lens host ints -> uint8 [B, max windows] (0 beyond an item's windows). The package is un-vendored: this
    raises ImportError where it is missing (callers then pass flags or opt out explicitly)."""
    import struct
    import webrtcvad
    spw = VAD_WINDOW_MS * sr // 1000
    x = wavs.detach().float().cpu().numpy() if torch.is_tensor(wavs) else np.asarray(wavs, dtype=np.float32)
    ns = [min(int(n), x.shape[1]) for n in lens]
    max_w = max(1, max(n // spw for n in ns))
    out = np.zeros((len(ns), max_w), dtype=np.uint8)
    for b, n in enumerate(ns):
        nw = n // spw
        w = x[b, :nw * spw]
        pcm = struct.pack("%dh" % len(w), *(np.round(w * 32767)).astype(np.int16))
        vad = webrtcvad.Vad(mode=3)
        for i in range(nw):
            out[b, i] = 1 if vad.is_speech(pcm[i * spw * 2:(i + 1) * spw * 2], sample_rate=sr) else 0
    return out


@torch.no_grad()
def trim_long_silences_device(wavs, lens, flags, sr=SAMPLING_RATE):
    """wavs fp32 [B, L] on the device (zero beyond lens[b]; host ints), flags [B, nW] (0/1; window w = samples [w * spw, (w + 1) * spw), spw = 30 ms)
    -> (trimmed [B, L'] fp32 zero padded, kept lengths as a list of host ints... computed on the device: int32 tensor [B])."""
    if wavs.device.type != "cuda":
        raise L.StyleSingerHipError("trim_long_silences_device needs device tensors: there is no CPU path")
    spw = VAD_WINDOW_MS * sr // 1000
    wavs = wavs.float().contiguous()
    B, Lx = wavs.shape
    ns = [min(int(n), Lx) for n in lens]
    max_w = max(1, max(n // spw for n in ns))
    fl = torch.as_tensor(np.asarray(flags)).to(torch.uint8)
    if fl.dim() != 2 or fl.shape[0] != B or fl.shape[1] < max_w:
        raise ValueError(f"trim_long_silences_device: flags must be [B, >= {max_w}] (one per {spw}-sample window), got {tuple(fl.shape)}")
    fl = fl.contiguous().to(wavs.device)
    if Lx < max_w * spw:
        raise ValueError("trim_long_silences_device: waveform buffer shorter than its windows")
    meta = torch.tensor(ns, dtype=torch.int32).to(wavs.device)
    out = torch.empty(B, max_w * spw, device=wavs.device, dtype=torch.float32)
    out_lens = torch.empty(B, device=wavs.device, dtype=torch.int32)
    win_dst = torch.empty(B, max_w, device=wavs.device, dtype=torch.int32)
    L.check(L.load().ss_vad_trim(L.ptr(wavs), Lx, L.ptr(meta), L.ptr(fl), fl.shape[1], B, max_w, spw, VAD_MOVING_AVERAGE_WIDTH, VAD_MAX_SILENCE_LENGTH,
                                 L.ptr(out), out.shape[1], L.ptr(out_lens), L.ptr(win_dst), L.stream_ptr()), "ss_vad_trim")
    for t in (meta, fl, win_dst):
        t.record_stream(torch.cuda.current_stream(wavs.device))
    return out, out_lens
