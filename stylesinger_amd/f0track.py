"""f0 tracker on the GPU: `inference/StyleSinger.py:112-135`

    f0 = parselmouth.Sound(wav, sr).to_pitch_ac(time_step=hop / sr, voicing_threshold=0.6, pitch_floor=80, pitch_ceiling=800).selected_array['frequency']
    f0 = np.pad(f0, [[2 * pad_size, len(mel) - len(f0) - 2 * pad_size]])       # onto the mel frame grid

parselmouth (praat-parselmouth==0.3.3) is an UN-VENDORED dependency of the reference: the kernels (`csrc/f0track.hip`, entry `ss_f0track`) follow
the published algorithm (Boersma 1993, Praat's "To Pitch (ac)" with a Hanning window) as `oracle/praat_pitch.py` restates it on the CPU.
PARITY UNPINNED: no output of the real package can be produced in this environment; the restatement is pinned by analytic known answers
(`tests/test_host_cpu.py`), the kernels by the restatement (`tests/test_gpu_round5.py`).

This module holds the host side only: the window / lag / frame geometry (a handful of float64 formulas per utterance, from the manual's
"as many frames as fit, centred" rule), the Hanning window and its normalised autocorrelation (computed once per geometry), the launch.
"""
import math

import numpy as np
import torch

from . import lib as L

# to_pitch_ac defaults the reference does not override (parselmouth 0.3.3)
MAX_CANDIDATES = 15
SILENCE_THRESHOLD = 0.03
OCTAVE_COST = 0.01
OCTAVE_JUMP_COST = 0.35
VOICED_UNVOICED_COST = 0.14
PERIODS_PER_WINDOW = 3.0


def geometry(sr, time_step, pitch_floor, pitch_ceiling):
    """The sample-count geometry of the analysis (independent of the sound's length)."""
    dx = 1.0 / sr
    nsamp_period = int(math.floor(1.0 / dx / pitch_floor))
    nsamp_window = int(math.floor(PERIODS_PER_WINDOW / pitch_floor / dx))
    halfw = nsamp_window // 2 - 1
    if halfw < 2:
        raise ValueError("f0 tracker: analysis window too short")
    nsamp_window = 2 * halfw
    hop = time_step / dx
    if abs(hop - round(hop)) > 1e-6:
        raise ValueError(f"f0 tracker: the time step must be a whole number of samples (got {hop})")
    return dict(dx=dx, sr=float(sr), time_step=float(time_step), pitch_floor=float(pitch_floor), pitch_ceiling=float(min(pitch_ceiling, 0.5 / dx)),
                nsamp_period=nsamp_period, halfnsamp_period=nsamp_period // 2 + 1, nsamp_window=nsamp_window, halfnsamp_window=halfw,
                maximum_lag=min(int(math.floor(nsamp_window / PERIODS_PER_WINDOW)) + 2, nsamp_window), nlag=nsamp_window // 2, hop=int(round(hop)),
                window_duration=PERIODS_PER_WINDOW / pitch_floor)


def frame_grid(g, n_samples):
    """(number of frames, 0-based index of the sample left of frame 0's centre): as many frames as fit, centred in the sound."""
    dx = g["dx"]
    duration = dx * n_samples
    if g["window_duration"] > duration:
        return 0, 0
    n_frames = int(math.floor((duration - g["window_duration"]) / g["time_step"])) + 1
    # first frame centre t1 = duration / 2 - (n_frames - 1) * time_step / 2, i.e. (n_samples - hop * (n_frames - 1)) / 2 samples: the sample left
    # of it in exact integer arithmetic (0-based; the manual's 1-based "low index" minus one). With an even sample count - always the case for
    # process_audio's n_mel * hop samples - the centres lie half way between samples; with an odd count they lie ON a sample and the manual's
    # float floor is decided by rounding noise frame by frame, which this grid does not imitate.
    left = (n_samples - g["hop"] * (n_frames - 1) - 1) // 2
    return n_frames, left


_tables = {}


def _window_tables(g, device):
    key = (g["nsamp_window"], g["nlag"], str(device))
    if key not in _tables:
        nw = g["nsamp_window"]
        i = np.arange(1, nw + 1, dtype=np.float64)
        w = 0.5 - 0.5 * np.cos(i * 2.0 * np.pi / (nw + 1))
        nfft = 1
        while nfft < nw * 1.5:
            nfft *= 2
        spec = np.fft.rfft(w, nfft)
        r = np.fft.irfft(spec.real ** 2 + spec.imag ** 2, nfft)
        wr = r[:g["nlag"] + 1] / r[0]
        _tables[key] = (torch.from_numpy(w).to(device), torch.from_numpy(np.ascontiguousarray(wr)).to(device))
    return _tables[key]


WS_CAP_BYTES = 256 << 20   # peak workspace of one tracker launch group (a 30 s item needs ~40 MB)
N_TRACK_CALLS = 0   # launches of the tracker since import (tests assert that the entry point tracks each reference audio ONCE)


@torch.no_grad()
def track_f0_device(wavs, n_samples, n_out, sr=48000, hop_size=256, pitch_floor=80.0, pitch_ceiling=800.0, voicing_threshold=0.6):
    """wavs fp32 [B, L] on the device (zero beyond n_samples[b]; host ints) -> f0 fp32 [B, n_out] in Hz (0 = unvoiced) on the mel frame grid:
    frame i of the tracker at column 2 * pad_size + i (pad_size = 2 at hop 256, 4 at hop 128), zeros elsewhere - what
    inference/StyleSinger.py:112-135 hands `norm_interp_f0`."""
    if wavs.device.type != "cuda":
        raise L.StyleSingerHipError("track_f0_device needs device tensors: there is no CPU path")
    global N_TRACK_CALLS
    N_TRACK_CALLS += 1
    pad_size = {128: 4, 256: 2}[int(hop_size)]
    time_step = hop_size / sr * 1000 / 1000            # the reference's own expression (ms and back)
    g = geometry(sr, time_step, pitch_floor, pitch_ceiling)
    wavs = wavs.float().contiguous()
    B = wavs.shape[0]
    ns = [int(v) for v in n_samples]
    grid = [frame_grid(g, n) for n in ns]
    for (nf, left), n in zip(grid, ns):
        if nf > 0 and (left + 1 - g["nsamp_period"] < 0 or left + 1 - g["halfnsamp_window"] < 0 or
                       left + (nf - 1) * g["hop"] + 1 + max(g["nsamp_period"], g["halfnsamp_window"]) > max(n, 0) or n > wavs.shape[1]):
            raise ValueError("f0 tracker: a frame window leaves the waveform buffer")
    max_frames = max(1, max(nf for nf, _ in grid))
    dev = wavs.device
    win, win_r = _window_tables(g, dev)
    prm = L.F0TrackParams()
    prm.sample_rate, prm.time_step, prm.pitch_floor, prm.pitch_ceiling = g["sr"], g["time_step"], g["pitch_floor"], g["pitch_ceiling"]
    prm.voicing_threshold, prm.silence_threshold = float(voicing_threshold), SILENCE_THRESHOLD
    prm.octave_cost, prm.octave_jump_cost, prm.voiced_unvoiced_cost = OCTAVE_COST, OCTAVE_JUMP_COST, VOICED_UNVOICED_COST
    for k in ("nsamp_window", "halfnsamp_window", "nsamp_period", "halfnsamp_period", "maximum_lag", "nlag", "hop"):
        setattr(prm, k, g[k])
    lib = L.load()
    # ONE int32 table [3][B] that stays referenced until the launches are queued (three temporaries would be freed - and their memory handed to
    # the next one - before the kernels read them)
    meta = torch.tensor([ns, [nf for nf, _ in grid], [lf for _, lf in grid]], dtype=torch.int32).to(dev)
    out = torch.empty(B, int(n_out), device=dev, dtype=torch.float32)
    import ctypes
    # The workspace holds the float64 autocorrelation of every frame (max_frames x (nlag + 1) x 8 B per item: 40 MB for a 30 s item). A batch is
    # tracked in groups of items whose workspace stays below WS_CAP_BYTES - 32 x 30 s would otherwise ask for 1.3 GB at once; items are independent,
    # so the grouping changes nothing but the peak allocation (the same kernels, the same per-item arithmetic).
    per_item = max(1, lib.ss_f0track_workspace_bytes(1, max_frames, g["nlag"]))
    group = max(1, min(B, WS_CAP_BYTES // per_item))
    ws = torch.empty(lib.ss_f0track_workspace_bytes(group, max_frames, g["nlag"]), device=dev, dtype=torch.uint8)
    for b0 in range(0, B, group):
        nb = min(group, B - b0)
        L.check(lib.ss_f0track(L.ptr(wavs[b0:]), wavs.shape[1], L.ptr(meta[0, b0:]), L.ptr(meta[1, b0:]), L.ptr(meta[2, b0:]), nb,
                               max_frames, ctypes.byref(prm), L.ptr(win), L.ptr(win_r), L.ptr(out[b0:]), int(n_out), 2 * pad_size, L.ptr(ws), ws.numel(),
                               L.stream_ptr()), "ss_f0track")
    meta.record_stream(torch.cuda.current_stream(dev))
    return out
