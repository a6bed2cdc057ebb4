"""CPU restatement (numpy, float64) of the f0 tracker the reference calls - TEST INFRASTRUCTURE ONLY.

`inference/StyleSinger.py:125-127`:
    parselmouth.Sound(wav, 48000).to_pitch_ac(time_step=hop/sr, voicing_threshold=0.6, pitch_floor=80, pitch_ceiling=800).selected_array['frequency']
parselmouth (praat-parselmouth==0.3.3, requirements.txt:17) is an UN-VENDORED dependency: neither the package nor Praat's sources are
under /root/reference, so PARITY IS UNPINNED - no output of the real package can be generated here. This file restates the PUBLISHED
algorithm - P. Boersma (1993), "Accurate short-term analysis of the fundamental frequency and the harmonics-to-noise ratio of a sampled
sound", IFA Proceedings 17, as Praat's "Sound: To Pitch (ac)..." implements it with a Hanning window - and is pinned only by analytic
known answers (tests/test_host_cpu.py: stationary sines / harmonic complexes come back at their f0, silence and low-level noise come
back unvoiced, the frame count and frame times follow the manual's formulas).

Algorithm (defaults of to_pitch_ac other than the three the reference passes: max_number_of_candidates 15, very_accurate False,
silence_threshold 0.03, octave_cost 0.01, octave_jump_cost 0.35, voiced_unvoiced_cost 0.14):
  1. window = 3 periods of the pitch floor, forced to an even number of samples; as many frames as fit, centred in the sound;
  2. per frame: subtract the local mean (one floor period to both sides), Hanning window, autocorrelation normalised by r(0) and divided
     by the window's own normalised autocorrelation (eq. 9 of the paper) for lags up to half the window;
  3. candidates = local maxima above half the voicing threshold, strength by windowed-sinc interpolation (depth 30), at most 15 kept
     (weakest replaced, with the octave-cost bias towards high frequencies), then each refined by maximising the sinc interpolation
     (depth 70, Brent's method) over [lag - 1, lag + 1]; strengths above 1 are reflected (1 / s);
  4. Viterbi over the candidates + the unvoiced candidate with the octave, octave-jump and voiced/unvoiced costs (eq. 24-27), time-step
     corrected by 0.01 / dt; candidates at or above the ceiling count as unvoiced in the path but keep their frequency in the output.
"""
import math

import numpy as np

MAX_CAND = 15
SILENCE_THRESHOLD = 0.03
OCTAVE_COST = 0.01
OCTAVE_JUMP_COST = 0.35
VOICED_UNVOICED_COST = 0.14


def geometry(n_samples, sr, time_step, pitch_floor, pitch_ceiling, periods_per_window=3.0):
    """Window / lag / frame geometry of the AC method with a Hanning window (interpolation depth 0.5)."""
    dx = 1.0 / sr
    duration = dx * n_samples
    nsamp_period = int(math.floor(1.0 / dx / pitch_floor))
    halfnsamp_period = nsamp_period // 2 + 1
    ceiling = min(pitch_ceiling, 0.5 / dx)
    dt_window = periods_per_window / pitch_floor
    nsamp_window = int(math.floor(dt_window / dx))
    halfnsamp_window = nsamp_window // 2 - 1
    if halfnsamp_window < 2:
        raise ValueError("analysis window too short")
    nsamp_window = halfnsamp_window * 2
    maximum_lag = min(int(math.floor(nsamp_window / periods_per_window)) + 2, nsamp_window)
    if dt_window > duration:
        raise ValueError("sound shorter than the analysis window")
    n_frames = int(math.floor((duration - dt_window) / time_step)) + 1
    mid = 0.5 * duration                                   # x1 - 0.5 dx + 0.5 duration with x1 = 0.5 dx
    t1 = mid - 0.5 * n_frames * time_step + 0.5 * time_step
    brent_ixmax = int(math.floor(nsamp_window * 0.5))
    nfft = 1
    while nfft < nsamp_window * 1.5:
        nfft *= 2
    return dict(dx=dx, nsamp_period=nsamp_period, halfnsamp_period=halfnsamp_period, ceiling=ceiling, nsamp_window=nsamp_window,
                halfnsamp_window=halfnsamp_window, maximum_lag=maximum_lag, n_frames=n_frames, t1=t1, brent_ixmax=brent_ixmax, nfft=nfft,
                time_step=time_step, pitch_floor=pitch_floor)


def hanning_window(n):
    i = np.arange(1, n + 1, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(i * 2.0 * np.pi / (n + 1))


def window_autocorrelation(window, nfft, n_lags):
    """normalised autocorrelation of the window, lags 0 .. n_lags (through the power spectrum, as the paper computes it)"""
    spec = np.fft.rfft(window, nfft)
    r = np.fft.irfft(spec.real ** 2 + spec.imag ** 2, nfft)
    return r[:n_lags + 1] / r[0]


def frame_start(g, i):
    """0-based first sample of frame i's window and of its local-mean span"""
    t = g["t1"] + i * g["time_step"]
    left = int(math.floor((t - 0.5 * g["dx"]) / g["dx"])) + 1     # 1-based index of the sample left of the frame centre
    right = left + 1
    return right - g["halfnsamp_window"] - 1, right - g["nsamp_period"] - 1, left + g["nsamp_period"]   # window start, mean start, mean end (excl.)


def interpolate_sinc(y, x, max_depth):
    """Windowed-sinc interpolation of y (1-based positions 1 .. len(y)) at the real position x: raised-cosine window of `max_depth`
    samples to either side, clipped at the array ends."""
    nx = len(y)
    midleft = int(math.floor(x))
    midright = midleft + 1
    if x > nx:
        return y[nx - 1]
    if x < 1:
        return y[0]
    if x == midleft:
        return y[midleft - 1]
    max_depth = min(max_depth, midright - 1, nx - midleft)
    if max_depth <= 0:
        return y[int(math.floor(x + 0.5)) - 1]
    if max_depth == 1:
        return y[midleft - 1] + (x - midleft) * (y[midright - 1] - y[midleft - 1])
    left, right = midright - max_depth, midleft + max_depth
    ix = np.arange(midleft, left - 1, -1)
    a = np.pi * (x - ix)
    sgn = np.where((midleft - ix) % 2 == 0, 1.0, -1.0)
    res = np.sum(y[ix - 1] * (0.5 * math.sin(math.pi * (x - midleft)) * sgn) / a * (1.0 + np.cos(a / (x - left + 1))))
    ix = np.arange(midright, right + 1)
    a = np.pi * (ix - x)
    sgn = np.where((ix - midright) % 2 == 0, 1.0, -1.0)
    res += np.sum(y[ix - 1] * (0.5 * math.sin(math.pi * (midright - x)) * sgn) / a * (1.0 + np.cos(a / (right - x + 1))))
    return float(res)


def brent_minimize(f, a, b, tol=1e-10, itermax=60):
    """Brent's minimiser (golden section + successive parabolic interpolation) of f on [a, b]: returns (x_min, f_min)."""
    golden = 1 - 0.6180339887498948482045868343656381177203
    sqrt_eps = math.sqrt(np.finfo(np.float64).eps)
    v = a + golden * (b - a)
    fv = f(v)
    x, w, fx, fw = v, v, fv, fv
    for _ in range(itermax):
        middle = 0.5 * (a + b)
        tol_act = sqrt_eps * abs(x) + tol / 3.0
        if abs(x - middle) + 0.5 * (b - a) <= 2.0 * tol_act:
            return x, fx
        new_step = golden * ((a - x) if x >= middle else (b - x))
        if abs(x - w) >= tol_act:   # try a parabola through x, v, w
            t = (x - w) * (fx - fv)
            q = (x - v) * (fx - fw)
            p = (x - v) * q - (x - w) * t
            q = 2.0 * (q - t)
            if q > 0.0:
                p = -p
            else:
                q = -q
            if abs(p) < abs(new_step * q) and p > q * (a - x + 2.0 * tol_act) and p < q * (b - x - 2.0 * tol_act):
                new_step = p / q
        if abs(new_step) < tol_act:
            new_step = tol_act if new_step > 0 else -tol_act
        t = x + new_step
        ft = f(t)
        if ft <= fx:
            if t < x:
                b = x
            else:
                a = x
            v, w, x = w, x, t
            fv, fw, fx = fw, fx, ft
        else:
            if t < x:
                a = t
            else:
                b = t
            if ft <= fw or w == x:
                v, w = w, t
                fv, fw = fw, ft
            elif ft <= fv or v == x or v == w:
                v, fv = t, ft
    return x, fx


def frame_candidates(r, g, voicing_threshold):
    """r: normalised, window-corrected autocorrelation for lags 0 .. brent_ixmax. Returns (freqs, strengths) of the voiced candidates."""
    bi, dx = g["brent_ixmax"], g["dx"]
    y = np.concatenate([r[:0:-1], r])                     # positions 1 .. 2 bi + 1 hold lags -bi .. bi
    off = bi + 1                                          # position = lag + off
    freqs, strengths, imax = [], [], []
    for i in range(2, min(g["maximum_lag"], bi)):
        if r[i] > 0.5 * voicing_threshold and r[i] > r[i - 1] and r[i] >= r[i + 1]:
            dr = 0.5 * (r[i + 1] - r[i - 1])
            d2r = 2.0 * r[i] - r[i - 1] - r[i + 1]
            f = 1.0 / dx / (i + dr / d2r)
            s = interpolate_sinc(y, 1.0 / dx / f + off, 30)
            if s > 1.0:
                s = 1.0 / s
            if len(freqs) < MAX_CAND - 1:                 # the voiceless candidate holds the first of the 15 places
                freqs.append(f), strengths.append(s), imax.append(i)
            else:
                weakest, place = 2.0, -1
                for k in range(len(freqs)):
                    local = strengths[k] - OCTAVE_COST * math.log2(g["pitch_floor"] / freqs[k])
                    if local < weakest:
                        weakest, place = local, k
                if s - OCTAVE_COST * math.log2(g["pitch_floor"] / f) > weakest:
                    freqs[place], strengths[place], imax[place] = f, s, i
    for k in range(len(freqs)):
        depth = 700 if freqs[k] > 0.3 / dx else 70
        xm, fm = brent_minimize(lambda x: -interpolate_sinc(y, x, depth), imax[k] + off - 1, imax[k] + off + 1)
        ymid = -fm
        freqs[k] = 1.0 / dx / (xm - off)
        strengths[k] = 1.0 / ymid if ymid > 1.0 else ymid
    return freqs, strengths


def analyse_frames(wav, sr, time_step, pitch_floor=80.0, pitch_ceiling=800.0, voicing_threshold=0.6):
    """-> (geometry, per-frame list of (freqs, strengths) incl. the voiceless candidate first, intensities, normalised autocorrelations)"""
    x = np.asarray(wav, dtype=np.float64)
    g = geometry(len(x), sr, time_step, pitch_floor, pitch_ceiling)
    nw, bi = g["nsamp_window"], g["brent_ixmax"]
    window = hanning_window(nw)
    window_r = window_autocorrelation(window, g["nfft"], bi)
    global_peak = float(np.max(np.abs(x - x.mean()))) if len(x) else 0.0
    frames, intens, acs = [], [], []
    lo = g["halfnsamp_window"] + 1 - g["halfnsamp_period"]
    hi = g["halfnsamp_window"] + g["halfnsamp_period"]
    lo, hi = max(lo, 1), min(hi, nw)
    for i in range(g["n_frames"]):
        if global_peak == 0.0:
            frames.append(([0.0], [0.0])), intens.append(0.0), acs.append(np.zeros(bi + 1))
            continue
        ws, ms, me = frame_start(g, i)
        frame = (x[ws:ws + nw] - x[ms:me].mean()) * window
        local_peak = float(np.max(np.abs(frame[lo - 1:hi])))
        intens.append(1.0 if local_peak > global_peak else local_peak / global_peak)
        if local_peak == 0.0:
            frames.append(([0.0], [0.0])), acs.append(np.zeros(bi + 1))
            continue
        spec = np.fft.rfft(frame, g["nfft"])
        ac = np.fft.irfft(spec.real ** 2 + spec.imag ** 2, g["nfft"])
        r = np.empty(bi + 1)
        r[0] = 1.0
        r[1:] = ac[1:bi + 1] / (ac[0] * window_r[1:bi + 1])
        f, s = frame_candidates(r, g, voicing_threshold)
        frames.append(([0.0] + f, [0.0] + s)), acs.append(r)
    return g, frames, np.asarray(intens), acs


def path_finder(frames, intens, g, voicing_threshold, silence_threshold=SILENCE_THRESHOLD, octave_cost=OCTAVE_COST,
                octave_jump_cost=OCTAVE_JUMP_COST, voiced_unvoiced_cost=VOICED_UNVOICED_COST):
    """Viterbi over the per-frame candidates -> selected frequency per frame (0 = unvoiced)."""
    n = len(frames)
    if n == 0:
        return np.zeros(0)
    ceiling = g["ceiling"]
    corr = 0.01 / g["time_step"]
    ojc, vuc = octave_jump_cost * corr, voiced_unvoiced_cost * corr
    voiced = lambda f: f > 0.0 and f < ceiling
    delta, psi = [], []
    for (fs, ss), it in zip(frames, intens):
        unv = 0.0 if silence_threshold <= 0 else 2.0 - it / (silence_threshold / (1.0 + voicing_threshold))
        unv = voicing_threshold + max(0.0, unv)
        delta.append([(s - octave_cost * math.log2(ceiling / f)) if voiced(f) else unv for f, s in zip(fs, ss)])
        psi.append([0] * len(fs))
    for i in range(1, n):
        f1s, f2s = frames[i - 1][0], frames[i][0]
        for c2, f2 in enumerate(f2s):
            best, place = -1e30, 0
            for c1, f1 in enumerate(f1s):
                v1, v2 = voiced(f1), voiced(f2)
                if not v2:
                    cost = 0.0 if not v1 else vuc
                else:
                    cost = vuc if not v1 else ojc * abs(math.log2(f1 / f2))
                val = delta[i - 1][c1] - cost + delta[i][c2]
                if val > best:
                    best, place = val, c1
            delta[i][c2] = best
            psi[i][c2] = place
    place = int(np.argmax(delta[-1]))
    out = np.zeros(n)
    for i in range(n - 1, -1, -1):
        out[i] = frames[i][0][place]
        place = psi[i][place]
    return out


def to_pitch_ac(wav, sr=48000, time_step=256 / 48000, pitch_floor=80.0, pitch_ceiling=800.0, voicing_threshold=0.6):
    """selected_array['frequency'] of Sound(wav, sr).to_pitch_ac(time_step, pitch_floor, voicing_threshold=..., pitch_ceiling=...)"""
    g, frames, intens, _ = analyse_frames(wav, sr, time_step, pitch_floor, pitch_ceiling, voicing_threshold)
    return path_finder(frames, intens, g, voicing_threshold)


def reference_f0(wav, n_mel, hop_size=256, sr=48000):
    """inference/StyleSinger.py:112-135: tracker contour padded (2 * pad_size frames left, the rest right) and fitted to the mel length."""
    pad_size = {128: 4, 256: 2}[hop_size]
    time_step = hop_size / sr * 1000
    f0 = to_pitch_ac(wav, sr, time_step / 1000, 80.0, 800.0, 0.6)
    lpad = pad_size * 2
    rpad = n_mel - len(f0) - lpad
    f0 = np.pad(f0, [[lpad, rpad]], mode="constant")
    delta = n_mel - len(f0)
    assert abs(delta) <= 8
    if delta > 0:
        f0 = np.concatenate([f0, [f0[-1]] * delta], 0)
    return f0[:n_mel]
