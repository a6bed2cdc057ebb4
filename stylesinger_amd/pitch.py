"""Reference-f0 conditioning: mirror of `utils/pitch_utils.py:34-62` (`norm_f0`, `norm_interp_f0`).

`inference/StyleSinger.py:152` feeds the tracker's contour (Hz, 0 = unvoiced) through `norm_interp_f0` before the model sees
it: log2(f0 + 1e-8) (`pitch_norm: log`), unvoiced frames replaced by a linear interpolation between their voiced neighbours
(`np.interp`: held flat beyond the first / last voiced frame), all-unvoiced contours become 0.

Two forms with the same contract:
  * `norm_interp_f0(f0, hparams)`      host numpy, what `StyleSingerInfer.input_to_batch` calls for one utterance;
  * `norm_interp_f0_device(f0, lens)`  `[B, T]` on the GPU (`ss_norm_interp_f0`), what `preprocess_batch` uses so a batch
                                       goes from tracker output to `infer_batch` without a host round trip.
"""
import numpy as np
import torch

from . import lib as L


def norm_f0(f0, uv, hparams):
    """utils/pitch_utils.py:34-45 for numpy input. Returns a new array."""
    f0 = np.array(f0, copy=True)
    if hparams.get("pitch_norm", "log") == "standard":
        f0 = (f0 - hparams.get("f0_mean", 400)) / hparams.get("f0_std", 100)
    if hparams.get("pitch_norm", "log") == "log":
        f0 = np.log2(f0 + 1e-8)
    if uv is not None and hparams.get("use_uv", True):
        f0[uv > 0] = 0
    return f0


def norm_interp_f0(f0, hparams):
    """utils/pitch_utils.py:47-62: f0 [T] in Hz (numpy or torch, any float dtype; the arithmetic runs in the input's dtype
    as numpy does there) -> (f0 [T] float32 tensor, uv [T] float32 tensor)."""
    is_torch = isinstance(f0, torch.Tensor)
    device = f0.device if is_torch else None
    x = f0.detach().cpu().numpy() if is_torch else np.asarray(f0)
    uv = x == 0
    y = norm_f0(x, uv, hparams)
    n_uv = int(uv.sum())
    if n_uv == len(y):
        y[uv] = 0
    elif n_uv > 0:
        y[uv] = np.interp(np.where(uv)[0], np.where(~uv)[0], y[~uv])
    f0_t = torch.from_numpy(np.asarray(y, dtype=np.float32))
    uv_t = torch.from_numpy(uv.astype(np.float32))
    if is_torch:
        f0_t, uv_t = f0_t.to(device), uv_t.to(device)
    return f0_t, uv_t


@torch.no_grad()
def norm_interp_f0_device(f0_hz, lens=None, hparams=None):
    """f0_hz fp32 [B, T] on the device (0 = unvoiced), lens int32 [B] valid frames per item (default T) ->
    (f0 [B, T], uv [B, T]) fp32: per item `norm_interp_f0` of its first lens[b] frames; frames >= lens[b] are 0. Contract: the entry takes
    FP32 Hz, i.e. what numpy computes on a float32 contour (log2 and the interpolation weights in double, the interpolation endpoints rounded to
    fp32 first); against the reference's float64 tracker output that is within 1 fp32 ulp of values in [6, 10] (tests allow 2e-6), not
    bit-identical to the host path `norm_interp_f0` of this module, which `input_to_batch` uses on the float64 contour."""
    hp = hparams or {}
    if hp.get("pitch_norm", "log") != "log" or not hp.get("use_uv", True):
        raise NotImplementedError("norm_interp_f0_device: only pitch_norm='log' with use_uv (the reference's setting)")
    if f0_hz.device.type != "cuda":
        raise L.StyleSingerHipError("norm_interp_f0_device needs device tensors: there is no CPU path")
    x = f0_hz.float().contiguous()
    B, T = x.shape
    out = torch.empty_like(x)
    uv = torch.empty_like(x)
    lp = None if lens is None else lens.to(device=x.device, dtype=torch.int32).contiguous()
    L.check(L.load().ss_norm_interp_f0(L.ptr(x), L.ptr(lp), L.ptr(out), L.ptr(uv), B, T, L.stream_ptr()), "ss_norm_interp_f0")
    return out, uv
