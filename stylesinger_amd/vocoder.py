"""HiFi-GAN-NSF vocoder plugin on the HIP path.

Mirrors the reference's plugin surface (tasks/tts/vocoder_infer/base_vocoder.py:6-18 and
tasks/tts/vocoder_infer/hifigan_nsf.py:46-75): a registry filled by `@register_vocoder(name)`,
`get_vocoder_cls(hparams)`, and an instance API `spec2wav(mel[T,80], f0=[T]) -> wav[T*hop]`.
`HifiGanGeneratorHIP` accepts the reference generator's checkpoint `state_dict` (weight-norm
parametrisation included) and folds/relayouts it once on the device.
"""
import ctypes

import numpy as np
import torch

from . import lib as L
from . import spec as _spec
from .config import make_hparams, make_vocoder_config

REGISTERED_VOCODERS = {}


def register_vocoder(name):
    def _f(cls):
        REGISTERED_VOCODERS[name] = cls
        return cls
    return _f


def get_vocoder_cls(hparams):
    return REGISTERED_VOCODERS[hparams["vocoder"]]


class HifiGanGeneratorHIP(torch.nn.Module):
    """Device-side HifiGanGenerator (modules/hifigan/hifigan_nsf.py:105-169), inference only."""

    def __init__(self, h=None):
        super().__init__()
        self.h = make_vocoder_config(h)
        self._names = []
        for name, shape in _spec.vocoder_spec(self.h):
            self._names.append(name)
            self.register_buffer("p__" + name.replace(".", "__"), torch.zeros(*shape))
        self._pk = None
        self._weights_version = 0
        self._packed_version = -1
        self.hop = int(np.prod(self.h["upsample_rates"]))

    def p(self, name):
        return getattr(self, "p__" + name.replace(".", "__"))

    def state_dict(self, *a, **k):
        return {n: self.p(n) for n in self._names}

    def load_state_dict(self, state_dict, strict=True):
        missing = [n for n in self._names if n not in state_dict]
        unexpected = [k for k in state_dict if k not in set(self._names)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"HifiGanGeneratorHIP.load_state_dict: missing={missing[:5]} unexpected={unexpected[:5]}")
        with torch.no_grad():
            for n in self._names:
                if n in state_dict:
                    self.p(n).copy_(state_dict[n])
        self._weights_version += 1
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def remove_weight_norm(self):
        """Kept for call-site compatibility (hifigan_nsf.py:171-178): folding happens in pack()."""
        return self

    def _wn(self, prefix):
        v, g = self.p(prefix + ".weight_v"), self.p(prefix + ".weight_g")
        return v, L.weight_norm_scale(v, g)

    def pack(self):
        h = self.h
        dev = self.p("conv_pre.bias").device
        if dev.type != "cuda":
            raise L.StyleSingerHipError("HifiGanGeneratorHIP needs its weights on a GPU: there is no CPU path")
        keep = []
        hg = L.HifiGan()
        rates, ks = h["upsample_rates"], h["upsample_kernel_sizes"]
        hg.n_ups, hg.n_kernels, hg.c0 = len(rates), len(h["resblock_kernel_sizes"]), h["upsample_initial_channel"]
        hg.sr, hg.harmonics = h["audio_sample_rate"], h["harmonic_num"]
        import os
        hg.mfma_bf16 = 1 if os.environ.get("SS_PRECISION", h.get("mfma_precision", "fp32")) == "bf16" else 0
        # grouped Winograd F(4,3) ResBlock convs (fp32 mode; SS_VOC_WINO=0 keeps the direct convs)
        hg.wino = 1 if (not hg.mfma_bf16 and os.environ.get("SS_VOC_WINO", str(h.get("vocoder_wino", 1))) != "0") else 0
        lib = L.load()
        for i, (u, k) in enumerate(zip(rates, ks)):
            hg.up_rate[i], hg.up_k[i] = u, k
        for i in range(len(rates), L.SS_HG_MAX_UPS):
            hg.up_rate[i], hg.up_k[i] = 1, 1
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            hg.rb_k[j] = k
            for m, d in enumerate(h["resblock_dilation_sizes"][j]):
                hg.rb_d[j][m] = d

        def hold(t):
            keep.append(t)
            return t.data_ptr()

        v, s0 = self._wn("conv_pre")
        hg.w_pre = hold(L.pack_conv_weight(v, scale0=s0))
        hg.b_pre = hold(L.pack_bias(self.p("conv_pre.bias")))
        nk = hg.n_kernels
        for i, u in enumerate(rates):
            v, s0 = self._wn(f"ups.{i}")
            for g in range(2):
                hg.w_up[i][g] = hold(L.pack_convtr_weight(v, s0, u, g))
            hg.b_up[i] = hold(L.pack_bias(self.p(f"ups.{i}.bias"), repeat=u))
            hg.w_noise[i] = hold(self.p(f"noise_convs.{i}.weight").contiguous())
            hg.b_noise[i] = hold(self.p(f"noise_convs.{i}.bias").contiguous())
            for j in range(nk):
                for m in range(3):
                    pfx = f"resblocks.{i * nk + j}"
                    k, d = h["resblock_kernel_sizes"][j], h["resblock_dilation_sizes"][j][m]
                    cout = hg.c0 >> (i + 1)
                    v, s0 = self._wn(f"{pfx}.convs1.{m}")
                    hg.w_rb1[i][j][m] = hold(L.pack_conv_weight(v, scale0=s0))
                    hg.b_rb1[i][j][m] = hold(L.pack_bias(self.p(f"{pfx}.convs1.{m}.bias")))
                    if hg.wino and lib.ss_wino43_conv_ok(cout, k, d):   # grouped F(4,3) pack of the folded weight
                        hg.w_rb1_wino[i][j][m] = hold(L.pack_conv_weight(L.wino43_group_weight(v * s0.view(-1, 1, 1))))
                    v, s0 = self._wn(f"{pfx}.convs2.{m}")
                    hg.w_rb2[i][j][m] = hold(L.pack_conv_weight(v, scale0=s0))
                    hg.b_rb2[i][j][m] = hold(L.pack_bias(self.p(f"{pfx}.convs2.{m}.bias")))
                    if hg.wino and lib.ss_wino43_conv_ok(cout, k, 1):
                        hg.w_rb2_wino[i][j][m] = hold(L.pack_conv_weight(L.wino43_group_weight(v * s0.view(-1, 1, 1))))
        v, s0 = self._wn("conv_post")
        hg.w_post = hold((v * s0.view(-1, 1, 1)).contiguous())  # 1 x c_last x 7 filter, consumed raw by conv_post_kernel
        hg.b_post = hold(self.p("conv_post.bias").contiguous())
        hg.src_w = hold(self.p("m_source.l_linear.weight").reshape(-1).contiguous())
        hg.src_b = hold(self.p("m_source.l_linear.bias").contiguous())
        self._pk = dict(hg=hg, keep=keep)
        self._packed_version = self._weights_version
        self._pack_device = dev
        torch.cuda.synchronize()

    def _ensure_packed(self):
        dev = self.p("conv_pre.bias").device
        if self._pk is None or self._packed_version != self._weights_version or self._pack_device != dev:
            self.pack()

    @torch.no_grad()
    def forward(self, mel, f0, lens=None, noise=None, seed=1234, return_source=False):
        """mel [B,T,80] (channels-last, already clipped), f0 [B,T] Hz -> wav [B, T*hop]."""
        self._ensure_packed()
        lib = L.load()
        hg = self._pk["hg"]
        mel = mel.contiguous().float()
        f0 = f0.contiguous().float()
        B, T, _ = mel.shape
        dev = mel.device
        Ls = T * self.hop
        wav = torch.empty(B, Ls, device=dev, dtype=torch.float32)
        har = torch.empty(B, Ls, device=dev, dtype=torch.float32) if return_source else None
        ws_bytes = lib.ss_hifigan_workspace_bytes(ctypes.addressof(hg), B, T)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        ri = sn = None
        if noise is not None:
            ri = noise["rand_ini"].to(dev).contiguous().float()
            sn = noise["sine_noise"].to(dev).contiguous().float()
        L.check(lib.ss_hifigan_forward(ctypes.addressof(hg), L.ptr(mel), L.ptr(f0), L.ptr(lens), B, T, L.ptr(ri), L.ptr(sn), seed,
                                       L.ptr(wav), L.ptr(har), L.ptr(ws), ws_bytes, L.stream_ptr()), "ss_hifigan_forward")
        return (wav, har) if return_source else wav


class BaseVocoder:
    def spec2wav(self, mel, **kwargs):
        raise NotImplementedError


@register_vocoder("HifiGAN_NSF")
class HifiGAN(BaseVocoder):
    """Drop-in for tasks/tts/vocoder_infer/hifigan_nsf.py::HifiGAN (B=1, numpy in/out) + a batched device API."""

    def __init__(self, config=None, state_dict=None, device="cuda", hparams=None):
        self.hparams = make_hparams(hparams)
        self.config = make_vocoder_config(config)
        self.device = torch.device(device)
        self.model = HifiGanGeneratorHIP(self.config)
        if state_dict is not None:
            self.model.load_state_dict(state_dict, strict=True)
        self.model.remove_weight_norm()
        self.model.eval().to(self.device)

    def spec2wav(self, mel, **kwargs):
        """mel [T,80] numpy, f0=[T] numpy Hz -> wav [T*hop] numpy (hifigan_nsf.py:62-75)."""
        f0 = kwargs.get("f0")
        if f0 is None or not self.hparams.get("use_nsf", True):
            raise NotImplementedError("the HIP vocoder implements the NSF path (use_nsf: true, f0 given)")
        c = torch.as_tensor(np.asarray(mel), dtype=torch.float32)[None].to(self.device)
        f = torch.as_tensor(np.asarray(f0), dtype=torch.float32)[None].to(self.device)
        y = self.model(c, f, noise=kwargs.get("noise"), seed=kwargs.get("seed", 1234))
        return y.view(-1).cpu().numpy()

    def spec2wav_batch(self, mel, f0, lens=None, **kwargs):
        """Device tensors in/out: mel [B,T,80], f0 [B,T], lens int32 [B] -> wav [B,T*hop]."""
        return self.model(mel, f0, lens=lens, **kwargs)
