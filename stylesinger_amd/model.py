"""`StyleSingerHIP` — host-side mirror of `modules/StyleSinger/stylesinger.py::StyleSinger` for inference.

Same constructor (`StyleSingerHIP(dictionary, out_dims=None)`), same `forward(...) -> dict` keys, and it
loads the reference `state_dict` unchanged (names + shapes are the weight contract), so
`inference/StyleSinger.py::build_model` / `tasks/StyleSinger/stylesinger.py::build_tts_model` can swap
the class and keep `load_ckpt(model, ..., 'model', strict=False)`.

All arithmetic runs in libstylesinger_hip.so (hand-written gfx950 kernels) through `lib.py`; torch is
used for device buffers, views/concats (data movement only) and the stream.  There is no CPU path.
"""
import collections
import math

import numpy as np
import torch

from . import lib as L
from . import spec as _spec
from .config import make_hparams

_lib = L.load


class _Packed:
    """A conv/linear weight in the MFMA kernel's layout + its metadata."""
    __slots__ = ("W", "bias", "Cout", "Cin", "k", "Np", "Kp", "half")

    def __init__(self, W, bias, Cout, Cin, k, half=0):
        self.W, self.bias, self.Cout, self.Cin, self.k, self.half = W, bias, Cout, Cin, k, half
        self.Np, self.Kp = W.shape[0], W.shape[1] // k


def _sin_table(n, dim):
    """SinusoidalPositionalEmbedding.get_embedding (common_layers.py:107-124), host fp32, padding row 0 zeroed."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(n, -1)
    e[0, :] = 0
    return e


def _step_emb_table(steps, dim):
    """SinusoidalPosEmb(t) for t = 0..steps-1 (modules/diff/net.py:32-44)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = torch.arange(steps)[:, None].float() * e[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1).contiguous()



class _DiffPlan:
    """Static device buffers (+ optional captured hipGraphs) of the three diffusion loops for one (B, T).

    The loops are ~13 000 launches per pass; for small batches they are launch-bound, so the launch sequence is
    captured once per shape with `torch.cuda.CUDAGraph` (the raw HIP launches go to the capture stream) and replayed.
    Noise stays fresh across replays through the device seed word (`seed_dev` of the C-ABI)."""

    def __init__(self, model, B, T, dev):
        hp, pk, lib = model.hp, model._pk, _lib()
        H, M = hp["hidden_size"], hp["audio_num_mel_bins"]
        f32 = dict(device=dev, dtype=torch.float32)
        self.B, self.T = B, T
        self.seed = torch.zeros(1, device=dev, dtype=torch.int64)
        self.nonfinite = torch.zeros(1, device=dev, dtype=torch.int32)   # set by ss_mel_denorm when a valid frame is NaN / inf
        # f0 pair: items [0,B) = agnostic net, [B,2B) = specific net (grouped launches)
        self.lens2 = torch.zeros(2 * B, device=dev, dtype=torch.int32)
        self.lens = self.lens2[:B]
        self.cond2 = torch.empty(2 * B, T, H, **f32)
        self.cond_a, self.cond_b = self.cond2[:B], self.cond2[B:]
        self.lo2 = torch.empty(2 * B, T, **f32)
        self.hi2 = torch.empty(2 * B, T, **f32)
        self.f02 = torch.empty(2 * B, T, **f32)
        self.uv2 = torch.zeros(2 * B, T, device=dev, dtype=torch.int32)
        self.f0 = [self.f02[:B], self.f02[B:]]
        self.uv = [self.uv2[:B], self.uv2[B:]]
        self.ws_f0_bytes = lib.ss_wavenet_workspace_bytes(C_byref(pk["f0_pair"]["net"]), 2 * B, T)
        self.ws_f0 = torch.empty(self.ws_f0_bytes, device=dev, dtype=torch.uint8)
        self.coarse_mel = torch.empty(B, T, M, **f32)
        self.cond_mel = torch.empty(B, T, H, **f32)
        self.xm = torch.empty(B, T, M, **f32)
        nsplit = 2 if (model.n_streams >= 2 and B >= 2) else 1
        self.bounds = [B * i // nsplit for i in range(nsplit + 1)]
        self.ws_mel = []
        for i in range(nsplit):
            nb = self.bounds[i + 1] - self.bounds[i]
            wsb = lib.ss_wavenet_workspace_bytes(C_byref(pk["mel"]["net"]), nb, T)
            self.ws_mel.append((wsb, torch.empty(wsb, device=dev, dtype=torch.uint8)))
        self.ws_prodiff = None   # full-batch workspace of the ProDiff decoder when ws_mel is split (allocated on first use, plan-owned)
        self.g_f0 = None
        self.g_mel = None
        self.g_ddim = {}
        self.plms_hist = None
        self.uses = 0   # forwards that asked for this shape (auto mode captures on the second one)
        self.recount()

    def recount(self):
        """Bytes this plan keeps alive, each storage once (cond_a / cond_b / lens / f0[i] / uv[i] are views of the pair buffers)."""
        seen, total = set(), 0

        def add(t):
            nonlocal total
            if not torch.is_tensor(t):
                return
            st = t.untyped_storage()
            if st.data_ptr() not in seen:
                seen.add(st.data_ptr())
                total += st.nbytes()
        for v in vars(self).values():
            if torch.is_tensor(v):
                add(v)
            elif isinstance(v, (list, tuple)):
                for e in v:
                    if isinstance(e, (list, tuple)):
                        for ee in e:
                            add(ee)
                    else:
                        add(e)
        self.bytes = total


def _pad_frames(x, T, dim=-1):
    """Zero-pad axis `dim` of x to T frames (hipGraph bucket padding; padded frames are masked everywhere)."""
    n = x.shape[dim]
    if n == T:
        return x
    pad = [0, 0] * x.dim()
    pad[2 * (x.dim() - 1 - (dim % x.dim())) + 1] = T - n
    return torch.nn.functional.pad(x, pad)


def _capture(fn):
    """Warm up `fn` on a side stream, then capture it into a CUDAGraph (hipGraph)."""
    cur = torch.cuda.current_stream()
    s = torch.cuda.Stream()
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        fn()
    cur.wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


class StyleSingerHIP(torch.nn.Module):
    def __init__(self, dictionary=None, out_dims=None, hparams=None):
        super().__init__()
        hp = make_hparams(hparams)
        if dictionary is not None:
            hp["vocab_size"] = len(dictionary)
        self.hp = hp
        self.prodiff = hp.get("decoder", "diffsinger") == "prodiff"
        self.hidden_size = hp["hidden_size"]
        self.out_dims = out_dims or hp["audio_num_mel_bins"]
        self._names = []
        for name, shape in _spec.acoustic_spec(hp):
            self._names.append(name)
            self.register_buffer(self._mangle(name), torch.zeros(tuple(shape)), persistent=True)
        self._packed_version = -1
        self._weights_version = 0
        self._pk = None
        self._pos_table = None
        self.training = False
        import os
        self.n_streams = int(os.environ.get("SS_STREAMS", "1"))  # 2 = split the mel batch over two streams (slower at C2: half-size launches balance worse)
        # hipGraph capture of the diffusion loops: "auto"/"on" = capture per (B, T) on first use, "off" = eager launches
        self.use_graphs = os.environ.get("SS_GRAPHS", "auto")
        # Winograd form of the denoisers' 3-tap dilated convs (fewer matrix ops, fp32-rounding-equal results); SS_WINO=0: direct conv
        self.use_wino = os.environ.get("SS_WINO", "1") not in ("0", "off", "false")
        # Winograd output tile of the dilated conv: 4 = F(4,3) (6 products per 4 frames), 2 = F(2,3) (4 per 2); direct form = 6 per 2
        self.wino_m = int(os.environ.get("SS_WINO_M", "4"))
        assert self.wino_m in (2, 4), "SS_WINO_M must be 2 or 4"
        # per-layer output projection = residual half only; skip sum of all layers as one K = L*C GEMM per step
        self.defer_skip = os.environ.get("SS_DEFER_SKIP", "1") not in ("0", "off", "false")
        prec = os.environ.get("SS_PRECISION", hp.get("mfma_precision", "fp32"))
        if prec not in ("fp32", "bf16", "bf16x2", "fp16x2", "fp16q4", "fp16sd", "bf16x3"):
            raise ValueError(f"mfma_precision={prec!r}: expected fp32 | bf16 | bf16x2 | fp16x2 | fp16q4 | fp16sd | bf16x3")
        # "bf16x2" (BASELINE config 4 at fp32-grade parity): the bf16 mode's data path (hidden GEMMs on the bf16 matrix cores, operands bf16
        # in HBM) with every operand a (hi, mid) PAIR of bf16 terms and three products hi*hi + hi*mid + mid*hi per GEMM; the step-invariant
        # conditioner projection in exact fp32, skip_projection folded into the K = L*C skip GEMM as in fp32 mode. Measured on the reference's
        # 1000-step golden: plain bf16 operands 2.5e-3 mel L1 (bar 1e-4), this mode 4e-6 (oracle/bf16x3_numerics.py, tools/../study in DESIGN 3.1h).
        # "fp16x2": the same data path with FP16 terms and only the WEIGHTS split (hi, lo of w * 2^FP16_WSHIFT): two products a*hi + a*lo per
        # GEMM instead of three. Over 1000 steps the weight rounding is the coherent error, the activation rounding averages out and fp16's is
        # 8x smaller than bf16's: 1.9e-5 on the same golden (oracle/bf16x2_numerics.py; plain fp16 operands 1.9e-4, bf16 with these two
        # products 1.6e-4). The residual stream is a true fp16 pair (22 bits).
        # "fp16q4": fp16x2 with the second product of the mel gate and of the skip GEMM on the block-scaled fp4 matrix instruction where the launch
        # fills the chip (ss_gemm_bf16_gate128q / _tile256q; validated on hardware in round 5: 2.6e-5 vs the real reference at T = 5625 x 1000 steps);
        # oracle contract set_matmul_rounding("fp16q4")
        # "fp16sd" (round 6): fp16x2's data path with ONE fp16 weight term per element - half the matrix work and half the weight bytes - and the weight
        # rounding NOISE-SHAPED over the loop's network evaluations: evaluation j uses weight set j % N, the N sets being a first-order sigma-delta
        # sequence of fp16 roundings of the same weight (their sum is N w up to one rounding), so the rounding averages out over the steps instead
        # of adding up coherently: 2.2e-5 on the reference's 1000-step golden with N = 32 (plain one-product fp16: 1.94e-4; fp16x2: 1.9e-5;
        # oracle/dither_numerics.py, oracle contract set_matmul_rounding("fp16sd")). The f0 denoisers keep bf16x2 as in the other fp16 modes.
        self.sd = prec == "fp16sd"
        self.sd_sets = max(1, int(os.environ.get("SS_SD_SETS", hp.get("fp16sd_sets", 32)))) if self.sd else 0
        # the step-invariant conditioner addend of the fused layer launch as fp16 sigma-delta sets cycled over the evaluations (0 = the fp32 slab)
        self.sd_e_sets = min(64, max(0, int(os.environ.get("SS_SD_E_SETS", hp.get("fp16sd_e_sets", 8))))) if self.sd else 0
        self.q4 = prec == "fp16q4"
        self.f16 = prec in ("fp16x2", "fp16q4", "fp16sd")
        self.split = prec in ("bf16x2", "fp16x2", "fp16q4", "fp16sd")
        self.bf16 = prec in ("bf16", "bf16x2", "fp16x2", "fp16q4", "fp16sd")
        # opt-in "bf16x3": fp32 products of the F(4,3) gate from operands split into three bf16 terms on the bf16 matrix cores
        # (ss_wino43_gate16x; fp32-grade results, oracle/bf16x3_numerics.py); everything else as the fp32 mode
        self.x3 = prec == "bf16x3"
        # fold skip_projection / sqrt(L) into the skip-all weights (fp32 mode only: in bf16 mode the operand rounding of the
        # two separate GEMMs is part of the stated arithmetic)
        self.fold_skip = self.defer_skip and (not self.bf16 or self.split) and os.environ.get("SS_FOLD_SKIP", "1") not in ("0", "off", "false")
        # bf16 mode: hidden-layer weights AND activations as bf16 in HBM (ss_gemm_bf16; SS_BF16_HBM=0 keeps the round-1 form
        # that rounds fp32 operands inside the fp32 kernel's BF16 template mode); needs the deferred-skip layout
        self.bf16_hbm = self.bf16 and self.defer_skip and (self.split or os.environ.get("SS_BF16_HBM", "1") not in ("0", "off", "false"))
        if self.split and not (self.defer_skip and self.fold_skip):
            raise ValueError(f"mfma_precision={prec} needs the deferred, folded skip form (SS_DEFER_SKIP / SS_FOLD_SKIP left on)")
        if self.bf16:
            self.use_wino = False  # the transform would amplify the operand rounding; the matrix pipe is not the limit in bf16
        # diffusion plans (workspaces + captured hipGraphs) are keyed by (B, T bucket): frames are padded up to a multiple of
        # `t_bucket` (padding = mel2ph 0, masked everywhere, so the valid frames are unchanged) and the cache is an LRU bounded
        # by bytes; in "auto" mode a shape is captured on its SECOND use (the first use of a shape runs eagerly: capture costs
        # a warm-up pass + ~13k recorded launches, which a one-off shape never earns back).
        self.t_bucket = max(1, int(os.environ.get("SS_T_BUCKET", hp.get("t_bucket", 64))))
        self.plan_bytes = int(float(os.environ.get("SS_PLAN_GIB", "24")) * 2 ** 30)
        self._plans = collections.OrderedDict()
        self.n_captures = 0

    # ---- state_dict contract ------------------------------------------------------------------
    @staticmethod
    def _mangle(name):
        return "p__" + name.replace(".", "__")

    def state_dict(self, *args, **kwargs):
        return {n: getattr(self, self._mangle(n)) for n in self._names}

    def load_state_dict(self, state_dict, strict=True):
        missing = [n for n in self._names if n not in state_dict]
        unexpected = [k for k in state_dict if k not in set(self._names)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"StyleSingerHIP.load_state_dict: missing={missing[:5]} unexpected={unexpected[:5]}")
        with torch.no_grad():
            for n in self._names:
                if n in state_dict:
                    dst = getattr(self, self._mangle(n))
                    src = state_dict[n]
                    if tuple(src.shape) != tuple(dst.shape):
                        raise RuntimeError(f"size mismatch for {n}: {tuple(src.shape)} vs {tuple(dst.shape)}")
                    dst.copy_(src)
        self._weights_version += 1
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def p(self, name):
        return getattr(self, self._mangle(name))

    def train(self, mode=True):
        if mode:
            raise RuntimeError("StyleSingerHIP is an inference-only drop-in (training is out of scope, SURVEY.md §8)")
        return super().train(False)

    # ---- weight packing ---------------------------------------------------------------------
    # "fp16x2": the weights' power-of-two shift. |w| 2^8 < 65504 for |w| < 255; lo = RNE16(w 2^8 - hi) stays a NORMAL fp16 number for every
    # |w| >= 2^-10 and below that is exact to 2^-32 in absolute terms (fp16 subnormals are fixed point) - no reliance on how the matrix cores
    # treat subnormal inputs for any weight that matters. oracle/restatement.py uses the same constant.
    FP16_WSHIFT = 8

    def _split_w(self, w, f0=False):
        """packed fp32 weight -> split 16-bit pack of the precision mode (pairs interleaved by 32 along every row). The two f0 denoisers keep the
        three-product bf16 form in "fp16x2" mode: their outputs feed DISCRETE voicing decisions (one flipped in 11 250 at T = 5625 with two
        products, none with three) and their 200 steps are ~1 % of a C4 batch."""
        if not self.f16 or f0:
            return L.split_bf16(w)
        if float(w.abs().max()) * 2.0 ** self.FP16_WSHIFT >= 32768.0:
            raise ValueError("mfma_precision=fp16x2: a hidden-layer weight exceeds 128 in magnitude (fp16 range after the 2^8 shift)")
        return L.split_f16(w, scale=2.0 ** self.FP16_WSHIFT)

    def _sd_sets(self, w):
        """"fp16sd": packed fp32 weight [rows][K] -> fp16 [N][rows][2 K], the N noise-shaped one-term weight sets in the pair layout with ZERO lo terms
        (the two-product kernels then compute the one-product result exactly). Sigma-delta in the scaled domain: r_0 = 0, W_k = RNE16(w 2^s + r_k),
        r_(k+1) = r_k + (w 2^s - W_k): sum_k W_k = N w 2^s - r_N, |r_N| <= half an fp16 ulp."""
        if float(w.abs().max()) * 2.0 ** self.FP16_WSHIFT >= 32768.0:
            raise ValueError("mfma_precision=fp16sd: a hidden-layer weight exceeds 128 in magnitude (fp16 range after the 2^8 shift)")
        ws = w.float() * 2.0 ** self.FP16_WSHIFT
        r = torch.zeros_like(ws)
        sets = []
        for _ in range(self.sd_sets):
            wk = (ws + r).to(torch.float16).float()
            r = r + (ws - wk)
            sets.append(L.split_f16(wk, scale=1.0))    # hi = W_k exactly (it is an fp16 number), lo = 0
        return torch.stack(sets).contiguous()

    def _pack_conv(self, wname, bname=None, *, half=0, scale0=None, row_scale=1.0, bias2=None):
        w = self.p(wname)
        if w.dim() == 2:
            Cout, Cin, k = w.shape[0], w.shape[1], 1
        else:
            Cout, Cin, k = w.shape
        W = L.pack_conv_weight(w, scale0=scale0, interleave_half=half, row_scale=row_scale)
        bias = None
        if bname is not None:
            bias = L.pack_bias(self.p(bname), b2=bias2, interleave_half=half)
        return _Packed(W, bias, Cout, Cin, k, half)

    def _pack_wn_conv(self, prefix, *, half=0):
        v, g = self.p(prefix + ".weight_v"), self.p(prefix + ".weight_g")
        s0 = L.weight_norm_scale(v, g)
        Cout, Cin, k = v.shape
        W = L.pack_conv_weight(v, scale0=s0, interleave_half=half)
        bias = L.pack_bias(self.p(prefix + ".bias"), interleave_half=half)
        return _Packed(W, bias, Cout, Cin, k, half)

    def _wino_form(self, C, cycle):
        """Which Winograd form a denoiser's dilated convs take, decided ONCE at pack time from what the kernels accept: F(4,3)
        (ss_wino43_gate / ss_wino43_gate16) needs C % 32 == 0 and dilations 2^(l % cycle) <= 64; otherwise F(2,3)."""
        return 4 if (self.wino_m == 4 and C % 32 == 0 and (1 << (max(int(cycle), 1) - 1)) <= 64) else 2

    def _pack_wavenet_tensors(self, prefix, C, Lyr, steps, f0, cycle=4):
        """Packed device tensors of one denoiser (DiffNet / DDiffNet), keyed like the ss_wavenet fields."""
        dev = self.p(prefix + ".mlp.0.weight").device
        t = {}
        if f0:
            t["w_in"] = self.p(prefix + ".input_projection.weight").reshape(-1).contiguous()
            t["b_in"] = self.p(prefix + ".input_projection.bias").contiguous()
            t["uv_embed"] = self.p(prefix + ".uv_embed.weight").contiguous()
        else:
            pin = self._pack_conv(prefix + ".input_projection.weight", prefix + ".input_projection.bias")
            t["w_in"], t["b_in"] = pin.W, pin.bias
        # dstep[s][l][:] = diffusion_projection_l(mlp(SinusoidalPosEmb(s)))  (net.py:66,118-119) — weights-only table
        emb = _step_emb_table(steps, C).to(dev)
        m0 = self._pack_conv(prefix + ".mlp.0.weight", prefix + ".mlp.0.bias")
        m2 = self._pack_conv(prefix + ".mlp.2.weight", prefix + ".mlp.2.bias")
        h1 = torch.empty(steps, 4 * C, device=dev)
        h2 = torch.empty(steps, C, device=dev)
        L.conv_gemm(emb, m0.W, h1, B=1, T=steps, Cin=C, N=4 * C, Np=m0.Np, Kp=m0.Kp, bias=m0.bias, act=L.ACT_MISH, mask_rows=False)
        L.conv_gemm(h1, m2.W, h2, B=1, T=steps, Cin=4 * C, N=C, Np=m2.Np, Kp=m2.Kp, bias=m2.bias, mask_rows=False)
        dstep = torch.empty(steps, Lyr, C, device=dev)
        wc_rows, bc_rows = [], []
        for l in range(Lyr):
            p = f"{prefix}.residual_layers.{l}"
            dp = self._pack_conv(p + ".diffusion_projection.weight", p + ".diffusion_projection.bias")
            L.conv_gemm(h2, dp.W, dstep[:, l], B=1, T=steps, Cin=C, N=C, Np=dp.Np, Kp=dp.Kp, bias=dp.bias, ldc=Lyr * C,
                        mask_rows=False)
            dil = self._pack_conv(p + ".dilated_conv.weight", None, half=C)
            out = self._pack_conv(p + ".output_projection.weight", p + ".output_projection.bias")
            cnd = self._pack_conv(p + ".conditioner_projection.weight", p + ".conditioner_projection.bias", half=C,
                                  bias2=self.p(p + ".dilated_conv.bias"))
            t[f"w_dil.{l}"], t[f"w_out.{l}"], t[f"b_out.{l}"] = dil.W, out.W, out.bias
            if self.defer_skip and not self.bf16 and C % 64 == 0:   # residual half in the fetch order of ss_gemm16_res
                t[f"w_out16.{l}"] = L.pack_gemm16_weights(out.W[:C].contiguous(), out.Kp)
            if self.use_wino:
                wsrc = self.p(p + ".dilated_conv.weight").contiguous()
                wt = L.wino43_weight(wsrc) if self._wino_form(C, cycle) == 4 else L.wino_weight(wsrc)
                t[f"w_dil_wino.{l}"] = L.pack_conv_weight(wt, interleave_half=C)
                if self._wino_form(C, cycle) == 4 and t[f"w_dil_wino.{l}"].shape[0] % 64 == 0:   # the 16x16x4 kernel's fetch order
                    t[f"w_dil_wino16.{l}"] = L.pack_gate16_weights(t[f"w_dil_wino.{l}"], dil.Kp)
                if self.x3 and self._wino_form(C, cycle) == 4:
                    t[f"w_dil_x3.{l}"] = L.split3_weights(t[f"w_dil_wino.{l}"], dil.Kp)
            if self.bf16_hbm:  # bf16 weight copies (rounded once, RNE): the operands of ss_gemm_bf16
                to_h = (lambda w_: self._split_w(w_, f0)) if self.split else L.to_bf16   # split: pairs interleaved by 32 along every row
                if self.sd and not f0:   # N one-term weight sets per tensor ([N][rows][2 K]; set 0 first)
                    to_h = self._sd_sets
                t[f"w_dil_h.{l}"] = to_h(dil.W)
                t[f"w_out_h.{l}"] = to_h(out.W)
                if self.q4 and not f0 and C == 256:   # the fp4 lo plane in the lane order of ss_gemm_bf16_gate128q
                    t[f"w_dil_q.{l}"] = L.pack_gate_q4(dil.W, shift=self.FP16_WSHIFT)[0]
                if self.f16 and not f0 and C == 256 and tuple(t[f"w_dil_h.{l}"].shape[-2:]) == (512, 3 * 256 * 2):
                    # the same terms in the fragment order ss_layer512 streams (one launch per layer at many-round sizes); fp16sd: one term, N sets
                    if self.sd:
                        t[f"w_dil_f.{l}"] = torch.stack([L.layer512_pack_gate(w_, 1) for w_ in t[f"w_dil_h.{l}"]]).contiguous()
                        t[f"w_out_f.{l}"] = torch.stack([L.layer512_pack_res(w_, 1) for w_ in t[f"w_out_h.{l}"]]).contiguous()
                    else:
                        t[f"w_dil_f.{l}"] = L.layer512_pack_gate(t[f"w_dil_h.{l}"])
                        t[f"w_out_f.{l}"] = L.layer512_pack_res(t[f"w_out_h.{l}"])
            wc_rows.append(cnd.W)
            bc_rows.append(cnd.bias)
        if self.defer_skip:  # skip halves of all output projections side by side: [C][L*C], column l*C + ci
            wsk = torch.cat([self.p(f"{prefix}.residual_layers.{l}.output_projection.weight")[C:, :, 0] for l in range(Lyr)], dim=1)
            bsk = torch.stack([self.p(f"{prefix}.residual_layers.{l}.output_projection.bias")[C:] for l in range(Lyr)]).sum(0)
            if self.fold_skip:  # skip_projection(sum/sqrt(L)) is linear in the g_l: fold it into the weights (float64 product)
                ws_ = self.p(prefix + ".skip_projection.weight")[:, :, 0].double()
                bs_ = self.p(prefix + ".skip_projection.bias").double()
                r = 1.0 / math.sqrt(Lyr)
                bsk = (ws_ @ bsk.double() * r + bs_).float()
                wsk = (ws_ @ wsk.double() * r).float()
            t["w_skipall"] = L.pack_conv_weight(wsk[:, :, None].contiguous())
            if self.x3 and self.fold_skip:
                t["w_skipall_x3"] = L.split3_gemm16_weights(t["w_skipall"], t["w_skipall"].shape[1])
            t["b_skipall"] = L.pack_bias(bsk.contiguous())
        t["dstep"] = dstep
        if self.f16 and not f0 and float(dstep.abs().max()) >= 16384.0:
            # activation-range contract of the fp16 modes: the stream enters every layer as fp16(x + dstep_l); a step embedding this large
            # leaves no headroom below 65504 (the bf16 modes have the fp32 exponent range)
            raise ValueError("mfma_precision=fp16x2: a diffusion-step embedding exceeds 16384 in magnitude - fp16 activations would overflow; use bf16x2")
        t["w_cond"] = torch.cat(wc_rows, 0).contiguous()
        t["b_cond"] = torch.cat(bc_rows, 0).contiguous()
        if self.bf16_hbm:
            t["w_cond_h"] = L.to_bf16(t["w_cond"])     # (unused in split mode: the hoisted projection runs in fp32 there)
            t["w_skipall_h"] = (self._sd_sets(t["w_skipall"]) if (self.sd and not f0) else self._split_w(t["w_skipall"], f0)) if self.split else L.to_bf16(t["w_skipall"])
            if self.sd and not f0 and t["w_skipall"].shape[1] % 64 == 0:   # the same sets without the zero plane: [N][Np][L C] (ss_wavenet.w_skipall_c)
                t["w_skipall_c"] = L.split_planes(t["w_skipall_h"])[0].to(torch.float16).contiguous()
            if self.q4 and not f0 and t["w_skipall"].shape[1] % 64 == 0:   # the fp4 lo plane in the lane order of ss_gemm_bf16_tile256q
                t["w_skipall_q"] = L.pack_skip_q4(t["w_skipall"], shift=self.FP16_WSHIFT)[0]
        skip = self._pack_conv(prefix + ".skip_projection.weight", prefix + ".skip_projection.bias")
        fin = self._pack_conv(prefix + ".output_projection.weight", prefix + ".output_projection.bias")
        t["w_skip"], t["b_skip"], t["w_final"], t["b_final"] = skip.W, skip.bias, fin.W, fin.bias
        torch.cuda.synchronize()
        return t

    def _pack_wavenet(self, prefixes, gen, C, Lyr, cycle, steps, in_dim, out_dim, f0):
        """Build the ss_wavenet descriptor of one net, or of a PAIR of same-shaped nets (grouped launches: every
        weight tensor is stacked [2][...] so that net g sits gs_* floats after net 0)."""
        hp = self.hp
        packs = [self._pack_wavenet_tensors(pf, C, Lyr, steps, f0, cycle) for pf in prefixes]
        keep = []
        net = L.WaveNet()
        net.C, net.L, net.cond_dim, net.dil_cycle, net.in_dim, net.out_dim, net.steps = C, Lyr, hp["hidden_size"], cycle, in_dim, out_dim, steps
        net.n_groups = len(packs)

        def place(key):
            if len(packs) == 1:
                tt = packs[0][key].contiguous()
                gs = 0
            else:
                tt = torch.stack([pk_[key] for pk_ in packs]).contiguous()
                gs = packs[0][key].numel()
            keep.append(tt)
            return tt.data_ptr(), gs

        for key in ("w_in", "b_in", "dstep", "w_cond", "b_cond", "w_skip", "b_skip", "w_final", "b_final") + (("uv_embed",) if f0 else ()) \
                + (("w_skipall", "b_skipall") if self.defer_skip else ()):
            ptr_, gs = place(key)
            setattr(net, key, ptr_)
            setattr(net, "gs_" + key, gs)
        for l in range(Lyr):
            for key, arr in (("w_dil", net.w_dil), ("w_out", net.w_out), ("b_out", net.b_out)):
                ptr_, gs = place(f"{key}.{l}")
                arr[l] = ptr_
                setattr(net, "gs_" + key, gs)
            if self.use_wino:
                ptr_, gs = place(f"w_dil_wino.{l}")
                net.w_dil_wino[l] = ptr_
                net.gs_w_dil_wino = gs
                net.wino_m = self._wino_form(C, cycle)
            if f"w_out16.{l}" in packs[0]:
                net.w_out16[l], net.gs_w_out16 = place(f"w_out16.{l}")
            if self.use_wino:
                if f"w_dil_wino16.{l}" in packs[0]:
                    net.w_dil_wino16[l], _ = place(f"w_dil_wino16.{l}")
                if self.x3 and f"w_dil_x3.{l}" in packs[0]:
                    ptr_, gs = place(f"w_dil_x3.{l}")
                    net.w_dil_x3[l] = ptr_
                    net.gs_w_dil_x3 = gs
                    net.mfma_x3 = 1
            if self.bf16_hbm:
                for key, arr in (("w_dil_h", net.w_dil_h), ("w_out_h", net.w_out_h)):
                    ptr_, gs = place(f"{key}.{l}")
                    arr[l] = ptr_
                    setattr(net, "gs_" + key, gs)
        if "w_skipall_x3" in packs[0]:
            net.w_skipall_x3, net.gs_w_skipall_x3 = place("w_skipall_x3")
        if self.bf16_hbm:
            for key in ("w_cond_h", "w_skipall_h"):
                ptr_, gs = place(key)
                setattr(net, key, ptr_)
                setattr(net, "gs_" + key, gs)
        net.mfma_bf16 = 1 if self.bf16 else 0
        net.mfma_split = (2 if (self.f16 and not f0) else 1) if self.split else 0
        net.mfma_out_scale = 2.0 ** -self.FP16_WSHIFT if (self.f16 and not f0) else 1.0
        if self.sd and not f0:   # every w_*_h / w_*_f tensor is [N][...]: pointer = set 0, ws_* = elements between sets
            assert len(packs) == 1
            net.n_wsets, net.mfma_products = self.sd_sets, 1
            net.n_esets = self.sd_e_sets
            net.ws_w_dil_h, net.ws_w_out_h = packs[0]["w_dil_h.0"][0].numel(), packs[0]["w_out_h.0"][0].numel()
            net.ws_w_skipall_h = packs[0]["w_skipall_h"][0].numel()
            if "w_skipall_c" in packs[0]:
                net.w_skipall_c, _ = place("w_skipall_c")
                net.ws_w_skipall_c = packs[0]["w_skipall_c"][0].numel()
            if "w_dil_f.0" in packs[0]:
                net.ws_w_dil_f, net.ws_w_out_f = packs[0]["w_dil_f.0"][0].numel(), packs[0]["w_out_f.0"][0].numel()
        if self.q4 and not f0:
            for l in range(Lyr):
                if f"w_dil_q.{l}" in packs[0]:
                    net.w_dil_q[l], net.gs_w_dil_q = place(f"w_dil_q.{l}")
            net.q_scale_gate = 2.0   # the stream x + dstep on a fixed fp4 scale (oracle/second_product_numerics.py)
            if "w_skipall_q" in packs[0]:
                net.w_skipall_q, net.gs_w_skipall_q = place("w_skipall_q")
                net.q_scale_z = 0.25   # gate outputs in (-1, 1)
        if len(packs) == 1 and all(f"w_dil_f.{l}" in packs[0] for l in range(Lyr)):
            for l in range(Lyr):
                net.w_dil_f[l], _ = place(f"w_dil_f.{l}")
                net.w_out_f[l], _ = place(f"w_out_f.{l}")
        net.skipall_folded = 1 if self.fold_skip else 0
        # schedule tables live on the host (the loop driver passes per-step scalars by value)
        def host(name):
            arr = np.ascontiguousarray(self.p(f"{gen}.{name}").detach().cpu().numpy().astype(np.float32))
            keep.append(arr)
            return arr.ctypes.data
        net.sqrt_recip_ac, net.sqrt_recipm1_ac = host("sqrt_recip_alphas_cumprod"), host("sqrt_recipm1_alphas_cumprod")
        net.post_c1, net.post_c2 = host("posterior_mean_coef1"), host("posterior_mean_coef2")
        net.post_logvar = host("posterior_log_variance_clipped")
        if f0:
            net.log_alpha, net.log_1m_alpha = host("log_alpha"), host("log_1_min_alpha")
            net.log_cumprod_alpha, net.log_1m_cumprod_alpha = host("log_cumprod_alpha"), host("log_1_min_cumprod_alpha")
        sched = {k: self.p(f"{gen}.{k}").detach().cpu() for k in ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod")}
        if not f0:
            sched["alphas_cumprod_np"] = np.ascontiguousarray(self.p(f"{gen}.alphas_cumprod").detach().cpu().numpy().astype(np.float32))
            # DDIM coefficients need 1 - alphas_cumprod at small t: rebuilt in float64 from the betas buffer as the reference builds its
            # own tables (shallow_diffusion_tts.py:77-80: np.cumprod(1 - betas)); the float32 buffer has only ~3 digits of 1 - ac_0
            betas64 = self.p(f"{gen}.betas").detach().cpu().numpy().astype(np.float64)
            sched["alphas_cumprod_f64"] = np.ascontiguousarray(np.cumprod(1.0 - betas64))
        return dict(net=net, keep=keep, sched=sched, packs=packs)

    def _pack_fft(self, prefix, n_layers):
        layers = []
        for i in range(n_layers):
            p = f"{prefix}.layers.{i}.op"
            layers.append(dict(
                ln1=(self.p(p + ".layer_norm1.weight"), self.p(p + ".layer_norm1.bias")),
                qkv=self._pack_conv(p + ".self_attn.in_proj_weight"),
                out=self._pack_conv(p + ".self_attn.out_proj.weight"),
                ln2=(self.p(p + ".layer_norm2.weight"), self.p(p + ".layer_norm2.bias")),
                ffn1=self._pack_conv(p + ".ffn.ffn_1.weight", p + ".ffn.ffn_1.bias"),
                ffn2=self._pack_conv(p + ".ffn.ffn_2.weight", p + ".ffn.ffn_2.bias")))
        return dict(layers=layers, ln=(self.p(prefix + ".layer_norm.weight"), self.p(prefix + ".layer_norm.bias")))

    def pack(self):
        """(Re)build every packed weight on the current device; called lazily by forward."""
        hp = self.hp
        dev = self.p("mel_out.weight").device
        if not dev.type == "cuda":
            raise L.StyleSingerHipError("StyleSingerHIP needs its weights on a GPU (model.to('cuda')): there is no CPU path")
        pk = {}
        pk["enc"] = self._pack_fft("encoder", hp["enc_layers"])
        pk["dec"] = self._pack_fft("decoder", hp["dec_layers"])
        pk["mel_out"] = self._pack_conv("mel_out.weight", "mel_out.bias")
        pk["spk"] = self._pack_conv("spk_embed_proj.weight", "spk_embed_proj.bias")
        pk["emo"] = self._pack_conv("emo_embed_proj.weight", "emo_embed_proj.bias")
        pk["dur"] = [dict(conv=self._pack_conv(f"dur_predictor.conv.{i}.1.weight", f"dur_predictor.conv.{i}.1.bias"),
                          ln=(self.p(f"dur_predictor.conv.{i}.3.weight"), self.p(f"dur_predictor.conv.{i}.3.bias")))
                     for i in range(hp["dur_predictor_layers"])]
        pk["dur_lin"] = self._pack_conv("dur_predictor.linear.weight", "dur_predictor.linear.bias")
        # RSA
        wn = []
        for i in range(4):
            inl = self._pack_wn_conv(f"style_extractor.wavenet.in_layers.{i}", half=80)
            v, g = self.p(f"style_extractor.wavenet.res_skip_layers.{i}.weight_v"), self.p(f"style_extractor.wavenet.res_skip_layers.{i}.weight_g")
            s0 = L.weight_norm_scale(v, g)
            b = self.p(f"style_extractor.wavenet.res_skip_layers.{i}.bias")
            if i < 3:
                res = _Packed(L.pack_conv_weight(v[:80], scale0=s0[:80].contiguous()), L.pack_bias(b[:80]), 80, 80, 1)
                skp = _Packed(L.pack_conv_weight(v[80:], scale0=s0[80:].contiguous()), L.pack_bias(b[80:]), 80, 80, 1)
            else:
                res = None
                skp = _Packed(L.pack_conv_weight(v, scale0=s0), L.pack_bias(b), 80, 80, 1)
            wn.append(dict(inl=inl, res=res, skip=skp))
        pk["wn"] = wn
        cb = []
        for rb in range(5):
            for blk in range(2):
                p = f"style_extractor.encoder.res_blocks.{rb}.blocks.{blk}"
                cb.append(dict(ln=(self.p(p + ".0.weight"), self.p(p + ".0.bias")), c1=self._pack_conv(p + ".1.weight", p + ".1.bias"),
                               c2=self._pack_conv(p + ".4.weight", p + ".4.bias")))
        pk["cb"] = cb
        pk["cb_ln"] = (self.p("style_extractor.encoder.last_norm.weight"), self.p("style_extractor.encoder.last_norm.bias"))
        pk["cb_post"] = self._pack_conv("style_extractor.encoder.post_net1.weight", "style_extractor.encoder.post_net1.bias")
        pk["codebooks"] = torch.stack([self.p(f"style_extractor.rqvae.codebooks.{d}.weight") for d in range(hp["rq_depth"])]).contiguous()
        pk["l1"] = self._pack_conv("l1.weight", "l1.bias")
        al = []
        H = hp["hidden_size"]
        for i in range(2):
            p = f"align.layers.{i}"
            w, b = self.p(p + ".multihead_attn.in_proj_weight"), self.p(p + ".multihead_attn.in_proj_bias")
            al.append(dict(
                q=_Packed(L.pack_conv_weight(w[:H]), L.pack_bias(b[:H]), H, H, 1),
                kv=_Packed(L.pack_conv_weight(w[H:]), L.pack_bias(b[H:]), 2 * H, H, 1),
                out=self._pack_conv(p + ".multihead_attn.out_proj.weight", p + ".multihead_attn.out_proj.bias"),
                n1=(self.p(p + ".norm1.weight"), self.p(p + ".norm1.bias")), n2=(self.p(p + ".norm2.weight"), self.p(p + ".norm2.bias")),
                l1=self._pack_conv(p + ".linear1.weight", p + ".linear1.bias"), l2=self._pack_conv(p + ".linear2.weight", p + ".linear2.bias")))
        pk["align"] = al
        f0_args = (hp["f0_residual_channels"], hp["f0_residual_layers"], hp["f0_dilation_cycle_length"], hp["f0_timesteps"], 1, 3, True)
        # the two f0 denoisers have identical shapes and schedules -> one grouped descriptor (items [0,B): agnostic
        # net, [B,2B): specific net): every launch of the f0 loops carries 2x the blocks.
        pk["f0_pair"] = self._pack_wavenet(["gm_diffnet", "gm_diffnet_inpainte"], "f0_gen", *f0_args)
        mel_args = (hp["residual_channels"], hp["residual_layers"], hp["dilation_cycle_length"], hp["timesteps"],
                    hp["audio_num_mel_bins"], hp["audio_num_mel_bins"], False)
        if self.prodiff:  # hparams['decoder'] == 'prodiff' (stylesinger.py:111-117): the DiffNet conditioned on decoder_inp itself
            pk["mel"] = self._pack_wavenet(["diff_decoder.denoise_fn"], "diff_decoder", *mel_args)
            g = lambda k: np.ascontiguousarray(self.p("diff_decoder." + k).detach().cpu().numpy().astype(np.float32))
            pk["prodiff_sched"] = dict(c1=g("posterior_mean_coef1"), c2=g("posterior_mean_coef2"),
                                       sigma=np.ascontiguousarray(np.exp(0.5 * g("posterior_log_variance_clipped")).astype(np.float32)))
        else:
            pk["mel"] = self._pack_wavenet(["postdiff.denoise_fn"], "postdiff", *mel_args)
            pk["ln_proj"] = self._pack_conv("ln_proj.weight", "ln_proj.bias")
            pk["spec_min"] = self.p("postdiff.spec_min").reshape(-1).contiguous()
            pk["spec_max"] = self.p("postdiff.spec_max").reshape(-1).contiguous()
        self._pk = pk
        self._packed_version = self._weights_version
        self._pack_device = dev
        self._pos_table = None
        # captured hipGraphs carry the OLD packed-weight pointers in their kernel arguments: drop every plan with them
        self._plans.clear()
        torch.cuda.synchronize()

    def _ensure_packed(self):
        dev = self.p("mel_out.weight").device
        if self._pk is None or self._packed_version != self._weights_version or self._pack_device != dev:
            self.pack()

    def _streams(self, n):
        """Side HIP streams: independent launch sequences (the two f0 samplers, batch halves of the mel sampler) run
        concurrently so that one sequence's kernel tails/launch gaps are filled by the other's blocks."""
        if not hasattr(self, "_side_streams") or len(self._side_streams) < n:
            self._side_streams = [torch.cuda.Stream() for _ in range(n)]
        return self._side_streams[:n]

    def _pos(self, n, dev):
        if self._pos_table is None or self._pos_table.shape[0] < n:
            # a blocking host->device copy (pageable source): complete on return, whichever stream built it, so forwards running
            # on other streams may read it without an event. A table that is being REPLACED may still be read by a forward in flight
            # on another stream: keep the old one alive until the device is idle.
            old = self._pos_table
            self._pos_table = _sin_table(max(n, 2048), self.hp["hidden_size"]).to(dev)
            if old is not None and old.is_cuda:
                torch.cuda.synchronize(dev)
        return self._pos_table

    def warm_caches(self, max_frames, device):
        """Build the lazily created shared state (packed weights, sinusoidal table) BEFORE forwards fork onto side streams."""
        self._ensure_packed()
        self._pos(int(max_frames) + 2, torch.device(device))

    # ---- op helpers -----------------------------------------------------------------------------
    def _gemm(self, x, pk, out, B, T, *, taps=None, lens=None, act=L.ACT_NONE, pre_scale=1.0, R=None, mask_rows=True,
              N=None, **kw):
        if taps is None:
            taps = [(j - (pk.k - 1) // 2) for j in range(pk.k)]
        N = pk.Cout if N is None else N
        L.conv_gemm(x, pk.W, out, B=B, T=T, Cin=pk.Cin, N=N, Np=pk.Np, Kp=pk.Kp, taps=taps, lens=lens, bias=pk.bias,
                    act=act, pre_scale=pre_scale, R=R, ldr=(N if R is not None else 0), mask_rows=mask_rows, **kw)
        return out

    def _fft_blocks(self, pkb, x, B, T, lens):
        """4 x EncSALayer + final LN (tts_modules.py:281-306, common_layers.py:649-673), in place on x [B,T,H]."""
        H = self.hp["hidden_size"]
        nh = self.hp["num_heads"]
        D = H // nh
        dev = x.device
        h = torch.empty(B, T, H, device=dev)
        qkv = torch.empty(B, T, 3 * H, device=dev)
        att = torch.zeros(B, T, H, device=dev)
        ff = torch.empty(B, T, 4 * H, device=dev)
        for ly in pkb["layers"]:
            L.layernorm(x, *ly["ln1"], B=B, T=T, C_=H, out=h)
            self._gemm(h, ly["qkv"], qkv, B, T, lens=lens, mask_rows=False)
            L.attention(qkv, qkv[:, :, H:], qkv[:, :, 2 * H:], att, B=B, H=nh, D=D, Tq=T, Tk=T, ldq=3 * H, ldk=3 * H, ldv=3 * H,
                        ldo=H, q_bs=T * 3 * H, k_bs=T * 3 * H, v_bs=T * 3 * H, o_bs=T * H, qlens=lens, klens=lens, scale=D ** -0.5)
            self._gemm(att, ly["out"], x, B, T, lens=lens, R=x)
            L.layernorm(x, *ly["ln2"], B=B, T=T, C_=H, out=h)
            k = ly["ffn1"].k
            self._gemm(h, ly["ffn1"], ff, B, T, lens=lens, act=L.ACT_GELU, pre_scale=k ** -0.5, mask_rows=False)
            self._gemm(ff, ly["ffn2"], x, B, T, lens=lens, R=x)
        L.layernorm(x, *pkb["ln"], B=B, T=T, C_=H, out=x, lens=lens, mask_rows=True)
        return x

    # ---- diffusion loops on a plan -------------------------------------------------------------------
    def bucket_frames(self, T):
        b = self.t_bucket
        return T if b <= 1 else (T + b - 1) // b * b

    def _plan(self, B, T, dev, slot=0):
        """LRU cache of diffusion plans keyed by (B, T, device, slot), bounded by `plan_bytes` of workspace. `slot` separates
        the workspaces of forwards that run CONCURRENTLY on different HIP streams (forward(plan_slot=...))."""
        key = (B, T, dev.index) if slot == 0 else (B, T, dev.index, slot)
        pl = self._plans.get(key)
        self.plan_lookups = getattr(self, "plan_lookups", 0) + 1
        if pl is None:
            self.plan_misses = getattr(self, "plan_misses", 0) + 1
            pl = _DiffPlan(self, B, T, dev)
            self._plans[key] = pl
            total = sum(p.bytes for p in self._plans.values())
            synced = False
            while total > self.plan_bytes and len(self._plans) > 1:
                if not synced:   # another slot's stream may still be replaying the victim's graph into its workspace
                    if torch.cuda.is_available():
                        torch.cuda.synchronize(dev)
                    synced = True
                _, old = self._plans.popitem(last=False)   # least recently used
                total -= old.bytes
                self.plan_evictions = getattr(self, "plan_evictions", 0) + 1
        else:
            self._plans.move_to_end(key)
        return pl

    def _want_graphs(self, pl):
        if self.use_graphs in ("1", "on", "true", True):
            return True
        if self.use_graphs in ("0", "off", "false", False):
            return False
        # auto (north_star: "the diffusion inner loop captured as a hipGraph"): capture once a shape comes back
        return pl.uses >= 2

    def _capture(self, fn):
        self.n_captures += 1
        return _capture(fn)

    # Philox keys: the host part of every key is a CONSTANT per call site and all per-call variation comes from the device
    # word pl.seed, so that a captured graph (host arguments frozen at capture) and the eager launches draw the same noise
    # for the same `seed`, whatever was run before.
    def _run_f0_pair(self, pl, tape=None):
        """Both joint f0/uv samplers in ONE grouped loop (they are independent given their conditions)."""
        lib, pk = _lib(), self._pk
        B, T = pl.B, pl.T
        sdp = L.ptr(pl.seed)
        net = pk["f0_pair"]["net"]
        zs = us = None
        if tape is not None:
            zs, us = tape  # [S][2B][T], [S][2B][2][T]
        else:
            L.check(lib.ss_fill_normal_rows(L.ptr(pl.f02), 2 * B, T, T, 11, sdp, L.stream_ptr()), "z0")
        L.check(lib.ss_f0diff_sample(C_byref(net), L.ptr(pl.f02), L.ptr(pl.uv2), L.ptr(pl.cond2), L.ptr(pl.lo2), L.ptr(pl.hi2),
                                     L.ptr(pl.lens2), 2 * B, T, L.ptr(zs), L.ptr(us), 17, sdp, 0, net.steps, 1,
                                     L.ptr(pl.ws_f0), pl.ws_f0_bytes, L.stream_ptr()), "f0 pair")

    def ddim_timesteps(self, n):
        """n network times, strictly decreasing from K-1 to 0 (uniform stride)."""
        K = self.hp["K_step"]
        return sorted({int(round(v)) for v in np.linspace(0, K - 1, max(1, min(n, K)))}, reverse=True)

    def _run_mel(self, pl, tape=None, ddim_ts=None, plms_interval=None, eta=0.0):
        """q_sample + the shallow reverse loop (batch halves on two streams); `ddim_ts` switches to the strided
        DDIM sampler (BASELINE config 5; `eta` = 0 deterministic ... 1 ancestral), `plms_interval` to the reference's PLMS sampler (pndm_speedup)."""
        lib, pk, hp = _lib(), self._pk, self.hp
        B, T, M = pl.B, pl.T, hp["audio_num_mel_bins"]
        net = pk["mel"]["net"]
        K = hp["K_step"]
        sa = float(pk["mel"]["sched"]["sqrt_alphas_cumprod"][K - 1])
        s1 = float(pk["mel"]["sched"]["sqrt_one_minus_alphas_cumprod"][K - 1])
        sdp = L.ptr(pl.seed)
        zq_n, zs_n = tape if tape is not None else (None, None)
        L.check(lib.ss_mel_qsample(L.ptr(pl.coarse_mel), L.ptr(pk["spec_min"]), L.ptr(pk["spec_max"]), sa, s1, L.ptr(zq_n), 23, sdp,
                                   L.ptr(pl.xm), B, T, M, L.stream_ptr()), "qsample")
        if ddim_ts is not None or plms_interval is not None:
            ac = pk["mel"]["sched"]["alphas_cumprod_np"]  # host table, read by the loop driver at launch time
            if len(pl.ws_mel) != 1:
                wsb = lib.ss_wavenet_workspace_bytes(C_byref(net), B, T)
                wsp = torch.empty(wsb, device=pl.xm.device, dtype=torch.uint8)
            else:
                wsb, wsp = pl.ws_mel[0]
        if plms_interval is not None:
            if pl.plms_hist is None:
                pl.plms_hist = torch.empty(6 * B * T * M, device=pl.xm.device, dtype=torch.float32)
                pl.recount()
            L.check(lib.ss_meldiff_sample_plms(C_byref(net), L.ptr(pl.xm), L.ptr(pl.cond_mel), L.ptr(pl.lens), B, T, K, int(plms_interval),
                                               L.hptr(ac), 1, L.ptr(pl.plms_hist), L.ptr(wsp), wsb, L.stream_ptr()), "meldiff plms")
            return
        if ddim_ts is not None:
            ts = np.ascontiguousarray(np.asarray(ddim_ts, dtype=np.int32))
            ac64 = pk["mel"]["sched"]["alphas_cumprod_f64"]
            L.check(lib.ss_meldiff_sample_ddim(C_byref(net), L.ptr(pl.xm), L.ptr(pl.cond_mel), L.ptr(pl.lens), B, T, L.hptr(ts), len(ts),
                                               L.hptr(ac64), float(eta), L.ptr(zs_n), 31, sdp, 1, L.ptr(wsp), wsb, L.stream_ptr()), "meldiff ddim")
            return
        nsplit = len(pl.ws_mel)
        main = torch.cuda.current_stream()
        side = self._streams(nsplit) if nsplit > 1 else [main]
        zparts = [zs_n[:, pl.bounds[i]:pl.bounds[i + 1]].contiguous() if (zs_n is not None and nsplit > 1) else zs_n for i in range(nsplit)]
        for sd_ in set(side) - {main}:
            sd_.wait_stream(main)
        for i, strm in enumerate(side):
            b0, nb = pl.bounds[i], pl.bounds[i + 1] - pl.bounds[i]
            wsb, wsp = pl.ws_mel[i]
            with torch.cuda.stream(strm):
                L.check(lib.ss_meldiff_sample(C_byref(net), L.ptr(pl.xm[b0:]), L.ptr(pl.cond_mel[b0:]), L.ptr(pl.lens[b0:]), nb, T,
                                              L.ptr(zparts[i]), 29 + 7919 * b0, sdp, 0, K, 1, L.ptr(wsp), wsb, L.stream_ptr()), "meldiff")
        for sd_ in set(side) - {main}:
            main.wait_stream(sd_)

    @torch.no_grad()
    def mel_stage(self, coarse_mel, cond, lens=None, z_q=None, z_steps=None, sampler="ddpm", ddim_steps=None, plms_interval=None, seed=1234,
                  eta=0.0):
        """The shallow mel diffusion alone (a11 output -> a12): coarse mel [B,T,80] + condition [B,T,256] -> mel [B,T,80].
        z_q [B,1,80,T] / z_steps [K,B,1,80,T]: optional recorded noise (reference layout); default device Philox."""
        self._ensure_packed()
        lib, pk, hp = _lib(), self._pk, self.hp
        dev = coarse_mel.device
        B, T, M = coarse_mel.shape
        K = hp["K_step"]
        pl = self._plan(B, T, dev)
        pl.lens.copy_(lens if lens is not None else torch.full((B,), T, device=dev, dtype=torch.int32))
        pl.seed.fill_(seed)
        pl.cond_mel.copy_(cond)
        pl.coarse_mel.copy_(coarse_mel)
        zq_n = None if z_q is None else z_q.to(dev).reshape(B, M, T).transpose(1, 2).contiguous().float()
        zs_n = None if z_steps is None else z_steps.to(dev).reshape(K, B, M, T).transpose(2, 3).contiguous().float()
        self._run_mel(pl, (zq_n, zs_n), ddim_ts=self.ddim_timesteps(ddim_steps) if sampler == "ddim" else None,
                      plms_interval=plms_interval if sampler == "plms" else None, eta=eta)
        mel_out = torch.empty(B, T, M, device=dev, dtype=torch.float32)
        L.check(lib.ss_mel_denorm(L.ptr(pl.xm), L.ptr(pk["spec_min"]), L.ptr(pk["spec_max"]), L.ptr(mel_out), B, T, M, L.ptr(pl.lens), None,
                                  L.stream_ptr()), "denorm")
        return mel_out

    # ---- forward ----------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_style(self, ref_mels, ref_f0):
        """Residual Style Adaptor + RQ lookup + `l1` (a5, a6 and the first line of a7; lse.py:103-129, RQ.py:226-270,
        stylesinger.py:189-196) for a batch of references. The result depends on the reference only, so a style-transfer
        sweep (BASELINE config 5) computes it once per reference and passes it to forward(style_cache=...)."""
        self._ensure_packed()
        lib, hp, pk = _lib(), self.hp, self._pk
        st = L.stream_ptr
        H = hp["hidden_size"]
        dev = ref_mels.device
        B = ref_mels.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        ref_mels = ref_mels.contiguous().float()
        Tr = ref_mels.shape[1]
        if ref_f0.dim() == 1:
            ref_f0 = ref_f0[None]
        ref_f0 = ref_f0.contiguous().float()
        lens_r = torch.empty(B, device=dev, dtype=torch.int32)
        L.check(lib.ss_ref_lens(L.ptr(ref_mels), B, Tr, 80, L.ptr(lens_r), st()), "ref_lens")
        xr = ref_mels.clone()
        acts = torch.empty(B, Tr, 80, **f32)
        wn_out = torch.zeros(B, Tr, 80, **f32)
        for i, w in enumerate(pk["wn"]):
            inl = w["inl"]
            L.conv_gemm(xr, inl.W, acts, B=B, T=Tr, Cin=80, N=80, Np=inl.Np, Kp=inl.Kp, taps=(-1, 0, 1), lens=lens_r,
                        epi=L.EPI_GATE, gate_mode=1, bias=inl.bias, ldc=80, mask_rows=False)
            if w["res"] is not None:
                self._gemm(acts, w["res"], xr, B, Tr, lens=lens_r, R=xr)
            self._gemm(acts, w["skip"], wn_out, B, Tr, lens=lens_r, accumulate=True)
        L.check(lib.ss_add_rowscalar(L.ptr(wn_out), L.ptr(ref_f0), B, Tr, 80, L.ptr(lens_r), st()), "add f0")
        h80 = torch.empty(B, Tr, 80, **f32)
        h160 = torch.empty(B, Tr, 160, **f32)
        for blk in pk["cb"]:
            L.layernorm(wn_out, *blk["ln"], B=B, T=Tr, C_=80, out=h80)
            self._gemm(h80, blk["c1"], h160, B, Tr, lens=lens_r, act=L.ACT_GELU, pre_scale=blk["c1"].k ** -0.5, mask_rows=False)
            self._gemm(h160, blk["c2"], wn_out, B, Tr, lens=lens_r, R=wn_out)
        L.layernorm(wn_out, *pk["cb_ln"], B=B, T=Tr, C_=80, out=h80, lens=lens_r, mask_rows=True)
        pre_rq = torch.empty(B, Tr, H, **f32)
        self._gemm(h80, pk["cb_post"], pre_rq, B, Tr, lens=lens_r)
        zq = torch.empty(B, Tr, H, **f32)
        codes = torch.empty(B, Tr, hp["rq_depth"], device=dev, dtype=torch.int64)
        L.check(lib.ss_rq_lookup(L.ptr(pre_rq), L.ptr(pk["codebooks"]), L.ptr(zq), L.ptr(codes), B * Tr, H, hp["nRQ"], hp["rq_depth"], st()), "rq")
        cat = torch.empty(B, Tr, 2 * H, **f32)
        cat[:, :, :H].copy_(zq)
        pos_r = torch.empty(B, Tr, device=dev, dtype=torch.int32)
        tabr = self._pos(Tr + 2, dev)
        L.check(lib.ss_make_positions(None, L.ptr(zq), H, Tr * H, L.ptr(pos_r), B, Tr, st()), "pos style")
        L.check(lib.ss_table_add(L.ptr(pos_r), L.ptr(tabr), tabr.shape[0], L.ptr(cat) + 4 * H, 2 * H, Tr * 2 * H, B, Tr, H, None, 1.0, 0, st()), "pos add")
        sty = torch.empty(B, Tr, H, **f32)
        self._gemm(cat, pk["l1"], sty, B, Tr, mask_rows=False)
        return dict(sty=sty, lens_r=lens_r, ref_f0=ref_f0, style_pre_rq=pre_rq, rq_codes=codes, style_rq=zq)

    @torch.no_grad()
    def forward(self, txt_tokens, mel2ph=None, spk_embed=None, emo_embed=None, ref_mels=None, ref_f0=None, f0=None, uv=None,
                skip_decoder=False, global_steps=0, infer=False, note=None, note_dur=None, note_type=None, **kwargs):
        """Mirror of StyleSinger.forward (modules/StyleSinger/stylesinger.py:119-187), inference branch only.

        Extra keyword arguments: `noise` (dict from synth.draw_acoustic_noise: a recorded noise tape for
        parity tests; default = on-device Philox), `seed` (Philox seed), `sampler="ddim", ddim_steps=n, eta=0.0` (strided
        DDIM mel sampler, BASELINE config 5; eta = 0 deterministic, eta = 1 with ddim_steps = K_step IS the reference's ancestral
        sampler; default = the reference's 100-step ancestral sampler),
        `sampler="plms", plms_interval=n` (the reference's PLMS sampler, hparams['pndm_speedup'],
        shallow_diffusion_tts.py:165-197), `plan_slot` (workspace/graph set to use: give concurrent forwards on different streams
        different slots), `style_cache` (the dict encode_style() returned for these references: skips
        the style encoder)."""
        if not infer or f0 is not None or uv is not None:
            raise NotImplementedError("StyleSingerHIP implements the inference path only (infer=True, f0/uv predicted)")
        self._ensure_packed()
        lib, hp, pk = _lib(), self.hp, self._pk
        st = L.stream_ptr
        dev = txt_tokens.device
        H = hp["hidden_size"]
        noise = kwargs.get("noise")
        seed = int(kwargs.get("seed", hp["seed"]))
        ret = {}
        B, Tp = txt_tokens.shape
        txt_tokens = txt_tokens.contiguous()
        f32 = dict(device=dev, dtype=torch.float32)
        lens_p = torch.empty(B, device=dev, dtype=torch.int32)
        L.check(lib.ss_count_nonzero_i64(L.ptr(txt_tokens), L.ptr(lens_p), B, Tp, st()), "count_nonzero")

        # ---- phoneme encoder (a1) + note encoder (a2) ----
        x = torch.empty(B, Tp, H, **f32)
        pos_p = torch.empty(B, Tp, device=dev, dtype=torch.int32)
        tab = self._pos(max(Tp, 8) + 2, dev)
        L.check(lib.ss_embedding(L.ptr(txt_tokens), L.ptr(self.p("encoder.embed_tokens.weight")), L.ptr(x), B * Tp, H,
                                 hp["vocab_size"], math.sqrt(H), 0, st()), "embedding")
        L.check(lib.ss_make_positions(L.ptr(txt_tokens), None, 0, 0, L.ptr(pos_p), B, Tp, st()), "make_positions")
        L.check(lib.ss_table_add(L.ptr(pos_p), L.ptr(tab), tab.shape[0], L.ptr(x), H, Tp * H, B, Tp, H, None, 1.0, 1, st()), "table_add")
        L.check(lib.ss_add_bcast_mask(L.ptr(x), None, None, None, None, L.ptr(x), B, Tp, H, L.ptr(lens_p), st()), "mask")
        enc = self._fft_blocks(pk["enc"], x, B, Tp, lens_p)
        ret["encoder_out_text"] = enc.clone()
        note_out = torch.empty(B, Tp, H, **f32)
        note, note_type, note_dur = note.contiguous(), note_type.contiguous(), note_dur.contiguous().float()
        L.check(lib.ss_embedding(L.ptr(note), L.ptr(self.p("note_encoder.emb.weight")), L.ptr(note_out), B * Tp, H, 100, math.sqrt(H), 0, st()), "note emb")
        L.check(lib.ss_note_dur_add(L.ptr(note_dur), L.ptr(self.p("note_encoder.dur_ln.weight").reshape(-1).contiguous()),
                                    L.ptr(self.p("note_encoder.dur_ln.bias")), L.ptr(note_out), B * Tp, H, st()), "note dur")
        L.check(lib.ss_embedding(L.ptr(note_type), L.ptr(self.p("note_encoder.type_emb.weight")), L.ptr(note_out), B * Tp, H, 5, math.sqrt(H), 1, st()), "type emb")
        L.check(lib.ss_add_bcast_mask(L.ptr(enc), None, L.ptr(note_out), None, None, L.ptr(enc), B, Tp, H, None, st()), "enc+note")

        # ---- speaker / emotion projections ----
        spk = torch.empty(B, H, **f32)
        emo = torch.empty(B, H, **f32)
        self._gemm(spk_embed.contiguous().float(), pk["spk"], spk, 1, B, mask_rows=False)
        self._gemm(emo_embed.contiguous().float(), pk["emo"], emo, 1, B, mask_rows=False)
        ret["spk_embed"], ret["emo_embed"] = spk[:, None, :], emo[:, None, :]

        # ---- duration predictor + length regulator (a3) ----
        dur_inp = torch.empty(B, Tp, H, **f32)
        L.check(lib.ss_add_bcast_mask(L.ptr(enc), L.ptr(spk), None, L.ptr(emo), None, L.ptr(dur_inp), B, Tp, H, L.ptr(lens_p), st()), "dur_inp")
        hbuf = torch.empty(B, Tp, H, **f32)
        cur = dur_inp
        for lyr in pk["dur"]:
            self._gemm(cur, lyr["conv"], hbuf, B, Tp, lens=lens_p, act=L.ACT_RELU, mask_rows=False)
            cur = L.layernorm(hbuf, *lyr["ln"], B=B, T=Tp, C_=H, out=torch.empty_like(hbuf), lens=lens_p, mask_rows=True)
        logdur = torch.empty(B, Tp, 4, **f32)
        L.conv_gemm(cur, pk["dur_lin"].W, logdur, B=B, T=Tp, Cin=H, N=1, Np=pk["dur_lin"].Np, Kp=pk["dur_lin"].Kp, lens=lens_p,
                    bias=pk["dur_lin"].bias, ldc=4, mask_rows=True)
        logdur = logdur[:, :, 0].contiguous()
        lens_t = torch.empty(B, device=dev, dtype=torch.int32)
        if mel2ph is None:
            dur = torch.empty(B, Tp, device=dev, dtype=torch.int64)
            L.check(lib.ss_length_regulate(L.ptr(logdur), L.ptr(txt_tokens), L.ptr(dur), None, L.ptr(lens_t), B, Tp, 0, st()), "dur")
            T = int(lens_t.max().item())  # host sync: the frame count is data dependent
            if T <= 0:
                raise L.StyleSingerHipError("predicted durations are all zero")
            mel2ph = torch.empty(B, T, device=dev, dtype=torch.int64)
            L.check(lib.ss_length_regulate(L.ptr(logdur), L.ptr(txt_tokens), L.ptr(dur), L.ptr(mel2ph), L.ptr(lens_t), B, Tp, T, st()), "lr")
            ret["dur"], ret["dur_choice"] = logdur[:, :, None], dur
        else:
            mel2ph = mel2ph.contiguous()
            T = mel2ph.shape[1]
            ret["dur"] = logdur
        # hipGraph bucket: run the frame axis padded to a multiple of t_bucket (padding = mel2ph 0 -> masked like any
        # batch padding); the plan/graph cache is then keyed by the bucket and every frame-level output is cropped back.
        T_out = T
        T = self.bucket_frames(T_out)
        if T != T_out:
            mel2ph = _pad_frames(mel2ph, T).contiguous()
        L.check(lib.ss_count_nonzero_i64(L.ptr(mel2ph), L.ptr(lens_t), B, T, st()), "lens_t")
        ret["mel2ph"] = mel2ph
        dec = torch.empty(B, T, H, **f32)
        L.check(lib.ss_gather_expand(L.ptr(enc), L.ptr(mel2ph), L.ptr(dec), B, Tp, T, H, st()), "expand")
        # UMLN (a4): DistributionUncertainty returns x unchanged when not training (umln.py:48-50)

        # ---- Residual Style Adaptor (a5,a6) + style-to-content attention (a7) ----
        sc = kwargs.get("style_cache")
        if sc is None:
            sc = self.encode_style(ref_mels, ref_f0)
        sty, lens_r, Tr = sc["sty"], sc["lens_r"], sc["sty"].shape[1]
        ret["ref_f0"], ret["style_pre_rq"] = sc["ref_f0"], sc["style_pre_rq"]
        ret["rq_codes"], ret["style_rq"], ret["rq_loss"] = sc["rq_codes"], sc["style_rq"], 0.0
        xs = dec.clone()
        q = torch.empty(B, T, H, **f32)
        kv = torch.empty(B, Tr, 2 * H, **f32)
        att = torch.zeros(B, T, H, **f32)
        ffh = torch.empty(B, T, 2048, **f32)
        for al in pk["align"]:
            self._gemm(xs, al["q"], q, B, T, mask_rows=False)
            self._gemm(sty, al["kv"], kv, B, Tr, mask_rows=False)
            L.attention(q, kv, kv[:, :, H:], att, B=B, H=2, D=H // 2, Tq=T, Tk=Tr, ldq=H, ldk=2 * H, ldv=2 * H, ldo=H,
                        q_bs=T * H, k_bs=Tr * 2 * H, v_bs=Tr * 2 * H, o_bs=T * H, qlens=None, klens=lens_r, scale=(H // 2) ** -0.5)
            self._gemm(att, al["out"], xs, B, T, R=xs, mask_rows=False)
            L.layernorm(xs, *al["n1"], B=B, T=T, C_=H)
            self._gemm(xs, al["l1"], ffh, B, T, act=L.ACT_RELU, mask_rows=False)
            self._gemm(ffh, al["l2"], xs, B, T, R=xs, mask_rows=False)
            L.layernorm(xs, *al["n2"], B=B, T=T, C_=H)
        ret["style"] = style = xs
        ret["gloss"] = 0.0

        # ---- pitch: two joint Gaussian/multinomial diffusions (a8) + post-processing (a9) ----
        midi = torch.empty(B, T, device=dev, dtype=torch.int64)
        L.check(lib.ss_gather_expand_i64(L.ptr(note), L.ptr(mel2ph), L.ptr(midi), B, Tp, T, st()), "midi")
        pl = self._plan(B, T, dev, int(kwargs.get("plan_slot", 0)))
        pl.uses += 1
        pl.lens2[:B].copy_(lens_t)
        pl.lens2[B:].copy_(lens_t)
        pl.seed.fill_(seed)
        L.check(lib.ss_f0_bounds(L.ptr(midi), L.ptr(pl.lo2), L.ptr(pl.hi2), B * T, st()), "bounds")
        pl.lo2[B:].copy_(pl.lo2[:B])
        pl.hi2[B:].copy_(pl.hi2[:B])
        pl.cond_a.copy_(dec)  # = decoder_inp * tgt_nonpadding (the gather already wrote 0 on padding)
        L.check(lib.ss_add_bcast_mask(L.ptr(dec), L.ptr(spk), None, L.ptr(emo), L.ptr(style), L.ptr(pl.cond_b), B, T, H, L.ptr(lens_t), st()), "cond_b")
        pl.uv2.zero_()
        graphs = noise is None and self._want_graphs(pl)

        def tape_t(x):  # recorded noise [..., T_out] (reference layout) -> device fp32, frame axis padded to the bucket
            return _pad_frames(x.to(dev).float(), T)
        if noise is not None:
            S = pk["f0_pair"]["net"].steps
            na, nb_ = noise["f0_a"], noise["f0_b"]
            pl.f0[0].copy_(tape_t(na["z0"]).reshape(B, T))
            pl.f0[1].copy_(tape_t(nb_["z0"]).reshape(B, T))
            zs = torch.cat([tape_t(na["z_steps"]).reshape(S, B, T), tape_t(nb_["z_steps"]).reshape(S, B, T)], 1).contiguous()
            us = torch.cat([tape_t(na["u_steps"]).reshape(S, B, 2, T), tape_t(nb_["u_steps"]).reshape(S, B, 2, T)], 1).contiguous()
            self._run_f0_pair(pl, (zs, us))
        elif graphs:
            if pl.g_f0 is None:
                pl.g_f0 = self._capture(lambda: self._run_f0_pair(pl))
                pl.uv2.zero_()
            pl.g_f0.replay()
        else:
            self._run_f0_pair(pl)
        res = {"f0_a": (pl.f0[0].clone(), pl.uv[0].clone()), "f0_b": (pl.f0[1].clone(), pl.uv[1].clone())}
        ret["gdiff1"] = ret["mdiff1"] = ret["gdiff2"] = ret["mdiff2"] = 0.0
        pitch_pred = torch.empty(B, T, 2, **f32)
        f0_denorm = torch.empty(B, T, **f32)
        coarse = torch.empty(B, T, device=dev, dtype=torch.int64)
        L.check(lib.ss_pitch_post(L.ptr(res["f0_a"][0]), L.ptr(res["f0_a"][1]), L.ptr(res["f0_b"][0]), L.ptr(res["f0_b"][1]),
                                  L.ptr(midi), L.ptr(mel2ph), L.ptr(pitch_pred), L.ptr(f0_denorm), L.ptr(coarse), B * T, st()), "pitch_post")
        ret["pitch_pred"], ret["f0_denorm"], ret["f0_denorm_pred"] = pitch_pred, f0_denorm, f0_denorm
        ret["pitch_coarse"] = coarse
        ret["f0_a"], ret["uv_a"], ret["f0_b"], ret["uv_b"] = res["f0_a"][0], res["f0_a"][1], res["f0_b"][0], res["f0_b"][1]
        pitch_emb = torch.empty(B, T, H, **f32)
        L.check(lib.ss_embedding(L.ptr(coarse), L.ptr(self.p("pitch_embed.weight")), L.ptr(pitch_emb), B * T, H, 300, 1.0, 0, st()), "pitch emb")
        dec_inp = torch.empty(B, T, H, **f32)
        L.check(lib.ss_add_bcast_mask(L.ptr(dec), L.ptr(spk), L.ptr(pitch_emb), L.ptr(emo), L.ptr(style), L.ptr(dec_inp), B, T, H, L.ptr(lens_t), st()), "dec_inp")
        ret["decoder_inp"] = dec_inp
        if skip_decoder:
            return self._crop_frames(ret, T, T_out)
        if self.prodiff:
            # ---- ProDiff teacher decoder (stylesinger.py:175-177, modules/diff/prodiff.py:205-221): decoder_inp is the condition
            M = hp["audio_num_mel_bins"]
            S = int(hp["timesteps"])
            pl.cond_mel.copy_(dec_inp)
            sch = pk["prodiff_sched"]
            # the workspace must outlive this call: a captured graph replays into it (a per-call temporary would be freed and its
            # address reused by the allocator while pl.g_mel still writes there)
            if len(pl.ws_mel) == 1:
                wsb, wsp = pl.ws_mel[0]
            else:
                if pl.ws_prodiff is None:
                    wsb = lib.ss_wavenet_workspace_bytes(C_byref(pk["mel"]["net"]), B, T)
                    pl.ws_prodiff = (wsb, torch.empty(wsb, device=dev, dtype=torch.uint8))
                    pl.recount()
                wsb, wsp = pl.ws_prodiff

            def run_prodiff(zs=None):
                if zs is None:
                    L.check(lib.ss_fill_normal_rows(L.ptr(pl.xm), B, T * M, T * M, 31, L.ptr(pl.seed), st()), "x_T")
                L.check(lib.ss_prodiff_sample(C_byref(pk["mel"]["net"]), L.ptr(pl.xm), L.ptr(pl.cond_mel), L.ptr(pl.lens), B, T, L.ptr(zs),
                                              37, L.ptr(pl.seed), S, L.hptr(sch["c1"]), L.hptr(sch["c2"]), L.hptr(sch["sigma"]), 1,
                                              L.ptr(wsp), wsb, st()), "prodiff")
            if noise is not None:
                to_btm = lambda x, lead: _pad_frames(x.to(dev).float().reshape(*lead, B, M, T_out), T).transpose(-1, -2).contiguous()
                pl.xm.copy_(to_btm(noise["mel"]["z_q"], ()))
                run_prodiff(to_btm(noise["mel"]["z_steps"], (S,)))
            elif graphs:
                if pl.g_mel is None:
                    pl.g_mel = self._capture(run_prodiff)
                pl.g_mel.replay()
            else:
                run_prodiff()
            mel_out = torch.empty(B, T, M, **f32)
            # the reference leaves padded frames unmasked (prodiff.py:219-220); frames past lens[b] are written as 0 here
            L.check(lib.ss_add_bcast_mask(L.ptr(pl.xm), None, None, None, None, L.ptr(mel_out), B, T, M, L.ptr(lens_t), st()), "mel mask")
            ret["mel_out"] = mel_out
            ret["lens"] = lens_t
            return self._crop_frames(ret, T, T_out)

        # ---- FFT decoder -> coarse mel (a10) ----
        xd = dec_inp.clone()
        pos_t = torch.empty(B, T, device=dev, dtype=torch.int32)
        tabt = self._pos(T + 2, dev)
        L.check(lib.ss_make_positions(None, L.ptr(dec_inp), H, T * H, L.ptr(pos_t), B, T, st()), "pos dec")
        L.check(lib.ss_table_add(L.ptr(pos_t), L.ptr(tabt), tabt.shape[0], L.ptr(xd), H, T * H, B, T, H,
                                 L.ptr(self.p("decoder.pos_embed_alpha")), 1.0, 1, st()), "pos add dec")
        xd = self._fft_blocks(pk["dec"], xd, B, T, lens_t)
        ret["decoder_out"] = xd
        M = hp["audio_num_mel_bins"]
        coarse_mel = torch.empty(B, T, M, **f32)
        self._gemm(xd, pk["mel_out"], coarse_mel, B, T, lens=lens_t)
        ret["fs2_mel"] = coarse_mel
        ret["x_mask"] = (mel2ph > 0).float()[:, :, None]
        if not (global_steps > hp["diff_start"]):
            ret["mel_out"] = coarse_mel
            return self._crop_frames(ret, T, T_out)

        # ---- condition projection (a11) + shallow mel diffusion (a12) ----
        gcat = torch.cat([coarse_mel, dec_inp, spk[:, None, :].expand(-1, T, -1), emo[:, None, :].expand(-1, T, -1), style], -1).contiguous()
        cond = pl.cond_mel
        self._gemm(gcat, pk["ln_proj"], cond, B, T, mask_rows=False)
        ret["diff_cond"] = cond.clone()
        pl.coarse_mel.copy_(coarse_mel)
        K = hp["K_step"]
        ddim_ts = self.ddim_timesteps(int(kwargs["ddim_steps"])) if kwargs.get("sampler") == "ddim" else None
        plms = kwargs.get("plms_interval", hp.get("pndm_speedup")) if kwargs.get("sampler", "plms" if hp.get("pndm_speedup") else None) == "plms" else None
        def mel_tape(x, lead):  # [*lead, B, 1, M, T_out] -> [*lead, B, T, M] on the device
            return _pad_frames(x.to(dev).float().reshape(*lead, B, M, T_out), T).transpose(-1, -2).contiguous()
        if plms:
            zq_n = None if noise is None else mel_tape(noise["mel"]["z_q"], ())
            self._run_mel(pl, (zq_n, None), plms_interval=int(plms))
        elif ddim_ts is not None:
            eta = float(kwargs.get("eta", 0.0))
            if noise is not None:
                nz = noise["mel"]
                zs = mel_tape(nz["z_steps"], (K,)) if (eta > 0.0 and "z_steps" in nz) else None
                self._run_mel(pl, (mel_tape(nz["z_q"], ()), zs), ddim_ts=ddim_ts, eta=eta)
            elif graphs:  # one captured graph per (B, T bucket, number of sampler steps, eta)
                key = (len(ddim_ts), eta)
                if key not in pl.g_ddim:
                    pl.g_ddim[key] = self._capture(lambda: self._run_mel(pl, ddim_ts=ddim_ts, eta=eta))
                pl.g_ddim[key].replay()
            else:
                self._run_mel(pl, ddim_ts=ddim_ts, eta=eta)
        elif noise is not None or not graphs:
            # "fp16q4": the kernels behind the mode convert their fp16 operand to fp4 on a FIXED scale (q_scale_gate / q_scale_z): on the first
            # (eager) forward of every plan the library reduces max |a| / (6 q_scale) over every operand those launches read (ss_set_q4_guard);
            # a checkpoint whose stream leaves the scale's range is refused instead of silently degrading the second product
            guard = torch.zeros(2, device=dev, dtype=torch.int32) if (self.q4 and pl.uses <= 1) else None
            if guard is not None:
                L.check(lib.ss_set_q4_guard(L.ptr(guard)), "ss_set_q4_guard")
            try:
                if noise is not None:
                    nz = noise["mel"]
                    self._run_mel(pl, (mel_tape(nz["z_q"], ()), mel_tape(nz["z_steps"], (K,))))
                else:
                    self._run_mel(pl)
            finally:
                if guard is not None:
                    L.check(lib.ss_set_q4_guard(None), "ss_set_q4_guard")
            if guard is not None:
                worst = guard.view(torch.float32).cpu()
                if float(worst.max()) > 1.0:
                    raise L.StyleSingerHipError(
                        f"mfma_precision=fp16q4: an operand of the fp4 second product leaves its fixed scale (max |a| / (6 q_scale): gate "
                        f"{float(worst[0]):.3g}, skip GEMM {float(worst[1]):.3g}; the stream x + dstep must stay within +-{6 * 2.0:g}) - this checkpoint "
                        f"needs mfma_precision='fp16x2' (no fixed activation scale)")
        else:
            if pl.g_mel is None:
                pl.g_mel = self._capture(lambda: self._run_mel(pl))
            pl.g_mel.replay()
        xm = pl.xm
        mel_out = torch.empty(B, T, M, **f32)
        # the reference does not mask padded frames here (shallow_diffusion_tts.py:305-306); with per-item
        # lengths the frames past lens[b] are not part of the utterance, so they are written as 0.
        pl.nonfinite.zero_()
        L.check(lib.ss_mel_denorm(L.ptr(xm), L.ptr(pk["spec_min"]), L.ptr(pk["spec_max"]), L.ptr(mel_out), B, T, M, L.ptr(lens_t), L.ptr(pl.nonfinite), st()), "denorm")
        # fp16 terms carry the residual stream x + dstep and the gate outputs inside the stack: unlike the bf16 modes they overflow beyond 65504.
        # The denorm kernel flags a non-finite valid frame on EVERY forward (ret["nonfinite"], a device word: `check_finite(ret)` wherever the
        # caller synchronises anyway - infer.py does); the first forward of every plan checks it here (one host sync).
        ret["nonfinite"] = pl.nonfinite.clone()
        if self.f16 and pl.uses <= 1:
            self.check_finite(ret)
        ret["mel_out"] = mel_out
        ret["diff"] = 0.0
        ret["lens"] = lens_t
        return self._crop_frames(ret, T, T_out)

    def check_finite(self, ret):
        """Raise if the forward that produced `ret` wrote a non-finite valid mel frame (reads one device word: a host sync)."""
        flag = ret.get("nonfinite")
        if flag is not None and int(flag.item()) != 0:
            if self.f16:
                raise L.StyleSingerHipError("mfma_precision=fp16x2: non-finite mel after the diffusion loop - the residual stream of this checkpoint "
                                            "leaves the fp16 range (|x + dstep| > 65504); use mfma_precision='bf16x2' (fp32 exponent range)")
            raise L.StyleSingerHipError("non-finite mel after the diffusion loop")

    _FRAME_KEYS = ("mel2ph", "style", "pitch_pred", "f0_denorm", "f0_denorm_pred", "pitch_coarse", "f0_a", "uv_a", "f0_b", "uv_b",
                   "decoder_inp", "decoder_out", "fs2_mel", "x_mask", "diff_cond", "mel_out")

    @classmethod
    def _crop_frames(cls, ret, T, T_out):
        """Undo the bucket padding: frame-level outputs back to the caller's frame count."""
        if T != T_out:
            for k in cls._FRAME_KEYS:
                if k in ret:
                    ret[k] = ret[k][:, :T_out].contiguous()
        return ret


def C_byref(struct):
    import ctypes
    return ctypes.addressof(struct)
