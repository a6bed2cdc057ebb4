"""CPU restatement of the StyleSinger inference hot path  —  TEST INFRASTRUCTURE ONLY.

This file is the *oracle* of the repo: a plain fp32 PyTorch-on-CPU restatement of what the reference
computes on the path  inference/StyleSinger.py -> StyleSinger.forward(infer=True) -> HiFi-GAN-NSF,
written functionally over the reference's own `state_dict` (no nn.Modules, no code from the reference).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product
(`stylesinger_amd/`) never does and fails loudly without its HIP library.

Pinned: `oracle/gen_golden.py` runs the REAL reference (imported from /root/reference in the build
container) on the same seeded weights / inputs / noise tape and stores per-stage outputs under
tests/golden/*.pt; `tests/test_oracle_golden.py` checks this restatement against those fixtures.

Every function cites the reference lines it follows (paths relative to the reference root).
All tensors are channels-last [B, T, C] unless noted.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# small pieces
# ------------------------------------------------------------------------------------------------
def sinusoidal_table(n, dim, padding_idx=0):
    """modules/commons/common_layers.py:107-124 (SinusoidalPositionalEmbedding.get_embedding)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(n, -1)
    if padding_idx is not None:
        e[padding_idx, :] = 0
    return e


def make_positions(probe_nonzero):
    """utils/tts_utils.py:6-18 with padding_idx = 0: cumsum of the non-pad mask, zero at pads."""
    m = probe_nonzero.int()
    return (torch.cumsum(m, dim=1).type_as(m) * m).long()


def sinpos(probe_nonzero, dim):
    """common_layers.py:126-148."""
    pos = make_positions(probe_nonzero)
    tab = sinusoidal_table(int(pos.max().item()) + 2, dim)
    return tab[pos]


# BASELINE config 4 ("bf16 MFMA"): with set_matmul_rounding("bf16") the GEMMs the HIP path runs on bf16 matrix cores see
# both operands rounded to bf16 (round-to-nearest-even, exactly what v_cvt_pk_bf16_f32 does) and accumulate in fp32; every
# other operation stays fp32. The reference has no such mode: this switch is the only oracle for it.
_ROUND = None


# "bf16x2" (the HIP path's mfma_precision of the same name): every operand of the denoisers' hidden GEMMs is a PAIR of bf16 terms
# v = hi + mid (hi = RNE(v), mid = RNE(v - hi)) and the product is hi*hi + hi*mid + mid*hi in fp32 (mid*mid, 2^-16 of the leading term, is
# dropped); the step-invariant conditioner projection stays exact fp32 (`rounded="hoisted"` call site). Not a reference mode either, but it IS
# pinned: against the reference's fp32 goldens it stays at fp32-grade distance (tests/test_oracle_golden.py, 4e-6 on the 1000-step chain).


# "fp16x2" (the HIP path's mfma_precision of the same name): the same GEMMs on the FP16 matrix cores with only the WEIGHTS split - the
# activation is one fp16 term a = RNE16(v), a weight the pair (hi, lo) of fp16 terms of w * 2^FP16_WSHIFT (the shift keeps lo out of the fp16
# subnormals; it is undone in fp32 after the accumulation), the product a*hi + a*lo: 2 matrix products instead of 3. Why that is enough
# (oracle/bf16x2_numerics.py): the WEIGHT rounding is the coherent error that adds up over 1000 steps, the activation rounding averages out,
# and fp16's 11 significand bits make the latter 8x smaller than bf16's. The residual stream lives as the fp16 pair of x + dstep_l (22 bits);
# the conditioner projection stays exact fp32; the two f0 denoisers stay in the bf16x2 form. Pinned like bf16x2: 1.9e-5 vs the reference's 1000-step golden (bar 1e-4).
FP16_WSHIFT = 8


# "fp16q4" (kernels written, not yet run - the arithmetic contract of the next precision mode as it is WIRED): fp16x2 with
# the SECOND product of the dilated conv and of the skip GEMM on fp4 (e2m1) operands - a*hi on fp16 terms as before, plus q4(a)*q4(lo) on
# v_mfma_scale_f32_32x32x64_f8f6f4 (3.7x the fp16 issue rate, tools/ubench/mfma_mx_layout.hip): the activation on a FIXED power-of-two scale
# (the stream x + dstep: 2, gate outputs / the skip GEMM's operand: 2^-2 - the kernels convert their own fragment registers, no block maximum),
# the weights' lo terms with one E8M0 scale per 32-element block. The output projection (HBM-bound on the GPU) keeps fp16x2's two fp16 products.
# lo is a correction of relative size 2^-12, two significant bits of it are enough: 3.4e-5 / 4.2e-5 on the reference's 1000- / 100-step goldens
# (oracle/second_product_numerics.py). (The GPU's blocks are the 32 elements a lane holds over a step pair, not 32 consecutive channels, and its
# skip GEMM is the folded K = L*C form: the same arithmetic class, not the same bits.)
# "fp16sd" (round 6; the HIP path's mfma_precision of the same name): ONE fp16 product per hidden GEMM of the mel denoiser - fp16 activations as
# in fp16x2, but a single fp16 weight term - with the weight rounding NOISE-SHAPED over the loop's network evaluations instead of corrected by a
# second product. Evaluation j (j = 0 for the first evaluation of a sampling loop) uses weight set W_(j mod N), the N sets being a first-order
# sigma-delta sequence of fp16 roundings of the same scaled fp32 weight: r_0 = 0, W_k = RNE16(w 2^s + r_k), r_(k+1) = r_k + (w 2^s - W_k), so that the
# sum of N consecutive sets is N w 2^s up to ONE fp16 rounding. The weight rounding - the coherent error that ruins the plain one-product mode
# (1.94e-4 on the reference's 1000-step golden) - then averages out over the steps like the activation rounding does: N = 2 / 4 / 8 / 16 / 32 / 64:
# 1.10e-4 / 5.3e-5 / 3.2e-5 / 2.5e-5 / 2.2e-5 / 2.1e-5 (oracle/dither_numerics.py; fp16x2: 1.9e-5; a never-repeating sequence: 2.0e-5). Half the matrix
# work and half the weight bytes of fp16x2. Everything else as fp16x2 (conditioner projection exact, stream, f0 denoisers in bf16x2).
# The step-invariant conditioner ADDEND of every layer (conditioner projection + dilated-conv bias, the largest stream of the HIP path's fused layer
# launch) is kept the same way at configs[3]'s size: FP16SD_E_SETS fp16 sigma-delta sets of the exactly computed addend, set j mod N_e in evaluation j
# (ss_layer512_tile_addend_f16; the GPU rounds the addend times the gate's exp2 constant, the restatement the addend itself: same error class, not the same
# bits). 1 set: 9.5e-5, 4: 3.3e-5, 8: 2.5e-5, 16: 2.3e-5 (oracle/dither_numerics.py --e-sets=N). 0 = exact addend (the HIP path's small-launch form).
FP16SD_SETS = 32
FP16SD_E_SETS = 8
_SD_CALLS = {}
_SD_CACHE = {}


def _sd_reset():
    """start of a sampling loop: evaluation counter of every weight back to 0"""
    _SD_CALLS.clear()


def _sd_addend(w_key, c):
    """the exactly computed addend c of the layer whose conditioner weight is w_key -> its fp16 set of this evaluation"""
    if FP16SD_E_SETS <= 0:
        return c
    key = ("addend", id(w_key))
    j = _SD_CALLS.get(key, 0)
    _SD_CALLS[key] = j + 1
    ent = _SD_CACHE.get(key)
    if ent is None or ent[0] is not w_key or ent[2].shape != c.shape or not torch.equal(ent[2], c):
        sets, r = [], torch.zeros_like(c)
        for _ in range(FP16SD_E_SETS):
            ck = (c + r).half().float()
            r = r + (c - ck)
            sets.append(ck)
        ent = _SD_CACHE[key] = (w_key, sets, c)
    return ent[1][j % FP16SD_E_SETS]


def _sd_weight(w):
    j = _SD_CALLS.get(id(w), 0)
    _SD_CALLS[id(w)] = j + 1
    ent = _SD_CACHE.get(id(w))
    if ent is None or ent[0] is not w:
        ws = w * float(2 ** FP16_WSHIFT)
        sets, r = [], torch.zeros_like(ws)
        for _ in range(FP16SD_SETS):
            wk = (ws + r).half().float()
            r = r + (ws - wk)
            sets.append(wk)
        ent = _SD_CACHE[id(w)] = (w, sets)
    return ent[1][j % FP16SD_SETS]


_FP4_GRID = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
_FP4_MID = (_FP4_GRID[1:] + _FP4_GRID[:-1]) / 2


def mxfp4(x, dim):
    """x -> its MXFP4 value: blocks of 32 along `dim` share the scale 2^(floor(log2(max |x|)) - 2) (the block's largest element lands in [4, 8),
    the e2m1 grid's top is 6), elements round to the nearest grid point, ties to the even mantissa (0.25 -> 0, 0.75 -> 1, 1.25 -> 1, 1.75 -> 2,
    2.5 -> 2, 3.5 -> 4, 5 -> 4) and saturate at 6: what v_cvt_scalef32_pk_fp4_f16 does (profiles/r04_ubench_cvt_fp4_probe.log)."""
    xs = x.movedim(dim, -1).contiguous()
    shp = xs.shape
    K = shp[-1]
    pad = (-K) % 32
    if pad:
        xs = F.pad(xs, (0, pad))
    b = xs.reshape(*xs.shape[:-1], -1, 32)
    s = torch.exp2(torch.floor(torch.log2(b.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30))) - 2)
    v = (b / s).abs().clamp(max=6.0).contiguous()
    idx = torch.bucketize(v, _FP4_MID) + ((v == 0.75) | (v == 1.75) | (v == 3.5)).long()   # bucketize sends ties down; these three go up (even mantissa)
    q = (_FP4_GRID[idx] * b.sign() * s).reshape(*xs.shape)[..., :K].reshape(shp)
    return q.movedim(-1, dim)


def fp4_fixed(x, scale):
    """x -> fp4(x / scale) * scale: e2m1 on a fixed power-of-two scale, ties to the even mantissa, saturating at 6 (v_cvt_scalef32_pk_fp4_f16)"""
    v = (x / scale).abs().clamp(max=6.0).contiguous()
    idx = torch.bucketize(v, _FP4_MID) + ((v == 0.75) | (v == 1.75) | (v == 3.5)).long()
    return _FP4_GRID[idx] * x.sign() * scale


FP4_SCALE = {"dil": 2.0, "skip": 0.25}   # the fixed activation scales of the fp16q4 mode (ss_wavenet.q_scale_gate / q_scale_z)


def set_matmul_rounding(mode):
    global _ROUND
    assert mode in (None, "fp32", "bf16", "bf16x2", "fp16x2", "fp16q4", "fp16sd")
    _ROUND = None if mode in (None, "fp32") else mode
    _sd_reset()


def _r(x):
    return x.bfloat16().float() if _ROUND == "bf16" else x


def _split2(x):
    hi = x.bfloat16().float()
    return hi, (x - hi).bfloat16().float()


def _split2h(x):
    hi = x.half().float()
    return hi, (x - hi).half().float()


def conv1d_cl(x, w, b, dilation=1, rounded=False):
    """'same' Conv1d on channels-last input; w is the torch [Cout, Cin, k] parameter. rounded: True or a site name ("dil" | "out" | "skip") = a GEMM
    the HIP path runs on the 16-bit matrix cores in those modes; "hoisted" = the same, but step-invariant (exact fp32 in the split modes)."""
    k = w.shape[-1]
    pad = (k - 1) // 2 * dilation
    xt = x.transpose(1, 2)
    site = rounded if rounded in ("dil", "out", "skip") else None
    if site is not None:
        rounded = True
    if rounded is True and _ROUND == "bf16x2":
        (xh, xm), (wh, wm) = _split2(xt), _split2(w)
        y = F.conv1d(xm, wh, None, padding=pad, dilation=dilation) + F.conv1d(xh, wm, None, padding=pad, dilation=dilation)
        y = y + F.conv1d(xh, wh, None, padding=pad, dilation=dilation)
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    if rounded is True and _ROUND == "fp16q4" and site in FP4_SCALE:
        xh = xt.half().float()
        ws = w * float(2 ** FP16_WSHIFT)
        wh = ws.half().float()
        y = F.conv1d(fp4_fixed(xh, FP4_SCALE[site]), mxfp4(ws - wh, 1), None, padding=pad, dilation=dilation) + F.conv1d(xh, wh, None, padding=pad, dilation=dilation)
        y = y * float(2.0 ** -FP16_WSHIFT)
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    if rounded is True and _ROUND == "fp16sd":
        y = F.conv1d(xt.half().float(), _sd_weight(w), None, padding=pad, dilation=dilation) * float(2.0 ** -FP16_WSHIFT)
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    if rounded is True and _ROUND in ("fp16x2", "fp16q4"):
        xh = xt.half().float()
        wh, wl = _split2h(w * float(2 ** FP16_WSHIFT))
        y = F.conv1d(xh, wl, None, padding=pad, dilation=dilation) + F.conv1d(xh, wh, None, padding=pad, dilation=dilation)
        y = y * float(2.0 ** -FP16_WSHIFT)
        if b is not None:
            y = y + b.view(1, -1, 1)
        return y.transpose(1, 2)
    if rounded and _ROUND == "bf16":
        xt, w = _r(xt), _r(w)
    return F.conv1d(xt, w, b, padding=pad, dilation=dilation).transpose(1, 2)


def weight_norm_fold(sd, prefix):
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v|| over all dims but 0."""
    v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(g.shape)
    return v * (g / n)


def mha(q_in, kv_in, w_in, b_in, w_out, b_out, n_heads, key_pad):
    """F.multi_head_attention_forward with need_weights=True (explicit softmax path)
    (modules/commons/common_layers.py:277-286; torch nn.MultiheadAttention used by lse.py:19,41).
    q_in [B,Tq,H], kv_in [B,Tk,H], key_pad [B,Tk] bool (True = masked)."""
    B, Tq, H = q_in.shape
    Tk = kv_in.shape[1]
    D = H // n_heads
    wq, wk, wv = w_in[:H], w_in[H:2 * H], w_in[2 * H:]
    bq = bk = bv = None
    if b_in is not None:
        bq, bk, bv = b_in[:H], b_in[H:2 * H], b_in[2 * H:]
    q = F.linear(q_in, wq, bq).view(B, Tq, n_heads, D).transpose(1, 2)
    k = F.linear(kv_in, wk, bk).view(B, Tk, n_heads, D).transpose(1, 2)
    v = F.linear(kv_in, wv, bv).view(B, Tk, n_heads, D).transpose(1, 2)
    s = torch.matmul(q * (D ** -0.5), k.transpose(-1, -2))
    if key_pad is not None:
        s = s.masked_fill(key_pad[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, Tq, H)
    return F.linear(o, w_out, b_out)


# ------------------------------------------------------------------------------------------------
# FFT blocks (modules/fastspeech/tts_modules.py:250-306, common_layers.py:624-673, :541-582)
# ------------------------------------------------------------------------------------------------
def fft_blocks(sd, prefix, x, pad_mask, n_layers, n_heads, use_pos_alpha):
    """x [B,T,H]; pad_mask [B,T] bool (True = padding)."""
    H = x.shape[-1]
    keep = (~pad_mask).float()[:, :, None]
    if use_pos_alpha:  # decoder: positions from x[..., 0] != 0 (tts_modules.py:289-291)
        x = x + sd[prefix + ".pos_embed_alpha"] * sinpos(x[..., 0] != 0, H)
    x = x * keep
    for i in range(n_layers):
        p = f"{prefix}.layers.{i}.op"
        res = x
        h = F.layer_norm(x, (H,), sd[p + ".layer_norm1.weight"], sd[p + ".layer_norm1.bias"], 1e-5)
        h = mha(h, h, sd[p + ".self_attn.in_proj_weight"], None, sd[p + ".self_attn.out_proj.weight"], None, n_heads, pad_mask)
        x = (res + h) * keep
        res = x
        h = F.layer_norm(x, (H,), sd[p + ".layer_norm2.weight"], sd[p + ".layer_norm2.bias"], 1e-5)
        w1 = sd[p + ".ffn.ffn_1.weight"]
        h = conv1d_cl(h, w1, sd[p + ".ffn.ffn_1.bias"]) * (w1.shape[-1] ** -0.5)
        h = F.gelu(h)
        h = F.linear(h, sd[p + ".ffn.ffn_2.weight"], sd[p + ".ffn.ffn_2.bias"])
        x = (res + h) * keep
        x = x * keep  # FFTBlocks.forward multiplies once more (tts_modules.py:297)
    x = F.layer_norm(x, (H,), sd[prefix + ".layer_norm.weight"], sd[prefix + ".layer_norm.bias"], 1e-5) * keep
    return x


def encoder(sd, hp, txt_tokens):
    """FastspeechEncoder.forward (tts_modules.py:326-346)."""
    H = hp["hidden_size"]
    x = math.sqrt(H) * sd["encoder.embed_tokens.weight"][txt_tokens]
    x = x + sinpos(txt_tokens != 0, H)
    return fft_blocks(sd, "encoder", x, txt_tokens == 0, hp["enc_layers"], hp["num_heads"], False)


def note_encoder(sd, hp, note, note_dur, note_type):
    """NoteEncoder.forward (modules/StyleSinger/stylesinger.py:31-36)."""
    H = hp["hidden_size"]
    x = sd["note_encoder.emb.weight"][note] * math.sqrt(H)
    ty = sd["note_encoder.type_emb.weight"][note_type] * math.sqrt(H)
    du = F.linear(note_dur.unsqueeze(-1), sd["note_encoder.dur_ln.weight"], sd["note_encoder.dur_ln.bias"])
    return x + du + ty


def duration_predictor(sd, hp, x, src_pad):
    """DurationPredictor._forward + out2dur (tts_modules.py:105-130)."""
    keep = (~src_pad).float()[:, :, None]
    h = x
    for i in range(hp["dur_predictor_layers"]):
        p = f"dur_predictor.conv.{i}"
        h = conv1d_cl(h, sd[p + ".1.weight"], sd[p + ".1.bias"])
        h = F.relu(h)
        h = F.layer_norm(h, (h.shape[-1],), sd[p + ".3.weight"], sd[p + ".3.bias"], 1e-5)
        h = h * keep
    xs = F.linear(h, sd["dur_predictor.linear.weight"], sd["dur_predictor.linear.bias"]) * keep
    dur = torch.clamp(torch.round(xs.squeeze(-1).exp() - 1.0), min=0).long()
    return dur, xs


def length_regulator(dur, src_pad):
    """LengthRegulator.forward (tts_modules.py:158-188)."""
    dur = dur * (1 - src_pad.long())
    B, Tp = dur.shape
    T = int(dur.sum(-1).max().item())
    mel2ph = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        pos = 0
        for i in range(Tp):
            d = int(dur[b, i])
            mel2ph[b, pos:pos + d] = i + 1
            pos += d
    return mel2ph


def expand_states(h, mel2ph):
    """fs2.py:258-262 / stylesinger.py:15-19."""
    h = F.pad(h, [0, 0, 1, 0])
    return torch.gather(h, 1, mel2ph[..., None].expand(-1, -1, h.shape[-1]))


# ------------------------------------------------------------------------------------------------
# Residual Style Adaptor (modules/StyleSinger/lse.py:93-129, wavenet.py:54-78, RQ.py)
# ------------------------------------------------------------------------------------------------
def wn_prenet(sd, x, keep):
    """WN.forward with g=None (wavenet.py:54-78). x [B,T,80], keep [B,T,1]."""
    Hc = x.shape[-1]
    out = torch.zeros_like(x)
    n_layers = 4
    for i in range(n_layers):
        w = weight_norm_fold(sd, f"style_extractor.wavenet.in_layers.{i}")
        xin = conv1d_cl(x, w, sd[f"style_extractor.wavenet.in_layers.{i}.bias"])
        acts = torch.tanh(xin[..., :Hc]) * torch.sigmoid(xin[..., Hc:])
        w2 = weight_norm_fold(sd, f"style_extractor.wavenet.res_skip_layers.{i}")
        rs = conv1d_cl(acts, w2, sd[f"style_extractor.wavenet.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[..., :Hc]) * keep
            out = out + rs[..., Hc:]
        else:
            out = out + rs
    return out * keep


def conv_blocks(sd, x):
    """ConvBlocks.forward / ResidualBlock.forward (lse.py:192-200,229-240). x [B,T,80] -> [B,T,256]."""
    C = x.shape[-1]
    nonpad = (x.abs().sum(-1) > 0).float()[:, :, None]
    for rb in range(5):
        nonpad_rb = (x.abs().sum(-1) > 0).float()[:, :, None]
        for blk in range(2):
            p = f"style_extractor.encoder.res_blocks.{rb}.blocks.{blk}"
            h = F.layer_norm(x, (C,), sd[p + ".0.weight"], sd[p + ".0.bias"], 1e-5)
            w = sd[p + ".1.weight"]
            h = conv1d_cl(h, w, sd[p + ".1.bias"]) * (w.shape[-1] ** -0.5)
            h = F.gelu(h)
            h = conv1d_cl(h, sd[p + ".4.weight"], sd[p + ".4.bias"])
            x = (x + h) * nonpad_rb
    x = x * nonpad
    x = F.layer_norm(x, (C,), sd["style_extractor.encoder.last_norm.weight"], sd["style_extractor.encoder.last_norm.bias"], 1e-5) * nonpad
    x = conv1d_cl(x, sd["style_extractor.encoder.post_net1.weight"], sd["style_extractor.encoder.post_net1.bias"]) * nonpad
    return x


def rq_lookup(sd, hp, x):
    """RQBottleneck.forward at eval (RQ.py:226-270, VQEmbedding :30-55)."""
    B, T, C = x.shape
    r = x.reshape(-1, C).clone()
    agg = torch.zeros_like(r)
    codes = []
    for d in range(hp["rq_depth"]):
        cb = sd[f"style_extractor.rqvae.codebooks.{d}.weight"][:-1]
        cbt = cb.t()
        dist = torch.addmm(r.pow(2.0).sum(1, keepdim=True) + cbt.pow(2.0).sum(0, keepdim=True), r, cbt, alpha=-2.0)
        k = dist.argmin(dim=-1)
        q = cb[k]
        r = r - q
        agg = agg + q
        codes.append(k)
    agg = agg.reshape(B, T, C)
    return x + (agg - x), torch.stack(codes, -1).reshape(B, T, -1)


def style_adaptor(sd, hp, ref_mels, ref_f0):
    """LocalStyleAdaptor.forward (lse.py:103-129)."""
    pad = ref_mels[:, :, 0] == 0
    keep = (~pad).float()[:, :, None]
    h = wn_prenet(sd, ref_mels, keep)
    if ref_f0.dim() == 1:  # the entrypoint passes a 1-D f0 (inference/StyleSinger.py:152-154)
        ref_f0 = ref_f0[None]
    h = h + ref_f0[:, :, None]
    h = conv_blocks(sd, h)
    z, codes = rq_lookup(sd, hp, h)
    return z, codes, h


def prosody_aligner(sd, hp, content, style):
    """get_style + ProsodyAligner/CrossAttenLayer, forcing=False (stylesinger.py:189-214, lse.py:28-81)."""
    H = hp["hidden_size"]
    pos = sinpos(style[:, :, 0] != 0, H)
    style = F.linear(torch.cat([style, pos], -1), sd["l1.weight"], sd["l1.bias"])
    key_pad = style[:, :, 0] == 0  # computed AFTER l1 (stylesinger.py:203): practically never true
    x = content
    for i in range(2):
        p = f"align.layers.{i}"
        a = mha(x, style, sd[p + ".multihead_attn.in_proj_weight"], sd[p + ".multihead_attn.in_proj_bias"],
                sd[p + ".multihead_attn.out_proj.weight"], sd[p + ".multihead_attn.out_proj.bias"], 2, key_pad)
        x = F.layer_norm(x + a, (H,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
        f = F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), sd[p + ".linear2.weight"],
                     sd[p + ".linear2.bias"])
        x = F.layer_norm(x + f, (H,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    return x


# ------------------------------------------------------------------------------------------------
# WaveNet-style denoisers (modules/diff/net.py:58-130,215-266; Mish diffusion.py:64-66)
# ------------------------------------------------------------------------------------------------
def step_embedding(sd, prefix, t, C):
    """SinusoidalPosEmb + mlp (net.py:32-44,98-102,118-119)."""
    half = C // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half) * -e)
    e = t[:, None].float() * e[None, :]
    e = torch.cat((e.sin(), e.cos()), dim=-1)
    h = F.linear(e, sd[prefix + ".mlp.0.weight"], sd[prefix + ".mlp.0.bias"])
    h = h * torch.tanh(F.softplus(h))
    return F.linear(h, sd[prefix + ".mlp.2.weight"], sd[prefix + ".mlp.2.bias"])


def residual_stack(sd, prefix, x, cond, demb, L, cycle):
    """x [B,T,C], cond [B,T,H], demb [B,C] -> relu(skip_projection(sum(skip)/sqrt(L)))  (net.py:66-78,120-127)."""
    C = x.shape[-1]
    skip = 0
    for l in range(L):
        p = f"{prefix}.residual_layers.{l}"
        d = 2 ** (l % cycle)
        ds = F.linear(demb, sd[p + ".diffusion_projection.weight"], sd[p + ".diffusion_projection.bias"])[:, None, :]
        c = conv1d_cl(cond, sd[p + ".conditioner_projection.weight"], sd[p + ".conditioner_projection.bias"], rounded="hoisted")
        if _ROUND == "fp16sd":   # the addend (+ the dilated conv's bias, added below in exact arithmetic here) as fp16 sigma-delta sets over the evaluations
            c = _sd_addend(sd[p + ".conditioner_projection.weight"], c)
        xin = x + ds
        if _ROUND == "bf16x2":   # the HIP path keeps the residual stream ONLY as the (hi, mid) pair of x + dstep_l (16 significant bits)
            hi, mid = _split2(xin)
            xin = hi + mid
            x = xin - ds
        if _ROUND in ("fp16x2", "fp16q4", "fp16sd"):   # ... as the fp16 pair (22 significant bits); the matrix cores read its hi term only
            hi, lo = _split2h(xin)
            xin = hi + lo
            x = xin - ds
        y = conv1d_cl(xin, sd[p + ".dilated_conv.weight"], sd[p + ".dilated_conv.bias"], dilation=d, rounded="dil") + c
        y = torch.sigmoid(y[..., :C]) * torch.tanh(y[..., C:])
        y = conv1d_cl(y, sd[p + ".output_projection.weight"], sd[p + ".output_projection.bias"], rounded="out")
        x = (x + y[..., :C]) / math.sqrt(2.0)
        skip = skip + y[..., C:]
    h = skip / math.sqrt(L)
    h = conv1d_cl(h, sd[prefix + ".skip_projection.weight"], sd[prefix + ".skip_projection.bias"], rounded="skip")
    return F.relu(h)


def diffnet(sd, hp, x, t, cond, prefix="postdiff.denoise_fn"):
    """DiffNet.forward (net.py:107-130). x [B,T,80]."""
    C = hp["residual_channels"]
    h = F.relu(conv1d_cl(x, sd[prefix + ".input_projection.weight"], sd[prefix + ".input_projection.bias"]))
    demb = step_embedding(sd, prefix, t, C)
    h = residual_stack(sd, prefix, h, cond, demb, hp["residual_layers"], hp["dilation_cycle_length"])
    return conv1d_cl(h, sd[prefix + ".output_projection.weight"], sd[prefix + ".output_projection.bias"])


def ddiffnet(sd, hp, f0, uv, t, cond, prefix):
    """DDiffNet.forward (net.py:242-266), nonpadding = ones. f0 [B,T], uv [B,T] long -> [B,T,3]."""
    C = hp["f0_residual_channels"]
    a = conv1d_cl(f0[:, :, None], sd[prefix + ".input_projection.weight"], sd[prefix + ".input_projection.bias"])
    e = sd[prefix + ".uv_embed.weight"][uv]
    h = torch.cat([a, e], dim=-1)
    demb = step_embedding(sd, prefix, t, C)
    global _ROUND
    saved = _ROUND
    if saved in ("fp16x2", "fp16q4", "fp16sd"):   # the f0 denoisers keep the three-product bf16 form in these modes (their outputs feed discrete voicing decisions)
        _ROUND = "bf16x2"
    try:
        h = residual_stack(sd, prefix, h, cond, demb, hp["f0_residual_layers"], hp["f0_dilation_cycle_length"])
    finally:
        _ROUND = saved
    return conv1d_cl(h, sd[prefix + ".output_projection.weight"], sd[prefix + ".output_projection.bias"])


def _log_add_exp(a, b):
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


def gm_sample(sd, hp, gen, net, cond, lo, hi, tape, trace=None):
    """GaussianMultinomialDiffusion.sample (gaussian_multinomial_diffusion.py:922-942) with
    gaussian_p_sample :326-333, p_sample/p_pred/q_posterior :374-413, log_sample_categorical :447-452.
    cond [B,T,H]; lo/hi [B,T].  Returns f0 [B,T], uv [B,T] long."""
    B, T, _ = cond.shape
    S = hp["f0_timesteps"]
    g = lambda k: sd[f"{gen}.{k}"]
    _ = tape.rand(B, 1, T)  # categorical init over a size-1 class dim: always class 0 (:924-926)
    uv = torch.zeros(B, T, dtype=torch.long)
    f0 = tape.randn(B, 1, T)[:, 0]
    log2 = np.log(2)
    for i in reversed(range(S)):
        t = torch.full((B,), i, dtype=torch.long)
        out = ddiffnet(sd, hp, f0, uv, t, cond, net)
        eps, logits = out[..., 0], out[..., 1:]
        # Gaussian part
        x0 = g("sqrt_recip_alphas_cumprod")[i] * f0 - g("sqrt_recipm1_alphas_cumprod")[i] * eps
        x0 = torch.min(torch.max(x0, lo), hi)
        mean = g("posterior_mean_coef1")[i] * x0 + g("posterior_mean_coef2")[i] * f0
        z = tape.randn(B, 1, T)[:, 0]
        nz = 0.0 if i == 0 else 1.0
        f0_new = mean + nz * (0.5 * g("posterior_log_variance_clipped")[i]).exp() * z
        # multinomial part (class dim last here)
        log_xt = torch.log(F.one_hot(uv, 2).float().clamp(min=1e-30))
        log_x0 = F.log_softmax(logits, dim=-1)
        tm1 = max(i - 1, 0)
        ev = _log_add_exp(log_x0 + g("log_cumprod_alpha")[tm1], g("log_1_min_cumprod_alpha")[tm1] - log2)
        if i == 0:
            ev = log_x0
        un = ev + _log_add_exp(log_xt + g("log_alpha")[i], g("log_1_min_alpha")[i] - log2)
        logp = un - torch.logsumexp(un, dim=-1, keepdim=True)
        u = tape.rand(B, 2, T).transpose(1, 2)
        gum = -torch.log(-torch.log(u + 1e-30) + 1e-30)
        uv = (gum + logp).argmax(dim=-1)
        f0 = f0_new
        if trace is not None:
            trace.append((i, f0.clone(), uv.clone()))
    return f0, uv


def f0_bounds(midi):
    """dyn_clip bounds (stylesinger.py:260-283). midi [B,T] long."""
    def nb(n):
        x = (2 ** ((n - 69) / 12) * 440).log2()
        x = torch.clamp(x, None, 10)
        v = (x - 6) / (10 - 6) * 2 - 1
        return v.clamp(-1, 1)
    return nb(midi - 3), nb(midi + 3)


def pitch_post(f0_a, uv_a, f0_b, uv_b, midi, mel2ph):
    """add_gmdiff_pitch tail + inpaint_pitch (stylesinger.py:216-247,286-311) + pitch_utils.py:22-31,65-78."""
    rest = midi == 0
    ua = uv_a.float().clone(); ua[rest] = 1
    ub = uv_b.float().clone(); ub[rest] = 1
    fa = (f0_a + 1) / 2 * (10 - 6) + 6
    fb = (f0_b + 1) / 2 * (10 - 6) + 6
    f = fb / 2 + fa / 2
    u = ub / 2 + ua / 2
    pitch_pred = torch.stack([f, u], -1)
    hz = 2 ** f
    hz = hz.masked_fill(u > 0, 0.0).masked_fill(mel2ph == 0, 0.0)
    f0_mel_min = 1127 * np.log(1 + 50.0 / 700)
    f0_mel_max = 1127 * np.log(1 + 1100.0 / 700)
    mel = 1127 * (1 + hz / 700).log()
    pos = mel > 0
    mel[pos] = (mel[pos] - f0_mel_min) * 254 / (f0_mel_max - f0_mel_min) + 1
    mel[mel <= 1] = 1
    mel[mel > 255] = 255
    return pitch_pred, hz, (mel + 0.5).long()


def mel_diffusion(sd, hp, coarse_mel, cond, tape, trace=None):
    """DiffusionDecoder.forward infer branch (shallow_diffusion_tts.py:285-307) incl. q_sample/p_sample/norm/denorm."""
    _sd_reset()
    g = lambda k: sd[f"postdiff.{k}"]
    smin, smax = g("spec_min")[0], g("spec_max")[0]  # [1,80]
    K = hp["K_step"]
    B, T, M = coarse_mel.shape
    x = (coarse_mel - smin) / (smax - smin) * 2 - 1
    zq = tape.randn(B, 1, M, T)[:, 0].transpose(1, 2)
    x = g("sqrt_alphas_cumprod")[K - 1] * x + g("sqrt_one_minus_alphas_cumprod")[K - 1] * zq
    for i in reversed(range(K)):
        t = torch.full((B,), i, dtype=torch.long)
        eps = diffnet(sd, hp, x, t, cond)
        x0 = g("sqrt_recip_alphas_cumprod")[i] * x - g("sqrt_recipm1_alphas_cumprod")[i] * eps
        x0 = x0.clamp(-1.0, 1.0)
        mean = g("posterior_mean_coef1")[i] * x0 + g("posterior_mean_coef2")[i] * x
        z = tape.randn(B, 1, M, T)[:, 0].transpose(1, 2)
        nz = 0.0 if i == 0 else 1.0
        x = mean + nz * (0.5 * g("posterior_log_variance_clipped")[i]).exp() * z
        if trace is not None:
            trace.append((i, x.clone()))
    return (x + 1) / 2 * (smax - smin) + smin


def prodiff_sample(sd, hp, cond, tape, trace=None):
    """ProDiffusion.forward infer branch (modules/diff/prodiff.py:205-221): x ~ N(0,1); for t = T-1..0 the denoiser predicts
    x0 DIRECTLY (:150-153, no clamp) and q_posterior_sample (:141-148) draws x_{t-1}; norm/denorm are identities (:223-227).
    The posterior noise is drawn at every step, t = 0 included (multiplied by 0 there)."""
    _sd_reset()
    g = lambda k: sd[f"diff_decoder.{k}"]
    B, T, _ = cond.shape
    M = hp["audio_num_mel_bins"]
    x = tape.randn(B, 1, M, T)[:, 0].transpose(1, 2)
    for i in reversed(range(int(hp["timesteps"]))):
        t = torch.full((B,), i, dtype=torch.long)
        x0 = diffnet(sd, hp, x, t, cond, prefix="diff_decoder.denoise_fn")
        mean = g("posterior_mean_coef1")[i] * x0 + g("posterior_mean_coef2")[i] * x
        z = tape.randn(B, 1, M, T)[:, 0].transpose(1, 2)
        nz = 0.0 if i == 0 else 1.0
        x = mean + nz * (0.5 * g("posterior_log_variance_clipped")[i]).exp() * z
        if trace is not None:
            trace.append((i, x.clone()))
    return x


def mel_ddim(sd, hp, coarse_mel, cond, tape, ts, eta=0.0):
    """Strided DDIM-family sampler (Song et al. 2021 eq. 12 with sigma of eq. 16) over the reference's DiffNet and schedule.
    eta = 0 (deterministic, BASELINE config 5) is NOT in the reference. eta = 1 with ts = K-1 ... 0 is algebraically the reference's
    ancestral p_sample (shallow_diffusion_tts.py:136-162: with eps recomputed from the CLAMPED x0 the update is
    posterior_mean_coef1 * x0 + posterior_mean_coef2 * x + sqrt(posterior_variance) * z) and draws its noise in the same order (one draw
    per step, t = 0 included) - tests/test_oracle_golden.py pins this function to the reference's golden `acoustic_t64_s100` that way.
    The schedule is rebuilt in float64 from `betas` as the reference builds its tables (:77-80)."""
    _sd_reset()
    g = lambda k: sd[f"postdiff.{k}"]
    smin, smax = g("spec_min")[0], g("spec_max")[0]
    K = hp["K_step"]
    B, T, M = coarse_mel.shape
    x = (coarse_mel - smin) / (smax - smin) * 2 - 1
    zq = tape.randn(B, 1, M, T)[:, 0].transpose(1, 2)
    x = g("sqrt_alphas_cumprod")[K - 1] * x + g("sqrt_one_minus_alphas_cumprod")[K - 1] * zq
    ac = np.cumprod(1.0 - g("betas").double().numpy())
    for i, t_ in enumerate(ts):
        t = torch.full((B,), int(t_), dtype=torch.long)
        eps = diffnet(sd, hp, x, t, cond)
        x0 = (g("sqrt_recip_alphas_cumprod")[t_] * x - g("sqrt_recipm1_alphas_cumprod")[t_] * eps).clamp(-1.0, 1.0)
        ac_t = float(ac[t_])
        ac_p = float(ac[ts[i + 1]]) if i + 1 < len(ts) else 1.0
        sig = eta * np.sqrt((1 - ac_p) / (1 - ac_t)) * np.sqrt(max(0.0, 1 - ac_t / ac_p))
        c2 = np.sqrt(max(0.0, 1 - ac_p - sig * sig) / (1 - ac_t))
        c1 = np.sqrt(ac_p) - c2 * np.sqrt(ac_t)
        x = np.float32(c1) * x0 + np.float32(c2) * x
        if eta > 0:
            z = tape.randn(B, 1, M, T)[:, 0].transpose(1, 2)
            x = x + np.float32(sig) * z
    return (x + 1) / 2 * (smax - smin) + smin


def mel_plms(sd, hp, coarse_mel, cond, tape, interval):
    """PLMS sampler: modules/diff/shallow_diffusion_tts.py:165-197 (p_sample_plms) driven as GaussianDiffusion.forward does
    with hparams['pndm_speedup'] = interval (:239-260). Pinned by tests/golden/plms_*.pt (the reference's own method)."""
    _sd_reset()
    g = lambda k: sd[f"postdiff.{k}"]
    smin, smax = g("spec_min")[0], g("spec_max")[0]
    K = hp["K_step"]
    B, T, M = coarse_mel.shape
    x = (coarse_mel - smin) / (smax - smin) * 2 - 1
    zq = tape.randn(B, 1, M, T)[:, 0].transpose(1, 2)
    x = g("sqrt_alphas_cumprod")[K - 1] * x + g("sqrt_one_minus_alphas_cumprod")[K - 1] * zq
    ac = g("alphas_cumprod")

    def x_pred(x, eps, t, tp):
        a_t, a_p = ac[t], ac[tp]
        a_t_sq, a_p_sq = a_t.sqrt(), a_p.sqrt()
        delta = (a_p - a_t) * ((1 / (a_t_sq * (a_t_sq + a_p_sq))) * x - 1 / (a_t_sq * (((1 - a_p) * a_t).sqrt() + ((1 - a_t) * a_p).sqrt())) * eps)
        return x + delta

    hist = []
    for t_ in reversed(range(0, K, interval)):
        tp = max(t_ - interval, 0)
        eps = diffnet(sd, hp, x, torch.full((B,), t_, dtype=torch.long), cond)
        if len(hist) == 0:
            xp = x_pred(x, eps, t_, tp)
            eps_prev = diffnet(sd, hp, xp, torch.full((B,), tp, dtype=torch.long), cond)
            prime = (eps + eps_prev) / 2
        elif len(hist) == 1:
            prime = (3 * eps - hist[-1]) / 2
        elif len(hist) == 2:
            prime = (23 * eps - 16 * hist[-1] + 5 * hist[-2]) / 12
        else:
            prime = (55 * eps - 59 * hist[-1] + 37 * hist[-2] - 9 * hist[-3]) / 24
        x = x_pred(x, prime, t_, tp)
        hist = (hist + [eps])[-4:]
    return (x + 1) / 2 * (smax - smin) + smin


# ------------------------------------------------------------------------------------------------
# top level: StyleSinger.forward(infer=True)  (modules/StyleSinger/stylesinger.py:119-187)
# ------------------------------------------------------------------------------------------------
def acoustic_forward(sd, hp, inp, tape, mel2ph=None, stages=None, mel_sampler=None):
    """inp: dict(txt_tokens, note, note_dur, note_type, spk_embed, emo_embed, ref_mels, ref_f0).
    Returns the reference's `ret` dict (inference keys)."""
    ret = {}
    txt = inp["txt_tokens"]
    ret["encoder_out"] = enc0 = encoder(sd, hp, txt)
    enc = enc0 + note_encoder(sd, hp, inp["note"], inp["note_dur"], inp["note_type"])
    src_keep = (txt > 0).float()[:, :, None]
    ret["spk_embed"] = spk = F.linear(inp["spk_embed"], sd["spk_embed_proj.weight"], sd["spk_embed_proj.bias"])[:, None, :]
    ret["emo_embed"] = emo = F.linear(inp["emo_embed"], sd["emo_embed_proj.weight"], sd["emo_embed_proj.bias"])[:, None, :]
    dur_inp = (enc + spk + emo) * src_keep
    if mel2ph is None:
        dur, xs = duration_predictor(sd, hp, dur_inp, txt == 0)
        ret["dur"], ret["dur_choice"] = xs, dur
        mel2ph = length_regulator(dur, txt == 0)
    else:
        _, xs = duration_predictor(sd, hp, dur_inp, txt == 0)
        ret["dur"] = xs.squeeze(-1)
    ret["mel2ph"] = mel2ph
    keep = (mel2ph > 0).float()[:, :, None]
    dec_inp = expand_states(enc, mel2ph)  # UMLN is the identity at eval (umln.py:48-50)
    z, codes, pre_rq = style_adaptor(sd, hp, inp["ref_mels"], inp["ref_f0"])
    ret["rq_codes"], ret["style_pre_rq"], ret["style_rq"] = codes, pre_rq, z
    ret["ref_f0"] = inp["ref_f0"]
    ret["style"] = style = prosody_aligner(sd, hp, dec_inp, z)
    midi = expand_states(inp["note"][:, :, None], mel2ph)[:, :, 0]
    lo, hi = f0_bounds(midi)
    cond_a = dec_inp * keep
    cond_b = (dec_inp + spk + emo + style) * keep
    f0_a, uv_a = gm_sample(sd, hp, "f0_gen", "gm_diffnet", cond_a, lo, hi, tape)
    f0_b, uv_b = gm_sample(sd, hp, "f0_gen_inpainte", "gm_diffnet_inpainte", cond_b, lo, hi, tape)
    ret["f0_a"], ret["uv_a"], ret["f0_b"], ret["uv_b"] = f0_a, uv_a, f0_b, uv_b
    ret["pitch_pred"], ret["f0_denorm"], coarse = pitch_post(f0_a, uv_a, f0_b, uv_b, midi, mel2ph)
    ret["f0_denorm_pred"] = ret["f0_denorm"]
    ret["pitch_coarse"] = coarse
    pitch_emb = sd["pitch_embed.weight"][coarse]
    ret["decoder_inp"] = dec_inp = (dec_inp + spk + pitch_emb + emo + style) * keep
    if hp.get("decoder", "diffsinger") == "prodiff":  # stylesinger.py:175-177: the diffusion decoder takes decoder_inp as its condition
        ret["mel_out"] = prodiff_sample(sd, hp, dec_inp, tape)
        return ret
    pad = dec_inp.abs().sum(-1) == 0
    ret["decoder_out"] = h = fft_blocks(sd, "decoder", dec_inp, pad, hp["dec_layers"], hp["num_heads"], True)
    ret["fs2_mel"] = coarse_mel = F.linear(h, sd["mel_out.weight"], sd["mel_out.bias"]) * keep
    ret["x_mask"] = keep
    T = coarse_mel.shape[1]
    gcat = torch.cat([coarse_mel, dec_inp, spk.expand(-1, T, -1), emo.expand(-1, T, -1), style], -1)
    ret["diff_cond"] = cond = F.linear(gcat, sd["ln_proj.weight"], sd["ln_proj.bias"])
    # mel_sampler: another sampler over the same denoiser with mel_diffusion's signature (tests pin mel_ddim(eta=1) to the goldens this way)
    ret["mel_out"] = (mel_sampler or mel_diffusion)(sd, hp, coarse_mel, cond, tape)
    return ret


# ------------------------------------------------------------------------------------------------
# emotion encoder (input producer): data_gen/tts/emotion/model.py:11-78, inference.py:39-53,139-151
# ------------------------------------------------------------------------------------------------
def emotion_lstm(esd, frames, layers=3):
    """nn.LSTM(40, 256, 3, batch_first=True) restated step by step (gate order i, f, g, o; torch.nn.LSTM docs);
    frames [P, n_frames, 40] -> final hidden state of the last layer [P, 256] (= EmotionEncoder.inference, model.py:62-78)."""
    x = frames
    for l in range(layers):
        w_ih, w_hh = esd[f"lstm.weight_ih_l{l}"], esd[f"lstm.weight_hh_l{l}"]
        b = esd[f"lstm.bias_ih_l{l}"] + esd[f"lstm.bias_hh_l{l}"]
        H = w_hh.shape[1]
        P, n, _ = x.shape
        xp = F.linear(x, w_ih, b)  # [P, n, 4H]
        h = torch.zeros(P, H)
        c = torch.zeros(P, H)
        outs = []
        for t in range(n):
            gts = xp[:, t] + F.linear(h, w_hh)
            i, f, g, o = gts[:, :H], gts[:, H:2 * H], gts[:, 2 * H:3 * H], gts[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        x = torch.stack(outs, 1)
    return x[:, -1]


def emotion_embed(esd, frames):
    """embed_utterance's tail (inference.py:147-151): mean of the partial embeddings, L2-normalised."""
    partial = emotion_lstm(esd, frames)
    raw = partial.mean(0)
    return raw / raw.norm(2), partial


def emotion_forward(esd, frames):
    """EmotionEncoder.forward (model.py:40-60): relu(linear(h_last)), L2-normalised per row (training-time embedding)."""
    e = F.relu(F.linear(emotion_lstm(esd, frames), esd["linear.weight"], esd["linear.bias"]))
    return e / e.norm(dim=1, keepdim=True)


def speaker_embed(ssd, frames):
    """resemblyzer.VoiceEncoder (un-vendored dependency of the reference, requirements.txt: resemblyzer==0.1.1.dev0; used at
    inference/StyleSinger.py:100,104): forward = ReLU(linear(h_last)) L2-normalised per partial - the architecture of the emotion encoder -
    and embed_utterance's tail = L2-normalised mean of the partial embeddings. PARITY UNPINNED: the package and its weights are not in
    /root/reference; this restates its published algorithm."""
    partial = emotion_forward(ssd, frames)
    raw = partial.mean(0)
    return raw / raw.norm(2), partial


# ------------------------------------------------------------------------------------------------
# HiFi-GAN-NSF (modules/hifigan/hifigan_nsf.py:105-169, modules/parallel_wavegan/models/source.py:311-531)
# ------------------------------------------------------------------------------------------------
def nsf_source(vsd, cfg, f0, tape):
    """f0 [B,T] Hz -> har_source [B,L] (SineGen.forward + SourceModuleHnNSF.forward)."""
    hop = int(np.prod(cfg["upsample_rates"]))
    sr = cfg["audio_sample_rate"]
    dim = cfg["harmonic_num"] + 1
    B, T = f0.shape
    f0u = f0[:, :, None].repeat(1, 1, hop).reshape(B, T * hop)  # nn.Upsample(nearest)
    fb = torch.zeros(B, T * hop, dim)
    fb[:, :, 0] = f0u
    for i in range(dim - 1):
        fb[:, :, i + 1] = fb[:, :, 0] * (i + 2)
    rad = (fb / sr) % 1
    ini = tape.rand(B, dim)
    ini[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + ini
    tmp = torch.cumsum(rad, 1) % 1
    over = (tmp[:, 1:, :] - tmp[:, :-1, :]) < 0
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * 0.1
    uv = (f0u > 0).float()[:, :, None]
    namp = uv * 0.003 + (1 - uv) * 0.1 / 3
    noise = namp * tape.randn(B, T * hop, dim)
    sw = sines * uv + noise
    merged = torch.tanh(F.linear(sw, vsd["m_source.l_linear.weight"], vsd["m_source.l_linear.bias"]))
    _ = tape.randn(B, T * hop, 1)
    return merged[:, :, 0]


def hifigan_forward(vsd, cfg, mel, f0, tape):
    """HifiGanGenerator.forward after remove_weight_norm. mel [B,T,80], f0 [B,T] Hz -> wav [B, T*hop], har [B,L]."""
    har = nsf_source(vsd, cfg, f0, tape)
    rates, ks = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    nk = len(cfg["resblock_kernel_sizes"])
    x = F.conv1d(_r(mel.transpose(1, 2)), _r(weight_norm_fold(vsd, "conv_pre")), vsd["conv_pre.bias"], padding=3)
    for i, (u, k) in enumerate(zip(rates, ks)):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(_r(x), _r(weight_norm_fold(vsd, f"ups.{i}")), vsd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i + 1 < len(rates):
            s = int(np.prod(rates[i + 1:]))
            xs_ = F.conv1d(har[:, None, :], vsd[f"noise_convs.{i}.weight"], vsd[f"noise_convs.{i}.bias"], stride=s, padding=s // 2)
        else:
            xs_ = F.conv1d(har[:, None, :], vsd[f"noise_convs.{i}.weight"], vsd[f"noise_convs.{i}.bias"])
        x = x + xs_
        acc = None
        for j, kk in enumerate(cfg["resblock_kernel_sizes"]):
            y = x
            for m, d in enumerate(cfg["resblock_dilation_sizes"][j]):
                p = f"resblocks.{i * nk + j}"
                t = F.leaky_relu(y, 0.1)
                t = F.conv1d(_r(t), _r(weight_norm_fold(vsd, f"{p}.convs1.{m}")), vsd[f"{p}.convs1.{m}.bias"], dilation=d,
                             padding=(kk * d - d) // 2)
                t = F.leaky_relu(t, 0.1)
                t = F.conv1d(_r(t), _r(weight_norm_fold(vsd, f"{p}.convs2.{m}")), vsd[f"{p}.convs2.{m}.bias"], padding=(kk - 1) // 2)
                y = t + y
            acc = y if acc is None else acc + y
        x = acc / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, weight_norm_fold(vsd, "conv_post"), vsd["conv_post.bias"], padding=3)
    return torch.tanh(x)[:, 0], har


def postprocess_for_vocoder(hp, mel_out, f0_denorm):
    """inference/StyleSinger.py:54-62 for one item: drop all-zero frames, clip mel."""
    keep = mel_out.abs().sum(-1) > 0
    mel = mel_out[keep].clamp(hp["mel_vmin"], hp["mel_vmax"])
    f0 = f0_denorm[:keep.shape[0]][keep]
    return mel, f0
