"""Parameter contract of the accelerated path: names and shapes of the reference `state_dict`s.

`load_ckpt(model, ..., 'model', strict=False)` (utils/commons/ckpt_utils.py:26-67) makes parameter
NAMES and SHAPES the weight interface.  This module regenerates that list from the
hyper-parameters; tests/test_host_cpu.py::test_param_spec_matches_reference_dump pins it against a dump of the reference's own
`StyleSinger(...).state_dict()` / `HifiGanGenerator(...).state_dict()` (tests/golden/param_spec.json).
"""


def _fft_block(prefix, H, k, out):
    out += [
        (f"{prefix}.op.layer_norm1.weight", (H,)), (f"{prefix}.op.layer_norm1.bias", (H,)),
        (f"{prefix}.op.self_attn.in_proj_weight", (3 * H, H)), (f"{prefix}.op.self_attn.out_proj.weight", (H, H)),
        (f"{prefix}.op.layer_norm2.weight", (H,)), (f"{prefix}.op.layer_norm2.bias", (H,)),
        (f"{prefix}.op.ffn.ffn_1.weight", (4 * H, H, k)), (f"{prefix}.op.ffn.ffn_1.bias", (4 * H,)),
        (f"{prefix}.op.ffn.ffn_2.weight", (H, 4 * H)), (f"{prefix}.op.ffn.ffn_2.bias", (H,)),
    ]


def _predictor(prefix, H, layers, k, odim, out, pos=False):
    if pos:
        out.append((f"{prefix}.pos_embed_alpha", (1,)))
    for i in range(layers):
        out += [(f"{prefix}.conv.{i}.1.weight", (H, H, k)), (f"{prefix}.conv.{i}.1.bias", (H,)),
                (f"{prefix}.conv.{i}.3.weight", (H,)), (f"{prefix}.conv.{i}.3.bias", (H,))]
    out += [(f"{prefix}.linear.weight", (odim, H)), (f"{prefix}.linear.bias", (odim,))]
    if pos:
        out.append((f"{prefix}.embed_positions._float_tensor", (1,)))


def _wavenet(prefix, C, L, H, in_dims, out_dims, f0, out):
    if f0:
        out += [(f"{prefix}.input_projection.weight", (C // 2, in_dims, 1)), (f"{prefix}.input_projection.bias", (C // 2,)),
                (f"{prefix}.uv_embed.weight", (2, C // 2))]
    else:
        out += [(f"{prefix}.input_projection.weight", (C, in_dims, 1)), (f"{prefix}.input_projection.bias", (C,))]
    out += [(f"{prefix}.mlp.0.weight", (4 * C, C)), (f"{prefix}.mlp.0.bias", (4 * C,)),
            (f"{prefix}.mlp.2.weight", (C, 4 * C)), (f"{prefix}.mlp.2.bias", (C,))]
    for l in range(L):
        p = f"{prefix}.residual_layers.{l}"
        out += [(f"{p}.dilated_conv.weight", (2 * C, C, 3)), (f"{p}.dilated_conv.bias", (2 * C,)),
                (f"{p}.diffusion_projection.weight", (C, C)), (f"{p}.diffusion_projection.bias", (C,)),
                (f"{p}.conditioner_projection.weight", (2 * C, H, 1)), (f"{p}.conditioner_projection.bias", (2 * C,)),
                (f"{p}.output_projection.weight", (2 * C, C, 1)), (f"{p}.output_projection.bias", (2 * C,))]
    out += [(f"{prefix}.skip_projection.weight", (C, C, 1)), (f"{prefix}.skip_projection.bias", (C,)),
            (f"{prefix}.output_projection.weight", (out_dims, C, 1)), (f"{prefix}.output_projection.bias", (out_dims,))]


GAUSS_BUFFERS = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                 "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                 "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
MULTI_BUFFERS = ["log_alpha", "log_1_min_alpha", "log_cumprod_alpha", "log_1_min_cumprod_alpha", "Lt_history", "Lt_count"]


def acoustic_spec(hp):
    """[(name, shape)] in the reference's state_dict order (modules/StyleSinger/stylesinger.py:46-117)."""
    H = hp["hidden_size"]
    V = hp["vocab_size"]
    M = hp["audio_num_mel_bins"]
    out = [("encoder_embed_tokens.weight", (V, H))]
    for i in range(hp["enc_layers"]):
        _fft_block(f"encoder.layers.{i}", H, hp["enc_ffn_kernel_size"], out)
    out += [("encoder.layer_norm.weight", (H,)), ("encoder.layer_norm.bias", (H,)),
            ("encoder.embed_tokens.weight", (V, H)), ("encoder.embed_positions._float_tensor", (1,)),
            ("decoder.pos_embed_alpha", (1,)), ("decoder.embed_positions._float_tensor", (1,))]
    for i in range(hp["dec_layers"]):
        _fft_block(f"decoder.layers.{i}", H, hp["dec_ffn_kernel_size"], out)
    out += [("decoder.layer_norm.weight", (H,)), ("decoder.layer_norm.bias", (H,)),
            ("mel_out.weight", (M, H)), ("mel_out.bias", (M,)),
            ("spk_embed_proj.weight", (H, 256)), ("spk_embed_proj.bias", (H,))]
    _predictor("dur_predictor", H, hp["dur_predictor_layers"], hp["dur_predictor_kernel"], 1, out)
    out.append(("pitch_embed.weight", (300, H)))
    _predictor("pitch_predictor", H, hp["predictor_layers"], hp["predictor_kernel"], 2, out, pos=True)  # unused at infer
    out += [("note_encoder.emb.weight", (100, H)), ("note_encoder.type_emb.weight", (5, H)),
            ("note_encoder.dur_ln.weight", (H, 1)), ("note_encoder.dur_ln.bias", (H,)),
            ("emo_embed_proj.weight", (H, hp["emo_size"])), ("emo_embed_proj.bias", (H,)),
            ("norm.affine_layer.linear_layer.weight", (2 * H, H)), ("norm.affine_layer.linear_layer.bias", (2 * H,))]
    for rb in range(5):
        for blk in range(2):
            p = f"style_extractor.encoder.res_blocks.{rb}.blocks.{blk}"
            out += [(f"{p}.0.weight", (80,)), (f"{p}.0.bias", (80,)), (f"{p}.1.weight", (160, 80, 5)), (f"{p}.1.bias", (160,)),
                    (f"{p}.4.weight", (80, 160, 1)), (f"{p}.4.bias", (80,))]
    out += [("style_extractor.encoder.last_norm.weight", (80,)), ("style_extractor.encoder.last_norm.bias", (80,)),
            ("style_extractor.encoder.post_net1.weight", (H, 80, 3)), ("style_extractor.encoder.post_net1.bias", (H,))]
    for d in range(hp["rq_depth"]):
        p = f"style_extractor.rqvae.codebooks.{d}"
        out += [(f"{p}.weight", (hp["nRQ"] + 1, H)), (f"{p}.cluster_size_ema", (hp["nRQ"],)), (f"{p}.embed_ema", (hp["nRQ"], H))]
    for i in range(4):
        p = f"style_extractor.wavenet.in_layers.{i}"
        out += [(f"{p}.bias", (160,)), (f"{p}.weight_g", (160, 1, 1)), (f"{p}.weight_v", (160, 80, 3))]
    for i in range(4):
        p = f"style_extractor.wavenet.res_skip_layers.{i}"
        n = 160 if i < 3 else 80
        out += [(f"{p}.bias", (n,)), (f"{p}.weight_g", (n, 1, 1)), (f"{p}.weight_v", (n, 80, 1))]
    out += [("style_extractor.wavenet.cond_layer.bias", (640,)), ("style_extractor.wavenet.cond_layer.weight_g", (640, 1, 1)),
            ("style_extractor.wavenet.cond_layer.weight_v", (640, 80, 1)),
            ("l1.weight", (H, 2 * H)), ("l1.bias", (H,))]
    for i in range(2):
        p = f"align.layers.{i}"
        out += [(f"{p}.multihead_attn.in_proj_weight", (3 * H, H)), (f"{p}.multihead_attn.in_proj_bias", (3 * H,)),
                (f"{p}.multihead_attn.out_proj.weight", (H, H)), (f"{p}.multihead_attn.out_proj.bias", (H,)),
                (f"{p}.linear1.weight", (2048, H)), (f"{p}.linear1.bias", (2048,)),
                (f"{p}.norm1.weight", (H,)), (f"{p}.norm1.bias", (H,)),
                (f"{p}.linear2.weight", (H, 2048)), (f"{p}.linear2.bias", (H,)),
                (f"{p}.norm2.weight", (H,)), (f"{p}.norm2.bias", (H,))]
    C0, L0, S0 = hp["f0_residual_channels"], hp["f0_residual_layers"], hp["f0_timesteps"]
    for net, gen in (("gm_diffnet", "f0_gen"), ("gm_diffnet_inpainte", "f0_gen_inpainte")):
        _wavenet(net, C0, L0, H, 1, 3, True, out)
        out += [(f"{gen}.{b}", (S0,)) for b in MULTI_BUFFERS + GAUSS_BUFFERS]
        _wavenet(f"{gen}._denoise_fn", C0, L0, H, 1, 3, True, out)
    out.append(("embed_positions._float_tensor", (1,)))
    if hp.get("decoder", "diffsinger") == "prodiff":
        # ProDiffusion (modules/diff/prodiff.py:59-117, built at stylesinger.py:111-117): no ln_proj / postdiff; the
        # schedule has timesteps + 1 entries (:69-71)
        n = hp["timesteps"] + 1
        out += [("diff_decoder.timesteps", ()), ("diff_decoder.timescale", ())]
        out += [(f"diff_decoder.{b}", (n,)) for b in GAUSS_BUFFERS]
        out += [("diff_decoder.spec_min", (1, 1, hp["keep_bins"])), ("diff_decoder.spec_max", (1, 1, hp["keep_bins"]))]
        _wavenet("diff_decoder.denoise_fn", hp["residual_channels"], hp["residual_layers"], H, M, M, False, out)
        return out
    cond_hs = M + 4 * H
    out += [("ln_proj.weight", (H, cond_hs)), ("ln_proj.bias", (H,))]
    out += [(f"postdiff.{b}", (hp["timesteps"],)) for b in GAUSS_BUFFERS]
    out += [("postdiff.spec_min", (1, 1, hp["keep_bins"])), ("postdiff.spec_max", (1, 1, hp["keep_bins"]))]
    _wavenet("postdiff.denoise_fn", hp["residual_channels"], hp["residual_layers"], H, M, M, False, out)
    return out


def emotion_spec(hidden=256, n_mel=40, layers=3, embed=256):
    """EmotionEncoder.state_dict() (data_gen/tts/emotion/model.py:11-31): the two cosine-similarity scalars (training
    loss only), the 3-layer LSTM and the linear layer."""
    out = [("similarity_weight", (1,)), ("similarity_bias", (1,))]
    for l in range(layers):
        cin = n_mel if l == 0 else hidden
        out += [(f"lstm.weight_ih_l{l}", (4 * hidden, cin)), (f"lstm.weight_hh_l{l}", (4 * hidden, hidden)),
                (f"lstm.bias_ih_l{l}", (4 * hidden,)), (f"lstm.bias_hh_l{l}", (4 * hidden,))]
    out += [("linear.weight", (embed, hidden)), ("linear.bias", (embed,))]
    return out


def reorder_like_reference(spec):
    """The reference registers gm_diffnet, f0_gen, gm_diffnet_inpainte, f0_gen_inpainte in that order."""
    return spec


def vocoder_spec(cfg):
    """HifiGanGenerator state_dict with weight-norm still applied, i.e. the checkpoint format
    (`ckpt['state_dict']['model_gen']`, tasks/tts/vocoder_infer/hifigan_nsf.py:26-38)."""
    out = [("m_source.l_linear.weight", (1, cfg["harmonic_num"] + 1)), ("m_source.l_linear.bias", (1,))]
    c0 = cfg["upsample_initial_channel"]
    rates, ks = cfg["upsample_rates"], cfg["upsample_kernel_sizes"]
    import numpy as np
    for i in range(len(rates)):
        c = c0 // (2 ** (i + 1))
        if i + 1 < len(rates):
            s = int(np.prod(rates[i + 1:]))
            out += [(f"noise_convs.{i}.weight", (c, 1, 2 * s)), (f"noise_convs.{i}.bias", (c,))]
        else:
            out += [(f"noise_convs.{i}.weight", (c, 1, 1)), (f"noise_convs.{i}.bias", (c,))]
    out += [("conv_pre.bias", (c0,)), ("conv_pre.weight_g", (c0, 1, 1)), ("conv_pre.weight_v", (c0, 80, 7))]
    for i in range(len(rates)):
        cin, c = c0 // (2 ** i), c0 // (2 ** (i + 1))
        out += [(f"ups.{i}.bias", (c,)), (f"ups.{i}.weight_g", (cin, 1, 1)), (f"ups.{i}.weight_v", (cin, c, ks[i]))]
    nk = len(cfg["resblock_kernel_sizes"])
    for i in range(len(rates)):
        c = c0 // (2 ** (i + 1))
        for j, k in enumerate(cfg["resblock_kernel_sizes"]):
            for grp in ("convs1", "convs2"):
                for m in range(3):
                    p = f"resblocks.{i * nk + j}.{grp}.{m}"
                    out += [(f"{p}.bias", (c,)), (f"{p}.weight_g", (c, 1, 1)), (f"{p}.weight_v", (c, c, k))]
    c_last = c0 // (2 ** len(rates))
    out += [("conv_post.bias", (1,)), ("conv_post.weight_g", (1, 1, 1)), ("conv_post.weight_v", (1, c_last, 7))]
    return out
