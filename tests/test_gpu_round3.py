"""Round-3 GPU tests: the C4 shape with full-length chains against the oracle on this box, the reference-f0 conditioning on the
device, the wired input producers (`preprocess_batch`), and the 16x16-tile Winograd gate kernel against the round-2 forms."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import record_measurement  # noqa: E402
from oracle import restatement as R  # noqa: E402
from stylesinger_amd import config, synth  # noqa: E402
from stylesinger_amd.model import StyleSingerHIP  # noqa: E402


def _model(hp, seed, dev="cuda:0"):
    m = StyleSingerHIP(None, hparams=hp)
    m.load_state_dict(synth.synth_acoustic_state_dict(hp, seed))
    m.eval().to(dev)
    return m


def _fwd(model, b, **kw):
    return model(b["txt_tokens"], mel2ph=b.get("mel2ph"), spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                 ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], **kw)


def test_norm_interp_f0_on_the_device_matches_the_reference_function(golden_dir):
    """ss_norm_interp_f0 (one launch for a ragged batch) vs the REAL utils/pitch_utils.py:47-62 outputs: voicing flags exact,
    contour within 1 fp32 ulp of values in [6, 10] (2e-6; the device entry takes fp32 Hz, numpy may see the tracker's float64)."""
    from stylesinger_amd import pitch
    cases = torch.load(os.path.join(golden_dir, "norm_interp_f0.pt"), weights_only=False)["cases"]
    keys = sorted(cases)
    Tm = max(cases[k]["hz"].numel() for k in keys)
    hz = torch.zeros(len(keys), Tm)
    lens = torch.zeros(len(keys), dtype=torch.int32)
    for i, k in enumerate(keys):
        n = cases[k]["hz"].numel()
        hz[i, :n] = cases[k]["hz"].float()
        hz[i, n:] = 123.0      # junk past the item's length must not leak in
        lens[i] = n
    f0, uv = pitch.norm_interp_f0_device(hz.cuda(), lens.cuda(), config.make_hparams())
    f0, uv = f0.cpu(), uv.cpu()
    worst = 0.0
    for i, k in enumerate(keys):
        n = int(lens[i])
        assert torch.equal(uv[i, :n], cases[k]["uv"]), k
        err = (f0[i, :n] - cases[k]["f0"]).abs().max().item()
        worst = max(worst, err)
        assert err <= 2e-6, (k, err)
        assert (f0[i, n:] == 0).all() and (uv[i, n:] == 0).all(), k
    record_measurement("norm_interp_f0_device_vs_reference", max_abs_err=worst, cases=len(keys))


def test_c4_shape_item_matches_oracle_with_100_step_chains():
    """BASELINE configs[3]'s SHAPE (30 s = 5625 frames, Tp=105, Tr=1500) with full-length 100 + 2x100 step chains in fp32 against the
    oracle's run on this box (~1 min of CPU): the multi-step check at this shape that round 2 only had at 2 steps. (The 1000-step
    schedule itself is pinned at T=32 by the real-reference golden `acoustic_t32_mel1000`.)"""
    S = 100
    hp = config.make_hparams(dict(timesteps=S, K_step=S, f0_timesteps=S))
    B, T, Tp, Tr = 1, 5625, 105, 1500
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 2025)
    sd = synth.synth_acoustic_state_dict(hp, 2025)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(78), B, T, S, S)
    model = _model(hp, 2025)
    got = _fwd(model, {k: v.cuda() for k, v in batch.items()}, noise=noise)
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(78), mel2ph=batch["mel2ph"])
    assert torch.equal(got["rq_codes"].cpu(), ref["rq_codes"])
    flips = (got["uv_a"].cpu().long() != ref["uv_a"]).sum().item() + (got["uv_b"].cpu().long() != ref["uv_b"]).sum().item()
    cf = (got["pitch_coarse"].cpu() != ref["pitch_coarse"])
    keep = torch.ones(B, T, dtype=torch.bool)
    for bb, tt in cf.nonzero().tolist():
        keep[bb, max(0, tt - 20):tt + 21] = False
    dm = (got["mel_out"].cpu() - ref["mel_out"]).abs()
    l1, mx = dm.mean().item(), dm.max().item()
    l1k, mxk = dm[keep].mean().item(), dm[keep].max().item()
    f0e = (got["f0_denorm"].cpu() - ref["f0_denorm"]).abs().max().item()
    print(f"C4 shape (T=5625, 100+100+100 steps): mel L1 {l1:.3e} max {mx:.3e}; away from flips {l1k:.3e} / {mxk:.3e}; voicing flips {flips}; "
          f"coarse flips {int(cf.sum())}/{T}; f0 max err {f0e:.3e} Hz")
    record_measurement("c4_shape_t5625_100steps_vs_oracle", mel_l1=l1, mel_max=mx, mel_l1_away_from_flips=l1k, mel_max_away_from_flips=mxk,
                       voicing_flips=flips, coarse_flips=int(cf.sum()), f0_max_err_hz=f0e)
    assert flips == 0
    assert cf.float().mean().item() <= 1e-3
    assert l1k <= 1e-5 and mxk <= 1e-3, (l1k, mxk)
    # the opt-in bf16x3 mode at this shape against the same oracle run (round-4 hardening; bf16x2 has its own test in test_gpu_round4.py)
    m3 = _model(dict(hp, mfma_precision="bf16x3"), 2025)
    got3 = _fwd(m3, {k: v.cuda() for k, v in batch.items()}, noise=noise)
    d3 = (got3["mel_out"].cpu() - ref["mel_out"]).abs()
    fl3 = (got3["uv_a"].cpu().long() != ref["uv_a"]).sum().item() + (got3["uv_b"].cpu().long() != ref["uv_b"]).sum().item()
    print(f"C4 shape in bf16x3 mode: mel L1 {d3.mean().item():.3e} max {d3.max().item():.3e}; voicing flips {fl3}")
    record_measurement("c4_shape_t5625_100steps_bf16x3_vs_oracle", mel_l1=d3.mean().item(), mel_max=d3.max().item(), voicing_flips=fl3)
    assert fl3 == 0 and d3.mean().item() <= 2e-5


@pytest.mark.parametrize("mt", [1, 2, 3])   # 1 = the small-launch tiling (round 4: one short utterance leaves most CUs idle at MT = 2)
def test_gate16_grouped_pair_launch_matches_the_32x32_kernel(mt):
    """The f0-pair form of the launch (C = 192, two weight sets selected by b // group_size, conditioner slab with a layer stride,
    ragged lens, a gate_mode-1 pass as the RSA uses it) on 16x16x4 tiles vs the round-2 32x32x2 kernel: same arithmetic up to the
    summation order over K -> 1e-5 on gate outputs in (-1, 1); rows past lens written as 0; the pick model returns a tiling."""
    import math
    from stylesinger_amd import lib as L
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    B, T, C, Lyr = 6, 333, 192, 3
    x = torch.randn(B, T, C, generator=g).to(dv)
    ws = [torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C) for _ in range(2)]
    Wt = torch.stack([L.pack_conv_weight(L.wino43_weight(w.to(dv)), interleave_half=C) for w in ws]).contiguous()
    Np = Wt.shape[1]
    ab = torch.randn(2, C, generator=g).to(dv)
    bias = (torch.randn(2, Np, generator=g) * 0.3).to(dv)
    E = torch.randn(B, T, Lyr * Np, generator=g).to(dv)
    lens = torch.tensor([T, T - 5, 1, T - 40, 17, T - 1], dtype=torch.int32).to(dv)
    for d, mode in ((1, 0), (4, 0), (8, 1)):
        kw = dict(dilation=d, B=B, T=T, Cin=C, N=C, Np=Np, Kp=C, lens=lens, a_bias=ab, bias=bias, E=E[:, :, Np:], lde=Lyr * Np,
                  e_bs=T * Lyr * Np, ldc=C, mask_rows=True, gate_mode=mode, group_size=3, w_gs=Wt[0].numel(), bias_gs=Np, a_bias_gs=C)
        want = torch.full((B, T, C), 7.0, device=dv)
        got = torch.full((B, T, C), 9.0, device=dv)
        L.wino43_gate(x, Wt, want, **kw)
        L.wino43_gate16(x, Wt, got, mt=mt, **kw)
        err = (got - want).abs().max().item()
        assert err <= 1e-5, (d, mode, err)
        for i in range(B):
            assert torch.all(got[i, int(lens[i]):] == 0)
    lib = L.load()
    assert lib.ss_wino43_gate16_pick(8, 1500, 512, 2) == 2 and lib.ss_wino43_gate16_pick(16, 1500, 384, 1) == 3   # mel: 768 x MT=2, f0 pair: 768 x MT=3
    assert lib.ss_wino43_gate16_pick(32, 5625, 512, 4) == 0      # many rounds per launch: the 32x32x2 tiles
    assert lib.ss_wino43_gate16_pick(1, 750, 512, 2) == 1 and lib.ss_wino43_gate16_pick(2, 750, 384, 8) == 1   # one 4 s utterance: 16-quad tiles


@pytest.mark.parametrize("mt", [2, 3])
def test_gate16_k_staged_form_is_bit_identical(mt):
    """`gate16_ks` (all six Winograd components of a K chunk staged at once, one barrier per K chunk) only changes WHEN a component
    is built, not the arithmetic or the order the products enter an accumulator: outputs equal the one-component-per-barrier form bit
    for bit - mel shape (C = 256, 8 K chunks), f0-pair shape (C = 192, 6, grouped weights, bias), C = 64 (2 chunks: no steady-state
    loop), ragged lens, every dilation of the cycle."""
    import math
    from stylesinger_amd import lib as L
    dv = torch.device("cuda:0")
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    before = lib.ss_get_tuning(b"gate16_ks")
    try:
        for (B, T, C, grouped) in ((3, 700, 256, False), (4, 333, 192, True), (2, 95, 64, False)):
            x = torch.randn(B, T, C, generator=g).to(dv)
            nw = 2 if grouped else 1
            ws = [torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C) for _ in range(nw)]
            Wt = torch.stack([L.pack_conv_weight(L.wino43_weight(w.to(dv)), interleave_half=C) for w in ws]).contiguous()
            Np = Wt.shape[1]
            ab = torch.randn(nw, C, generator=g).to(dv)
            bias = (torch.randn(nw, Np, generator=g) * 0.3).to(dv) if grouped else None
            E = torch.randn(B, T, 2 * Np, generator=g).to(dv)
            lens = torch.tensor([T, T - 7, 5, T - 1][:B], dtype=torch.int32).to(dv)
            for d in (1, 2, 4, 8):
                kw = dict(dilation=d, B=B, T=T, Cin=C, N=C, Np=Np, Kp=C, lens=lens, a_bias=ab, bias=bias, E=E[:, :, Np:], lde=2 * Np,
                          e_bs=T * 2 * Np, ldc=C, mask_rows=True)
                if grouped:
                    kw.update(group_size=2, w_gs=Wt[0].numel(), bias_gs=Np, a_bias_gs=C)
                outs = []
                W16 = torch.stack([L.pack_gate16_weights(Wt[i], C) for i in range(nw)]).contiguous()
                for ks, w16 in ((0, None), (1, None), (1, W16), (0, W16)):   # K staging x weight layout: the same arithmetic, the same order
                    L.check(lib.ss_set_tuning(b"gate16_ks", ks), "ss_set_tuning")
                    o = torch.full((B, T, C), 9.0, device=dv)
                    L.wino43_gate16(x, Wt if grouped else Wt[0], o, mt=mt, W16=None if w16 is None else (w16 if grouped else w16[0]), **kw)
                    outs.append(o)
                for o in outs[1:]:
                    assert torch.equal(outs[0], o), (B, T, C, d, (outs[0] - o).abs().max().item())
    finally:
        L.check(lib.ss_set_tuning(b"gate16_ks", before), "ss_set_tuning")


def test_preprocess_batch_feeds_infer_batch_from_device_buffers(golden_dir):
    """`StyleSingerInfer.preprocess_batch` (reference audio -> ref mel, emotion embedding, normalised f0) (a) equals the
    batch assembled by hand from the stand-alone producers bit for bit, tensors and the mel `infer_batch` makes of them; (b) every
    producer agrees with its oracle: librosa-restated mels (oracle/frontend.py), the emotion LSTM restatement on the oracle's 40-mel
    partials, and the REAL reference's norm_interp_f0 (host mirror, pinned bit-exactly by tests/golden/norm_interp_f0.pt)."""
    from oracle import frontend as OF
    from stylesinger_amd import pitch
    from stylesinger_amd.emotion import EmotionEncoderHIP, compute_partial_slices
    from stylesinger_amd.frontend import EmotionMelFrontendHIP, MelFrontendHIP
    from stylesinger_amd.infer import StyleSingerInfer
    dev = torch.device("cuda:0")
    hp = config.make_hparams(dict(timesteps=3, K_step=3, f0_timesteps=3))
    esd = synth.synth_emotion_state_dict(5)
    inf = StyleSingerInfer(hp, device=dev, model_state=synth.synth_acoustic_state_dict(hp, 5), vocoder_state=synth.synth_vocoder_state_dict(None, 5),
                           emotion_state=esd)
    g = torch.Generator().manual_seed(11)
    lens = [43200, 33711]                      # 0.9 s and 0.7 s of "48 kHz" audio, the second not a multiple of the hop
    B, Lmax = len(lens), max(lens)
    wav = torch.zeros(B, Lmax)
    for b, n in enumerate(lens):
        t = torch.arange(n) / 48000.0
        wav[b, :n] = (0.05 * torch.sin(2 * np.pi * (180.0 + 40 * b) * t) + 0.02 * torch.sin(2 * np.pi * 1234.0 * t) + 0.01 * torch.randn(n, generator=g)) * (0.2 if b == 0 else 1.0)   # item 0 is below -30 dBFS: normalize_volume raises it
    Tr = 1 + Lmax // 256
    frames = [1 + n // 256 for n in lens]
    f0 = torch.zeros(B, Tr, dtype=torch.float64)
    for b in range(B):
        f0[b, :frames[b]] = synth.synth_f0_hz(b, frames[b], 5)
    T, Tp = 48, 6
    it = synth.synth_batch(B, T, Tp, 8, hp, 5)
    args = dict(txt_tokens=it["txt_tokens"], note=it["note"], note_dur=it["note_dur"], note_type=it["note_type"], mel2ph=it["mel2ph"])
    batch = inf.preprocess_batch(wav.to(dev), lens, it["spk_embed"], f0.float(), **args)
    # (a) hand-assembled from the stand-alone producers
    mf, ef, enc = MelFrontendHIP(hp, device=dev), EmotionMelFrontendHIP(dev), EmotionEncoderHIP(esd, device=dev)
    mel_h, fr_h = mf.wav2mel(wav.to(dev), torch.tensor(lens))
    f0_h, _ = pitch.norm_interp_f0_device(f0.float().to(dev), fr_h, hp)
    ew = ef.normalize_volume(wav.to(dev), torch.tensor(lens))
    emo_h = []
    for b, n in enumerate(lens):
        ws, ms = compute_partial_slices(n)
        need = max(n, ws[-1].stop)
        one = torch.zeros(1, need, device=dev)
        one[0, :n] = ew[b, :n]
        m40, _ = ef.wav2mel(one, [need])
        emo_h.append(enc.embed_utterance_frames(m40[0].cpu(), n_samples=n))
    emo_h = torch.stack(emo_h)
    assert torch.equal(batch["ref_mels"], mel_h) and torch.equal(batch["ref_f0"], f0_h)
    assert torch.equal(batch["emo_embed"], emo_h.to(dev)), (batch["emo_embed"] - emo_h.to(dev)).abs().max().item()
    hand = dict(batch, ref_mels=mel_h, ref_f0=f0_h, emo_embed=emo_h.to(dev))
    r1 = inf.infer_batch(batch, seed=3, vocode=False)
    r2 = inf.infer_batch(hand, seed=3, vocode=False)
    assert torch.equal(r1["mel"], r2["mel"]) and torch.isfinite(r1["mel"]).all()
    # (b) against the oracles
    worst = dict(mel=0.0, f0=0.0, mel40=0.0, emo=0.0)
    for b, n in enumerate(lens):
        ref_mel = torch.from_numpy(OF.wav2mel(wav[b, :n].numpy()))
        assert ref_mel.shape[0] == frames[b]
        worst["mel"] = max(worst["mel"], (batch["ref_mels"][b, :frames[b]].cpu() - ref_mel).abs().max().item())
        assert (batch["ref_mels"][b, frames[b]:] == 0).all()
        f0_ref, _ = pitch.norm_interp_f0(f0[b, :frames[b]].numpy(), hp)
        worst["f0"] = max(worst["f0"], (batch["ref_f0"][b, :frames[b]].cpu() - f0_ref).abs().max().item())
        ew_ref = OF.normalize_volume(wav[b, :n].numpy())
        fr_ref = torch.from_numpy(OF.embed_utterance_frames(ew_ref))
        ws, ms = compute_partial_slices(n)
        need = max(n, ws[-1].stop)
        one = torch.zeros(1, need, device=dev)
        one[0, :n] = ew[b, :n]
        m40 = ef.wav2mel(one, [need])[0][0].cpu()
        got_fr = torch.stack([m40[s] for s in ms])
        worst["mel40"] = max(worst["mel40"], ((got_fr - fr_ref).abs().max() / fr_ref.abs().max()).item())
        emo_ref, _ = R.emotion_embed(esd, fr_ref)
        worst["emo"] = max(worst["emo"], (batch["emo_embed"][b].cpu() - emo_ref).abs().max().item())
    print("preprocess_batch vs oracles:", worst)
    record_measurement("preprocess_batch_vs_oracles", **worst)
    assert worst["mel"] <= 5e-5 and worst["f0"] <= 2e-6 and worst["mel40"] <= 1e-5 and worst["emo"] <= 2e-5, worst


@pytest.mark.parametrize("mt", [0, 4, 6, 8])
@pytest.mark.parametrize("C,B,T,groups", [(256, 3, 333, 1), (192, 4, 200, 2), (256, 1, 1536, 1), (192, 2, 97, 1)])
def test_gemm16_res_matches_torch_and_the_generic_kernel(C, B, T, groups, mt):
    """ss_gemm16_res (16x16x4 tiles, LDS-DMA A ring, register-resident weights): x <- (x + g . W_res^T + b) / sqrt(2) in place, ragged
    lens (rows past lens written as 0, the DMA zero-fills rows past lens), grouped weight sets, only the first C of the 2C packed rows
    used - vs torch float64 and vs ss_conv_gemm's RESSKIP epilogue."""
    import math
    from stylesinger_amd import lib as L
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + T + mt)
    Lyr = 3
    ga = torch.randn(B, T, Lyr * C, generator=g)           # gate outputs of all layers: the operand is a column slice (lda = L*C)
    x = torch.randn(B, T, C, generator=g)
    ws = [torch.randn(2 * C, C, 1, generator=g) / math.sqrt(C) for _ in range(groups)]
    bs = [torch.randn(2 * C, generator=g) * 0.1 for _ in range(groups)]
    lens = torch.tensor([max(1, T - 13 * i) for i in range(B)], dtype=torch.int32)
    s = 1.0 / math.sqrt(2.0)
    ref = torch.zeros(B, T, C)
    for i in range(B):
        n, gi = int(lens[i]), i * groups // B
        v = ga[i, :n, C:2 * C].double() @ ws[gi][:C, :, 0].double().t() + bs[gi][:C].double()
        ref[i, :n] = ((x[i, :n].double() + v) * s).float()
    Wp = torch.stack([L.pack_conv_weight(w.to(dv)) for w in ws]).contiguous()
    bp = torch.stack([L.pack_bias(b_.to(dv)) for b_ in bs]).contiguous()
    kw = dict(B=B, T=T, Cin=C, N=C, Np=Wp.shape[1], Kp=Wp.shape[2], lda=Lyr * C, a_bs=T * Lyr * C, lens=lens.to(dv), bias=bp, ldr=C, ldc=C,
              post_scale=s, mask_rows=True, group_size=(B // groups if groups > 1 else 0), w_gs=Wp[0].numel(), bias_gs=bp[0].numel())
    A = ga.to(dv)[:, :, C:]
    x1 = x.to(dv).clone()
    L.gemm16_res(A, Wp, x1, mt=mt, R=x1, **kw)
    err = (x1.cpu() - ref).abs().max().item()
    assert err <= 2e-5, err
    for i in range(B):
        assert torch.all(x1[i, int(lens[i]):] == 0)
    x2 = x.to(dv).clone()
    S = torch.zeros(B, T, C, device=dv)
    L.conv_gemm(A, Wp, x2, epi=L.EPI_RESSKIP, R=x2, Nh=C, C2=S, ldc2=C, c2_bs=T * C, tile=3, **kw)
    assert (x1 - x2).abs().max().item() <= 1e-5
    # the same launch with the weights in the kernel's fetch order (ss_gemm16_resw): bit-identical
    W16 = torch.stack([L.pack_gemm16_weights(Wp[i][:C].contiguous(), Wp.shape[2]) for i in range(groups)]).contiguous()
    x3 = x.to(dv).clone()
    L.gemm16_res(A, Wp, x3, mt=mt, R=x3, W16=W16, **dict(kw, w_gs=W16[0].numel()))
    assert torch.equal(x1, x3), (x1 - x3).abs().max().item()


@pytest.mark.parametrize("mt", [0, 4, 6, 8])
@pytest.mark.parametrize("K,N,B,T,groups,relu", [(1920, 192, 4, 150, 2, True), (5120, 256, 2, 333, 1, True), (96, 80, 1, 70, 1, False), (32, 64, 3, 17, 1, False)])
def test_gemm16_store_matches_torch(K, N, B, T, groups, relu, mt):
    """ss_gemm16_store (both operands streamed by LDS-DMA): C = act(A . W^T + bias) for the skip-GEMM shapes (K = L*C), odd and even
    chunk counts (K = 96 -> 3 chunks, 32 -> 1), N not a multiple of 64, ragged lens, grouped weights - vs torch float64."""
    import math
    from stylesinger_amd import lib as L
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(K + N + mt)
    A = torch.randn(B, T, K, generator=g)
    ws = [torch.randn(N, K, 1, generator=g) / math.sqrt(K) for _ in range(groups)]
    bs = [torch.randn(N, generator=g) * 0.1 for _ in range(groups)]
    lens = torch.tensor([max(1, T - 9 * i) for i in range(B)], dtype=torch.int32)
    ref = torch.zeros(B, T, N)
    for i in range(B):
        n, gi = int(lens[i]), i * groups // B
        v = A[i, :n].double() @ ws[gi][:, :, 0].double().t() + bs[gi].double()
        ref[i, :n] = (v.clamp_min(0) if relu else v).float()
    Wp = torch.stack([L.pack_conv_weight(w.to(dv)) for w in ws]).contiguous()
    bp = torch.stack([L.pack_bias(b_.to(dv)) for b_ in bs]).contiguous()
    out = torch.full((B, T, N), 3.0, device=dv)
    L.gemm16_store(A.to(dv), Wp, out, mt=mt, B=B, T=T, Cin=K, N=N, Np=Wp.shape[1], Kp=Wp.shape[2], lens=lens.to(dv), bias=bp,
                   act=L.ACT_RELU if relu else L.ACT_NONE, mask_rows=True, group_size=(B // groups if groups > 1 else 0), w_gs=Wp[0].numel(),
                   bias_gs=bp[0].numel())
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 3e-5, err
    for i in range(B):
        assert torch.all(out[i, int(lens[i]):] == 0)
    # the bf16x3 form (ss_gemm16x_store: operands as three bf16 terms, six exact bf16 MFMA products, fp32 accumulate): the same bar
    Wx = torch.stack([L.split3_gemm16_weights(Wp[i], Wp.shape[2]) for i in range(groups)]).contiguous()
    outx = torch.full((B, T, N), 5.0, device=dv)
    L.gemm16x_store(A.to(dv), Wp, Wx, outx, mt=mt, B=B, T=T, Cin=K, N=N, Np=Wp.shape[1], Kp=Wp.shape[2], lens=lens.to(dv), bias=bp,
                    act=L.ACT_RELU if relu else L.ACT_NONE, mask_rows=True, group_size=(B // groups if groups > 1 else 0), w_gs=Wx[0].numel(),
                    bias_gs=bp[0].numel())
    errx = (outx.cpu() - ref).abs().max().item()
    assert errx <= 3e-5, errx
    for i in range(B):
        assert torch.all(outx[i, int(lens[i]):] == 0)


@pytest.mark.parametrize("C,d,T,B", [(256, 1, 300, 2), (256, 2, 777, 3), (256, 4, 256, 1), (256, 8, 1100, 2), (192, 8, 530, 2)])
def test_bf16_gate256_kernel_matches_the_generic_bf16_kernel(C, d, T, B):
    """ss_gemm_bf16_gate256 (256x256 tiles, 8 waves, LDS-DMA, A staged once with its dilation halo) vs ss_gemm_bf16's generic GATE kernel on
    the same bf16 operands: ragged lens (zero padding by the DMA's range check, also for NEGATIVE rows of the halo), T not a multiple
    of 256, every dilation of the cycle, C = 192 (3 channel chunks, Np = 384 -> not a multiple of 256: rejected) - outputs are bf16
    gate values in (-1, 1): equal up to one bf16 ulp of the K-order difference (4e-3 abs), rows past lens exactly 0."""
    import math
    from stylesinger_amd import lib as L
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(C + d + T)
    x = torch.randn(B, T, C, generator=g).to(dv)
    lens = torch.tensor([max(1, T - 41 * i) for i in range(B)], dtype=torch.int32).to(dv)
    for b in range(B):
        x[b, int(lens[b]):] = 0
    xh = L.to_bf16(x)
    w = (torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C)).to(dv)
    Wh = L.to_bf16(L.pack_conv_weight(w, interleave_half=C))
    Np = Wh.shape[0]
    E = (torch.randn(B, T, 2 * Np, generator=g) * 0.5).to(dv)
    bias = (torch.randn(Np, generator=g) * 0.2).to(dv)
    kw = dict(B=B, T=T, K=C, taps=(-d, 0, d), N=C, Np=Np, epi=L.HEPI_GATE, lens=lens, E=E[:, :, Np:], lde=2 * Np, e_bs=T * 2 * Np, bias=bias)
    want = torch.full((B, T, C), 3.0, device=dv, dtype=torch.bfloat16)
    L.gemm_bf16(xh, Wh, out=want, **kw)
    if Np % 256 != 0:
        with pytest.raises(L.StyleSingerHipError):
            L.gemm_bf16(xh, Wh, out=torch.empty_like(want), gate256=True, **kw)
        return
    got = torch.full((B, T, C), 5.0, device=dv, dtype=torch.bfloat16)
    L.gemm_bf16(xh, Wh, out=got, gate256=True, **kw)
    err = (got.float() - want.float()).abs().max().item()
    assert err <= 4e-3, err
    assert (got.float() - want.float()).abs().mean().item() <= 2e-4
    for b in range(B):
        assert torch.all(got[b, int(lens[b]):] == 0)


@pytest.mark.parametrize("C,k,d", [(64, 3, 1), (64, 7, 3), (64, 11, 5), (128, 3, 5), (128, 11, 1), (256, 7, 1), (256, 3, 3)])
def test_wino43_conv_matches_torch_conv1d_and_the_direct_kernel(C, k, d):
    """ss_wino43_conv (grouped Winograd F(4,3): ceil(k/3) tap groups, six accumulators over all of them) vs torch fp32 conv1d on the same
    leaky-relu'd input and vs ss_conv_gemm: ragged lens (frames >= len read as zero padding and are written as 0), input leaky-relu,
    bias, in-place residual, post_scale + accumulate, leaky-relu of the output. Tolerance 5e-5 abs on O(1) outputs that are sums of up to
    11 x 256 products (measured: <= 2.1e-5 vs torch, the same as the direct kernel's distance; emulation in
    oracle/wino_vocoder_numerics.py: 2.7e-6 vs float64 at C = 32)."""
    import math
    import torch.nn.functional as F
    from stylesinger_amd import lib as L
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 * C + 10 * k + d)
    B, T = 3, 777
    lens_l = [T, T - 123, 41]
    x = torch.randn(B, T, C, generator=g)
    for b, n in enumerate(lens_l):
        x[b, n:] = 0
    w = torch.randn(C, C, k, generator=g) / math.sqrt(C * k)
    bias = torch.randn(C, generator=g) * 0.3
    res = torch.randn(B, T, C, generator=g)
    prev = torch.randn(B, T, C, generator=g)
    lens = torch.tensor(lens_l, dtype=torch.int32).to(dv)
    Ww = L.pack_conv_weight(L.wino43_group_weight(w.to(dv)))
    Wd = L.pack_conv_weight(w.to(dv))
    bp = L.pack_bias(bias.to(dv))
    xd = x.to(dv)
    kw = dict(B=B, T=T, Cin=C, N=C, Np=C, Kp=C, lens=lens, bias=bp, ldc=C, mask_rows=True)
    taps = [(j - (k - 1) // 2) * d for j in range(k)]

    def ref(slope_in, act_out, R, post, acc):
        outs = []
        for b, n in enumerate(lens_l):
            xi = F.leaky_relu(x[b:b + 1, :n], slope_in) if slope_in != 1.0 else x[b:b + 1, :n]
            y = F.conv1d(xi.transpose(1, 2), w, bias, padding=(k - 1) // 2 * d, dilation=d).transpose(1, 2)
            if act_out:
                y = F.leaky_relu(y, 0.1)
            if R is not None:
                y = y + R[b:b + 1, :n]
            y = y * post
            if acc is not None:
                y = y + acc[b:b + 1, :n]
            outs.append(F.pad(y, (0, 0, 0, T - n)))
        return torch.cat(outs)

    # (1) first conv of a ResBlock pair: input leaky-relu, output leaky-relu, no residual
    want = ref(0.1, True, None, 1.0, None)
    got = torch.full((B, T, C), 9.0, device=dv)
    L.wino43_conv(xd, Ww, got, k=k, dilation=d, a_lrelu=0.1, act=L.ACT_LRELU, act_slope=0.1, **kw)
    direct = torch.full((B, T, C), 7.0, device=dv)
    L.conv_gemm(xd, Wd, direct, taps=taps, a_lrelu=0.1, act=L.ACT_LRELU, act_slope=0.1, **kw)
    e1, e1d = (got.cpu() - want).abs().max().item(), (got - direct).abs().max().item()
    assert e1 <= 5e-5 and e1d <= 5e-5, (e1, e1d)
    for b, n in enumerate(lens_l):
        assert torch.all(got[b, n:] == 0)
    # (2) second conv: no input activation, residual IN PLACE (R == C), then (3) the accumulating form with post_scale
    want = ref(1.0, False, res, 1.0, None)
    buf = res.to(dv).clone()
    L.wino43_conv(xd, Ww, buf, k=k, dilation=d, R=buf, ldr=C, **kw)
    e2 = (buf.cpu() - want).abs().max().item()
    assert e2 <= 5e-5, e2
    want = ref(1.0, False, res, 1.0 / 3.0, prev)
    for b, n in enumerate(lens_l):
        want[b, n:] = 0
    acc = prev.to(dv).clone()
    L.wino43_conv(xd, Ww, acc, k=k, dilation=d, R=res.to(dv), ldr=C, post_scale=1.0 / 3.0, accumulate=True, **kw)
    e3 = (acc.cpu() - want).abs().max().item()
    assert e3 <= 5e-5, e3
    record_measurement(f"wino43_conv_C{C}_k{k}_d{d}", max_err_vs_torch=max(e1, e2, e3), max_err_vs_direct_kernel=e1d)


@pytest.mark.parametrize("mt", [2, 3])
def test_gate16x_split_bf16_products_are_fp32_grade(mt):
    """ss_wino43_gate16x ("bf16x3" mode: every fp32 product of the F(4,3) gate from operands split into three bf16 terms, six exact bf16
    MFMA products, fp32 accumulation) against the exact-fp32 16x16x4 kernel on the same inputs - mel shape (C = 256), grouped f0-pair
    shape (C = 192, bias, two weight sets), ragged lens, every dilation: the two agree to fp32 rounding (<= 2e-5 on gate outputs in
    (-1, 1); measured ~2e-6), rows past lens are written as 0."""
    import math
    from stylesinger_amd import lib as L
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(23)
    worst = 0.0
    for (B, T, C, grouped) in ((3, 700, 256, False), (4, 333, 192, True)):
        x = torch.randn(B, T, C, generator=g).to(dv)
        nw = 2 if grouped else 1
        ws = [torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C) for _ in range(nw)]
        Wt = torch.stack([L.pack_conv_weight(L.wino43_weight(w.to(dv)), interleave_half=C) for w in ws]).contiguous()
        Wx = torch.stack([L.split3_weights(Wt[i], C) for i in range(nw)]).contiguous()
        Np = Wt.shape[1]
        ab = torch.randn(nw, C, generator=g).to(dv)
        bias = (torch.randn(nw, Np, generator=g) * 0.3).to(dv) if grouped else None
        E = torch.randn(B, T, 2 * Np, generator=g).to(dv)
        lens = torch.tensor([T, T - 7, 5, T - 1][:B], dtype=torch.int32).to(dv)
        for d in (1, 2, 4, 8):
            kw = dict(dilation=d, B=B, T=T, Cin=C, N=C, Np=Np, Kp=C, lens=lens, a_bias=ab, bias=bias, E=E[:, :, Np:], lde=2 * Np,
                      e_bs=T * 2 * Np, ldc=C, mask_rows=True)
            kx = dict(kw)
            if grouped:
                kw.update(group_size=2, w_gs=Wt[0].numel(), bias_gs=Np, a_bias_gs=C)
                kx.update(group_size=2, w_gs=Wx[0].numel(), bias_gs=Np, a_bias_gs=C)
            want = torch.full((B, T, C), 7.0, device=dv)
            got = torch.full((B, T, C), 9.0, device=dv)
            L.wino43_gate16(x, Wt if grouped else Wt[0], want, mt=mt, **kw)
            L.wino43_gate16x(x, Wx if grouped else Wx[0], got, mt=mt, **kx)
            err = (got - want).abs().max().item()
            worst = max(worst, err)
            assert err <= 2e-5, (B, T, C, d, err)
            for i in range(B):
                assert torch.all(got[i, int(lens[i]):] == 0)
    record_measurement(f"gate16x_vs_exact_fp32_mt{mt}", max_abs_diff=worst)


def test_bf16x3_mode_matches_the_reference_golden_chain():
    """Whole path in the opt-in "bf16x3" precision mode (F(4,3) gates on the bf16 matrix cores from split operands, everything else as the
    fp32 mode) against the REAL reference's 100-step golden and its 1000-step golden: the same 1e-5 the fp32 mode is held to (CPU emulation
    in oracle/bf16x3_numerics.py: 3.1e-7 / 3.4e-7, indistinguishable from fp32)."""
    from oracle import harness
    dev = torch.device("cuda:0")
    out = {}
    for name in ("acoustic_t64_s100", "acoustic_t32_mel1000"):
        case = harness.load_case(name)
        meta, gold = case["meta"], case["out"]
        hp, sd, batch = harness.case_setup(meta)
        hp = dict(hp, mfma_precision="bf16x3")
        model = StyleSingerHIP(None, hparams=hp)
        assert model.x3
        model.load_state_dict(sd, strict=True)
        model.eval().to(dev)
        noise = synth.draw_acoustic_noise(synth.NoiseTape(meta["tape_seed"]), meta["B"], meta["T"], meta["steps_f0"], meta["steps_mel"])
        b = {k: v.to(dev) for k, v in batch.items()}
        ret = _fwd(model, b, noise=noise)
        assert model._pk["mel"]["net"].mfma_x3 == 1
        l1 = (ret["mel_out"].cpu() - gold["mel_out"]).abs().mean().item()
        mx = (ret["mel_out"].cpu() - gold["mel_out"]).abs().max().item()
        flips = ((ret["pitch_pred"][..., 1].cpu() > 0) != (gold["pitch_pred"][..., 1] > 0)).float().mean().item()
        out[name] = dict(mel_l1=l1, mel_max=mx, voicing_flips=flips)
        assert flips == 0.0 and l1 <= 1e-5, (name, l1, mx, flips)
    record_measurement("bf16x3_mode_vs_reference_goldens", **{f"{k}_{kk}": vv for k, v in out.items() for kk, vv in v.items()})
