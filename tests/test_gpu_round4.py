"""Round-4 GPU tests: the DDIM entry point pinned to the REAL reference (eta = 1 / stride 1 == p_sample), fp32 waveform parity at
realistic lengths (T = 1500 and T = 5625 frames: 384 000 / 1 440 000 samples of NSF phase integration + the 4-stage generator), the
vocoder's direct-kernel fallback for items beyond 32-bit offsets, and the RCCL collective driven once on this 1-GPU box."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import record_measurement  # noqa: E402
from oracle import harness  # noqa: E402
from oracle import restatement as R  # noqa: E402
from stylesinger_amd import config, synth  # noqa: E402
from stylesinger_amd import lib as L  # noqa: E402
from stylesinger_amd.model import StyleSingerHIP  # noqa: E402
from stylesinger_amd.vocoder import HifiGAN  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WAV_TOL = 1e-5


def _fwd(model, b, **kw):
    return model(b["txt_tokens"], mel2ph=b.get("mel2ph"), spk_embed=b["spk_embed"], emo_embed=b["emo_embed"], ref_mels=b["ref_mels"],
                 ref_f0=b["ref_f0"], global_steps=320000, infer=True, note=b["note"], note_dur=b["note_dur"], note_type=b["note_type"], **kw)


def test_ddim_eta1_stride1_reproduces_the_reference_ancestral_chain():
    """ss_meldiff_sample_ddim(eta = 1, ts = K-1 ... 0) against the REAL reference's 100-step golden: the DDIM update with eps recomputed
    from the clamped x0 is GaussianDiffusion.p_sample (/root/reference/modules/diff/shallow_diffusion_tts.py:136-162), so the sampler
    entry point of BASELINE config 5 is pinned to the reference here (the eta = 0 form shares every line of it but sigma)."""
    case = harness.load_case("acoustic_t64_s100")
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    K = hp["K_step"]
    model = StyleSingerHIP(None, hparams=hp)
    model.load_state_dict(sd, strict=True)
    model.eval().to("cuda:0")
    noise = synth.draw_acoustic_noise(synth.NoiseTape(meta["tape_seed"]), meta["B"], meta["T"], meta["steps_f0"], meta["steps_mel"])
    b = {k: v.cuda() for k, v in batch.items()}
    got = _fwd(model, b, noise=noise, sampler="ddim", ddim_steps=K, eta=1.0)
    assert model.ddim_timesteps(K) == list(range(K - 1, -1, -1))
    d = (got["mel_out"].cpu() - gold["mel_out"]).abs()
    ddpm = _fwd(model, b, noise=noise)
    d2 = (got["mel_out"] - ddpm["mel_out"]).abs()
    print(f"ddim eta=1 stride 1 vs the reference golden: mel L1 {d.mean().item():.3e} max {d.max().item():.3e}; vs the DDPM entry point {d2.mean().item():.3e}")
    record_measurement("ddim_eta1_vs_reference_golden_t64_s100", mel_l1=d.mean().item(), mel_max=d.max().item(), vs_ddpm_entry_l1=d2.mean().item())
    assert d.mean().item() <= 1e-5 and d.max().item() <= 2e-4
    # eta = 0 on the same inputs is deterministic: a second call with another Philox seed gives the same mel
    a0 = _fwd(model, b, noise=noise, sampler="ddim", ddim_steps=10)["mel_out"]
    a1 = _fwd(model, b, noise=noise, sampler="ddim", ddim_steps=10, seed=999)["mel_out"]
    assert torch.equal(a0, a1)


def _voc_inputs(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    mel = (torch.randn(B, T, 80, generator=g) * 0.8 - 3.0).clamp(-6, 1.5)
    f0 = torch.stack([synth.synth_f0_hz(i, T, seed, dtype=torch.float32) for i in range(B)])
    return mel, f0


@pytest.mark.parametrize("T", [1500, 5625])
def test_fp32_waveform_matches_oracle_at_full_length(T):
    """north_star: 'match the reference PyTorch mel AND waveform'. The whole generator (hifigan_nsf.py:144-169: conv_pre, 4 x (transposed conv
    + noise conv + 3 ResBlocks on the grouped-Winograd kernel), conv_post, tanh) and the NSF source (source.py:342-441: phase integrated over
    T * 256 samples) at one BASELINE configs[1] item (T = 1500) and one configs[3] item (T = 5625), through StyleSingerInfer.vocode, against
    the pinned restatement on this box with the same noise tape."""
    from stylesinger_amd.infer import StyleSingerInfer
    cfg = config.make_vocoder_config()
    vsd = synth.synth_vocoder_state_dict(cfg, 77)
    hp = config.make_hparams()
    voc = HifiGAN(cfg, vsd, device="cuda:0")
    mel, f0 = _voc_inputs(1, T, 4000 + T)
    noise = synth.draw_vocoder_noise(synth.NoiseTape(90 + T), 1, T * 256)
    inf = StyleSingerInfer.__new__(StyleSingerInfer)
    inf.hparams, inf.vocoder = hp, voc
    lens = torch.tensor([T], dtype=torch.int32).cuda()
    wav = inf.vocode(mel.cuda(), f0.cuda(), lens, noise=noise)
    _, har = voc.model(mel.cuda().clamp(hp["mel_vmin"], hp["mel_vmax"]), f0.cuda(), noise=noise, return_source=True)
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref_wav, ref_har = R.hifigan_forward(vsd, cfg, mel.clamp(hp["mel_vmin"], hp["mel_vmax"]), f0, synth.NoiseTape(90 + T))
    e_h = (har.cpu() - ref_har).abs().max().item()
    dw = (wav.cpu() - ref_wav).abs()
    print(f"vocoder T={T} ({T * 256} samples): har max err {e_h:.3e}; wav max err {dw.max().item():.3e} mean {dw.mean().item():.3e}; |wav| max {ref_wav.abs().max().item():.3f}")
    record_measurement(f"vocoder_fp32_t{T}_vs_oracle", wav_max_err=dw.max().item(), wav_mean_err=dw.mean().item(), har_max_err=e_h, samples=T * 256)
    assert wav.shape == (1, T * 256)
    assert e_h <= 2e-5, e_h        # phase is integrated over up to 1.44 M samples (golden-size measurement: 3e-8)
    assert dw.max().item() <= WAV_TOL, dw.max().item()


def test_vocoder_direct_kernel_fallback_for_items_beyond_32bit_offsets():
    """hifigan.hip: a stage panel of >= 2 GiB per item cannot be addressed by the grouped-Winograd conv's 32-bit offsets and takes the direct
    conv kernel. The 'voc_wino_max_mb' knob lowers that limit so the branch runs at test size: every ResBlock conv then goes the fallback
    way, and the waveform still matches the oracle and the Winograd path."""
    cfg = config.make_vocoder_config()
    vsd = synth.synth_vocoder_state_dict(cfg, 78)
    voc = HifiGAN(cfg, vsd, device="cuda:0")
    B, T = 2, 150
    mel, f0 = _voc_inputs(B, T, 5150)
    noise = synth.draw_vocoder_noise(synth.NoiseTape(91), B, T * 256)
    lens = torch.tensor([T, T - 31], dtype=torch.int32).cuda()
    lib = L.load()
    assert lib.ss_get_tuning(b"voc_wino_max_mb") == 2048
    wav_w = voc.model(mel.cuda(), f0.cuda(), lens=lens, noise=noise)
    try:
        L.check(lib.ss_set_tuning(b"voc_wino_max_mb", 1), "knob")
        wav_d = voc.model(mel.cuda(), f0.cuda(), lens=lens, noise=noise)
    finally:
        L.check(lib.ss_set_tuning(b"voc_wino_max_mb", 2048), "knob")
    assert not torch.equal(wav_w, wav_d), "the knob must have switched kernels (Winograd and direct forms round differently)"
    with torch.no_grad():
        ref1, _ = R.hifigan_forward(vsd, cfg, mel[1:, :T - 31], f0[1:, :T - 31], _ItemTape(noise, 1, (T - 31) * 256))
        ref0, _ = R.hifigan_forward(vsd, cfg, mel[:1], f0[:1], _ItemTape(noise, 0, T * 256))
    for name, w in (("winograd", wav_w), ("direct fallback", wav_d)):
        e0 = (w[0].cpu() - ref0[0]).abs().max().item()
        e1 = (w[1, :(T - 31) * 256].cpu() - ref1[0]).abs().max().item()
        print(f"{name}: wav max err item0 {e0:.3e} item1 (ragged) {e1:.3e}")
        assert max(e0, e1) <= WAV_TOL
    assert lib.ss_set_tuning(b"voc_wino_max_mb", 0) != 0 and lib.ss_get_tuning(b"no_such_knob") < 0


class _ItemTape:
    """Replays item `i` of a pre-drawn vocoder noise dict in the order hifigan_forward draws (rand_ini, sine noise, unused source noise)."""
    def __init__(self, noise, i, n):
        self.q = [noise["rand_ini"][i:i + 1], noise["sine_noise"][i:i + 1, :n], torch.zeros(1, n, 1)]

    def rand(self, *shape):
        return self.q.pop(0).clone()

    def randn(self, *shape):
        return self.q.pop(0).clone()


_RCCL_CHILD = r"""
import os, sys, torch, torch.distributed as td
sys.path.insert(0, sys.argv[1])
os.environ["SS_FORCE_COLLECTIVE"] = "1"
from stylesinger_amd import dist as D
torch.cuda.set_device(0)
td.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % int(sys.argv[2]), world_size=1, rank=0, device_id=torch.device("cuda:0"))
assert td.get_backend() == "nccl" and D._is_dist()
g = torch.Generator().manual_seed(3)
Bl, T, M = 8, 1500, 80
mel = torch.randn(Bl, T, M, generator=g).cuda(); f0 = (torch.rand(Bl, T, generator=g) * 400).cuda()
lens = torch.tensor([1500, 1499, 1, 0, 777, 1500, 64, 1023], dtype=torch.int32).cuda()
side = torch.cuda.Stream()
with torch.cuda.stream(side):      # a step stream other than the default one, as bench.py --streams 3 uses
    m, f, l = D.gather_mels(mel, f0, lens)
side.synchronize()
assert m.shape == (Bl, T, M) and torch.equal(m, mel) and torch.equal(f, f0) and torch.equal(l, lens) and l.dtype == torch.int32
assert D.global_max_int(1234) == 1234
res = D.run_sharded(lambda items: (mel[:len(items)], f0[:len(items)], lens[:len(items)]), lambda a, b, c: a.sum(-1), list(range(Bl)), 0, 1, T)
assert torch.equal(res["mel_all"], mel) and torch.equal(res["lens_all"], lens)
torch.cuda.synchronize()
td.destroy_process_group()
print("RCCL_OK", torch.cuda.get_device_name(0))
"""


def test_rccl_all_gather_branch_runs_on_this_box(tmp_path):
    """the data path's ONE collective. No multi-GPU node is available to these tests, so the RCCL branch of dist.gather_mels
    (`all_gather_into_tensor` on the communication stream, backend "nccl") is driven with a process group of one rank on cuda:0
    (SS_FORCE_COLLECTIVE=1 lifts the W = 1 short cut): payload packing, the bit-cast lengths, stream ordering and the gathered order must give back
    the inputs exactly. The multi-rank ordering logic is covered by the gloo world-size-2 tests (tests/test_dist_cpu.py)."""
    script = tmp_path / "rccl_child.py"
    script.write_text(_RCCL_CHILD)
    port = 29500 + os.getpid() % 400
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, str(script), ROOT, str(port)], capture_output=True, text=True, timeout=300, env=env)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "RCCL_OK" in r.stdout
    record_measurement("rccl_world1_all_gather", ok=True, payload_bytes=8 * 1500 * 82 * 4)


# ------------------------------------------------------------------------------------------------------------------------------
# "bf16x2" precision: BASELINE configs[3] ("bf16 MFMA") at fp32-grade parity - operands as (hi, mid) bf16 pairs, three products
# ------------------------------------------------------------------------------------------------------------------------------
def _split_ref(x):
    hi = x.to(torch.bfloat16).float()
    return hi, (x - hi).to(torch.bfloat16).float()


@pytest.mark.parametrize("T,K,force256", [(200, 256, False), (333, 192, False), (5600, 256, False), (5600, 256, True), (777, 256, True)])
def test_gemm_bf16_split_operands_match_float64_of_the_same_three_products(T, K, force256):
    """ss_gemm_bf16 with split = 1 (A and W as (hi, mid) bf16 pairs in one row; hi*hi + hi*mid + mid*hi, fp32 accumulate) against float64
    math on the SAME split terms: GATE (3-tap dilated conv + addend, outputs written as (hi, mid) pairs), RESX (fp32 residual stream + the next
    layer's split operand) and STORE. Also: the split outputs reconstruct the fp32 value to 2^-16, and the result is fp32-grade against the
    UNSPLIT fp32 operands (the point of the mode). force256: the 256x256 LDS-DMA kernel (ss_gemm_bf16_gate256, many-round launches)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(T + K + 1)
    B, C = 3, K
    if force256 and not hasattr(L.load(), "ss_gemm_bf16_gate256"):
        pytest.skip("no gate256")
    lens = torch.tensor([T, T - 37, 5], dtype=torch.int32, device=dev)
    x = torch.randn(B, T, C, generator=g).to(dev) * 3.0
    for b in range(B):
        x[b, lens[b]:] = 0
    xs = L.split_bf16(x)                                   # [B,T,2C]: (hi, mid) pairs interleaved by 32 channels
    xh, xm = _split_ref(x)
    ph, pm = L.split_planes(xs)
    assert torch.equal(ph, xh) and torch.equal(pm, xm)
    assert torch.equal(xs[..., 64:96].float(), xh[..., 32:64]) and torch.equal(xs[..., 32:64].float(), xm[..., :32])   # the layout itself
    assert (xh + xm - x).abs().max().item() <= 3.0 * 6 * 2.0 ** -17
    d = 4
    w = (torch.randn(2 * C, C, 3, generator=g) / (3 * C) ** 0.5).to(dev)
    Wp = L.pack_conv_weight(w, interleave_half=C)          # [2C][3*Kp] gate-interleaved, fp32
    Ws = L.split_bf16(Wp)                                  # [2C][2*3*Kp]
    wh, wm = _split_ref(w)
    conv = lambda a, ww: torch.nn.functional.conv1d(a.double().transpose(1, 2), ww.double(), padding=d, dilation=d).transpose(1, 2)
    y3 = conv(xm, wh) + conv(xh, wm) + conv(xh, wh)        # the three products, float64
    y_exact = conv(x, w)
    E = torch.randn(B, T, 2 * C, generator=g).to(dev) * 0.5
    Ep = torch.empty_like(E)
    for p in range(C // 32):
        Ep[..., 64 * p:64 * p + 32] = E[..., 32 * p:32 * p + 32]
        Ep[..., 64 * p + 32:64 * p + 64] = E[..., C + 32 * p:C + 32 * p + 32]
    Lyr = 2                                                # the output lands in layer slot 1 (physical columns [2C, 4C)) of a [rows][2 * Lyr * C] buffer
    GA = torch.full((B, T, 2 * Lyr * C), 7.0, device=dev, dtype=torch.bfloat16)
    L.gemm_bf16(xs, Ws, B=B, T=T, K=C, taps=(-d, 0, d), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens, E=Ep, lde=2 * C, out=GA[..., 2 * C:],
                ldc=2 * Lyr * C, c_bs=T * 2 * Lyr * C, lda=2 * C, split=1, gate256=force256)
    z = y3 + E.double()
    g_ref = (torch.sigmoid(z[..., :C]) * torch.tanh(z[..., C:])).float()
    z2 = y_exact + E.double()
    g_exact = (torch.sigmoid(z2[..., :C]) * torch.tanh(z2[..., C:])).float()
    for b in range(B):
        g_ref[b, lens[b]:] = 0
        g_exact[b, lens[b]:] = 0
    gah, gam = L.split_planes(GA)                          # logical [B,T,Lyr*C] planes
    got = gah[..., C:] + gam[..., C:]
    e3, ex = (got - g_ref).abs().max().item(), (got - g_exact).abs().max().item()
    print(f"split GATE T={T} K={K} gate256={force256}: vs float64 of the 3 products {e3:.2e}, vs exact operands {ex:.2e}")
    assert e3 <= 2e-5 and ex <= 2e-4, (e3, ex)             # e3: (hi, mid) output pair = 2^-17 of values in (-1, 1) + hardware exp/rcp; ex: + the dropped mid*mid terms of |x| <= 12 sums over K = 768 (plain bf16 operands: ~1e-2)
    assert torch.all(GA[..., :2 * C].float() == 7.0), "the neighbouring layer slot is untouched"
    # RESX on the layer-slot operand: x <- (x + G . Wo^T + b) / sqrt(2); Y = split(x + next_bias)
    wo = (torch.randn(C, C, 1, generator=g) / C ** 0.5).to(dev)
    Wos = L.split_bf16(L.pack_conv_weight(wo))
    bo = (torch.randn(C, generator=g) * 0.1).to(dev)
    nb = torch.randn(C, generator=g).to(dev)
    X = torch.randn(B, T, C, generator=g).to(dev)
    X0 = X.clone()
    Y = torch.empty(B, T, 2 * C, device=dev, dtype=torch.bfloat16)
    L.gemm_bf16(GA[..., 2 * C:], Wos, B=B, T=T, K=C, taps=(0,), N=C, Np=Wos.shape[0], epi=L.HEPI_RESX, lens=lens, bias=L.pack_bias(bo), X=X,
                post_scale=0.5 ** 0.5, next_bias=nb, Y=Y, lda=2 * Lyr * C, a_bs=T * 2 * Lyr * C, split=1)
    gh, gm = gah[..., C:].double(), gam[..., C:].double()
    woh, wom = (t.double() for t in _split_ref(wo[:, :, 0]))
    x_ref = ((X0.double() + (gm @ woh.t() + gh @ wom.t() + gh @ woh.t() + bo.double())) * (0.5 ** 0.5)).float()
    for b in range(B):
        x_ref[b, lens[b]:] = 0
    assert (X - x_ref).abs().max().item() <= 4e-6 * (C ** 0.5)
    y_ref = x_ref + nb
    for b in range(B):
        y_ref[b, lens[b]:] = 0
    yh, ym = L.split_planes(Y)
    assert ((yh + ym) - y_ref).abs().max().item() <= 1e-4
    # the pair-only residual stream (X = NULL): Y holds x + cur_bias as a pair, is read, updated and rewritten in place
    cb = torch.randn(C, generator=g).to(dev)
    Yp = L.split_bf16(X0 + cb)
    for b in range(B):
        Yp[b, lens[b]:] = 0
    y0h, y0m = L.split_planes(Yp)
    L.gemm_bf16(GA[..., 2 * C:], Wos, B=B, T=T, K=C, taps=(0,), N=C, Np=Wos.shape[0], epi=L.HEPI_RESX, lens=lens, bias=L.pack_bias(bo), X=None,
                post_scale=0.5 ** 0.5, next_bias=nb, Y=Yp, lda=2 * Lyr * C, a_bs=T * 2 * Lyr * C, split=1, cur_bias=cb,
                gate256=force256)   # force256: the 256-row LDS-DMA kernel (ss_gemm_bf16_tile256) instead of the generic one
    x_in = (y0h + y0m) - cb
    xp_ref = ((x_in.double() + (gm @ woh.t() + gh @ wom.t() + gh @ woh.t() + bo.double())) * (0.5 ** 0.5)).float() + nb
    for b in range(B):
        xp_ref[b, lens[b]:] = 0
    y1h, y1m = L.split_planes(Yp)
    assert ((y1h + y1m) - xp_ref).abs().max().item() <= 1e-4
    # STORE with ReLU on the K = Lyr * C (skip-GEMM form) operand
    w2 = (torch.randn(C, Lyr * C, 1, generator=g) / (Lyr * C) ** 0.5).to(dev)
    S = torch.empty(B, T, C, device=dev)
    GA[..., :2 * C] = L.split_bf16(torch.randn(B, T, C, generator=g).to(dev))   # fill layer slot 0 with real operands
    L.gemm_bf16(GA, L.split_bf16(L.pack_conv_weight(w2)), B=B, T=T, K=Lyr * C, taps=(0,), N=C, Np=L.round_up(C, 32), epi=L.HEPI_STORE,
                lens=lens, act=L.ACT_RELU, out=S, lda=2 * Lyr * C, split=1, bias=L.pack_bias(bo), gate256=force256)
    ah, am = (t.double() for t in L.split_planes(GA))
    w2h, w2m = (t.double() for t in _split_ref(w2[:, :, 0]))
    s_ref = torch.relu(am @ w2h.t() + ah @ w2m.t() + ah @ w2h.t() + bo.double()).float()
    for b in range(B):
        s_ref[b, lens[b]:] = 0
    assert (S - s_ref).abs().max().item() <= 2e-5 * (Lyr * C) ** 0.5


def _model(hp, sd):
    m = StyleSingerHIP(None, hparams=hp)
    m.load_state_dict(sd)
    m.eval().to("cuda:0")
    return m


def test_bf16x2_mode_meets_north_star_on_the_1000_step_golden():
    """BASELINE configs[3] AS SPECIFIED - 1000 mel diffusion steps on the bf16 matrix cores - against the REAL reference's fp32 golden
    `acoustic_t32_mel1000` (/root/reference/modules/diff/shallow_diffusion_tts.py:99-162, coefficients up to 3e6): plain bf16 operands end
    2.5e-3 away (tests/test_gpu_round2.py), the split-operand mode must meet north_star's mel L1 <= 1e-4 - asserted 5x tighter."""
    case = harness.load_case("acoustic_t32_mel1000")
    meta, gold = case["meta"], case["out"]
    hp, sd, batch = harness.case_setup(meta)
    b = {k: v.cuda() for k, v in batch.items()}
    noise = synth.draw_acoustic_noise(synth.NoiseTape(meta["tape_seed"]), meta["B"], meta["T"], meta["steps_f0"], meta["steps_mel"])
    m = _model(dict(hp, mfma_precision="bf16x2"), sd)
    assert m.bf16 and m.split and m.bf16_hbm and m.fold_skip and not m.use_wino
    got = _fwd(m, b, noise=noise)
    d = (got["mel_out"].cpu() - gold["mel_out"]).abs()
    uv = ((got["pitch_pred"][..., 1].cpu() > 0) != (gold["pitch_pred"][..., 1] > 0)).sum().item()
    f0e = (got["f0_denorm"].cpu() - gold["f0_denorm"]).abs().max().item()
    print(f"bf16x2 mode, 1000-step golden of the real reference: mel L1 {d.mean().item():.3e} max {d.max().item():.3e}; voicing flips {uv}; f0 max err {f0e:.3e} Hz")
    record_measurement("c4_bf16x2_t32_1000steps_vs_fp32_reference", mel_l1=d.mean().item(), mel_max=d.max().item(), voicing_flips=uv,
                       f0_max_err_hz=f0e, pinned=True, north_star=1e-4)
    assert uv == 0
    assert d.mean().item() <= 2e-5, d.mean().item()
    # the 100-step goldens too (the f0 denoisers run in this mode as well): T = 64 and T = 300 frames of the real reference
    for name in ("acoustic_t64_s100", "acoustic_t300_s100"):
        case = harness.load_case(name)
        meta, gold = case["meta"], case["out"]
        hp, sd, batch = harness.case_setup(meta)
        noise = synth.draw_acoustic_noise(synth.NoiseTape(meta["tape_seed"]), meta["B"], meta["T"], meta["steps_f0"], meta["steps_mel"])
        got = _fwd(_model(dict(hp, mfma_precision="bf16x2"), sd), {k: v.cuda() for k, v in batch.items()}, noise=noise)
        d = (got["mel_out"].cpu() - gold["mel_out"]).abs()
        uv = ((got["pitch_pred"][..., 1].cpu() > 0) != (gold["pitch_pred"][..., 1] > 0)).sum().item()
        print(f"bf16x2 mode, {name}: mel L1 {d.mean().item():.3e} max {d.max().item():.3e}; voicing flips {uv}")
        record_measurement(f"bf16x2_{name.split('_', 1)[1]}_vs_fp32_reference", mel_l1=d.mean().item(), mel_max=d.max().item(), voicing_flips=uv)
        assert uv == 0 and d.mean().item() <= 2e-5


def test_bf16x2_mode_at_the_c4_shape_matches_the_fp32_oracle():
    """The C4 SHAPE (one 30 s item, T = 5625: the 256x256 LDS-DMA gate kernel and the 128-row tiles run here) with 100 + 2 x 100 step chains in
    bf16x2 mode against the fp32 oracle on this box - mel L1 <= 1e-4 asserted 5x tighter, voicing decisions identical."""
    S = 100
    hp = config.make_hparams(dict(timesteps=S, K_step=S, f0_timesteps=S))
    B, T, Tp, Tr = 1, 5625, 105, 1500
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 2025)
    sd = synth.synth_acoustic_state_dict(hp, 2025)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(78), B, T, S, S)
    got = _fwd(_model(dict(hp, mfma_precision="bf16x2"), sd), {k: v.cuda() for k, v in batch.items()}, noise=noise)
    torch.cuda.synchronize()
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(78), mel2ph=batch["mel2ph"])
    flips = (got["uv_a"].cpu().long() != ref["uv_a"]).sum().item() + (got["uv_b"].cpu().long() != ref["uv_b"]).sum().item()
    cf = (got["pitch_coarse"].cpu() != ref["pitch_coarse"])
    dm = (got["mel_out"].cpu() - ref["mel_out"]).abs()
    print(f"bf16x2 at T=5625, 100+100+100 steps vs the fp32 oracle: mel L1 {dm.mean().item():.3e} max {dm.max().item():.3e}; voicing flips {flips}; coarse flips {int(cf.sum())}")
    record_measurement("c4_shape_t5625_100steps_bf16x2_vs_fp32_oracle", mel_l1=dm.mean().item(), mel_max=dm.max().item(), voicing_flips=flips,
                       coarse_flips=int(cf.sum()))
    assert flips == 0 and cf.float().mean().item() <= 1e-3
    assert dm.mean().item() <= 2e-5


# ------------------------------------------------------------------------------------------------------------------------------
# the B = 1 latency shape (inference/StyleSinger.py:175-186 runs ONE utterance): small-launch tilings
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N,B,T,groups,ksplit", [(5120, 256, 1, 750, 1, 5), (1920, 192, 2, 333, 2, 3), (5120, 256, 1, 64, 1, 16), (96, 80, 1, 70, 1, 3)])
def test_gemm16_store_splitk_matches_torch_and_is_deterministic(K, N, B, T, groups, ksplit):
    """ss_gemm16_store_splitk: K split over workgroup slices + fixed-order reduction (bias, ReLU, row mask in the reduction) vs torch float64
    and vs the unsplit kernel; two runs give identical bits (hipGraph replay == eager depends on it)."""
    import math
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(K + N + ksplit)
    A = torch.randn(B, T, K, generator=g)
    ws = [torch.randn(N, K, 1, generator=g) / math.sqrt(K) for _ in range(groups)]
    bs = [torch.randn(N, generator=g) * 0.1 for _ in range(groups)]
    lens = torch.tensor([max(1, T - 9 * i) for i in range(B)], dtype=torch.int32)
    ref = torch.zeros(B, T, N)
    for i in range(B):
        n, gi = int(lens[i]), i * groups // B
        ref[i, :n] = (A[i, :n].double() @ ws[gi][:, :, 0].double().t() + bs[gi].double()).clamp_min(0).float()
    Wp = torch.stack([L.pack_conv_weight(w.to(dv)) for w in ws]).contiguous()
    bp = torch.stack([L.pack_bias(b_.to(dv)) for b_ in bs]).contiguous()
    kw = dict(B=B, T=T, Cin=K, N=N, Np=Wp.shape[1], Kp=Wp.shape[2], lens=lens.to(dv), bias=bp, act=L.ACT_RELU, mask_rows=True,
              group_size=(B // groups if groups > 1 else 0), w_gs=Wp[0].numel(), bias_gs=bp[0].numel())
    outs = []
    for _ in range(2):
        out = torch.full((B, T, N), 3.0, device=dv)
        L.gemm16_store_splitk(A.to(dv), Wp, out, ksplit=ksplit, **kw)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    one = torch.full((B, T, N), 4.0, device=dv)
    L.gemm16_store(A.to(dv), Wp, one, **kw)
    err, err1 = (outs[0].cpu() - ref).abs().max().item(), (outs[0] - one).abs().max().item()
    assert err <= 3e-5 and err1 <= 3e-5, (err, err1)
    for i in range(B):
        assert torch.all(outs[0][i, int(lens[i]):] == 0)
    lib = L.load()
    assert lib.ss_gemm16_ksplit_pick(1, 750, 256, 5120) == 5 and lib.ss_gemm16_ksplit_pick(8, 1500, 256, 5120) == 1
    assert lib.ss_gemm16_ksplit_pick(2, 750, 192, 1920) >= 2


def test_one_utterance_latency_shape_matches_the_oracle():
    """B = 1, T = 750 (BASELINE configs[0]'s shape: one 4 s utterance - what inference/StyleSinger.py:175-186 runs) through the small-launch
    tilings (16-quad gate tiles, split-K skip GEMM), 20 + 2 x 20 steps, against the oracle on this box; graph replay == eager bit for bit."""
    S = 20
    hp = config.make_hparams(dict(timesteps=S, K_step=S, f0_timesteps=S))
    B, T, Tp, Tr = 1, 750, 14, 750
    batch = synth.synth_batch(B, T, Tp, Tr, hp, 31)
    sd = synth.synth_acoustic_state_dict(hp, 31)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(32), B, T, S, S)
    m = _model(hp, sd)
    b = {k: v.cuda() for k, v in batch.items()}
    got = _fwd(m, b, noise=noise)
    with torch.no_grad():
        ref = R.acoustic_forward(sd, hp, batch, synth.NoiseTape(32), mel2ph=batch["mel2ph"])
    dm = (got["mel_out"].cpu() - ref["mel_out"]).abs()
    flips = (got["uv_a"].cpu().long() != ref["uv_a"]).sum().item() + (got["uv_b"].cpu().long() != ref["uv_b"]).sum().item()
    print(f"B=1 T=750, {S} steps vs oracle: mel L1 {dm.mean().item():.3e} max {dm.max().item():.3e}; voicing flips {flips}")
    record_measurement("c1_shape_b1_t750_vs_oracle", mel_l1=dm.mean().item(), mel_max=dm.max().item(), voicing_flips=flips)
    assert flips == 0 and dm.mean().item() <= 1e-5
    m.use_graphs = "on"
    g1 = _fwd(m, b, seed=5)["mel_out"]
    g2 = _fwd(m, b, seed=5)["mel_out"]
    m.use_graphs = "off"
    e1 = _fwd(m, b, seed=5)["mel_out"]
    assert torch.equal(g1, g2) and torch.equal(g1, e1)


@pytest.mark.parametrize("mt", [1, 2, 3])
def test_gate16_with_the_addend_in_fetch_order_is_bit_identical(mt):
    """ss_gate16_tile_addend + ss_conv_gemm_args.e_tiled: the conditioner addend re-laid in the 16x16x4 gate kernel's fetch order (1 KB
    contiguous per wave instruction instead of 8 lines x 32 B) feeds the SAME values into the same arithmetic - outputs equal the row-major
    form bit for bit, for the mel shape and the grouped f0-pair shape, every dilation of the cycle, ragged lens (zero-filled tile tails)."""
    import math
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 + mt)
    for (B, T, C, Lyr, groups) in ((3, 333, 256, 3, 1), (4, 150, 192, 2, 2)):
        x = torch.randn(B, T, C, generator=g).to(dv)
        ws = [torch.randn(2 * C, C, 3, generator=g) / math.sqrt(3 * C) for _ in range(groups)]
        Wt = torch.stack([L.pack_conv_weight(L.wino43_weight(w.to(dv)), interleave_half=C) for w in ws]).contiguous()
        Np = Wt.shape[1]
        W16 = torch.stack([L.pack_gate16_weights(Wt[i], C) for i in range(groups)]).contiguous()
        ab = torch.randn(groups, C, generator=g).to(dv)
        E = torch.randn(B, T, Lyr * Np, generator=g).to(dv)
        lens = torch.tensor([T, T - 5, 1, T - 40][:B], dtype=torch.int32).to(dv)
        for d in (1, 2, 4, 8):
            kw = dict(dilation=d, B=B, T=T, Cin=C, N=C, Np=Np, Kp=C, lens=lens, a_bias=ab, ldc=C, mask_rows=True, W16=W16,
                      group_size=(B // groups if groups > 1 else 0), w_gs=W16[0].numel(), a_bias_gs=C)
            want = torch.full((B, T, C), 7.0, device=dv)
            got = torch.full((B, T, C), 9.0, device=dv)
            L.wino43_gate16(x, Wt, want, mt=mt, E=E[:, :, Np:], lde=Lyr * Np, e_bs=T * Lyr * Np, **kw)
            E16 = L.gate16_tile_addend(E[:, :, Np:], B=B, T=T, Np=Np, lde=Lyr * Np, e_bs=T * Lyr * Np, dilation=d, mt=mt)
            L.wino43_gate16(x, Wt, got, mt=mt, E=E16, e_tiled=True, **kw)
            assert torch.equal(got, want), (mt, C, d, (got - want).abs().max().item())


def test_small_launch_tail_kernels_agree_with_the_matrix_core_launches():
    """`mel_tail` knob A/B at the B = 1 shape: output projection + DDPM update + next input projection as ONE VALU launch (mel_tail_kernel, with the
    split-K slices of the skip GEMM added inside) vs the two matrix-core launches + reduction launch. Same exact-fp32 products, another summation
    order: mel within 1e-6, integer outputs identical; the knob really switches paths (the two results differ in the last bits)."""
    S = 12
    hp = config.make_hparams(dict(timesteps=S, K_step=S, f0_timesteps=3))
    B, T = 1, 333
    batch = synth.synth_batch(B, T, 8, 200, hp, 71)
    sd = synth.synth_acoustic_state_dict(hp, 71)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(72), B, T, 3, S)
    m = _model(hp, sd)
    b = {k: v.cuda() for k, v in batch.items()}
    lib = L.load()
    assert lib.ss_get_tuning(b"mel_tail") == 1
    fused = _fwd(m, b, noise=noise)
    try:
        L.check(lib.ss_set_tuning(b"mel_tail", 0), "knob")
        split = _fwd(m, b, noise=noise)
    finally:
        L.check(lib.ss_set_tuning(b"mel_tail", 1), "knob")
    d = (fused["mel_out"] - split["mel_out"]).abs()
    print(f"mel tail kernel vs matrix-core launches: mel max diff {d.max().item():.3e}, mean {d.mean().item():.3e}")
    assert torch.equal(fused["pitch_coarse"], split["pitch_coarse"]) and torch.equal(fused["uv_a"], split["uv_a"])
    assert 0.0 < d.max().item() <= 1e-5 and d.mean().item() <= 1e-6
