""""fp16q4" (ss_gemm_bf16_args.split = 3): the fp16x2 gate with its second product (activation x weight-lo) on gfx950's block-scaled fp4 matrix
instruction - per pair of 32-channel steps 32 fp16 MFMAs + 8 fp4 ones instead of 64 (gemm_bf16_gate128q.hip). The kernel converts its own fp16
A fragments to fp4 in registers; the weights' lo plane is packed once in the kernel's lane order (lib.pack_gate_q4).

Written at the end of round 4 from measured instruction semantics (tools/ubench/mfma_mx_layout.hip, cvt_fp4_probe.hip), CPU numerics
(oracle/second_product_numerics.py: 3.4e-5 / 4.2e-5 on the real reference's goldens, bar 1e-4; tests/test_oracle_golden.py pins the contract) and a
host check of the kernel's addressing (tools/layout_check_gate128.cpp). Round 5, first GPU session: all six tests below passed on their first run
on an MI355X (profiles/r05_session1_tests.log) - the env gate is gone. Parity of the MODE against the real reference: tests/test_gpu_round5.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from stylesinger_amd import lib as L  # noqa: E402

WS = 8


@pytest.mark.parametrize("T,d,qs", [(5600, 4, 2.0), (777, 8, 2.0), (1000, 1, 4.0)])
def test_gate128q_matches_float64_of_the_same_terms(T, d, qs):
    """ss_gemm_bf16_gate128q against float64 math on exactly the terms the matrix cores see: a_hi (fp16) x w_hi (fp16 of w * 2^8) on the fp16
    instruction + fp4(a_hi / qs) x fp4-with-block-scale(w_lo) on the block-scaled one, the sum scaled by 2^-8, + addend, sigmoid * tanh, fp16 out."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(T + d)
    B, C = 3, 256
    sc, osc = float(2 ** WS), float(2.0 ** -WS)
    lens_l = [T, T - 37, 5]
    lens = torch.tensor(lens_l, dtype=torch.int32, device=dev)
    x = torch.randn(B, T, C, generator=g).to(dev) * 3.0
    for b in range(B):
        x[b, lens_l[b]:] = 0
    xs = L.split_f16(x)
    xh = x.to(torch.float16).float()
    w = (torch.randn(2 * C, C, 3, generator=g) / (3 * C) ** 0.5).to(dev)
    Wp = L.pack_conv_weight(w, interleave_half=C)          # [2C][3 * 256] fp32, gate-interleaved rows, tap-major
    assert Wp.shape == (2 * C, 3 * C)
    pack, lo_q = L.pack_gate_q4(Wp, shift=WS)
    hi = (Wp * sc).to(torch.float16).float()
    lo = Wp * sc - hi
    # the block-scaled fp4 image of lo keeps two to three significant bits of a correction that is 2^-11 of the weight
    rel = ((lo_q - lo).abs().max() / lo.abs().max()).item()
    print(f"fp4 image of the weights' lo plane: max |lo_q - lo| / max |lo| = {rel:.3f}")
    assert rel <= 0.26
    # A as the kernel's three taps see it (zero outside [0, len)), and its in-register fp4 image with the fixed scale qs
    idx, mag = L.fp4_rne(xh / qs)
    xq = mag * torch.sign(xh) * qs

    def unfold(a):
        z = torch.zeros(B, T + 2 * d, C, device=dev, dtype=torch.float64)
        z[:, d:d + T] = a.double()
        return torch.cat([z[:, 0:T], z[:, d:d + T], z[:, 2 * d:2 * d + T]], dim=-1)   # taps (-d, 0, +d), tap-major K
    y = (unfold(xh) @ hi.double().t() + unfold(xq) @ lo_q.double().t()) * osc       # packed columns
    y_exact = unfold(x) @ Wp.double().t()
    E = torch.randn(B, T, 2 * C, generator=g).to(dev) * 0.5                        # addend, already in packed column order
    z = y + E.double()
    zx = y_exact + E.double()

    def gate(v):
        v = v.view(B, T, C // 32, 2, 32)
        return (torch.sigmoid(v[..., 0, :]) * torch.tanh(v[..., 1, :])).reshape(B, T, C).float()
    g_ref, g_exact = gate(z), gate(zx)
    for b in range(B):
        g_ref[b, lens_l[b]:] = 0
        g_exact[b, lens_l[b]:] = 0
    GA = torch.full((B, T, 2 * C), 7.0, device=dev, dtype=torch.float16)
    L.gemm_bf16(xs, pack, B=B, T=T, K=C, taps=(-d, 0, d), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens, E=E, lde=2 * C, out=GA, ldc=2 * C, lda=2 * C,
                split=3, out_scale=osc, q_scale=qs, gate256=128)
    got, second = L.split_planes(GA)
    e_same, e_exact = (got - g_ref).abs().max().item(), (got - g_exact).abs().max().item()
    print(f"gate128q T={T} d={d} qs={qs}: vs float64 of the same terms {e_same:.2e}, vs exact operands {e_exact:.2e}")
    assert torch.all(second == 7.0), "the second plane of the output rows is not written"
    assert e_same <= 3e-4 and e_exact <= 4e-3, (e_same, e_exact)
    # and against the fp16x2 kernel (lo plane as fp16 terms): the two second products differ by the fp4 rounding of a 2^-11 correction
    GA2 = torch.full((B, T, 2 * C), 7.0, device=dev, dtype=torch.float16)
    L.gemm_bf16(xs, L.split_f16(Wp, scale=sc), B=B, T=T, K=C, taps=(-d, 0, d), N=C, Np=2 * C, epi=L.HEPI_GATE, lens=lens, E=E, lde=2 * C, out=GA2,
                ldc=2 * C, lda=2 * C, split=2, out_scale=osc, gate256=128)
    dd = (got - L.split_planes(GA2)[0]).abs().max().item()
    print(f"  vs gate128 (fp16x2): {dd:.2e}")
    assert dd <= 1.5e-3


def test_fp16q4_mode_stays_close_to_fp16x2_at_the_c4_shape():
    """BASELINE configs[3]'s shape (B = 32 x T = 5625: the only shape whose gate launches qualify for gate128q), 20 + 2 x 20 steps: the fp16q4 path
    against the fp16x2 path on the same inputs and noise - the two differ by the fp4 rounding of a 2^-11 correction (CPU restatement vs the real
    reference: 3.9e-5; fp16x2: 3.3e-5 on the same golden)."""
    from stylesinger_amd import config, synth
    from stylesinger_amd.model import StyleSingerHIP
    S = 20
    hp = config.make_hparams(dict(timesteps=S, K_step=S, f0_timesteps=S))
    B, T, Tp, Tr = 32, 5625, 105, 1500
    batch = {k: v.cuda() for k, v in synth.synth_batch(B, T, Tp, Tr, hp, 2025).items()}
    sd = synth.synth_acoustic_state_dict(hp, 2025)
    noise = synth.draw_acoustic_noise(synth.NoiseTape(78), B, T, S, S)
    outs = {}
    for prec in ("fp16x2", "fp16q4"):
        m = StyleSingerHIP(None, hparams=dict(hp, mfma_precision=prec))
        m.load_state_dict(sd)
        m.eval().to("cuda:0")
        assert m.f16 and (m.q4 == (prec == "fp16q4"))
        r = m(batch["txt_tokens"], mel2ph=batch.get("mel2ph"), spk_embed=batch["spk_embed"], emo_embed=batch["emo_embed"], ref_mels=batch["ref_mels"],
              ref_f0=batch["ref_f0"], global_steps=320000, infer=True, note=batch["note"], note_dur=batch["note_dur"], note_type=batch["note_type"], noise=noise)
        outs[prec] = r["mel_out"].float().cpu()
        del m
        torch.cuda.empty_cache()
    d = (outs["fp16q4"] - outs["fp16x2"]).abs()
    print(f"fp16q4 vs fp16x2 at B=32 x T=5625, {S} steps: mel L1 {d.mean().item():.3e} max {d.max().item():.3e}")
    assert d.mean().item() > 0, "the two modes must not be the same code path at this shape"
    assert d.mean().item() <= 1e-4


@pytest.mark.parametrize("T,K", [(5600, 512), (777, 1024)])
def test_tile256q_store_matches_float64_of_the_same_terms(T, K):
    """ss_gemm_bf16_tile256q (the fp16q4 skip GEMM: STORE + ReLU, A = gate outputs in the pair layout, hi term only) against float64 of the terms
    the matrix cores see: a (fp16) x w_hi (fp16) + fp4(a / 2^-2) x block-scaled fp4 (w_lo), scaled by 2^-8, + bias, ReLU."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(T + K)
    B, N, qs = 3, 256, 0.25
    sc, osc = float(2 ** WS), float(2.0 ** -WS)
    lens_l = [T, T - 37, 5]
    lens = torch.tensor(lens_l, dtype=torch.int32, device=dev)
    a = (torch.rand(B, T, K, generator=g) * 2 - 1).to(dev)          # gate outputs live in (-1, 1)
    for b in range(B):
        a[b, lens_l[b]:] = 0
    As = L.split_f16(a)
    ah = a.to(torch.float16).float()
    idx, mag = L.fp4_rne(ah / qs)
    aq = mag * torch.sign(ah) * qs
    w = (torch.randn(N, K, 1, generator=g) / K ** 0.5).to(dev)
    Wp = L.pack_conv_weight(w)
    pack, lo_q = L.pack_skip_q4(Wp, shift=WS)
    hi = (Wp * sc).to(torch.float16).float()
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    ref = torch.relu((ah.double() @ hi.double().t() + aq.double() @ lo_q.double().t())[..., :N] * osc + bias.double()).float()
    exact = torch.relu((a.double() @ Wp.double().t())[..., :N] + bias.double()).float()
    for b in range(B):
        ref[b, lens_l[b]:] = 0
        exact[b, lens_l[b]:] = 0
    S = torch.empty(B, T, N, device=dev)
    L.gemm_bf16(As, pack, B=B, T=T, K=K, taps=(0,), N=N, Np=Wp.shape[0], epi=L.HEPI_STORE, lens=lens, act=L.ACT_RELU, out=S, lda=2 * K, split=3,
                out_scale=osc, q_scale=qs, bias=L.pack_bias(bias), gate256=True)
    e_same, e_exact = (S - ref).abs().max().item(), (S - exact).abs().max().item()
    print(f"tile256q T={T} K={K}: vs float64 of the same terms {e_same:.2e}, vs exact operands {e_exact:.2e}")
    assert e_same <= 2e-5 * K ** 0.5 and e_exact <= 5e-3
