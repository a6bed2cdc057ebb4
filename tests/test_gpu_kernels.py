"""GPU unit parity: each HIP kernel against a plain fp32 torch CPU statement of the same op, through the C-ABI."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from stylesinger_amd import lib as L  # noqa: E402


def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _ref_conv(x, w, b, dil, lens):
    """channels-last same conv with per-item zero padding beyond lens."""
    B, T, C = x.shape
    out = torch.zeros(B, T, w.shape[0])
    k = w.shape[-1]
    for i in range(B):
        n = int(lens[i])
        xi = x[i:i + 1, :n].transpose(1, 2)
        out[i, :n] = F.conv1d(xi, w, b, padding=(k - 1) // 2 * dil, dilation=dil).transpose(1, 2)[0]
    return out


@pytest.mark.parametrize("B,T,Cin,Cout,k,dil,tile", [
    (2, 70, 256, 256, 3, 2, 0), (1, 130, 80, 160, 5, 1, 0), (3, 33, 256, 1024, 9, 1, 0), (2, 200, 192, 96, 1, 1, 0),
    (1, 300, 32, 32, 11, 5, 5), (1, 300, 64, 64, 7, 3, 4), (2, 129, 128, 128, 3, 1, 1), (2, 129, 128, 128, 3, 1, 2),
    (2, 129, 128, 128, 3, 1, 3), (1, 40, 1104, 256, 1, 1, 0), (1, 64, 256, 3, 1, 1, 0),
])
def test_conv_gemm_store(B, T, Cin, Cout, k, dil, tile):
    d = dev()
    x = _rand(B, T, Cin, seed=1)
    w = _rand(Cout, Cin, k, seed=2, scale=1 / math.sqrt(Cin * k))
    b = _rand(Cout, seed=3, scale=0.1)
    r = _rand(B, T, Cout, seed=4)
    lens = torch.tensor([T - 3 * i for i in range(B)], dtype=torch.int32)
    ref = F.gelu(_ref_conv(x, w, b, dil, lens) * 0.5)
    ref = ref + r
    for i in range(B):
        ref[i, int(lens[i]):] = 0
    W = L.pack_conv_weight(w.to(d))
    bias = L.pack_bias(b.to(d))
    out = torch.full((B, T, Cout), 7.0, device=d)
    L.conv_gemm(x.to(d), W, out, B=B, T=T, Cin=Cin, N=Cout, Np=W.shape[0], Kp=W.shape[1] // k,
                taps=[(j - (k - 1) // 2) * dil for j in range(k)], lens=lens.to(d), bias=bias, pre_scale=0.5, act=L.ACT_GELU,
                R=r.to(d), ldr=Cout, tile=tile)
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5, err


def test_conv_gemm_prologue_bias_lrelu_scale_accumulate():
    d = dev()
    B, T, C = 2, 90, 64
    x = _rand(B, T, C, seed=5)
    ab = _rand(C, seed=6)
    w = _rand(C, C, 3, seed=7, scale=0.1)
    b = _rand(C, seed=8, scale=0.1)
    lens = torch.tensor([90, 61], dtype=torch.int32)
    xa = F.leaky_relu((x + ab) * 0.7, 0.1)
    prev = _rand(B, T, C, seed=9)
    ref = prev + _ref_conv(xa, w, b, 1, lens) * (1 / 3)
    for i in range(B):
        ref[i, int(lens[i]):] = 0
    W = L.pack_conv_weight(w.to(d))
    out = prev.to(d).clone()
    L.conv_gemm(x.to(d), W, out, B=B, T=T, Cin=C, N=C, Np=W.shape[0], Kp=W.shape[1] // 3, taps=(-1, 0, 1), lens=lens.to(d),
                a_bias=ab.to(d), a_scale=0.7, a_lrelu=0.1, bias=L.pack_bias(b.to(d)), post_scale=1 / 3, accumulate=True)
    assert (out.cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("C,mode", [(256, 0), (192, 0), (80, 1)])
def test_conv_gemm_gate_and_resskip(C, mode):
    d = dev()
    B, T = 2, 100
    x = _rand(B, T, C, seed=11)
    w = _rand(2 * C, C, 3, seed=12, scale=1 / math.sqrt(3 * C))
    b = _rand(2 * C, seed=13, scale=0.1)
    e = _rand(B, T, 2 * C, seed=14)
    lens = torch.tensor([T, T - 17], dtype=torch.int32)
    y = _ref_conv(x, w, b, 2, lens) + e
    g_ref = torch.sigmoid(y[..., :C]) * torch.tanh(y[..., C:]) if mode == 0 else torch.tanh(y[..., :C]) * torch.sigmoid(y[..., C:])
    W = L.pack_conv_weight(w.to(d), interleave_half=C)
    bias = L.pack_bias(b.to(d), interleave_half=C)
    Np = W.shape[0]
    # E must be in packed column order
    ep = torch.zeros(B, T, Np)
    Cp = Np // 2
    for p in range(Cp // 32):
        n = min(32, C - p * 32)
        if n <= 0:
            break
        ep[..., (2 * p) * 32:(2 * p) * 32 + n] = e[..., p * 32:p * 32 + n]
        ep[..., (2 * p + 1) * 32:(2 * p + 1) * 32 + n] = e[..., C + p * 32:C + p * 32 + n]
    g = torch.empty(B, T, C, device=d)
    L.conv_gemm(x.to(d), W, g, B=B, T=T, Cin=C, N=C, Np=Np, Kp=W.shape[1] // 3, taps=(-2, 0, 2), lens=lens.to(d), epi=L.EPI_GATE,
                gate_mode=mode, bias=bias, E=ep.to(d), lde=Np, e_bs=T * Np, ldc=C, mask_rows=False)
    for i in range(B):
        n = int(lens[i])
        assert (g[i, :n].cpu() - g_ref[i, :n]).abs().max().item() < 2e-5
    if C % 32 == 0:
        wo = _rand(2 * C, C, 1, seed=15, scale=1 / math.sqrt(C))
        bo = _rand(2 * C, seed=16, scale=0.1)
        xs = _rand(B, T, C, seed=17)
        sk = _rand(B, T, C, seed=18)
        gv = g_ref.clone()
        for i in range(B):
            gv[i, int(lens[i]):] = 0
        yo = F.conv1d(gv.transpose(1, 2), wo, bo).transpose(1, 2)
        x_ref = (xs + yo[..., :C]) / math.sqrt(2.0)
        s_ref = sk + yo[..., C:]
        Wo = L.pack_conv_weight(wo.to(d))
        xd, sd_ = xs.to(d).clone(), sk.to(d).clone()
        L.conv_gemm(gv.to(d), Wo, xd, B=B, T=T, Cin=C, N=2 * C, Np=Wo.shape[0], Kp=Wo.shape[1], lens=lens.to(d), epi=L.EPI_RESSKIP,
                    bias=L.pack_bias(bo.to(d)), Nh=C, R=xd, ldr=C, ldc=C, post_scale=1 / math.sqrt(2.0), C2=sd_, ldc2=C, c2_bs=T * C,
                    accumulate=True, mask_rows=False)
        assert (xd.cpu() - x_ref).abs().max().item() < 2e-5
        assert (sd_.cpu() - s_ref).abs().max().item() < 2e-5


def test_convtranspose_polyphase():
    d = dev()
    for (Cin, Cout, u) in [(64, 32, 8), (32, 16, 2)]:
        k = 2 * u
        B, T = 2, 37
        x = _rand(B, T, Cin, seed=21)
        v = _rand(Cin, Cout, k, seed=22, scale=0.1)
        g = torch.rand(Cin, 1, 1) + 0.5
        b = _rand(Cout, seed=23, scale=0.1)
        w = v * (g / v.reshape(Cin, -1).norm(dim=1).reshape(Cin, 1, 1))
        ref = F.conv_transpose1d(F.leaky_relu(x, 0.1).transpose(1, 2), w, b, stride=u, padding=(k - u) // 2).transpose(1, 2)
        s0 = L.weight_norm_scale(v.to(d), g.to(d))
        out = torch.zeros(B, T * u, Cout, device=d)
        pad = (k - u) // 2
        nph0 = u - pad
        bias = L.pack_bias(b.to(d), repeat=u)
        for grp in range(2):
            W = L.pack_convtr_weight(v.to(d), s0, u, grp)
            nph = nph0 if grp == 0 else u - nph0
            o = out.view(B, T, u * Cout)[:, :, (0 if grp == 0 else nph0 * Cout):]
            L.conv_gemm(x.to(d), W, o, B=B, T=T, Cin=Cin, N=nph * Cout, Np=W.shape[0], Kp=W.shape[1] // 2,
                        taps=(0, -1) if grp == 0 else (1, 0), a_lrelu=0.1, bias=bias, ldc=u * Cout, c_bs=T * u * Cout)
        assert (out.cpu() - ref).abs().max().item() < 2e-5


def test_layernorm_and_masks():
    d = dev()
    for C in (80, 256):
        B, T = 2, 50
        x = _rand(B, T, C, seed=31) * 3 + 1
        g, b = _rand(C, seed=32) + 1, _rand(C, seed=33)
        lens = torch.tensor([50, 20], dtype=torch.int32)
        ref = F.layer_norm(x, (C,), g, b, 1e-5)
        ref[1, 20:] = 0
        y = L.layernorm(x.to(d), g.to(d), b.to(d), B=B, T=T, C_=C, out=torch.empty(B, T, C, device=d), lens=lens.to(d), mask_rows=True)
        assert (y.cpu() - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize("Tq,Tk,B", [(70, 70, 2), (33, 200, 1), (257, 129, 2)])
def test_attention(Tq, Tk, B):
    d = dev()
    H, D = 2, 128
    q, k, v = _rand(B, Tq, H * D, seed=41), _rand(B, Tk, H * D, seed=42), _rand(B, Tk, H * D, seed=43)
    klens = torch.tensor([Tk - 5 * i for i in range(B)], dtype=torch.int32)
    scale = D ** -0.5
    ref = torch.zeros(B, Tq, H * D)
    for b in range(B):
        n = int(klens[b])
        for h in range(H):
            s = (q[b, :, h * D:(h + 1) * D] * scale) @ k[b, :n, h * D:(h + 1) * D].t()
            ref[b, :, h * D:(h + 1) * D] = torch.softmax(s, -1) @ v[b, :n, h * D:(h + 1) * D]
    o = torch.zeros(B, Tq, H * D, device=d)
    L.attention(q.to(d), k.to(d), v.to(d), o, B=B, H=H, D=D, Tq=Tq, Tk=Tk, ldq=H * D, ldk=H * D, ldv=H * D, ldo=H * D, q_bs=Tq * H * D,
                k_bs=Tk * H * D, v_bs=Tk * H * D, o_bs=Tq * H * D, klens=klens.to(d), scale=scale)
    assert (o.cpu() - ref).abs().max().item() < 1e-5


def test_rq_lookup_exact_codes():
    d = dev()
    lib = L.load()
    rows, C, n, depth = 203, 256, 128, 4
    x = _rand(rows, C, seed=51)
    cb = torch.stack([_rand(n + 1, C, seed=60 + i) * (0.8 * 0.6 ** i) for i in range(depth)])
    r, agg, codes = x.clone(), torch.zeros_like(x), []
    for i in range(depth):
        c = cb[i, :-1]
        dist = torch.addmm(r.pow(2).sum(1, keepdim=True) + c.t().pow(2).sum(0, keepdim=True), r, c.t(), alpha=-2.0)
        kk = dist.argmin(-1)
        r = r - c[kk]
        agg = agg + c[kk]
        codes.append(kk)
    ref = x + (agg - x)
    out = torch.empty(rows, C, device=d)
    cod = torch.empty(rows, depth, device=d, dtype=torch.int64)
    xd, cbd = x.to(d), cb.to(d).contiguous()  # keep alive: L.ptr() of a temporary would dangle
    L.check(lib.ss_rq_lookup(L.ptr(xd), L.ptr(cbd), L.ptr(out), L.ptr(cod), rows, C, n, depth, L.stream_ptr()))
    assert torch.equal(cod.cpu(), torch.stack(codes, -1))  # integer work: bit-exact
    assert (out.cpu() - ref).abs().max().item() < 1e-5


def test_positions_embedding_lengthreg():
    d = dev()
    lib = L.load()
    B, Tp = 2, 9
    tok = torch.tensor([[5, 3, 0, 7, 9, 0, 0, 2, 1], [4, 4, 4, 0, 0, 0, 0, 0, 0]])
    pos = torch.empty(B, Tp, dtype=torch.int32, device=d)
    tokd = tok.to(d)
    L.check(lib.ss_make_positions(L.ptr(tokd), None, 0, 0, L.ptr(pos), B, Tp, L.stream_ptr()))
    m = (tok != 0).int()
    assert torch.equal(pos.cpu().long(), (torch.cumsum(m, 1) * m).long())
    logdur = torch.tensor([[1.2, 0.1, 3.0, 1.7, -2.0, 0.5, 0.5, 1.0986123, 2.2], [0.9, 1.5, 1.1, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]])
    dur = torch.clamp(torch.round(logdur.exp() - 1), min=0).long() * (tok != 0).long()
    T = int(dur.sum(-1).max())
    dur_o = torch.empty(B, Tp, dtype=torch.int64, device=d)
    lens = torch.empty(B, dtype=torch.int32, device=d)
    m2p = torch.empty(B, T, dtype=torch.int64, device=d)
    ldd = logdur.to(d)
    L.check(lib.ss_length_regulate(L.ptr(ldd), L.ptr(tokd), L.ptr(dur_o), None, L.ptr(lens), B, Tp, 0, L.stream_ptr()))
    assert torch.equal(dur_o.cpu(), dur) and lens.cpu().tolist() == dur.sum(-1).tolist()
    L.check(lib.ss_length_regulate(L.ptr(ldd), L.ptr(tokd), L.ptr(dur_o), L.ptr(m2p), L.ptr(lens), B, Tp, T, L.stream_ptr()))
    ref = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        p = 0
        for i in range(Tp):
            ref[b, p:p + int(dur[b, i])] = i + 1
            p += int(dur[b, i])
    assert torch.equal(m2p.cpu(), ref)


@pytest.mark.parametrize("tile", [2, 3])
@pytest.mark.parametrize("C,d,T,B", [(256, 1, 150, 2), (256, 2, 203, 1), (256, 8, 97, 2), (192, 4, 260, 2)])
def test_winograd_gate_equals_direct_conv(C, d, T, B, tile):
    """ss_wino_gate (Winograd F(2,3)) vs a plain torch statement of conv(x + bias) + E -> sigmoid*tanh, with ragged lens,
    odd T (partial last group) and every dilation of the cycle."""
    dv = dev()
    x = _rand(B, T, C, seed=71)
    ab = _rand(C, seed=72)
    w = _rand(2 * C, C, 3, seed=73, scale=1 / math.sqrt(3 * C))
    Lyr = 3
    e = _rand(B, T, Lyr * 2 * C, seed=74)
    lens = torch.tensor([T - 11 * i for i in range(B)], dtype=torch.int32)
    ref = torch.zeros(B, T, C)
    for i in range(B):
        n = int(lens[i])
        y = (x[i:i + 1, :n] + ab).transpose(1, 2)
        z = F.conv1d(y, w, None, padding=d, dilation=d).transpose(1, 2)[0] + e[i, :n, 2 * C:4 * C]
        ref[i, :n] = torch.sigmoid(z[:, :C]) * torch.tanh(z[:, C:])
    Wt = L.pack_conv_weight(L.wino_weight(w.to(dv)), interleave_half=C)
    Np = Wt.shape[0]
    # E in packed column order (layer slab 1 of 3)
    ep = torch.zeros(B, T, Lyr * Np)
    for p in range(C // 32):
        ep[..., Np + (2 * p) * 32:Np + (2 * p) * 32 + 32] = e[..., 2 * C + p * 32:2 * C + p * 32 + 32]
        ep[..., Np + (2 * p + 1) * 32:Np + (2 * p + 1) * 32 + 32] = e[..., 3 * C + p * 32:3 * C + p * 32 + 32]
    epd = ep.to(dv)
    g = torch.full((B, T, C), 5.0, device=dv)
    L.wino_gate(x.to(dv), Wt, g, dilation=d, B=B, T=T, Cin=C, N=C, Np=Np, Kp=C, lens=lens.to(dv), a_bias=ab.to(dv),
                E=epd[:, :, Np:], lde=Lyr * Np, e_bs=T * Lyr * Np, ldc=C, mask_rows=True, tile=tile)  # 2: 64x128, 3: 64x64
    err = (g.cpu() - ref).abs().max().item()
    assert err < 5e-6, err


@pytest.mark.parametrize("mt", [None, 2, 3, 0])
@pytest.mark.parametrize("C,d,T,B", [(256, 1, 150, 2), (256, 2, 203, 1), (256, 8, 97, 2), (192, 4, 260, 2), (256, 4, 1536, 3), (64, 1, 5, 1),
                                      (64, 8, 5, 2), (96, 16, 300, 2), (32, 2, 1, 1)])
def test_winograd_f43_gate_equals_direct_conv(C, d, T, B, mt):
    """ss_wino43_gate (Winograd F(4,3): 6 products per 4 frames; mt None = 32x32x2 tiles) and ss_wino43_gate16 (16x16x4 tiles of 16*mt
    quads; mt 0 = the library's pick) vs a plain torch statement of conv(x + bias) + E -> sigmoid*tanh, with ragged lens, T not a
    multiple of the 4d frame group, every dilation of the cycle and a tile-spanning length."""
    dv = dev()
    x = _rand(B, T, C, seed=171)
    ab = _rand(C, seed=172)
    w = _rand(2 * C, C, 3, seed=173, scale=1 / math.sqrt(3 * C))
    bias = _rand(2 * C, seed=175, scale=0.3)
    Lyr = 3
    e = _rand(B, T, Lyr * 2 * C, seed=174)
    lens = torch.tensor([max(1, T - 11 * i) for i in range(B)], dtype=torch.int32)
    ref = torch.zeros(B, T, C)
    for i in range(B):
        n = int(lens[i])
        y = (x[i:i + 1, :n] + ab).transpose(1, 2)
        z = F.conv1d(y.double(), w.double(), bias.double(), padding=d, dilation=d).transpose(1, 2)[0] + e[i, :n, 2 * C:4 * C].double()
        ref[i, :n] = (torch.sigmoid(z[:, :C]) * torch.tanh(z[:, C:])).float()
    Wt = L.pack_conv_weight(L.wino43_weight(w.to(dv)), interleave_half=C)
    Np = Wt.shape[0]
    assert Wt.shape[1] == 6 * C
    Cp = Np // 2
    def packed(v):  # [..., 2C] -> packed column order [..., Np]: 32 first-operand channels, then the 32 second-operand ones
        out = torch.zeros(*v.shape[:-1], Np)
        for p in range(Cp // 32):
            n = min(32, C - p * 32)
            out[..., (2 * p) * 32:(2 * p) * 32 + n] = v[..., p * 32:p * 32 + n]
            out[..., (2 * p + 1) * 32:(2 * p + 1) * 32 + n] = v[..., C + p * 32:C + p * 32 + n]
        return out
    ep = torch.zeros(B, T, Lyr * Np)
    ep[..., Np:2 * Np] = packed(e[..., 2 * C:4 * C])
    epd = ep.to(dv)
    g = torch.full((B, T, C), 5.0, device=dv)
    kw = dict(dilation=d, B=B, T=T, Cin=C, N=C, Np=Np, Kp=C, lens=lens.to(dv), a_bias=ab.to(dv),
              bias=packed(bias).to(dv), E=epd[:, :, Np:], lde=Lyr * Np, e_bs=T * Lyr * Np, ldc=C, mask_rows=True)
    if mt is None:
        L.wino43_gate(x.to(dv), Wt, g, **kw)
    else:
        L.wino43_gate16(x.to(dv), Wt, g, mt=mt, **kw)
    err = (g.cpu() - ref).abs().max().item()
    # max over up to 1.2 M outputs of ONE F(4,3) layer vs a float64 conv: 9.2e-6 for exact-fp32 products in any summation order
    # (CPU emulation of the transforms, tools/wino43_numerics.py; mean error 7e-7) - the bound leaves 2x for the order over K
    assert err < (2e-5 if B * T * C > 500000 else 1e-5), err
    # rows past an item's length are written as zeros (mask_rows), never left stale
    for i in range(B):
        assert torch.all(g[i, int(lens[i]):] == 0)


def test_grouped_launch_uses_per_item_weight_sets():
    dv = dev()
    B, T, C = 4, 70, 64
    x = _rand(B, T, C, seed=81)
    ws = [_rand(C, C, 3, seed=82 + g_, scale=0.1) for g_ in range(2)]
    bs = [_rand(C, seed=84 + g_, scale=0.1) for g_ in range(2)]
    lens = torch.full((B,), T, dtype=torch.int32)
    ref = torch.stack([_ref_conv(x[i:i + 1], ws[i // 2], bs[i // 2], 1, lens[i:i + 1])[0] for i in range(B)])
    Wp = torch.stack([L.pack_conv_weight(w_.to(dv)) for w_ in ws]).contiguous()
    bp = torch.stack([L.pack_bias(b_.to(dv)) for b_ in bs]).contiguous()
    out = torch.empty(B, T, C, device=dv)
    L.conv_gemm(x.to(dv), Wp, out, B=B, T=T, Cin=C, N=C, Np=Wp.shape[1], Kp=Wp.shape[2] // 3, taps=(-1, 0, 1), lens=lens.to(dv),
                bias=bp, group_size=2, w_gs=Wp[0].numel(), bias_gs=bp[0].numel())
    assert (out.cpu() - ref).abs().max().item() < 2e-5


def _bf(x):
    return x.bfloat16().float()


@pytest.mark.parametrize("B,T,Cin,Cout,k,dil,tile", [
    (2, 70, 256, 256, 3, 2, 0), (1, 130, 32, 64, 1, 1, 0), (1, 130, 64, 64, 1, 1, 3), (2, 90, 32, 128, 3, 1, 2), (1, 300, 32, 32, 11, 5, 5),
    (1, 300, 64, 64, 7, 3, 4), (2, 129, 128, 128, 5, 1, 1), (1, 64, 80, 256, 7, 1, 0), (3, 33, 256, 512, 9, 1, 0),
])
def test_conv_gemm_bf16_operands_store(B, T, Cin, Cout, k, dil, tile):
    """mfma_bf16 = 1: operands rounded to bf16 (RNE) after the fp32 prologue, fp32 accumulate and epilogue. Reference = fp32
    conv of the rounded operands; the only difference left is the accumulation order (tolerance 2e-5 on O(1) outputs).
    Chunk counts 1, 2, 3, 5, 7, 11, 20, 21, 24, 72 exercise the two-stage prefetch loop for odd and even trip counts."""
    d = dev()
    x = _rand(B, T, Cin, seed=21)
    ab = _rand(Cin, seed=22, scale=0.3)
    w = _rand(Cout, Cin, k, seed=23, scale=1 / math.sqrt(Cin * k))
    b = _rand(Cout, seed=24, scale=0.1)
    r = _rand(B, T, Cout, seed=25)
    lens = torch.tensor([T - 3 * i for i in range(B)], dtype=torch.int32)
    xa = _bf(F.leaky_relu((x + ab) * 0.9, 0.1))
    ref = F.gelu(_ref_conv(xa, _bf(w), b, dil, lens) * 0.5) + r
    for i in range(B):
        ref[i, int(lens[i]):] = 0
    W = L.pack_conv_weight(w.to(d))
    out = torch.full((B, T, Cout), 7.0, device=d)
    L.conv_gemm(x.to(d), W, out, B=B, T=T, Cin=Cin, N=Cout, Np=W.shape[0], Kp=W.shape[1] // k,
                taps=[(j - (k - 1) // 2) * dil for j in range(k)], lens=lens.to(d), a_bias=ab.to(d), a_scale=0.9, a_lrelu=0.1,
                bias=L.pack_bias(b.to(d)), pre_scale=0.5, act=L.ACT_GELU, R=r.to(d), ldr=Cout, tile=tile, bf16=True)
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-5, err
    # and it is NOT the fp32 result (the rounding is really applied)
    ref32 = F.gelu(_ref_conv(F.leaky_relu((x + ab) * 0.9, 0.1), w, b, dil, lens) * 0.5) + r
    for i in range(B):
        ref32[i, int(lens[i]):] = 0
    assert (out.cpu() - ref32).abs().max().item() > 1e-4


@pytest.mark.parametrize("C", [256, 192])
def test_conv_gemm_bf16_operands_gate_and_resskip(C):
    d = dev()
    B, T = 2, 100
    x = _rand(B, T, C, seed=31)
    w = _rand(2 * C, C, 3, seed=32, scale=1 / math.sqrt(3 * C))
    b = _rand(2 * C, seed=33, scale=0.1)
    e = _rand(B, T, 2 * C, seed=34)
    lens = torch.tensor([T, T - 17], dtype=torch.int32)
    y = _ref_conv(_bf(x), _bf(w), b, 2, lens) + e
    g_ref = torch.sigmoid(y[..., :C]) * torch.tanh(y[..., C:])
    W = L.pack_conv_weight(w.to(d), interleave_half=C)
    bias = L.pack_bias(b.to(d), interleave_half=C)
    Np = W.shape[0]
    ep = torch.zeros(B, T, Np)
    for p in range(Np // 64):
        n = min(32, C - p * 32)
        if n <= 0:
            break
        ep[..., (2 * p) * 32:(2 * p) * 32 + n] = e[..., p * 32:p * 32 + n]
        ep[..., (2 * p + 1) * 32:(2 * p + 1) * 32 + n] = e[..., C + p * 32:C + p * 32 + n]
    g = torch.empty(B, T, C, device=d)
    L.conv_gemm(x.to(d), W, g, B=B, T=T, Cin=C, N=C, Np=Np, Kp=W.shape[1] // 3, taps=(-2, 0, 2), lens=lens.to(d), epi=L.EPI_GATE,
                bias=bias, E=ep.to(d), lde=Np, e_bs=T * Np, ldc=C, mask_rows=False, bf16=True)
    for i in range(B):
        n = int(lens[i])
        assert (g[i, :n].cpu() - g_ref[i, :n]).abs().max().item() < 2e-5
    wo = _rand(2 * C, C, 1, seed=35, scale=1 / math.sqrt(C))
    bo = _rand(2 * C, seed=36, scale=0.1)
    xs = _rand(B, T, C, seed=37)
    sk = _rand(B, T, C, seed=38)
    gv = g_ref.clone()
    for i in range(B):
        gv[i, int(lens[i]):] = 0
    yo = F.conv1d(_bf(gv).transpose(1, 2), _bf(wo), bo).transpose(1, 2)
    x_ref = (xs + yo[..., :C]) / math.sqrt(2.0)
    s_ref = sk + yo[..., C:]
    Wo = L.pack_conv_weight(wo.to(d))
    xd, sd_ = xs.to(d).clone(), sk.to(d).clone()
    L.conv_gemm(gv.to(d), Wo, xd, B=B, T=T, Cin=C, N=2 * C, Np=Wo.shape[0], Kp=Wo.shape[1], lens=lens.to(d), epi=L.EPI_RESSKIP,
                bias=L.pack_bias(bo.to(d)), Nh=C, R=xd, ldr=C, ldc=C, post_scale=1 / math.sqrt(2.0), C2=sd_, ldc2=C, c2_bs=T * C,
                accumulate=True, mask_rows=False, bf16=True)
    assert (xd.cpu() - x_ref).abs().max().item() < 2e-5
    assert (sd_.cpu() - s_ref).abs().max().item() < 2e-5


def test_clock_probe_reports_a_plausible_shader_clock_and_changes_nothing():
    """ss_set_clock_probe: the Winograd gate kernel's first wave reports shader cycles and 100 MHz ticks; their ratio is the
    sustained clock (nominal 2.4 GHz, lower under load). The gate output must not depend on the probe."""
    dv = dev()
    B, T, C = 4, 512, 256
    x = _rand(B, T, C, seed=91).to(dv)
    w = _rand(2 * C, C, 3, seed=92, scale=1 / math.sqrt(3 * C)).to(dv)
    Wt = L.pack_conv_weight(L.wino_weight(w), interleave_half=C)
    lens = torch.full((B,), T, dtype=torch.int32, device=dv)
    kw = dict(dilation=2, B=B, T=T, Cin=C, N=C, Np=Wt.shape[0], Kp=C, lens=lens, ldc=C)
    g0 = torch.empty(B, T, C, device=dv)
    L.wino_gate(x, Wt, g0, **kw)
    probe = torch.zeros(2, dtype=torch.int64, device=dv)
    lib = L.load()
    L.check(lib.ss_set_clock_probe(probe.data_ptr()), "ss_set_clock_probe")
    try:
        g1 = torch.empty(B, T, C, device=dv)
        for _ in range(8):
            L.wino_gate(x, Wt, g1, **kw)
        torch.cuda.synchronize()
    finally:
        L.check(lib.ss_set_clock_probe(None), "ss_set_clock_probe")
    cyc, ticks = (int(v) for v in probe.cpu())
    assert ticks > 0 and 0.5 < cyc / ticks / 10.0 < 2.6, (cyc, ticks)
    assert torch.equal(g0, g1)
    before = probe.clone()
    L.wino_gate(x, Wt, g1, **kw)  # probe off again: nothing is written
    torch.cuda.synchronize()
    assert torch.equal(before, probe)
