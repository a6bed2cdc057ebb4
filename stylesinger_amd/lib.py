"""ctypes binding of libstylesinger_hip.so (the C-ABI in include/stylesinger_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be resolved this
module raises, and every op raises `StyleSingerHipError` on a non-zero status.  PyTorch is used only
for device memory (`tensor.data_ptr()`) and streams.
"""
import ctypes as C
import os
import re

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS_LIB_PATH") or os.path.join(HERE, "libstylesinger_hip.so")  # SS_LIB_PATH: debug builds (tools/ablate.sh)
HEADER = os.path.join(os.path.dirname(HERE), "include", "stylesinger_hip.h")

SS_MAX_TAPS = 16
SS_MAX_LAYERS = 32
SS_HG_MAX_UPS = 6
SS_HG_MAX_KERNELS = 4

ABI_VERSION = 19  # include/stylesinger_hip.h SS_ABI_VERSION
EPI_STORE, EPI_GATE, EPI_RESSKIP, EPI_DDPM = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_GELU, ACT_MISH, ACT_TANH, ACT_LRELU = 0, 1, 2, 3, 4, 5

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p


class StyleSingerHipError(RuntimeError):
    pass


class ConvGemmArgs(C.Structure):
    _fields_ = [
        ("A", _vp), ("a_batch_stride", C.c_int64), ("lda", C.c_int32), ("Cin", C.c_int32), ("ntaps", C.c_int32),
        ("tap_off", C.c_int32 * SS_MAX_TAPS), ("lens", _vp), ("B", C.c_int32), ("T", C.c_int32), ("a_bias", _vp),
        ("a_scale", C.c_float), ("a_lrelu", C.c_float),
        ("W", _vp), ("N", C.c_int32), ("Np", C.c_int32), ("Kp", C.c_int32),
        ("epi", C.c_int32), ("bias", _vp), ("pre_scale", C.c_float), ("act", C.c_int32), ("act_slope", C.c_float),
        ("E", _vp), ("lde", C.c_int32), ("e_batch_stride", C.c_int64), ("gate_mode", C.c_int32),
        ("R", _vp), ("ldr", C.c_int32), ("r_batch_stride", C.c_int64), ("post_scale", C.c_float),
        ("accumulate", C.c_int32), ("mask_rows", C.c_int32),
        ("C", _vp), ("ldc", C.c_int32), ("c_batch_stride", C.c_int64),
        ("C2", _vp), ("ldc2", C.c_int32), ("c2_batch_stride", C.c_int64), ("Nh", C.c_int32),
        ("ddpm_recip", C.c_float), ("ddpm_recipm1", C.c_float), ("ddpm_c1", C.c_float), ("ddpm_c2", C.c_float),
        ("ddpm_sigma", C.c_float), ("noise", _vp), ("seed", C.c_uint64), ("seed_dev", _vp), ("step", C.c_uint32), ("tile", C.c_int32),
        ("group_size", C.c_int32), ("w_group_stride", C.c_int64), ("bias_group_stride", C.c_int64), ("a_bias_group_stride", C.c_int64),
        ("mfma_bf16", C.c_int32), ("ddpm_x0_pred", C.c_int32), ("e_tiled", C.c_int32), ("reserved2_", C.c_int32),
    ]


class WaveNet(C.Structure):
    _fields_ = [
        ("C", C.c_int32), ("L", C.c_int32), ("cond_dim", C.c_int32), ("dil_cycle", C.c_int32), ("in_dim", C.c_int32),
        ("out_dim", C.c_int32), ("steps", C.c_int32),
        ("w_in", _vp), ("b_in", _vp), ("uv_embed", _vp), ("dstep", _vp),
        ("w_dil", _vp * SS_MAX_LAYERS), ("w_out", _vp * SS_MAX_LAYERS), ("b_out", _vp * SS_MAX_LAYERS),
        ("w_cond", _vp), ("b_cond", _vp), ("w_skip", _vp), ("b_skip", _vp), ("w_final", _vp), ("b_final", _vp),
        ("sqrt_recip_ac", _vp), ("sqrt_recipm1_ac", _vp), ("post_c1", _vp), ("post_c2", _vp), ("post_logvar", _vp),
        ("log_alpha", _vp), ("log_1m_alpha", _vp), ("log_cumprod_alpha", _vp), ("log_1m_cumprod_alpha", _vp),
        ("n_groups", C.c_int32), ("w_dil_wino", _vp * SS_MAX_LAYERS), ("gs_w_dil_wino", C.c_int64),
    ] + [(n, C.c_int64) for n in ("gs_w_in", "gs_b_in", "gs_uv_embed", "gs_dstep", "gs_w_dil", "gs_w_out", "gs_b_out", "gs_w_cond",
                                  "gs_b_cond", "gs_w_skip", "gs_b_skip", "gs_w_final", "gs_b_final")] \
        + [("mfma_bf16", C.c_int32), ("wino_m", C.c_int32), ("w_skipall", _vp), ("b_skipall", _vp),
           ("gs_w_skipall", C.c_int64), ("gs_b_skipall", C.c_int64), ("skipall_folded", C.c_int32), ("mfma_x3", C.c_int32),
           ("w_dil_h", _vp * SS_MAX_LAYERS), ("w_out_h", _vp * SS_MAX_LAYERS), ("w_skipall_h", _vp), ("w_cond_h", _vp),
           ("gs_w_dil_h", C.c_int64), ("gs_w_out_h", C.c_int64), ("gs_w_skipall_h", C.c_int64), ("gs_w_cond_h", C.c_int64),
           ("w_dil_x3", _vp * SS_MAX_LAYERS), ("gs_w_dil_x3", C.c_int64), ("w_dil_wino16", _vp * SS_MAX_LAYERS),
           ("w_out16", _vp * SS_MAX_LAYERS), ("gs_w_out16", C.c_int64), ("w_skipall_x3", _vp), ("gs_w_skipall_x3", C.c_int64),
           ("mfma_split", C.c_int32), ("mfma_out_scale", C.c_float),
           ("w_dil_q", _vp * SS_MAX_LAYERS), ("gs_w_dil_q", C.c_int64), ("q_scale_gate", C.c_float), ("q_scale_z", C.c_float),
           ("w_skipall_q", _vp), ("gs_w_skipall_q", C.c_int64),
           ("w_dil_f", _vp * SS_MAX_LAYERS), ("w_out_f", _vp * SS_MAX_LAYERS),
           ("n_wsets", C.c_int32), ("mfma_products", C.c_int32)] + [(n, C.c_int64) for n in ("ws_w_dil_h", "ws_w_out_h", "ws_w_skipall_h", "ws_w_dil_f", "ws_w_out_f")] \
        + [("w_skipall_c", _vp), ("ws_w_skipall_c", C.c_int64), ("n_esets", C.c_int32), ("reserved3_", C.c_int32)]


class GemmBf16Args(C.Structure):
    _fields_ = [
        ("A", _vp), ("a_batch_stride", C.c_int64), ("lda", C.c_int32), ("K", C.c_int32), ("ntaps", C.c_int32), ("tap_off", C.c_int32 * 4),
        ("lens", _vp), ("B", C.c_int32), ("T", C.c_int32), ("W", _vp), ("w_group_stride", C.c_int64), ("N", C.c_int32), ("Np", C.c_int32),
        ("epi", C.c_int32), ("act", C.c_int32), ("bias", _vp), ("bias_group_stride", C.c_int64), ("E", _vp), ("lde", C.c_int32),
        ("gate_mode", C.c_int32), ("e_batch_stride", C.c_int64), ("X", _vp), ("x_batch_stride", C.c_int64), ("ldx", C.c_int32),
        ("post_scale", C.c_float), ("next_bias", _vp), ("next_bias_group_stride", C.c_int64), ("Y", _vp), ("y_batch_stride", C.c_int64),
        ("ldy", C.c_int32), ("ldc", C.c_int32), ("C", _vp), ("c_batch_stride", C.c_int64), ("mask_rows", C.c_int32), ("group_size", C.c_int32),
        ("split", C.c_int32), ("out_scale", C.c_float), ("q_scale", C.c_float), ("one_product", C.c_int32), ("cur_bias", _vp), ("cur_bias_group_stride", C.c_int64),
        ("a_compact", C.c_int32), ("reserved2_", C.c_int32),
    ]


HEPI_STORE, HEPI_GATE, HEPI_RESX = 0, 1, 2


class Layer512Args(C.Structure):
    _fields_ = [
        ("Hin", _vp), ("d", C.c_int32), ("n_products", C.c_int32), ("Hout", _vp), ("P", _vp),
        ("lens", _vp), ("B", C.c_int32), ("T", C.c_int32), ("Wg", _vp), ("Wr", _vp), ("E512", _vp), ("G", _vp), ("g_batch_stride", C.c_int64),
        ("ldg", C.c_int32), ("mask_rows", C.c_int32), ("bias_r", _vp), ("next_bias", _vp), ("g_compact", C.c_int32), ("e_f16", C.c_int32), ("out_scale", C.c_float),
        ("post_scale", C.c_float), ("cur_bias", _vp),
    ]


class HifiGan(C.Structure):
    _fields_ = [
        ("n_ups", C.c_int32), ("n_kernels", C.c_int32), ("c0", C.c_int32), ("sr", C.c_int32), ("harmonics", C.c_int32),
        ("up_rate", C.c_int32 * SS_HG_MAX_UPS), ("up_k", C.c_int32 * SS_HG_MAX_UPS),
        ("rb_k", C.c_int32 * SS_HG_MAX_KERNELS), ("rb_d", (C.c_int32 * 3) * SS_HG_MAX_KERNELS),
        ("w_pre", _vp), ("b_pre", _vp),
        ("w_up", (_vp * 2) * SS_HG_MAX_UPS), ("b_up", _vp * SS_HG_MAX_UPS),
        ("w_noise", _vp * SS_HG_MAX_UPS), ("b_noise", _vp * SS_HG_MAX_UPS),
        ("w_rb1", ((_vp * 3) * SS_HG_MAX_KERNELS) * SS_HG_MAX_UPS), ("b_rb1", ((_vp * 3) * SS_HG_MAX_KERNELS) * SS_HG_MAX_UPS),
        ("w_rb2", ((_vp * 3) * SS_HG_MAX_KERNELS) * SS_HG_MAX_UPS), ("b_rb2", ((_vp * 3) * SS_HG_MAX_KERNELS) * SS_HG_MAX_UPS),
        ("w_post", _vp), ("b_post", _vp), ("src_w", _vp), ("src_b", _vp), ("mfma_bf16", C.c_int32), ("wino", C.c_int32),
        ("w_rb1_wino", ((_vp * 3) * SS_HG_MAX_KERNELS) * SS_HG_MAX_UPS), ("w_rb2_wino", ((_vp * 3) * SS_HG_MAX_KERNELS) * SS_HG_MAX_UPS),
    ]


class F0TrackParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("sample_rate", "time_step", "pitch_floor", "pitch_ceiling", "voicing_threshold", "silence_threshold",
                                          "octave_cost", "octave_jump_cost", "voiced_unvoiced_cost")] \
        + [(n, C.c_int32) for n in ("nsamp_window", "halfnsamp_window", "nsamp_period", "halfnsamp_period", "maximum_lag", "nlag", "hop", "reserved_")]


_CTYPE = {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "uint32_t": C.c_uint32,
          "float": C.c_float, "void": None}


def declarations():
    """{name: (restype, [argtypes])} parsed from the public header, so the binding cannot drift from it."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    out = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(ss_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        restype = C.c_char_p if "char" in ret else _CTYPE[ret]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(C.c_void_p)
                else:
                    base = [t for t in a.replace("const", " ").split() if t in _CTYPE]
                    assert base, f"cannot parse parameter '{a}' of {name}"
                    argtypes.append(_CTYPE[base[0]])
        out[name] = (restype, argtypes)
    return out


def declared_symbols():
    """Every `ss_*` function the public header declares."""
    return sorted(declarations())


_lib = None


def load():
    """Load the shared library (once). Raises if it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StyleSingerHipError(
            f"{LIB_PATH} not found: build it with `python -m stylesinger_amd.build` (hipcc, gfx950). "
            "The StyleSinger HIP path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in declarations().items():
        if not hasattr(lib, name):
            raise StyleSingerHipError(f"libstylesinger_hip.so does not export {name}")
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.ss_abi_version() != ABI_VERSION:
        raise StyleSingerHipError(f"libstylesinger_hip.so has ABI {lib.ss_abi_version()}, this binding expects {ABI_VERSION}: rebuild it")
    sizes = (C.c_int64 * 6)()
    if lib.ss_struct_sizes(sizes, 6) != 0:
        raise StyleSingerHipError("ss_struct_sizes failed")
    mine = (C.sizeof(ConvGemmArgs), C.sizeof(WaveNet), C.sizeof(HifiGan), C.sizeof(GemmBf16Args), C.sizeof(F0TrackParams), C.sizeof(Layer512Args))
    if tuple(sizes) != mine:
        raise StyleSingerHipError(f"ctypes mirror out of sync with include/stylesinger_hip.h: C={tuple(sizes)} py={mine}")
    _lib = lib
    for env, key in (("SS_WAVE_PRIO", b"wave_prio"), ("SS_GATE16", b"gate16"), ("SS_GATE16_KS", b"gate16_ks"), ("SS_GATE256", b"gate256"), ("SS_RES16", b"res16"), ("SS_SKIP16", b"skip16"), ("SS_RES_TILE", b"res_tile"), ("SS_SKIP_TILE", b"skip_tile"), ("SS_E16", b"e16"), ("SS_MEL_TAIL", b"mel_tail"), ("SS_HTILE", b"htile"),
                     ("SS_WINO_TN", b"wino_tn"), ("SS_WINO_V1", b"wino_v1"), ("SS_GATE128", b"gate128"), ("SS_Q4_FORCE", b"q4_force"), ("SS_SKIP_DENSE", b"skip_dense"), ("SS_LAYER512", b"layer512"), ("SS_LAYER512_TAIL", b"layer512_tail")):
        val = os.environ.get(env)
        if val is not None:   # validated by the library; an out-of-range value is an error, not a silent different tile
            check(lib.ss_set_tuning(key, int(val)), f"ss_set_tuning({env}={val})")
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ss_last_error().decode(errors="replace")
        raise StyleSingerHipError(f"{what} failed ({rc}): {msg}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a torch tensor (or None) as a plain int."""
    if t is None:
        return None
    assert t.is_cuda, "HIP path needs device tensors"
    return t.data_ptr()


def hptr(a):
    """Host pointer of a contiguous numpy array."""
    return a.ctypes.data


def _f(x):
    return float(x)


def round_up(x, m):
    return (x + m - 1) // m * m


# ---------------------------------------------------------------------------------------------
# thin op wrappers (argument marshalling only)
# ---------------------------------------------------------------------------------------------
def conv_gemm(A, W, out, **kw):
    a = _fill_args(A, W, out, **kw)
    check(load().ss_conv_gemm(C.byref(a), stream_ptr()), "ss_conv_gemm")


def _fill_args(A, W, out, *, B, T, Cin, N, Np, Kp, lda=None, a_bs=None, taps=(0,), lens=None, a_bias=None, a_scale=1.0,
               a_lrelu=1.0, epi=EPI_STORE, bias=None, pre_scale=1.0, act=ACT_NONE, act_slope=0.0, E=None, lde=0, e_bs=0,
               gate_mode=0, R=None, ldr=0, r_bs=None, post_scale=1.0, accumulate=False, mask_rows=True, ldc=None, c_bs=None,
               C2=None, ldc2=0, c2_bs=0, Nh=0, tile=0, group_size=0, w_gs=0, bias_gs=0, a_bias_gs=0, bf16=False, e_tiled=False):
    a = ConvGemmArgs()
    a.A = ptr(A); a.lda = lda if lda is not None else Cin
    a.a_batch_stride = a_bs if a_bs is not None else T * a.lda
    a.Cin = Cin; a.ntaps = len(taps)
    for i, o in enumerate(taps):
        a.tap_off[i] = int(o)
    a.lens = ptr(lens); a.B = B; a.T = T; a.a_bias = ptr(a_bias); a.a_scale = a_scale; a.a_lrelu = a_lrelu
    a.W = ptr(W); a.N = N; a.Np = Np; a.Kp = Kp
    a.epi = epi; a.bias = ptr(bias); a.pre_scale = pre_scale; a.act = act; a.act_slope = act_slope
    a.E = ptr(E); a.lde = lde; a.e_batch_stride = e_bs; a.gate_mode = gate_mode
    a.R = ptr(R); a.ldr = ldr; a.r_batch_stride = r_bs if r_bs is not None else T * ldr
    a.post_scale = post_scale; a.accumulate = int(accumulate); a.mask_rows = int(mask_rows)
    a.C = ptr(out); a.ldc = ldc if ldc is not None else N
    a.c_batch_stride = c_bs if c_bs is not None else T * a.ldc
    a.C2 = ptr(C2); a.ldc2 = ldc2; a.c2_batch_stride = c2_bs; a.Nh = Nh; a.tile = tile
    a.group_size = group_size; a.w_group_stride = w_gs; a.bias_group_stride = bias_gs; a.a_bias_group_stride = a_bias_gs
    a.mfma_bf16 = 1 if bf16 else 0
    a.e_tiled = 1 if e_tiled else 0
    return a


def gate16_tile_addend(E, *, B, T, Np, lde, e_bs, dilation, mt):
    """E (this layer's Np packed columns, row stride lde) -> the addend in the 16x16x4 gate kernel's fetch order (ss_gate16_tile_addend)."""
    n = load().ss_gate16_tiled_floats(B, T, Np, int(dilation), int(mt))
    out = torch.empty(n, device=E.device, dtype=torch.float32)
    check(load().ss_gate16_tile_addend(ptr(E), lde, e_bs, ptr(out), B, T, Np, int(dilation), int(mt), stream_ptr()), "ss_gate16_tile_addend")
    return out


def wino_gate(A, Wt, out, *, dilation, **kw):
    """Winograd F(2,3) dilated conv + gate (ss_wino_gate); Wt = packed transformed weights (4 'taps')."""
    kw.setdefault("epi", EPI_GATE)
    a = _fill_args(A, Wt, out, **kw)
    check(load().ss_wino_gate(C.byref(a), int(dilation), stream_ptr()), "ss_wino_gate")


def wino43_gate(A, Wt, out, *, dilation, **kw):
    """Winograd F(4,3) dilated conv + gate (ss_wino43_gate); Wt = packed transformed weights (6 'taps')."""
    kw.setdefault("epi", EPI_GATE)
    a = _fill_args(A, Wt, out, **kw)
    check(load().ss_wino43_gate(C.byref(a), int(dilation), stream_ptr()), "ss_wino43_gate")


def pack_gemm16_weights(Wp, Kp):
    """packed weight rows [Np][Kp] (Np % 64 == 0) -> the same floats in the fetch order of ss_gemm16_res (ss_pack_gemm16_weights)."""
    Wp = Wp.contiguous().float()
    out = torch.empty_like(Wp)
    check(load().ss_pack_gemm16_weights(ptr(Wp), ptr(out), Wp.shape[0], Kp, stream_ptr()), "ss_pack_gemm16_weights")
    return out


def pack_gate16_weights(Wp, Kp):
    """packed F(4,3) weights [Np][6 * Kp] -> the same floats in the fetch order of the 16x16x4 gate kernel (ss_pack_gate16_weights)."""
    Wp = Wp.contiguous().float()
    out = torch.empty_like(Wp)
    check(load().ss_pack_gate16_weights(ptr(Wp), ptr(out), Wp.shape[0], Kp, stream_ptr()), "ss_pack_gate16_weights")
    return out


def wino43_gate16(A, Wt, out, *, dilation, mt=0, W16=None, **kw):
    """The F(4,3) gate on 16x16x4 MFMA tiles (ss_wino43_gate16); mt = 0 lets the library pick, 2 / 3 force the row-tile count.
    W16 = pack_gate16_weights(Wt): the weights in the kernel's fetch order (ss_wino43_gate16w)."""
    kw.setdefault("epi", EPI_GATE)
    a = _fill_args(A, Wt, out, **kw)
    if W16 is not None:
        check(load().ss_wino43_gate16w(C.byref(a), ptr(W16), int(dilation), int(mt), stream_ptr()), "ss_wino43_gate16w")
        return
    check(load().ss_wino43_gate16(C.byref(a), int(dilation), int(mt), stream_ptr()), "ss_wino43_gate16")


def gemm16_res(A, W, out, *, mt=0, W16=None, **kw):
    """Residual projection C = (R + A.W^T + bias) * post_scale on 16x16x4 tiles (ss_gemm16_res); same keyword arguments as conv_gemm.
    W16 = pack_gemm16_weights(first N rows of W): the weights in the kernel's fetch order (ss_gemm16_resw)."""
    a = _fill_args(A, W, out, **kw)
    if W16 is not None:
        check(load().ss_gemm16_resw(C.byref(a), ptr(W16), int(mt), stream_ptr()), "ss_gemm16_resw")
        return
    check(load().ss_gemm16_res(C.byref(a), int(mt), stream_ptr()), "ss_gemm16_res")


def gemm16_store(A, W, out, *, mt=0, **kw):
    """C = act(A.W^T + bias) on 16x16x4 tiles with both operands streamed by LDS-DMA (ss_gemm16_store)."""
    a = _fill_args(A, W, out, **kw)
    check(load().ss_gemm16_store(C.byref(a), int(mt), stream_ptr()), "ss_gemm16_store")


def gemm16_store_splitk(A, W, out, *, ksplit, mt=0, **kw):
    """ss_gemm16_store_splitk: the long-K GEMM with K split over `ksplit` workgroup slices + a fixed-order reduction (small launches)."""
    a = _fill_args(A, W, out, **kw)
    part = torch.empty(max(1, ksplit) * a.B * a.T * a.N, device=out.device, dtype=torch.float32)
    check(load().ss_gemm16_store_splitk(C.byref(a), int(mt), int(ksplit), ptr(part), stream_ptr()), "ss_gemm16_store_splitk")


def wino43_weight(w):
    """conv weight [Cout][Cin][3] (device) -> F(4,3)-transformed [Cout][Cin][6]."""
    w = w.contiguous().float()
    out = torch.empty(w.shape[0], w.shape[1], 6, device=w.device, dtype=torch.float32)
    check(load().ss_wino43_weight_transform(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "ss_wino43_weight_transform")
    return out


def wino43_group_weight(w):
    """conv weight [Cout][Cin][k] (device, k odd) -> [Cout][Cin][6 * ceil(k/3)]: the taps in groups of three (zero padded), every group
    F(4,3)-transformed - the B operand of ss_wino43_conv after ss_pack_conv_weight."""
    w = w.contiguous().float()
    Cout, Cin, k = w.shape
    G = (k + 2) // 3
    wp = torch.zeros(Cout, Cin, 3 * G, device=w.device, dtype=torch.float32)
    wp[..., :k] = w
    return torch.cat([wino43_weight(wp[..., 3 * g:3 * g + 3]) for g in range(G)], dim=-1).contiguous()


def wino43_conv(A, W, out, *, k, dilation, **kw):
    """Grouped Winograd F(4,3) conv (ss_wino43_conv); keyword arguments as conv_gemm (taps are implied by k and dilation)."""
    taps = [(j - (k - 1) // 2) * dilation for j in range(k)]
    a = _fill_args(A, W, out, taps=taps, **kw)
    check(load().ss_wino43_conv(C.byref(a), k, dilation, stream_ptr()), "ss_wino43_conv")
    return out


def split3_weights(Wp, Kp):
    """packed F(4,3) weights [Np][6 * Kp] fp32 -> Np * 18 * Kp bf16: every element as its three bf16 terms, in the fetch order of
    ss_wino43_gate16x ([n tile][wave][K chunk][component][plane][lane][8])."""
    Wp = Wp.contiguous().float()
    Np = Wp.shape[0]
    out = torch.empty(Np, 3 * Wp.shape[1], device=Wp.device, dtype=torch.bfloat16)
    check(load().ss_split3_weights(ptr(Wp), ptr(out), Np, Kp, stream_ptr()), "ss_split3_weights")
    return out


def split3_gemm16_weights(Wp, Kp):
    """packed weight rows [Np][Kp] fp32 -> their three bf16 terms in the fetch order of ss_gemm16x_store."""
    Wp = Wp.contiguous().float()
    Np = Wp.shape[0]
    out = torch.empty(int(load().ss_split3_gemm16_elems(Np, Kp)), device=Wp.device, dtype=torch.bfloat16)
    check(load().ss_split3_gemm16_weights(ptr(Wp), ptr(out), Np, Kp, stream_ptr()), "ss_split3_gemm16_weights")
    return out


def gemm16x_store(A, W, Wx, out, *, mt=0, **kw):
    """bf16x3 form of gemm16_store (ss_gemm16x_store); W only supplies Np / Kp for the argument struct."""
    a = _fill_args(A, W, out, **kw)
    check(load().ss_gemm16x_store(C.byref(a), ptr(Wx), int(mt), stream_ptr()), "ss_gemm16x_store")


def wino43_gate16x(A, Wx, out, *, dilation, mt=0, **kw):
    """bf16x3 form of the F(4,3) gate (ss_wino43_gate16x); Wx = split3_weights(packed F(4,3) weight); keyword arguments as wino43_gate16
    (w_gs, if given, in bf16 elements)."""
    kw.setdefault("epi", EPI_GATE)
    a = _fill_args(A, Wx, out, **kw)
    check(load().ss_wino43_gate16x(C.byref(a), ptr(Wx), dilation, mt, stream_ptr()), "ss_wino43_gate16x")
    return out


def wino_weight(w):
    """conv weight [Cout][Cin][3] (device) -> transformed [Cout][Cin][4]."""
    w = w.contiguous().float()
    out = torch.empty(w.shape[0], w.shape[1], 4, device=w.device, dtype=torch.float32)
    check(load().ss_wino_weight_transform(ptr(w), ptr(out), w.shape[0], w.shape[1], stream_ptr()), "ss_wino_weight_transform")
    return out


def pack_conv_weight(w, *, scale0=None, interleave_half=0, row_scale=1.0):
    """torch conv/linear weight [Cout, Cin(, k)] (device) -> packed [Np][k*Kp] device tensor."""
    if w.dim() == 2:
        w = w[:, :, None]
    w = w.contiguous().float()
    Cout, Cin, k = w.shape
    Kp = round_up(Cin, 32)
    Np = 2 * round_up(interleave_half, 32) if interleave_half else round_up(Cout, 32)
    dst = torch.empty(Np, k * Kp, device=w.device, dtype=torch.float32)
    check(load().ss_pack_conv_weight(ptr(w), ptr(scale0), ptr(dst), Cout, Cin, k, Np, Kp, interleave_half, _f(row_scale),
                                     stream_ptr()), "ss_pack_conv_weight")
    return dst


def weight_norm_scale(v, g):
    v = v.contiguous().float()
    g = g.contiguous().float().reshape(-1)
    rows = v.shape[0]
    out = torch.empty(rows, device=v.device, dtype=torch.float32)
    check(load().ss_weight_norm_scale(ptr(v), ptr(g), ptr(out), rows, v.numel() // rows, stream_ptr()), "ss_weight_norm_scale")
    return out


def pack_convtr_weight(w, scale0, u, group):
    w = w.contiguous().float()
    Cin, Cout, k = w.shape
    pad = (k - u) // 2
    nph = (u - pad) if group == 0 else pad
    Np = round_up(nph * Cout, 32)
    Kp = round_up(Cin, 32)
    dst = torch.empty(Np, 2 * Kp, device=w.device, dtype=torch.float32)
    check(load().ss_pack_convtr_weight(ptr(w), ptr(scale0), ptr(dst), Cin, Cout, k, u, group, Np, Kp, stream_ptr()),
          "ss_pack_convtr_weight")
    return dst


def pack_bias(b, *, b2=None, Np=None, interleave_half=0, repeat=1):
    b = b.contiguous().float()
    n = b.numel()
    if Np is None:
        Np = 2 * round_up(interleave_half, 32) if interleave_half else round_up(n * repeat, 32)
    dst = torch.empty(Np, device=b.device, dtype=torch.float32)
    check(load().ss_pack_bias(ptr(b), ptr(b2), ptr(dst), n, Np, interleave_half, repeat, stream_ptr()), "ss_pack_bias")
    return dst


def to_bf16(x, bias=None, lens=None):
    """fp32 device tensor [..., C] -> bf16 bits (torch.bfloat16 tensor, RNE) through ss_to_bf16 (B*T rows)."""
    x = x.contiguous().float()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(load().ss_to_bf16(ptr(x), ptr(bias), ptr(y), 1, rows, Cc, Cc, Cc, ptr(lens), 0, 0, stream_ptr()), "ss_to_bf16")
    return y


def split_bf16(x, bias=None, lens=None):
    """fp32 device tensor [..., C] -> [..., 2C] bf16 bits, pairs interleaved by 32: per 32-channel chunk the 32 hi = RNE(v) terms, then the 32
    mid = RNE(v - hi) terms (ss_split_bf16; the operand layout of ss_gemm_bf16_args.split)."""
    x = x.contiguous().float()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty(tuple(x.shape[:-1]) + (2 * Cc,), device=x.device, dtype=torch.bfloat16)
    check(load().ss_split_bf16(ptr(x), ptr(bias), ptr(y), 1, rows, Cc, Cc, 2 * Cc, ptr(lens), 0, 0, stream_ptr()), "ss_split_bf16")
    return y


def split_f16(x, bias=None, lens=None, scale=1.0):
    """fp32 device tensor [..., C] -> [..., 2C] fp16 bits in the same pairs-interleaved-by-32 layout: hi = RNE16(v), lo = RNE16(v - hi) of
    v = (x + bias) * scale (ss_split_f16; the operand layout of ss_gemm_bf16_args.split = 2, scale = the weights' power-of-two shift)."""
    x = x.contiguous().float()
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    y = torch.empty(tuple(x.shape[:-1]) + (2 * Cc,), device=x.device, dtype=torch.float16)
    check(load().ss_split_f16(ptr(x), ptr(bias), _f(scale), ptr(y), 1, rows, Cc, Cc, 2 * Cc, ptr(lens), 0, 0, stream_ptr()), "ss_split_f16")
    return y


_FP4_GRID = (0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)


def fp4_rne(v):
    """|v| <= 6 (float tensor) -> (code 0..7, value) of the nearest e2m1 magnitude, ties to the even mantissa (0.25 -> 0, 0.75 -> 1, 1.25 -> 1,
    1.75 -> 2, 2.5 -> 2, 3.5 -> 4, 5 -> 4): what v_cvt_scalef32_pk_fp4_f16 does (profiles/r04_ubench_cvt_fp4_probe.log)."""
    grid = torch.tensor(_FP4_GRID, device=v.device, dtype=v.dtype)
    mid = (grid[1:] + grid[:-1]) / 2
    a = v.abs().clamp(max=6.0).contiguous()
    idx = torch.bucketize(a, mid) + ((a == 0.75) | (a == 1.75) | (a == 3.5)).long()
    return idx, grid[idx]


def gate128q_kindex():
    """[12 pairs][2 lane halves][32 elements] -> K index (tap * 256 + channel): the element order of the fp16q4 gate's block-scaled second
    product (csrc/gate128_layout.h, g128q::q_kindex - the same function the kernel's layout is checked against)."""
    out = (C.c_int32 * 768)()
    n = load().ss_gate128q_kindex(out, 768)
    if n != 768:
        raise StyleSingerHipError("ss_gate128q_kindex: " + load().ss_last_error().decode(errors="replace"))
    return torch.tensor(list(out), dtype=torch.long).view(12, 2, 32)


def tile256q_kindex(n_pairs):
    """[n_pairs][2][32] -> K index: the element order of the fp16q4 skip GEMM's block-scaled second product (csrc/gate128_layout.h t128q_kindex)."""
    out = (C.c_int32 * (n_pairs * 64))()
    if load().ss_tile256q_kindex(out, n_pairs) != n_pairs * 64:
        raise StyleSingerHipError("ss_tile256q_kindex: " + load().ss_last_error().decode(errors="replace"))
    return torch.tensor(list(out), dtype=torch.long).view(n_pairs, 2, 32)


def _pack_q4(Wp, tab, odd_lines, shift):
    """shared by pack_gate_q4 / pack_skip_q4: the ss_split_f16 pack of w * 2^shift with its lo plane replaced by block-scaled fp4 terms; tab
    [pairs][2][32] = K index of every element, odd_lines [pairs] = the 64-element weight line that carries pair p's terms"""
    Np, K = Wp.shape
    nl = K // 32
    dev = Wp.device
    ws = Wp.float() * float(2 ** shift)
    if Wp.is_cuda:
        pack = split_f16(Wp, scale=float(2 ** shift))
    else:   # host form of ss_split_f16 (pairs interleaved by 32), so that the packer can be checked without a GPU
        hi = ws.half()
        lo16 = (ws - hi.float()).half()
        pack = torch.stack([hi.view(Np, nl, 32), lo16.view(Np, nl, 32)], dim=2).reshape(Np, 2 * K).contiguous()
    lo = ws - ws.half().float()
    npairs = tab.shape[0]
    tab = tab.to(dev)
    blocks = lo[:, tab.reshape(-1)].view(Np, npairs, 2, 32)
    amax = blocks.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -120))) - 2      # the block's largest element lands in [4, 8): the grid's top is 6
    e = torch.where(amax > 0, e, torch.full_like(e, -120.0))
    scale = torch.exp2(e)
    idx, mag = fp4_rne(blocks / scale)
    code = idx | ((blocks < 0).long() << 3)
    nib = code.view(Np, npairs, 2, 16, 2)
    qbytes = (nib[..., 0] | (nib[..., 1] << 4)).to(torch.uint8)       # element e in nibble e & 1 of byte e >> 1
    sbyte = (e.squeeze(-1) + 127).clamp(0, 254).to(torch.uint8)       # [Np][pairs][2]
    wb = pack.view(torch.uint8).view(Np, nl, 128)
    wb[:, :, 64:] = 0                                                 # nothing in the second half of any line ...
    line = torch.as_tensor(odd_lines, device=dev, dtype=torch.long)   # ... except the line that closes every pair
    for h in range(2):
        wb[:, line, 64 + 16 * h:64 + 16 * h + 16] = qbytes[:, :, h]
        wb[:, line, 96 + h] = sbyte[:, :, h]
    lo_q = torch.zeros_like(lo)
    lo_q[:, tab.reshape(-1)] = (mag * torch.sign(blocks) * scale).view(Np, -1)
    return pack, lo_q


def pack_skip_q4(Wp, *, shift=8):
    """Packed fp32 1-tap weight [Np][K] (K a multiple of 64) -> the operand of ss_gemm_bf16_tile256q (split = 3): as pack_gate_q4, pairs =
    consecutive 32-channel chunks, the pair's fp4 terms in the line of its odd chunk. Returns (pack [Np][2 K] fp16 bits, lo_q [Np][K] fp32)."""
    Np, K = Wp.shape
    assert K % 64 == 0
    return _pack_q4(Wp, tile256q_kindex(K // 64), [2 * p + 1 for p in range(K // 64)], shift)


def pack_gate_q4(Wp, *, shift=8):
    """Packed fp32 gate weight [Np][3 * 256] (ss_pack_conv_weight, gate-interleaved rows, tap-major) -> the operand of ss_gemm_bf16_gate128q
    (ss_gemm_bf16_args.split = 3): the ss_split_f16 pack of w * 2^shift whose LO plane is replaced by the fp4 (e2m1) terms of
    lo = w * 2^shift - fp16(w * 2^shift) in the kernel's lane order with one E8M0 scale per 32-element block (block = lane half h of step pair p).
    Returns (pack [Np][2 * 768] fp16 bits, lo_q [Np][768] fp32 = the values the matrix cores will see for lo, in K order - for references)."""
    Np, K3 = Wp.shape
    assert K3 == 768, "the fp16q4 gate is built for K = 256, three taps"
    S_odd = [2 * p + 1 for p in range(12)]
    return _pack_q4(Wp, gate128q_kindex(), [(S % 3) * 8 + S // 3 for S in S_odd], shift)   # g128q::step_line of the pair's odd step


def unpack_skip_q4(pack):
    """the ss_gemm_bf16_tile256q view of a pack_skip_q4 pack: (hi [Np][K], lo_q [Np][K]) in K order"""
    Np, K = pack.shape[0], pack.shape[1] // 2
    nl = K // 32
    wb = pack.contiguous().view(torch.uint8).view(Np, nl, 128)
    hi = pack.view(Np, nl, 64)[:, :, :32].float().reshape(Np, K)
    tab = tile256q_kindex(K // 64).to(pack.device)
    grid = torch.tensor(_FP4_GRID, device=pack.device)
    lo_q = torch.zeros(Np, K, device=pack.device)
    for p in range(K // 64):
        for h in range(2):
            by = wb[:, 2 * p + 1, 64 + 16 * h:64 + 16 * h + 16].long()
            nib = torch.stack([by & 15, by >> 4], dim=-1).reshape(Np, 32)
            lo_q[:, tab[p, h]] = grid[nib & 7] * torch.where((nib & 8) != 0, -1.0, 1.0) * torch.exp2(wb[:, 2 * p + 1, 96 + h].float() - 127.0).unsqueeze(-1)
    return hi, lo_q


def unpack_gate_q4(pack):
    """What ss_gemm_bf16_gate128q reads from a pack_gate_q4 pack, decoded the way the kernel addresses it (csrc/gate128_layout.h): for packed
    column n, step pair p, lane half h the 16 bytes at logical slot 4 + h of the weight line of step 2 p + 1 are 32 e2m1 nibbles (element e in
    nibble e & 1 of byte e >> 1), byte h of slot 6 their E8M0 scale. Returns (hi [Np][768] fp32 in K order, lo_q [Np][768] fp32 in K order)."""
    Np = pack.shape[0]
    wb = pack.contiguous().view(torch.uint8).view(Np, 24, 128)
    hi = pack.view(Np, 24, 64)[:, :, :32].float().reshape(Np, 768)
    tab = gate128q_kindex().to(pack.device)
    grid = torch.tensor(_FP4_GRID, device=pack.device)
    lo_q = torch.zeros(Np, 768, device=pack.device)
    for p in range(12):
        S = 2 * p + 1
        line = (S % 3) * 8 + S // 3
        for h in range(2):
            by = wb[:, line, 64 + 16 * h:64 + 16 * h + 16].long()                  # [Np][16]
            nib = torch.stack([by & 15, by >> 4], dim=-1).reshape(Np, 32)             # element e = 2 * byte + nibble
            val = grid[nib & 7] * torch.where((nib & 8) != 0, -1.0, 1.0) * torch.exp2(wb[:, line, 96 + h].float() - 127.0).unsqueeze(-1)
            lo_q[:, tab[p, h]] = val
    return hi, lo_q


def split_planes(y):
    """[..., 2C] pairs-interleaved-by-32 bf16 / fp16 -> (hi [..., C], mid [..., C]) as float (host-side view for tests / debugging)."""
    v = y.float().reshape(*y.shape[:-1], y.shape[-1] // 64, 2, 32)
    return v[..., 0, :].reshape(*y.shape[:-1], -1), v[..., 1, :].reshape(*y.shape[:-1], -1)


def gemm_bf16(A, Wh, *, B, T, K, taps, N, Np, epi, lens=None, bias=None, act=ACT_NONE, E=None, lde=0, e_bs=None, X=None, post_scale=1.0,
              next_bias=None, Y=None, out=None, ldc=None, c_bs=None, lda=None, a_bs=None, mask_rows=True, gate_mode=0, gate256=False,
              split=0, cur_bias=None, out_scale=1.0, q_scale=0.0, one_product=False, a_compact=False):
    """ss_gemm_bf16: A, Wh = bf16 device tensors (A [B,T,lda], Wh packed [Np][len(taps)*K]); see include/stylesinger_hip.h."""
    a = GemmBf16Args()
    a.A = ptr(A); a.lda = lda if lda is not None else A.shape[-1]
    a.a_batch_stride = a_bs if a_bs is not None else T * a.lda
    a.K = K; a.ntaps = len(taps)
    for i, o in enumerate(taps):
        a.tap_off[i] = int(o)
    a.lens = ptr(lens); a.B = B; a.T = T; a.W = ptr(Wh); a.N = N; a.Np = Np; a.epi = epi; a.act = act
    a.bias = ptr(bias); a.E = ptr(E); a.lde = lde; a.e_batch_stride = e_bs if e_bs is not None else T * lde; a.gate_mode = gate_mode
    a.X = ptr(X)
    if X is not None:
        a.ldx = X.shape[-1]; a.x_batch_stride = T * a.ldx
    a.post_scale = post_scale; a.next_bias = ptr(next_bias); a.Y = ptr(Y)
    if Y is not None:
        a.ldy = Y.shape[-1]; a.y_batch_stride = T * a.ldy
    a.C = ptr(out); a.ldc = ldc if ldc is not None else (out.shape[-1] if out is not None else 0)
    a.c_batch_stride = c_bs if c_bs is not None else T * a.ldc
    a.mask_rows = int(mask_rows)
    a.split = split
    a.out_scale = out_scale
    a.q_scale = q_scale
    a.one_product = int(one_product)
    a.a_compact = int(a_compact)
    a.cur_bias = ptr(cur_bias)
    if gate256 and epi == HEPI_STORE and split == 3:   # the fp16q4 skip GEMM
        check(load().ss_gemm_bf16_tile256q(C.byref(a), stream_ptr()), "ss_gemm_bf16_tile256q")
        return
    if gate256 == 128 and epi == HEPI_GATE and split == 3:   # ... with the second product on the block-scaled fp4 instruction
        check(load().ss_gemm_bf16_gate128q(C.byref(a), stream_ptr()), "ss_gemm_bf16_gate128q")
        return
    if gate256 == 128 and epi == HEPI_GATE:   # the fp16x2 gate on 256 x 128 tiles, two workgroups per CU
        check(load().ss_gemm_bf16_gate128(C.byref(a), stream_ptr()), "ss_gemm_bf16_gate128")
        return
    if gate256:   # the 256-row LDS-DMA kernels directly (ss_gemm_bf16 picks them by itself for many-round launches)
        if epi == HEPI_GATE:
            check(load().ss_gemm_bf16_gate256(C.byref(a), stream_ptr()), "ss_gemm_bf16_gate256")
        else:
            check(load().ss_gemm_bf16_tile256(C.byref(a), stream_ptr()), "ss_gemm_bf16_tile256")
        return
    check(load().ss_gemm_bf16(C.byref(a), stream_ptr()), "ss_gemm_bf16")


def layer512_pack_gate(w_pairs, n_products=2):
    """ss_split_f16 pack [512][3*256*2] of the gate-interleaved dilated-conv weights -> the fragment order ss_layer512 streams (fp16 [786432];
    n_products = 1: the hi terms only, [393216])."""
    assert w_pairs.dtype == torch.float16 and tuple(w_pairs.shape) == (512, 3 * 256 * 2) and w_pairs.is_contiguous(), tuple(w_pairs.shape)
    out = torch.empty(8 * 48 * 2 * n_products * 64 * 8, device=w_pairs.device, dtype=torch.float16)
    check(load().ss_layer512_pack_gate(ptr(w_pairs), ptr(out), n_products, stream_ptr()), "ss_layer512_pack_gate")
    return out


def layer512_pack_res(w_pairs, n_products=2):
    """ss_split_f16 pack [>= 256][256*2] of the output projection (first 256 rows = residual half) -> fragment order (fp16 [131072]; n_products = 1: [65536])."""
    assert w_pairs.dtype == torch.float16 and w_pairs.shape[0] >= 256 and w_pairs.shape[1] == 512 and w_pairs.is_contiguous(), tuple(w_pairs.shape)
    out = torch.empty(8 * 16 * n_products * 64 * 8, device=w_pairs.device, dtype=torch.float16)
    check(load().ss_layer512_pack_res(ptr(w_pairs), ptr(out), n_products, stream_ptr()), "ss_layer512_pack_res")
    return out


def layer512_tile_addend(E, *, B, T, lde=None, e_bs=None, out=None):
    """E fp32 [B][T][lde] (the layer's 512 packed addend columns start at E) -> the tiled slab ss_layer512 reads."""
    lde = lde if lde is not None else E.shape[-1]
    n = load().ss_layer512_addend_floats(B, T)
    out = torch.empty(n, device=E.device, dtype=torch.float32) if out is None else out
    check(load().ss_layer512_tile_addend(ptr(E), lde, e_bs if e_bs is not None else T * lde, ptr(out), B, T, stream_ptr()), "ss_layer512_tile_addend")
    return out


def layer512_tile_addend_f16(E, n_sets, *, B, T, lde=None, e_bs=None):
    """E fp32 [B][T][lde] -> fp16 [n_sets][ss_layer512_addend_halfs]: the addend slab as sigma-delta sets (ss_layer512 with e_f16 reads ONE of them per launch)."""
    lde = lde if lde is not None else E.shape[-1]
    n = load().ss_layer512_addend_halfs(B, T)
    out = torch.empty(n_sets, n, device=E.device, dtype=torch.float16)
    check(load().ss_layer512_tile_addend_f16(ptr(E), lde, e_bs if e_bs is not None else T * lde, ptr(out), n_sets, n, B, T, stream_ptr()), "ss_layer512_tile_addend_f16")
    return out


def layer512_addend_values(E512, *, B, T, f16=False):
    """a tiled addend slab (fp32 form, or ONE fp16 set) -> [B][T][512] in the packed column order, as the gate's exp2 arguments (test helper)"""
    nt = (T + 127) // 128
    if f16:
        v = E512.view(B * nt, 4, 4, 8, 2, 32, 2, 4).float()            # tile, m, q, wave, lh, l31, nb, e
        v = v.permute(0, 1, 5, 3, 6, 2, 4, 7)                          # tile, m, l31, wave, nb, q, lh, e
    else:
        v = E512.view(B * nt, 2, 4, 4, 8, 2, 32, 4)                    # tile, nb, m, q, wave, lh, l31, e
        v = v.permute(0, 2, 6, 4, 1, 3, 5, 7)                          # tile, m, l31, wave, nb, q, lh, e
    return v.reshape(B, nt * 128, 512)[:, :T]


def layer512_entry(X, bias=None, *, B, T, lens=None):
    """ss_layer512_entry: X fp32 [B][T][256] -> (H = fp16(X + bias) in slot-major tiles (fp16 [elems]), P = X in accumulator order (uint8 buffer of fp32))"""
    H = torch.empty(load().ss_layer512_h_elems(B, T), device=X.device, dtype=torch.float16)
    P = torch.empty(load().ss_layer512_stream_bytes(B, T), device=X.device, dtype=torch.uint8)
    check(load().ss_layer512_entry(ptr(X), X.shape[-1], T * X.shape[-1], ptr(bias), ptr(lens), ptr(H), ptr(P), B, T, stream_ptr()), "ss_layer512_entry")
    return H, P


def layer512_h_values(H, *, B, T):
    """H (slot-major tiles [tile][slot 32][row 128][8]) -> fp16 [B][T][256] (test helper)"""
    nt = (T + 127) // 128
    return H.view(B * nt, 32, 128, 8).permute(0, 2, 1, 3).reshape(B, nt * 128, 256)[:, :T]


def layer512_stream_values(P, H, bias=None, *, B, T, lens=None):
    """the stream x = (H - bias) + R from its two terms: H (slot-major tiles, = fp16(x + bias)) and P = R, the fp16 remainder in accumulator order
    -> fp32 [B][T][256], rows >= lens[b] zero (test helper: the index map and the arithmetic of include/stylesinger_hip.h)"""
    nt = (T + 127) // 128
    v = P.view(torch.float16).view(B * nt, 4, 4, 8, 2, 32, 4).float()  # tile, m, q, wave, lh, l31, e
    r = v.permute(0, 1, 5, 3, 2, 4, 6).reshape(B, nt * 128, 256)[:, :T]   # tile, (m, l31) = row, (wave, q, lh, e) = channel
    h = layer512_h_values(H, B=B, T=T).float()
    x = (h - bias) + r if bias is not None else h + r
    if lens is not None:
        x = x * (torch.arange(T, device=x.device)[None, :, None] < lens.to(x.device)[:, None, None])
    return x


def layer512(Hin, Wg, E512, G, *, B, T, d, lens=None, Hout=None, P=None, Wr=None, bias_r=None, next_bias=None, out_scale=1.0 / 256.0,
             post_scale=0.70710678118654752440, ldg=None, g_bs=None, mask_rows=True, n_products=2, g_compact=False, e_f16=False, cur_bias=None):
    """ss_layer512: one launch per residual layer (gate + residual projection) of the fp16x2 mel denoiser; see include/stylesinger_hip.h."""
    a = Layer512Args()
    a.Hin = ptr(Hin); a.d = d; a.n_products = n_products; a.g_compact = int(g_compact); a.e_f16 = int(e_f16)
    a.Hout = ptr(Hout); a.P = ptr(P)
    a.lens = ptr(lens); a.B = B; a.T = T; a.Wg = ptr(Wg); a.Wr = ptr(Wr); a.E512 = ptr(E512)
    a.G = ptr(G); a.ldg = ldg if ldg is not None else G.shape[-1]; a.g_batch_stride = g_bs if g_bs is not None else T * a.ldg
    a.mask_rows = int(mask_rows); a.bias_r = ptr(bias_r); a.next_bias = ptr(next_bias)
    a.out_scale = out_scale; a.post_scale = post_scale; a.cur_bias = ptr(cur_bias)
    check(load().ss_layer512(C.byref(a), stream_ptr()), "ss_layer512")


def layernorm(x, gamma, beta, *, B, T, C_, out=None, lens=None, mask_rows=False, eps=1e-5):
    out = x if out is None else out
    check(load().ss_layernorm(ptr(x), ptr(out), ptr(gamma), ptr(beta), B, T, C_, C_, C_, T * C_,
                              T * C_, _f(eps), ptr(lens), int(mask_rows), stream_ptr()), "ss_layernorm")
    return out


def attention(Q, K, V, O, *, B, H, D, Tq, Tk, ldq, ldk, ldv, ldo, q_bs, k_bs, v_bs, o_bs, qlens=None, klens=None, scale):
    check(load().ss_attention(ptr(Q), ptr(K), ptr(V), ptr(O), B, H, D, Tq, Tk, ldq, ldk, ldv, ldo, q_bs,
                              k_bs, v_bs, o_bs, ptr(qlens), ptr(klens), _f(scale),
                              stream_ptr()), "ss_attention")
