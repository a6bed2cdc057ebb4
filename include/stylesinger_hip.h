/*
 * libstylesinger_hip — C-ABI boundary of the MI355X-native StyleSinger inference hot path.
 *
 * The reference (AaronZ345/StyleSinger) has no FFI: its operator surface is the Python classes
 * `modules/StyleSinger/stylesinger.py::StyleSinger` and the vocoder plugin
 * `tasks/tts/vocoder_infer/hifigan_nsf.py::HifiGAN`.  This header is what a
 * ctypes binding for that path binds instead of the stock torch ops; every entry point cites the
 * reference code it replaces.  `stylesinger_amd/lib.py` is that binding.
 *
 * Conventions
 *   - plain C, raw device pointers + sizes, no torch types; the caller (PyTorch) owns every buffer;
 *   - every launch goes to the `hipStream_t` passed in (as `void*`), nothing synchronises, nothing
 *     allocates, so every call is hipGraph-capturable;
 *   - return 0 on success, <0 on error; `ss_last_error()` gives the message (thread-local);
 *   - activations are fp32, channels-last: `[B][T_pad][C]` with `lens[b]` valid rows per item
 *     (rows >= lens[b] behave as the zero padding a B=1 reference run would see);
 *   - weights are packed once (ss_pack_*) into the K-contiguous `[N][taps*Kp]` layout the MFMA
 *     kernel streams.
 */
#ifndef STYLESINGER_HIP_H
#define STYLESINGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 19 /* 19: the residual stream of ss_layer512 as (H, fp16 remainder) instead of an fp32 copy (ss_layer512_args.cur_bias; P halves); 18: fp16 addend sets of ss_layer512 (ss_layer512_args.e_f16, ss_layer512_tile_addend_f16, ss_layer512_addend_halfs, ss_wavenet.n_esets), knob skip_dense; 17: compact gate rows (ss_layer512_args.g_compact, ss_gemm_bf16_args.a_compact), compact one-term skip weights (ss_gemm_bf16_args.one_product = 2, ss_wavenet.w_skipall_c); 16: ss_layer512 (one launch per residual layer of the fp16x2 mel denoiser: gate + residual projection with G kept in LDS), ss_layer512_pack_gate / _pack_res / _tile_addend, ss_wavenet.w_dil_f / w_out_f, knob layer512, ss_round_f16_rows takes the items' own lengths, ss_set_q4_guard, ss_mel_denorm reports non-finite frames, "fp16sd" (ss_wavenet.n_wsets / mfma_products / ws_*, ss_layer512_args.n_products); 15: ss_f0track (f0 tracker of the input producers), ss_f0track_params, ss_vad_trim, ss_normalize_volume, ss_round_f16_rows, knob q4_force; removed ss_gemm_bf16_tile128 and the knobs tile128 / skip_deep (measured: no gain); 14: fp16q4 gate (ss_gemm_bf16_args.split = 3 + q_scale, ss_gemm_bf16_gate128q, ss_gate128q_kindex), ss_gemm_bf16_gate128 / _tile128, tuning knobs gate128 / tile128 / skip_deep; 13: fp16x2 mode (ss_gemm_bf16_args.split = 2 + out_scale, ss_split_f16, ss_wavenet.mfma_split = 2 + mfma_out_scale); 12: bf16x2 split-operand mode (ss_gemm_bf16_args.split, ss_split_bf16, ss_wavenet.mfma_split), DDIM eta + double schedule, tuning knob table; 2: grouped launches + Winograd weights; 3: mfma_bf16 fields, samplers, front end, writer; 4: deferred skip; 5: folded skip projection; 6: PLMS step_hi, per-item Philox counters, ss_fill_normal_rows, ProDiff sampler, emotion LSTM; 7: bf16-in-HBM GEMM + bf16 weight copies; 9: input producers (ss_norm_interp_f0, ss_spec_power, ss_reflect_pad), ss_wino43_gate16, ss_gemm16_res; 10: ss_wino43_conv + Winograd packs in ss_hifigan; 11: bf16x3 mode (ss_wino43_gate16x, ss_split3_weights, ss_wavenet.w_dil_x3), ss_wino43_gate16w + ss_pack_gate16_weights */
#define SS_MAX_TAPS 16
#define SS_MAX_LAYERS 32

const char* ss_last_error(void);
int ss_abi_version(void);
/* number of compute units / name of device `dev` (sanity: must be gfx950) */
int ss_device_info(int dev, int* n_cu, char* arch, int arch_len);
/* out[0..2] = sizeof(ss_conv_gemm_args), sizeof(ss_wavenet), sizeof(ss_hifigan) (+ out[3] = sizeof(ss_gemm_bf16_args) when n >= 4):
 * lets a binding verify its mirror */
int ss_struct_sizes(int64_t* out, int n);
/* process-wide performance knobs (results never change): "wave_prio" = 0|1|2 static per-workgroup wave priority in the MFMA
 * kernels (0 = none, 1 = (blockIdx/256)%3, 2 = blockIdx%3); "gate16" = 0|1|2|3 tiling of the F(4,3) gate launches inside the
 * denoiser loops (1 = per-launch pick, default; 0 = 32x32x2 tiles; 2|3 = force 16x16x4 tiles of 16*MT quads); "gate16_ks" = 0|1
 * the 16x16x4 gate kernel stages one Winograd component per barrier (0) or all six of a K chunk at once (1, default); "res_tile" /
 * "skip_tile" = SS_TILE_* override for the residual-half projection / the K = L*C skip GEMM (0 = built-in choice); "gate256" = 0|1 bf16 GATE launches on the 256x256 LDS-DMA kernel when they
 * qualify (default 1); "res16" / "skip16" = 0|1|4|6|8
 * residual-half projection on ss_gemm16_res / skip GEMM on ss_gemm16_store (1 = on, row tile picked per launch, default; 0 =
 * ss_conv_gemm; 4|6|8 = force 16*mt rows); experiment switches "htile" = 0|64|128 (row tile of the generic bf16 kernel), "wino_tn" = 0|1|2 and
 * "wino_v1" = 0|1 (F(2,3) gate: column tile, round-1 kernel); "voc_wino_max_mb" = 1..2048: vocoder items whose stage panel reaches this many MiB
 * take the direct conv kernel instead of the grouped-Winograd one (32-bit offsets; default 2048 = the real limit, tests lower it); "e16" = 0|1
 * the fp32 denoiser loops hand the 16x16x4 gate its conditioner addend in fetch order (ss_gate16_tile_addend once per forward; default 1);
 * "mel_tail" = 0|1 small launches (<= 8 frames per CU) run the mel sampler's output projection + update + next input projection as one launch;
 * "gate128" = 0|1 fp16x2 GATE launches of very many tiles on ss_gemm_bf16_gate128 (two workgroups per CU; default 1); "q4_force" = 0|1 the
 * fp16q4 kernels (ss_gemm_bf16_gate128q / _tile256q) take any launch they can compute, not only those that fill the chip (default 0; the parity
 * tests run one 30 s item through them); "layer512" = 0|1|2 the fp16x2 mel stack as one ss_layer512 launch per layer when the shape qualifies
 * (default 1; 0 = the gate + residual-projection launch pair; 2 = also for launches that do not fill the chip: the parity tests run one item through it); "layer512_tail" = 0|1|2 ss_layer512 runs the tiles of
 * an under-filled last round as half tiles (default 1: the even workgroups take their half tile FIRST, which puts the two halves of the chip half a tile period out of phase - one
 * streams through HBM while the other multiplies; 2: every half tile last; 0: whole tiles only; identical results); "skip_dense" = 0|1 (default 1) the skip GEMM with both operands compact (a_compact and one_product = 2) runs 64 channels per
 * step with every DMA lane live; 0 keeps the 32-channel steps (identical results). The library reads NO environment variable: a direct C caller sets knobs here (the Python binding
 * forwards SS_* variables once at load). */
int ss_set_tuning(const char* key, int value);
/* current value of a tuning knob (>= 0), or < 0 for an unknown key */
int ss_get_tuning(const char* key);
/* Measurement aid (bench.py's roofline block): while `dev_u64x2` is non-null, wave 0 of workgroup 0 of every Winograd gate launch
 * adds its lifetime to dev_u64x2[0] in shader cycles (s_memtime) and to dev_u64x2[1] in ticks of the constant 100 MHz counter
 * (s_memrealtime): [0] / [1] / 10 = the shader clock in GHz the chip sustained under that load (it clocks to its power budget:
 * 2.4 GHz nominal, ~2.0 GHz measured in these loops). Pass NULL to switch it off. Results never change. */
int ss_set_clock_probe(void* dev_u64x2);
/* Range guard of the "fp16q4" precision: its kernels convert their fp16 A operand to fp4 with a FIXED power-of-two scale (q_scale), which
 * saturates at |a| > 6 q_scale. While dev_u32x2 is non-null every ss_gemm_bf16_gate128q / ss_gemm_bf16_tile256q call first reduces
 * max |a| / (6 q_scale) over the operand it is about to read into dev_u32x2[0] (gate) / [1] (skip GEMM) as float bits (atomicMax; the caller
 * zeroes the words): a value > 1 means the second product of that launch is degraded - use "fp16x2". One extra HBM pass per guarded launch:
 * the model arms it for the first (eager) forward of every plan only. Pass NULL to switch it off. Results never change. */
int ss_set_q4_guard(void* dev_u32x2);
/* Measurement aid (tools/launch_floor.py): an EMPTY kernel launched with the given geometry - a captured chain of these with the launch count and
 * grid sizes of a diffusion loop is the floor the loop's dependent launch edges cost, whatever the kernels do. */
int ss_debug_null_launch(int grid, int block, void* stream);

/* ------------------------------------------------------------------------------------------
 * Generic fp32-MFMA implicit-GEMM 1-D convolution / linear layer.
 *   out[b][t][n] = epilogue( sum_{j<ntaps} sum_{ci<Cin} A'[b][t+tap_off[j]][ci] * W[n][j][ci] )
 *   A' = lrelu((A + a_bias) * a_scale) for rows inside [0, lens[b]), 0 outside (zero padding).
 * Replaces torch conv1d / conv_transpose1d (polyphase) / linear / addmm at every call site of
 * the hot path, e.g. modules/diff/net.py:61-64, modules/hifigan/hifigan_nsf.py:33-47,
 * modules/commons/common_layers.py:548-582.
 * ------------------------------------------------------------------------------------------ */
enum {
  SS_EPI_STORE = 0,   /* v=(acc+bias)*pre_scale; act; +R; *post_scale; (+=C); row mask          */
  SS_EPI_GATE = 1,    /* paired 32-col blocks: gate_mode 0: sigmoid(v0)*tanh(v1) (net.py:72-73);   *
                       *                       gate_mode 1: tanh(v0)*sigmoid(v1) (wavenet.py:6-11) */
  SS_EPI_RESSKIP = 2, /* n<Nh: C=(R+v)*post_scale ; n>=Nh: C2 (+)= v  (net.py:75-77)              */
  SS_EPI_DDPM = 3     /* v = predicted noise -> fused DDPM posterior step on C (shallow_diffusion_tts.py:130-162) */
};
enum { SS_ACT_NONE_ = 0, SS_ACT_RELU_ = 1, SS_ACT_GELU_ = 2, SS_ACT_MISH_ = 3, SS_ACT_TANH_ = 4, SS_ACT_LRELU_ = 5 };

/* (the 96x256 / 96x128 / 64x256 / 256x32 / 256x64 / 2-wave / 8-wave tiles measured in round 1 were all slower at this
 * path's shapes and are no longer built - DESIGN.md §6) */
enum { SS_TILE_AUTO = 0, SS_TILE_128x128 = 1, SS_TILE_64x128 = 2, SS_TILE_64x64 = 3, SS_TILE_128x64 = 4, SS_TILE_128x32 = 5 };

typedef struct ss_conv_gemm_args {
  /* A operand */
  const float* A;
  int64_t a_batch_stride; /* floats */
  int32_t lda;            /* floats, multiple of 4 */
  int32_t Cin;            /* multiple of 4 */
  int32_t ntaps;
  int32_t tap_off[SS_MAX_TAPS];
  const int32_t* lens; /* [B] or NULL (= T) */
  int32_t B, T;
  const float* a_bias; /* [Cin] or NULL */
  float a_scale;       /* 1.0 = none */
  float a_lrelu;       /* slope, 1.0 = none */
  /* B operand (packed by ss_pack_conv_weight) */
  const float* W;
  int32_t N;  /* logical columns written */
  int32_t Np; /* packed rows, multiple of 32 */
  int32_t Kp; /* packed channels per tap, multiple of 32 */
  /* epilogue */
  int32_t epi;
  const float* bias; /* [Np] packed order or NULL */
  float pre_scale;
  int32_t act;
  float act_slope;
  const float* E; /* GATE: pre-activation addend [B][T][lde] in packed column order, or NULL */
  int32_t lde;
  int64_t e_batch_stride;
  int32_t gate_mode;
  const float* R; /* STORE/RESSKIP: residual [B][T][ldr] */
  int32_t ldr;
  int64_t r_batch_stride;
  float post_scale;
  int32_t accumulate; /* STORE: C += ; RESSKIP: C2 += */
  int32_t mask_rows;  /* write 0 to rows >= lens[b] */
  float* C;
  int32_t ldc;
  int64_t c_batch_stride;
  float* C2; /* RESSKIP second half */
  int32_t ldc2;
  int64_t c2_batch_stride;
  int32_t Nh; /* RESSKIP split point */
  /* DDPM epilogue (C is x_t in/out, [B][T][N]) */
  float ddpm_recip, ddpm_recipm1, ddpm_c1, ddpm_c2, ddpm_sigma;
  const float* noise; /* [B][T][N] or NULL -> Philox(seed, step) */
  uint64_t seed;
  const uint64_t* seed_dev; /* optional device word added to `seed` at run time (keeps hipGraph replays fresh) */
  uint32_t step;
  /* tiling: 0 = auto, else one of SS_TILE_* (BMxBN) */
  int32_t tile;
  /* grouped launch: batch item b uses weight set g = b / group_size (0 = one set). Strides in floats. Lets the two
   * independent f0 denoisers (same shapes, different weights) share every launch: 2x the blocks per kernel. */
  int32_t group_size;
  int64_t w_group_stride, bias_group_stride, a_bias_group_stride;
  /* matrix precision: 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32); 1 = both operands rounded to bf16 (RNE) on the way
   * into v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 prologue/epilogue (BASELINE config 4) */
  int32_t mfma_bf16;
  /* DDPM epilogue variant: 0 = v is the predicted NOISE (x0 = clamp(recip*x - recipm1*v, -1, 1), shallow_diffusion_tts.py:130-153);
   * 1 = v is the predicted x0 itself, no clamp (ProDiffusion.p_sample, modules/diff/prodiff.py:150-153) */
  int32_t ddpm_x0_pred;
  /* ss_wino43_gate16 / 16w only: 1 = E is this launch's conditioner addend in the kernel's FETCH ORDER (ss_gate16_tile_addend with the same
   * dilation and mt; mt must then be given explicitly): per workgroup tile [wave][row tile m][quad r][lane][frame o] floats, so a wave reads
   * 1 KB contiguous per instruction (16 bytes per lane) instead of 8 cache lines x 32 B. lde / e_batch_stride are ignored. */
  int32_t e_tiled;
  int32_t reserved2_;
} ss_conv_gemm_args;

int ss_conv_gemm(const ss_conv_gemm_args* args, void* stream);

/* Small-K GEMM on 16x16x4 fp32 MFMA tiles for the residual half of the denoisers' output projection (net.py:75-77, deferred-skip
 * form):  C = (R + A . W^T + bias) * post_scale,  K = Cin = Kp in {192, 256}, one tap. A workgroup is 16*mt rows x 64 columns
 * (mt = 4 | 6 | 8, 0 = ss_gemm16_pick); the wave's whole weight slice is register-resident, the A tile arrives by LDS-DMA.
 * Uses A, lda, Cin, Kp, lens, B, T, W, N, Np, bias, R, ldr, C, ldc, post_scale, mask_rows, batch strides and the group fields of
 * ss_conv_gemm_args; C may alias R (in-place residual update). Same result as ss_conv_gemm(epi STORE + R) up to the K order. */
int ss_gemm16_res(const ss_conv_gemm_args* args, int mt, void* stream);
int ss_gemm16_pick(int B, int T, int N);
/* Streaming form for large K (the K = L*C skip GEMM of the deferred-skip loops): C = act(A . W^T + bias), act none | relu, rows >=
 * lens[b] written as 0 when mask_rows. Both operands arrive by LDS-DMA (shared A ring, per-wave B ring), no VALU work in the loop.
 * K = Cin = Kp multiple of 32, one tap; same 16*mt x 64 tiles as ss_gemm16_res. */
int ss_gemm16_store(const ss_conv_gemm_args* args, int mt, void* stream);
/* Split-K form of ss_gemm16_store for launches that leave most CUs idle (one short utterance - the B = 1 latency shape of
 * inference/StyleSinger.py:175-186: the K = L*C skip GEMM is 48 workgroups x 160 K chunks): `ksplit` slices of K run as separate workgroups
 * into `partials` (ksplit * B * T * N floats) and a second launch adds them in slice order (deterministic) + bias / activation / row mask.
 * ksplit = 1 is ss_gemm16_store. ss_gemm16_ksplit_pick: the slice count the denoiser loops use for a launch (1 = do not split). */
int ss_gemm16_store_splitk(const ss_conv_gemm_args* args, int mt, int ksplit, float* partials, void* stream);
int ss_gemm16_ksplit_pick(int B, int T, int N, int K);

/* ss_gemm16_res with its weights in fetch order: W16 = ss_pack_gemm16_weights(first N rows of args->W) ([n tile][wave][K chunk][half]
 * [lane][4 floats], N % 64 == 0); with grouped launches args->w_group_stride is the stride of W16 */
int ss_gemm16_resw(const ss_conv_gemm_args* args, const float* W16, int mt, void* stream);
int ss_pack_gemm16_weights(const float* src, float* dst, int Np, int Kp, void* stream);
/* ss_wino43_gate16 with the weights of the 16x16x4 tilings in their fetch order: W16 = ss_pack_gate16_weights(args->W) (same size;
 * [n tile][wave][K chunk][component][half][lane][4 floats]: one fetch instruction of a wave = 1 KB contiguous). args->W (packed rows) is
 * still what the 32x32x2 fallback reads when the pick returns 0; with grouped launches both use args->w_group_stride (floats). */
int ss_wino43_gate16w(const ss_conv_gemm_args* args, const float* W16, int dilation, int mt, void* stream);
int ss_pack_gate16_weights(const float* src, float* dst, int Np, int Kp, void* stream);
/* "bf16x3" form of ss_wino43_gate16 (opt-in precision mode): the same F(4,3) gate with every fp32 product computed on the BF16 matrix cores
 * from operands split into three bf16 terms (a = hi + mid + lo, round-to-nearest each; six partial products hi.hi, hi.mid, mid.hi, hi.lo,
 * lo.hi, mid.mid accumulated in fp32, smallest first). Wx = ss_split3_weights of the packed F(4,3) weights (Np * 18 * Kp bf16; with
 * grouped launches args->w_group_stride counts bf16 elements); args->W is ignored. mt as ss_wino43_gate16.
 * ss_split3_weights: packed F(4,3) weights [Np][6 * Kp] fp32 -> the three bf16 terms of every element in the kernel's fetch order
 * [n tile][wave][K chunk][component][plane][lane][8] (Np % 64 == 0, Kp % 32 == 0). */
int ss_wino43_gate16x(const ss_conv_gemm_args* args, const void* Wx, int dilation, int mt, void* stream);
int ss_split3_weights(const float* src, void* dst, int Np, int Kp, void* stream);
/* "bf16x3" form of ss_gemm16_store (same argument rules): Wx = ss_split3_gemm16_weights(args->W as [Np][Kp] fp32) - the three bf16 terms of
 * every weight in fetch order [n tile (64 columns, zero padded)][wave][K chunk][plane][lane][8]; A stays fp32 in HBM and is split when it is
 * staged. args->W is ignored; with grouped launches args->w_group_stride counts bf16 elements (ss_split3_gemm16_elems per weight set). */
int ss_gemm16x_store(const ss_conv_gemm_args* args, const void* Wx, int mt, void* stream);
int ss_split3_gemm16_weights(const float* src, void* dst, int Np, int Kp, void* stream);
int64_t ss_split3_gemm16_elems(int Np, int Kp);
/* Grouped Winograd F(4,3) form of a k-tap (3 | 7 | 11) dilated (1 | 3 | 5) C -> C conv with the SS_EPI_STORE epilogue (act none | leaky-relu,
 * bias, residual R, post_scale, accumulate, row mask) and the input leaky-relu of the HiFi-GAN ResBlocks (hifigan_nsf.py:54-61 /
 * hifigan.py ResBlock1): the taps are split into ceil(k/3) groups of three, each an F(4,3) product, six accumulators over all groups.
 * args as for ss_conv_gemm with W = ss_pack_conv_weight of the transformed taps [C][C][6*ceil(k/3)] (ss_wino43_weight_transform per
 * group, zero-padded last group), Kp == Np == N == Cin == C (multiple of 64), fp32 only. ss_wino43_conv_ok: 1 if (C, k, dilation) is covered. */
int ss_wino43_conv(const ss_conv_gemm_args* args, int k, int dilation, void* stream);
int ss_wino43_conv_ok(int C, int k, int dilation);
/* Winograd F(2,3) form of the 3-tap dilated conv + SS_EPI_GATE epilogue (net.py:66-73): same arguments as the
 * direct call except that W is the TRANSFORMED weight packed as a 4-"tap" tensor (ss_wino_weight_transform then
 * ss_pack_conv_weight(k=4, interleave_half=C)) and the dilation is passed explicitly (power of two). 1.5x fewer
 * matrix ops; results equal the direct form to fp32 rounding. Uses A, lda, Cin, lens, B, T, a_bias, W, N, Np, Kp, E,
 * lde, gate_mode, bias, mask_rows, C, ldc, batch strides and the group fields of ss_conv_gemm_args. */
int ss_wino_gate(const ss_conv_gemm_args* args, int dilation, void* stream);
/* src [Cout][Cin][3] -> dst [Cout][Cin][4]: g0=w0, g1=(w0+w1+w2)/2, g2=(w0-w1+w2)/2, g3=w2 */
int ss_wino_weight_transform(const float* src, float* dst, int Cout, int Cin, void* stream);
/* Winograd F(4,3) form of the same layer: four frames (t, t+d, t+2d, t+3d) from 6 products instead of 12 (2x fewer matrix ops
 * than the direct form). W is the transformed weight packed as a 6-"tap" tensor (ss_wino43_weight_transform then
 * ss_pack_conv_weight(k=6, interleave_half=C)). Needs Cin % 32 == 0 and Kp == Cin. Same argument use as ss_wino_gate; results
 * equal the direct form to fp32 rounding (single-layer error about 2x that of F(2,3); tests/test_gpu_kernels.py). */
int ss_wino43_gate(const ss_conv_gemm_args* args, int dilation, void* stream);
/* The same layer on 16x16x4 fp32 MFMA tiles: a wave tile is 16*mt quads x 16 packed columns (8 channels, both gate operands), a
 * workgroup 16*mt quads x 64 packed columns; mt = 2 | 3, or 0 = ss_wino43_gate16_pick decides per launch and falls back to
 * ss_wino43_gate when the 32x32x2 tiles fit better (many rounds of workgroups per launch). For single-round launches: at BASELINE
 * config 2 the mel launch is 512 workgroups (2 per CU) and the f0-pair launch 768 (3 per CU) instead of 384 / 564 uneven ones.
 * Same arguments, same weights (ss_wino43_weight_transform + ss_pack_conv_weight(k=6, interleave_half=C)), same arithmetic up to
 * the summation order over K (tests/test_gpu_round3.py). */
int ss_wino43_gate16(const ss_conv_gemm_args* args, int dilation, int mt, void* stream);
/* The conditioner addend of one layer in the fetch order of the 16x16x4 gate kernel (ss_conv_gemm_args.e_tiled): E [B][T][lde] (this layer's
 * Np packed columns at E, row stride lde, batch stride e_batch_stride floats) -> E16, ss_gate16_tiled_floats(B, T, Np, dilation, mt) floats.
 * Frames >= T of the last tile of an item are written as 0. */
int ss_gate16_tile_addend(const float* E, int lde, int64_t e_batch_stride, float* E16, int B, int T, int Np, int dilation, int mt, void* stream);
int64_t ss_gate16_tiled_floats(int B, int T, int Np, int dilation, int mt);
/* the tiling ss_wino43_gate16(mt = 0) uses for a launch of B items x T frames x Np packed columns: 2 | 3, or 0 = the 32x32x2 kernel */
int ss_wino43_gate16_pick(int B, int T, int Np, int dilation);
/* src [Cout][Cin][3] -> dst [Cout][Cin][6]: g0=w0/4, g1=-(w0+w1+w2)/6, g2=-(w0-w1+w2)/6, g3=w0/24+w1/12+w2/6,
 * g4=w0/24-w1/12+w2/6, g5=w2 */
int ss_wino43_weight_transform(const float* src, float* dst, int Cout, int Cin, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16-operand GEMM/conv for the denoisers' hidden layers (BASELINE config 4): A and W are bf16 IN HBM (rounded once where
 * they are produced: weights by ss_to_bf16 at pack time = the "bf16 weight copies", activations by the producing epilogue),
 * fp32 accumulate on v_mfma_f32_32x32x16_bf16, K chunks of 64. Same tap / zero-padding / grouped-launch semantics as
 * ss_conv_gemm. Arithmetic contract: RNE rounding of both matmul operands, everything else fp32 (net.py:58-130 otherwise).
 * ------------------------------------------------------------------------------------------ */
enum {
  SS_HEPI_STORE = 0, /* C fp32 = act(acc + bias), rows >= lens[b] written as 0 when mask_rows                               */
  SS_HEPI_GATE = 1,  /* paired 32-col blocks + fp32 addend E -> sigmoid*tanh (gate_mode as SS_EPI_GATE), C is BF16 [.][ldc]  */
  SS_HEPI_RESX = 2   /* X fp32 in place: x = (x + acc + bias) * post_scale ; Y bf16 = x + next_bias (next layer's operand)   */
};
typedef struct ss_gemm_bf16_args {
  const uint16_t* A;      /* bf16 [B][T][lda] */
  int64_t a_batch_stride; /* elements */
  int32_t lda;            /* elements, multiple of 8 */
  int32_t K;              /* channels per tap, multiple of 64 */
  int32_t ntaps;          /* 1..4 */
  int32_t tap_off[4];
  const int32_t* lens;
  int32_t B, T;
  const uint16_t* W; /* bf16 packed [Np][ntaps*K] (ss_pack_conv_weight layout, converted by ss_to_bf16) */
  int64_t w_group_stride;
  int32_t N, Np;
  int32_t epi;
  int32_t act; /* STORE: SS_ACT_* */
  const float* bias;
  int64_t bias_group_stride;
  const float* E;
  int32_t lde;
  int32_t gate_mode;
  int64_t e_batch_stride;
  float* X;
  int64_t x_batch_stride;
  int32_t ldx;
  float post_scale;
  const float* next_bias; /* [N] or NULL */
  int64_t next_bias_group_stride;
  uint16_t* Y; /* bf16 [B][T][ldy] or NULL */
  int64_t y_batch_stride;
  int32_t ldy;
  int32_t ldc;
  void* C; /* STORE: float*, GATE: uint16_t* */
  int64_t c_batch_stride;
  int32_t mask_rows;
  int32_t group_size;
  /* split-operand form ("bf16x2" precision, BASELINE config 4 at fp32-grade parity): split = 1 -> every bf16 operand is a PAIR of bf16 terms
   * v = hi + mid (hi = RNE(v), mid = RNE(v - hi): 16 significand bits) and the matrix cores run the three products hi*hi + hi*mid + mid*hi
   * (fp32 accumulate). Layout ("pairs interleaved by 32"): a logical row of n channels is 2n bf16 - for every 32-channel chunk j the 32 hi terms
   * at [64j, 64j+32) followed by the 32 mid terms at [64j+32, 64j+64), i.e. one 128-byte line = one K chunk of both planes. That holds for A
   * (lda = physical row stride; K = logical channels per tap, a multiple of 32), for W (rows of 2*ntaps*K, ss_split_bf16 of the packed fp32
   * weights) and for the bf16 OUTPUTS (GATE's C, RESX's Y: logical channel c of the output lands at (c >> 5) * 64 + (c & 31), its mid term 32
   * further; ldc / ldy are physical strides). */
  int32_t split;
  /* split = 2 ("fp16x2" precision: BASELINE config 4 at parity with TWO products instead of three): the same pair layouts with FP16 terms, and
   * only the WEIGHTS are read as pairs - W holds (hi, lo) = ss_split_f16 of w * 2^s, the A operand's hi term alone feeds the matrix cores
   * (v_mfma_f32_32x32x16_f16: a*hi + a*lo), and the fp32 accumulator is multiplied by out_scale = 2^-s before bias / addend / residual are added.
   * The A operand's second plane is never fetched. Outputs: GATE writes fp16(g) into the hi slots and leaves the second plane of its output rows
   * untouched; RESX reads and writes the stream as a true fp16 pair (22 significant bits). Why two products suffice:
   * over 1000 reverse steps the weight rounding is the coherent error, the activation rounding averages out, and fp16's is 8x smaller than
   * bf16's (oracle/bf16x2_numerics.py: 1.9e-5 vs the reference's 1000-step golden, bar 1e-4). */
  float out_scale;
  /* split = 3 ("fp16q4", SS_HEPI_GATE through ss_gemm_bf16_gate128q only - written at the end of round 4, not yet run on hardware): split = 2
   * with the second product a * lo on the block-scaled fp4 matrix instruction. W is a ss_split_f16 pack whose lo plane has been replaced by
   * the fp4 terms of lo in the kernel's lane order + their E8M0 block scales (stylesinger_amd.lib.pack_gate_q4, layout: csrc/gate128_layout.h
   * g128q); the kernel converts its own fp16 A fragments to fp4 with the fixed power-of-two scale q_scale: q = fp4(a / q_scale). */
  float q_scale;
  /* split = 2 with ONE weight term ("fp16sd": W is an ss_split_f16 pack whose lo terms are zero): kernels that know the flag skip the second product
   * and do not fetch the lo plane (ss_gemm_bf16_tile256, SS_HEPI_STORE); the others compute the same result with a product of zeros.
   * one_product = 2 (ss_gemm_bf16_tile256 STORE only; ss_gemm_bf16 refuses it elsewhere): W is the COMPACT one-term pack, fp16 [Np][K] - the rows of
   * the hi plane alone, half the bytes per tile pass and small enough to stay in an XCD's L2. */
  int32_t one_product;
  /* RESX with split operands and X == NULL ("pair-only residual stream"): the stream lives ONLY as the (hi, mid) pair Y = x + cur_bias (16
   * significand bits - measured harmless on the reference's 1000-step golden: 2.4e-6 either way, oracle/bf16x2_numerics.py). The epilogue reads
   * its element of Y, recovers x = hi + mid - cur_bias, and writes Y = pair(x_new + next_bias) in place: 3 instead of 4 KB per row of traffic. */
  const float* cur_bias;
  int64_t cur_bias_group_stride;
  /* split = 2, SS_HEPI_STORE through ss_gemm_bf16_tile256 only (round 6): the A operand is COMPACT - a row holds its K fp16 hi terms contiguously
   * (lda >= K elements), no interleaved second plane. The fp16x2 / fp16sd matrix cores never read that plane, but in the pair layout it shares every
   * 128-byte line with the hi terms, so the K = L C skip GEMM pulled twice the bytes it used from HBM (ss_layer512 writes G in this form with
   * g_compact = 1). ss_gemm_bf16 refuses the flag for launches its other kernels would take. */
  int32_t a_compact;
  int32_t reserved2_;
} ss_gemm_bf16_args;
int ss_gemm_bf16(const ss_gemm_bf16_args* args, void* stream);
/* The SS_HEPI_GATE form of ss_gemm_bf16 for many-round launches (BASELINE config 4): three taps (-d, 0, d) with d <= 8, K = 256,
 * Np a multiple of 256. 256 rows x 256 packed columns per workgroup, 8 waves, both operands by LDS-DMA, the A tile staged once per
 * channel chunk with its dilation halo. ss_gemm_bf16 dispatches here when ss_gemm_bf16_gate256_ok(args) (and the "gate256" tuning knob)
 * say so; same arithmetic contract, results equal up to the K summation order. */
int ss_gemm_bf16_gate256(const ss_gemm_bf16_args* args, void* stream);
int ss_gemm_bf16_gate256_ok(const ss_gemm_bf16_args* args);
/* The same launch for split = 2 ("fp16x2") operands on 256 x 128 tiles with TWO workgroups per CU (4 waves, 80 KB of LDS each: the A image is
 * compact - only the hi plane of the A operand is staged - and half the columns halve the weight tile), so that one workgroup's MFMAs run under
 * the other's epilogue and barrier waits. Same arithmetic and summation order as ss_gemm_bf16_gate256: bit-identical outputs
 * (tests/test_gpu_fp16x2.py). ss_gemm_bf16 dispatches here when the "gate128" tuning knob is 1 (default) and ss_gemm_bf16_gate128_ok(args):
 * launches of >= 2048 such tiles (BASELINE config 4: 360 -> 333 us back to back, batch 11.9 -> 11.2 s). */
int ss_gemm_bf16_gate128(const ss_gemm_bf16_args* args, void* stream);
int ss_gemm_bf16_gate128_ok(const ss_gemm_bf16_args* args);
/* ss_gemm_bf16_gate128 for split = 3 ("fp16q4") operands: per pair of 32-channel steps 32 fp16 MFMAs (a * hi) + 8 block-scaled fp4 ones
 * (a_q * lo_q, v_mfma_scale_f32_32x32x64_f8f6f4) instead of 64. NOT YET RUN ON HARDWARE; nothing dispatches to it.
 * ss_gate128q_kindex(out): the K index (tap * 256 + channel) of element e of lane half h in step pair p, out[(p * 2 + h) * 32 + e], 12 pairs -
 * the order the weights' fp4 lo terms are packed in (returns the number of entries written: 768). */
int ss_gemm_bf16_gate128q(const ss_gemm_bf16_args* args, void* stream);
int ss_gemm_bf16_gate128q_ok(const ss_gemm_bf16_args* args);
int ss_gate128q_kindex(int32_t* out, int n);
/* The long-K STORE GEMM (the skip GEMM) of ss_gemm_bf16_tile256 for split = 3 ("fp16q4") operands: the same second product on the block-scaled
 * fp4 instruction, pairs = consecutive 32-channel chunks, A scale q_scale (gate outputs: 2^-2). NOT YET RUN ON HARDWARE; nothing dispatches
 * to it. ss_tile256q_kindex(out, n_pairs): K index of element e of lane half h in chunk pair p, out[(p * 2 + h) * 32 + e] (returns the count). */
int ss_gemm_bf16_tile256q(const ss_gemm_bf16_args* args, void* stream);
int ss_gemm_bf16_tile256q_ok(const ss_gemm_bf16_args* args);
int ss_tile256q_kindex(int32_t* out, int n_pairs);
/* The split-operand 1-tap forms of ss_gemm_bf16 for many-round launches (BASELINE config 4 in bf16x2 precision): SS_HEPI_RESX on the pair-only
 * stream (X = NULL) and SS_HEPI_STORE (the K = L*C skip GEMM), N <= 256, K a multiple of 64. 256 rows x all columns per workgroup, 8 waves, both
 * operands by LDS-DMA, epilogues through LDS as 16-byte vectors. ss_gemm_bf16 dispatches here when ss_gemm_bf16_tile256_ok(args) (and the
 * "gate256" knob) say so; same arithmetic contract, results equal up to the K summation order. */
int ss_gemm_bf16_tile256(const ss_gemm_bf16_args* args, void* stream);
int ss_gemm_bf16_tile256_ok(const ss_gemm_bf16_args* args);
/* ONE launch per residual layer of the mel denoiser in "fp16x2" precision for many-round launches (BASELINE configs[3]); replaces the
 * SS_HEPI_GATE launch + the SS_HEPI_RESX launch of a layer (modules/diff/net.py:66-78: dilated conv + conditioner addend -> sigmoid * tanh ->
 * residual half of output_projection -> (x + r) / sqrt(2)). A workgroup owns 128 rows x all 512 pre-activation columns (C = 256 fixed): the
 * gate output stays in LDS as the residual projection's operand and leaves the CU once (G, the skip GEMM's operand); persistent workgroups,
 * weight fragments streamed L2 -> registers in the order ss_layer512_pack_gate / _pack_res lay them out, the conditioner addend in the
 * accumulator order of ss_layer512_tile_addend. Same arithmetic contract as ss_gemm_bf16 with split = 2 (results equal up to the fp32
 * summation order).
 * The residual stream lives in a layout of its own, written by ss_layer512_entry and by this launch only:
 *   H  = fp16(x + dstep_l), the conv's operand, in slot-major tiles: [tile = b * ceil(T / 128) + t / 128][slot 32][row 128] x 8 channels
 *      (slot s = channels 8 s .. 8 s + 7), ss_layer512_h_elems(B, T) elements. DOUBLE BUFFERED: Hout must differ from Hin (a tile reads halo
 *      rows its neighbours rewrite). Rows >= lens[b] are zero (every producer masks them);
 *   P  = R, the fp16 REMAINDER of the stream: x = (H - dstep_l) + R with H = fp16(x + dstep_l) as above and R = fp16(x - (H - dstep_l)) - 22
 *      significant bits, as the fp16 pair of the two-launch form - in accumulator order, ss_layer512_stream_bytes(B, T) bytes, updated in place:
 *      [tile][m 4][q 4][wave 8][lane 64] x 4 fp16; lane (l31, lh) of (wave, m, q) holds channels 32 wave + 8 q + 4 lh .. + 3 of row 32 m + l31 of
 *      the tile. The launch takes the H term of its own rows from the activation tile it staged anyway, so the stream costs 2 bytes per element of
 *      HBM traffic each way (first form of round 6: an fp32 copy, 4 bytes each way - 256 of the 584 KB a tile moved). cur_bias = dstep_l and
 *      next_bias = dstep_(l+1) are the two biases of that representation.
 * Hout == NULL: gate only (the last layer: its residual stream is never read); P and Wr are then unused. */
typedef struct ss_layer512_args {
  const uint16_t* Hin;      /* fp16(x + dstep_l), slot-major tiles */
  int32_t d;                /* dilation, 1..8: taps (-d, 0, d) */
  int32_t n_products;       /* weight terms per element = matrix products per GEMM: 2 (or 0) = "fp16x2" (hi, lo); 1 = "fp16sd": Wg / Wr hold ONE fp16 term
                             * (packs made with n_products = 1; the caller cycles noise-shaped weight sets over the evaluations, ss_wavenet.n_wsets) */
  uint16_t* Hout;           /* fp16(x' + next_bias), or NULL */
  void* P;                  /* fp32 stream x, read and rewritten in place (x') */
  const int32_t* lens;
  int32_t B, T;
  const uint16_t* Wg;       /* ss_layer512_pack_gate of the layer's ss_split_f16 dilated-conv pack (786 432 elements) */
  const uint16_t* Wr;       /* ss_layer512_pack_res of the residual half of the ss_split_f16 output-projection pack (131 072 elements) */
  const float* E512;        /* ss_layer512_tile_addend of this layer's 512 addend columns: ss_layer512_addend_floats(B, T) floats */
  uint16_t* G;              /* gate output fp16 [B][T][ldg]: hi slots of ss_gemm_bf16's pair layout (the second plane is not written; ldg >= 512), or with
                             * g_compact the 256 channels contiguously (ldg >= 256) - the skip GEMM's a_compact operand */
  int64_t g_batch_stride;
  int32_t ldg;
  int32_t mask_rows;        /* rows >= lens[b]: G = 0, stream = 0 */
  const float* bias_r;      /* [256] residual half of the output-projection bias, or NULL */
  const float* next_bias;   /* [256] dstep_{l+1}, or NULL */
  int32_t g_compact;        /* 1: G rows are compact (see G) */
  int32_t e_f16;            /* 1 (n_products = 1 only): E512 is ONE SET of the fp16 form made by ss_layer512_tile_addend_f16 */
  float out_scale;          /* 2^-s of the weight packs */
  float post_scale;         /* 1 / sqrt(2) */
  const float* cur_bias;    /* [256] dstep_l (what Hin's values carry on top of x), or NULL */
} ss_layer512_args;
int ss_layer512(const ss_layer512_args* args, void* stream);
/* 1 if the shape fits the kernel's fixed geometry and fills the chip (>= 4 rounds of 128-row tiles per CU; any size with the knob layer512 = 2) */
int ss_layer512_ok(int B, int T, int C, int d_max, int ldg);
/* stack entry: X fp32 [B][T][ldx] -> H = fp16(x + bias) and P = R = fp16(x - (H - bias)) (bias = dstep_0; NULL = none) as above; rows >= lens[b] zero */
int ss_layer512_entry(const float* X, int ldx, int64_t x_batch_stride, const float* bias, const int32_t* lens, uint16_t* H, void* P, int B, int T,
                      void* stream);
int64_t ss_layer512_stream_bytes(int B, int T);
int64_t ss_layer512_h_elems(int B, int T);
int64_t ss_layer512_addend_floats(int B, int T);
/* E [B][T][lde] (the layer's 512 packed addend columns start at E) -> the tiled slab the kernel reads (once per forward and layer); stored as
 * the gate's exp2 arguments: E * -log2(e) in the sigmoid blocks, E * -2 log2(e) in the tanh blocks */
int ss_layer512_tile_addend(const float* E, int lde, int64_t e_batch_stride, float* out, int B, int T, void* stream);
/* The same addend as n_sets fp16 SETS for ss_layer512_args.e_f16 (the one-product form, "fp16sd"): set k at out + k * set_stride, each
 * ss_layer512_addend_halfs(B, T) fp16 elements ([tile][m 4][q 4][wave 8][lane 64] x (4 values of the sigmoid block | 4 of the tanh block): half the bytes
 * of the fp32 slab, the largest stream of the launch). The addend is the same in every evaluation of a sampling loop, so one fp16 rounding of it would be a
 * fixed bias (9.5e-5 on the reference's 1000-step golden); the sets are a first-order sigma-delta sequence of roundings of the scaled value (r_0 = 0,
 * E_k = RNE16(e + r_k), r_(k+1) = r_k + (e - E_k)) and evaluation j of a loop reads set j % n_sets (8 sets: 2.5e-5, exact fp32 slab: 2.2e-5;
 * oracle/dither_numerics.py --e-sets=N). */
int64_t ss_layer512_addend_halfs(int B, int T);
int ss_layer512_tile_addend_f16(const float* E, int lde, int64_t e_batch_stride, uint16_t* out, int n_sets, int64_t set_stride, int B, int T, void* stream);
/* ss_split_f16 pack [512][3 * 256 * 2] of the gate-interleaved dilated-conv weights -> fragment order (786 432 elements; n_products = 1: the hi
 * terms only, 393 216 elements) */
int ss_layer512_pack_gate(const uint16_t* w_pairs, uint16_t* out, int n_products, void* stream);
/* ss_split_f16 pack [>= 256][256 * 2] of the output projection (first 256 rows = residual half) -> fragment order (131 072 elements; n_products = 1: 65 536) */
int ss_layer512_pack_res(const uint16_t* w_pairs, uint16_t* out, int n_products, void* stream);
/* y = bf16(x + bias) (RNE; bias per column, optional, per weight group), rows >= lens[b] -> 0. Also converts packed weights
 * (B = 1, T = rows): the bf16 weight copies of the checkpoint packer. */
int ss_to_bf16(const float* x, const float* bias, uint16_t* y, int B, int T, int C, int ldx, int ldy, const int32_t* lens,
               int group_size, int64_t bias_group_stride, void* stream);
/* The split form of ss_to_bf16 (C a multiple of 32, ldy >= 2C): v = x + bias -> hi = RNE(v) at y[.][(c >> 5) * 64 + (c & 31)], mid = RNE(v - hi)
 * 32 elements further ("pairs interleaved by 32", see ss_gemm_bf16_args.split). */
int ss_split_bf16(const float* x, const float* bias, uint16_t* y, int B, int T, int C, int ldx, int ldy, const int32_t* lens,
                  int group_size, int64_t bias_group_stride, void* stream);
/* The same with FP16 terms of v = (x + bias) * scale (scale: a power of two - the weight shift 2^s of ss_gemm_bf16_args.split = 2, which keeps
 * the lo terms of ordinary weights out of the fp16 subnormals; 1 for activations): hi = RNE16(v), lo = RNE16(v - hi), same interleaved layout. */
int ss_split_f16(const float* x, const float* bias, float scale, uint16_t* y, int B, int T, int C, int ldx, int ldy, const int32_t* lens,
                 int group_size, int64_t bias_group_stride, void* stream);

/* Weight packing (device -> device).  src is the torch parameter layout [Cout][Cin][k] (conv1d,
 * k=1 for nn.Linear [out][in]).  dst is [Np][k*Kp] with zero fill.  If scale0 != NULL (from
 * ss_weight_norm_scale) the weight-norm reparametrisation w = g * v / ||v|| (torch.nn.utils.weight_norm,
 * dim=0; hifigan_nsf.py:171-178, wavenet.py:37-52) is folded: row o is multiplied by scale0[o].  interleave_half > 0 reorders rows for SS_EPI_GATE: packed 32-row
 * block 2p holds rows [p*32, p*32+32) of the first half, block 2p+1 those of the second half
 * (half = interleave_half rows; rows beyond the half are zero).  row_scale multiplies every row. */
int ss_pack_conv_weight(const float* src, const float* scale0, float* dst, int Cout, int Cin, int k, int Np, int Kp,
                        int interleave_half, float row_scale, void* stream);
/* scale[r] = g[r] / ||v[r,:]||_2 : the weight-norm factor along dim 0 (rows x cols view of v) */
int ss_weight_norm_scale(const float* v, const float* g, float* scale, int rows, int cols, void* stream);
/* ConvTranspose1d weight [Cin][Cout][k] (stride u, padding (k-u)/2, k == 2u) -> two polyphase groups,
 * each a 2-tap conv with N = nph*Cout columns (hifigan_nsf.py:121-123).  group 0: phases [0,u-pad) taps (t, t-1);
 * group 1: phases [u-pad,u) taps (t+1, t).  dst = [Np][2*Kp]. */
int ss_pack_convtr_weight(const float* src, const float* scale0, float* dst, int Cin, int Cout, int k, int u, int group,
                          int Np, int Kp, void* stream);
/* bias [n] -> packed order [Np] (same interleave rule), optional second bias added (b + b2) */
int ss_pack_bias(const float* src, const float* src2, float* dst, int n, int Np, int interleave_half, int repeat,
                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Row-wise / elementwise pieces
 * ------------------------------------------------------------------------------------------ */
/* LayerNorm over the last dim C (eps inside sqrt, torch.nn.LayerNorm; common_layers.py:75-82,
 * tts_modules.py:37-56).  y may alias x.  mask_rows: rows >= lens[b] are written as 0. */
int ss_layernorm(const float* x, float* y, const float* gamma, const float* beta, int B, int T, int C, int ldx, int ldy,
                 int64_t x_batch_stride, int64_t y_batch_stride, float eps, const int32_t* lens, int mask_rows,
                 void* stream);

/* Multi-head attention core, flash-style, fp32 MFMA: O = softmax(Q K^T * scale + key_mask) V.
 * Q [B][Tq][ldq] (head h at column h*D), K/V [B][Tk][ldk|ldv]; key j valid iff j < klens[b].
 * Replaces the bmm/softmax/bmm inside F.multi_head_attention_forward (common_layers.py:277-286)
 * and nn.MultiheadAttention (lse.py:19,41).  D must be 128. */
int ss_attention(const float* Q, const float* K, const float* V, float* O, int B, int H, int D, int Tq, int Tk, int ldq,
                 int ldk, int ldv, int ldo, int64_t q_bs, int64_t k_bs, int64_t v_bs, int64_t o_bs,
                 const int32_t* qlens, const int32_t* klens, float scale, void* stream);

/* out[b][t][:] = scale * table[idx[b][t]][:]  (+ out if accumulate).  idx int64 (torch LongTensor).
 * nn.Embedding lookups: tts_modules.py:339-346, stylesinger.py:31-36,246 */
int ss_embedding(const int64_t* idx, const float* table, float* out, int rows, int C, int n_table, float scale,
                 int accumulate, void* stream);
/* fairseq make_positions (utils/tts_utils.py:6-18) + sinusoidal table lookup
 * (common_layers.py:129-148): pos = cumsum(nz)*nz where nz = (probe != 0) ; out (+)= alpha*alpha_dev[0]*table[pos].
 * probe is either int64 tokens (probe_i64) or the first channel of a float tensor (probe_f32, row stride ldp). */
int ss_make_positions(const int64_t* probe_i64, const float* probe_f32, int ldp, int64_t probe_batch_stride,
                      int32_t* pos, int B, int T, void* stream);
int ss_table_add(const int32_t* pos, const float* table, int table_rows, float* out, int ldo, int64_t out_batch_stride,
                 int B, int T, int C, const float* alpha_dev, float alpha, int accumulate, void* stream);
/* out[b][t][c] = ((((x + v1[b][c]) + y1) + v2[b][c]) + y2) * (t < lens[b]); v1/y1/v2/y2 optional.
 * The fixed order reproduces the reference's in-place adds (stylesinger.py:139-142,158-177). */
int ss_add_bcast_mask(const float* x, const float* v1, const float* y1, const float* v2, const float* y2, float* out,
                      int B, int T, int C, const int32_t* lens, void* stream);
/* gather rows: out[b][t][:] = (idx[b][t] > 0) ? src[b][idx[b][t]-1][:] : 0  (expand_states, fs2.py:258-262) */
int ss_gather_expand(const float* src, const int64_t* mel2ph, float* out, int B, int Tsrc, int T, int C, void* stream);
int ss_gather_expand_i64(const int64_t* src, const int64_t* mel2ph, int64_t* out, int B, int Tsrc, int T, void* stream);
/* NoteEncoder dur_ln: out[r][c] += dur[r]*w[c] + b[c]  (stylesinger.py:33-35) */
int ss_note_dur_add(const float* dur, const float* w, const float* b, float* out, int rows, int C, void* stream);
/* DurationPredictor.out2dur + LengthRegulator (tts_modules.py:122-130,158-188):
 * dur = max(round(exp(x)-1),0)*(tok!=0); mel2ph[b][t] = phoneme index (1-based) or 0; lens[b] = frames.
 * Call with Tmax = 0 first (durations + lens only), read lens, then with Tmax = max(lens) to fill mel2ph. */
int ss_length_regulate(const float* logdur, const int64_t* tokens, int64_t* dur_out, int64_t* mel2ph, int32_t* lens, int B,
                       int Tp, int Tmax, void* stream);
/* lens[b] = number of t with mel2ph[b][t] > 0 */
int ss_count_nonzero_i64(const int64_t* x, int32_t* lens, int B, int T, void* stream);

/* ------------------------------------------------------------------------------------------
 * Residual Style Adaptor pieces
 * ------------------------------------------------------------------------------------------ */
/* RQ codebook lookup (modules/StyleSinger/RQ.py:30-55,117-128,226-270): depth residual nearest-code
 * search (first-min tie-break), out = x + (sum_q - x); codes int64 [rows][depth]. codebooks [depth][n_embed+1][C] */
int ss_rq_lookup(const float* x, const float* codebooks, float* out, int64_t* codes, int rows, int C, int n_embed,
                 int depth, void* stream);
/* x[b][t][c] += f0[b][t] on valid rows (lse.py:119-122) */
int ss_add_rowscalar(float* x, const float* s, int B, int T, int C, const int32_t* lens, void* stream);
/* lens[b] = 1 + last t with ref_mels[b][t][0] != 0  (padding mask of the reference mel, lse.py:109) */
int ss_ref_lens(const float* ref_mels, int B, int T, int C, int32_t* lens, void* stream);
/* x[r][:] = 0 where ref[r*ldref] == 0  (per-frame padding mask inside the valid range, lse.py:109,193) */
int ss_mask_rows_by_ref(float* x, const float* ref, int ldref, int rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Diffusion samplers.  A `ss_wavenet` describes one denoiser (DiffNet / DDiffNet, net.py:58-130,215-266)
 * with packed weights.  All pointers are device pointers owned by the caller.
 * ------------------------------------------------------------------------------------------ */
typedef struct ss_wavenet {
  int32_t C;        /* residual channels */
  int32_t L;        /* residual layers */
  int32_t cond_dim; /* encoder hidden */
  int32_t dil_cycle;
  int32_t in_dim;  /* 80 (mel) / 1 (f0) */
  int32_t out_dim; /* 80 / 3 */
  int32_t steps;
  const float* w_in; /* mel: packed [C][Kp(in_dim)] ; f0: raw input_projection weight [C/2] */
  const float* b_in; /* [C] (mel) / [C/2] (f0) */
  const float* uv_embed; /* f0 only: [2][C/2] */
  const float* dstep;    /* [steps][L][C]: diffusion_projection_l(mlp(sinemb(step))) */
  const float* w_dil[SS_MAX_LAYERS]; /* packed gate-interleaved [2C][3*C] */
  const float* w_out[SS_MAX_LAYERS]; /* packed [2C][C] */
  const float* b_out[SS_MAX_LAYERS]; /* [2C] */
  const float* w_cond;               /* packed [L*2C][cond_dim], per-layer gate interleave */
  const float* b_cond;               /* [L*2C] = conditioner bias + dilated conv bias, packed order */
  const float* w_skip;               /* packed [C][C] */
  const float* b_skip;
  const float* w_final; /* packed [Np(out_dim)][C] */
  const float* b_final;
  /* schedule tables: HOST pointers, fp32 [steps] each (the loop driver reads them on the host and passes
   * per-step scalars by value; shallow_diffusion_tts.py:99-119, gaussian_multinomial_diffusion.py:237-284) */
  const float* sqrt_recip_ac;
  const float* sqrt_recipm1_ac;
  const float* post_c1;
  const float* post_c2;
  const float* post_logvar;
  const float* log_alpha;         /* f0 only */
  const float* log_1m_alpha;      /* f0 only */
  const float* log_cumprod_alpha; /* f0 only */
  const float* log_1m_cumprod_alpha;
  /* paired nets (n_groups = 2): every weight pointer above is net 0's; net g's tensor lives gs_* floats further.
   * Both nets must share shapes and schedules (the two DDiffNets of stylesinger.py:69-73 do). */
  int32_t n_groups;
  /* optional Winograd-transformed dilated-conv weights, gate-interleaved; NULL -> direct conv. Packed [2C][4*Kp] for F(2,3)
   * (wino_m = 0 or 2) or [2C][6*Kp] for F(4,3) (wino_m = 4, below) */
  const float* w_dil_wino[SS_MAX_LAYERS];
  int64_t gs_w_dil_wino;
  int64_t gs_w_in, gs_b_in, gs_uv_embed, gs_dstep, gs_w_dil, gs_w_out, gs_b_out, gs_w_cond, gs_b_cond, gs_w_skip, gs_b_skip,
      gs_w_final, gs_b_final;
  /* 1 = run the hidden-to-hidden GEMMs (conditioner projection, dilated conv, output projection, skip projection) with
   * bf16 operands (ss_conv_gemm_args.mfma_bf16); the input and final projections, which touch the diffusion state, the
   * sampler update and all accumulations stay fp32. Winograd weights are ignored in this mode. */
  int32_t mfma_bf16;
  int32_t wino_m; /* output tile of the Winograd weights in w_dil_wino: 0 or 2 = F(2,3), 4 = F(4,3) */
  /* optional deferred-skip form: w_skipall = the skip halves of all output projections side by side, packed
   * [round_up32(C)][L*C] (column l*C + ci = output_projection_l.weight[C + n][ci]); b_skipall[n] = sum_l bias_l[C + n].
   * When set, the per-layer output projection only computes the residual half and the skip sum is one GEMM per step. */
  const float* w_skipall;
  const float* b_skipall;
  int64_t gs_w_skipall, gs_b_skipall;
  /* 1 = w_skipall / b_skipall are pre-multiplied by skip_projection / sqrt(L) (no nonlinearity sits between the skip sum
   * and skip_projection, net.py:124-126): the K = L*C GEMM + ReLU then IS the stack output and the skip_projection
   * launch disappears. */
  int32_t skipall_folded;
  /* 1 = "bf16x3" precision mode (opt-in): layers with a w_dil_x3 pack run the F(4,3) gate on the BF16 matrix cores from split operands
   * (ss_wino43_gate16x: three bf16 terms per operand, six exact products, fp32 accumulation - fp32-grade results at 6/16 of the fp32
   * matrix time); everything else of the loop is unchanged */
  int32_t mfma_x3;
  /* optional bf16 copies of the hidden-layer weights (same packed layouts, ss_to_bf16). When w_dil_h[0] is set and mfma_bf16 = 1
   * the residual stack runs on ss_gemm_bf16: activations travel between the layers as bf16 (y = x + dstep, gate outputs) and
   * the workspace carries the bf16 planes. w_out_h = residual half only ([C][C]); needs the deferred-skip form (w_skipall_h). */
  const uint16_t* w_dil_h[SS_MAX_LAYERS];
  const uint16_t* w_out_h[SS_MAX_LAYERS];
  const uint16_t* w_skipall_h;
  const uint16_t* w_cond_h;
  int64_t gs_w_dil_h, gs_w_out_h, gs_w_skipall_h, gs_w_cond_h;
  /* optional split copies of the F(4,3) gate weights: ss_split3_weights of w_dil_wino (2C * 18 * Kp bf16, fetch order of the kernel); gs in bf16 elements */
  const uint16_t* w_dil_x3[SS_MAX_LAYERS];
  int64_t gs_w_dil_x3;
  /* optional copies of w_dil_wino (F(4,3) only) in the fetch order of the 16x16x4 gate kernel (ss_pack_gate16_weights); same group stride */
  const float* w_dil_wino16[SS_MAX_LAYERS];
  /* optional copies of the RESIDUAL half of w_out ([C][Kp], C a multiple of 64) in the fetch order of ss_gemm16_res
   * (ss_pack_gemm16_weights); gs_w_out16 = floats between the two nets of a pair */
  const float* w_out16[SS_MAX_LAYERS];
  int64_t gs_w_out16;
  /* bf16x3 mode: ss_split3_gemm16_weights of w_skipall (used with skipall_folded); gs in bf16 elements */
  const uint16_t* w_skipall_x3;
  int64_t gs_w_skipall_x3;
  /* 1 = "bf16x2" precision (BASELINE config 4 at fp32-grade parity; needs mfma_bf16 = 1 and the w_*_h packs): the w_*_h tensors are SPLIT packs
   * (ss_split_bf16 of the packed fp32 weights: rows of 2*ntaps*K, pairs interleaved by 32) and the hidden activations travel as (hi, mid) bf16 pairs;
   * every hidden GEMM runs hi*hi + hi*mid + mid*hi on the bf16 matrix cores (ss_gemm_bf16_args.split), the hoisted conditioner projection in
   * exact fp32 (w_cond_h unused). With skipall_folded the K = L*C GEMM + ReLU is the stack output, as in fp32 mode. */
  int32_t mfma_split;
  /* mfma_split = 2 = "fp16x2" precision: the w_*_h tensors are ss_split_f16 packs of the weights times 2^s, activations fp16 (pair layout, hi
   * term read by the matrix cores, the stream a true fp16 pair), two products per GEMM (ss_gemm_bf16_args.split = 2);
   * mfma_out_scale = 2^-s (every ss_gemm_bf16 launch of the stack passes it as out_scale) */
  float mfma_out_scale;
  /* optional, with mfma_split = 2 ("fp16q4" precision; the kernel behind it has not run on hardware yet): per layer the dilated-conv weights as
   * stylesinger_amd.lib.pack_gate_q4 packs (hi plane fp16, lo plane as block-scaled fp4 in the lane order of ss_gemm_bf16_gate128q). Layers that
   * have one run their gate launch with split = 3 / q_scale = q_scale_gate when ss_gemm_bf16_gate128q_ok says so; every other launch of the mode
   * is fp16x2's. gs in 16-bit elements. */
  const uint16_t* w_dil_q[SS_MAX_LAYERS];
  int64_t gs_w_dil_q;
  float q_scale_gate; /* power of two: the fixed fp4 scale of the gate's A operand (the stream x + dstep: 2.0 in the numerics study) */
  float q_scale_z;    /* ... of the skip GEMM's A operand (gate outputs in (-1, 1): 0.25) */
  /* optional, as w_dil_q: the folded skip-all weights as a stylesinger_amd.lib.pack_skip_q4 pack; the K = L*C skip GEMM then runs on
   * ss_gemm_bf16_tile256q (split = 3, q_scale = q_scale_z) when ss_gemm_bf16_tile256q_ok says so. gs in 16-bit elements. */
  const uint16_t* w_skipall_q;
  int64_t gs_w_skipall_q;
  /* optional, with mfma_split = 2 and C = 256, one group: per layer the ss_layer512_pack_gate / _pack_res fragment-order packs of w_dil_h / w_out_h.
   * When every layer has them, the "layer512" knob is on and ss_layer512_ok(B, T, ...) holds, the residual stack runs ONE ss_layer512 launch per
   * layer (w_out_f of the last layer is unused). */
  const uint16_t* w_dil_f[SS_MAX_LAYERS];
  const uint16_t* w_out_f[SS_MAX_LAYERS];
  /* "fp16sd" precision (round 6; mfma_split = 2 data path with ONE fp16 weight term and a noise-shaped rounding): n_wsets > 1 -> every w_*_h / w_*_f
   * pointer above addresses set 0 of n_wsets weight sets laid out ws_* elements apart; network evaluation j of a sampling loop (j = 0 for its first
   * evaluation) uses set j % n_wsets. The sets are a first-order sigma-delta sequence of fp16 roundings of the same scaled fp32 weight
   * (r_0 = 0, W_k = RNE16(w 2^s + r_k), r_(k+1) = r_k + (w 2^s - W_k)): their sum is n_wsets * w 2^s up to one fp16 rounding, so the weight rounding
   * averages out over the steps instead of adding up (oracle/dither_numerics.py: 2.2e-5 on the reference's 1000-step golden with 32 sets; plain
   * one-product fp16: 1.94e-4; fp16x2: 1.9e-5). The pair-layout packs (w_*_h) carry zero lo terms - the generic two-product kernels then compute the
   * one-product result exactly - and the fragment-order packs (w_*_f) one term only (ss_layer512 with n_products = 1). mfma_products = 1 says so. */
  int32_t n_wsets;
  int32_t mfma_products;   /* 0 | 2: two weight terms (fp16x2); 1: one (fp16sd) */
  int64_t ws_w_dil_h, ws_w_out_h, ws_w_skipall_h, ws_w_dil_f, ws_w_out_f;
  /* mfma_products = 1, optional: the skip GEMM's weight sets WITHOUT the zero second plane - fp16 [n_wsets][Np][L C], ws_w_skipall_c elements apart.
   * Used (with ss_gemm_bf16_args.one_product = 2) whenever the stack runs in the compact-G form; 2.6 MB per set at C = 256, L = 20 stays in one
   * XCD's 4 MB L2 across the row tiles of a launch, which the 5.2 MB pair-layout pack does not. NULL: the pair-layout pack is used. */
  const uint16_t* w_skipall_c;
  int64_t ws_w_skipall_c;
  /* mfma_products = 1, fused-layer form: > 0 -> the conditioner addend slab of ss_layer512 is kept as n_esets fp16 sets (ss_layer512_tile_addend_f16; the
   * workspace grows by n_esets x L x ss_layer512_addend_halfs x 2 bytes: 29.5 GB for 8 sets at 32 x 5625 rows) and evaluation j reads set j % n_esets;
   * 0 = the fp32 slab. */
  int32_t n_esets;
  int32_t reserved3_;
} ss_wavenet;

/* bytes of scratch the samplers need for (B, T) */
int64_t ss_wavenet_workspace_bytes(const ss_wavenet* net, int B, int T);

/* Shallow mel diffusion, whole reverse loop (DiffusionDecoder.forward infer branch,
 * shallow_diffusion_tts.py:296-306 + p_sample :155-162).
 *   x      [B][T][80] in: x_K (already q_sampled) ; out: x_0 (normalised)
 *   cond   [B][T][cond_dim]
 *   noise  [steps][B][T][80] tape (index s = loop step t) or NULL -> Philox(seed + *seed_dev)
 *   seed_dev: optional device word added to the seed when the kernels run, so a captured hipGraph of this call
 *             draws fresh noise on every replay (NULL = use `seed` alone)
 *   step_lo/step_hi: run t = step_hi-1 ... step_lo (full loop: 0, steps) */
int ss_meldiff_sample(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T,
                      const float* noise, uint64_t seed, const uint64_t* seed_dev, int step_lo, int step_hi,
                      int precompute_cond, void* ws, int64_t ws_bytes, void* stream);
/* Strided DDIM-family sampler over the same denoiser (Song et al. 2021, eq. 12/16): network times ts[0] > ts[1] > ... (HOST array),
 * alphas_cumprod = HOST schedule table [steps] in DOUBLE precision (cumprod(1 - betas); 1 - ac loses its digits in float at small t).
 *   eta = 0: deterministic sampler of BASELINE config 5 (the reference has no such sampler; its strided option is PLMS).
 *   eta = 1 with ts = K-1 ... 0: the reference's ancestral p_sample (shallow_diffusion_tts.py:136-162) - c1/c2 become
 *            posterior_mean_coef1/2, sigma^2 the posterior variance - which pins this entry point to golden acoustic_t64_s100.
 *   noise [steps][B][T][80] tape (index = network time t) or NULL -> Philox(seed + *seed_dev); only read when eta > 0. */
int ss_meldiff_sample_ddim(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T,
                           const int32_t* ts, int n_ts, const double* alphas_cumprod, float eta, const float* noise, uint64_t seed,
                           const uint64_t* seed_dev, int precompute_cond, void* ws, int64_t ws_bytes, void* stream);

/* PLMS ("pndm_speedup") sampler of the reference: modules/diff/shallow_diffusion_tts.py:165-197 (p_sample_plms) driven as
 * in :254-260 - network times reversed(range(0, step_hi, interval)) with step_hi = K_step (the shallow-diffusion depth x
 * was q-sampled to; <= steps), 4-deep eps history, two network evaluations on the first step. `hist` = device scratch of
 * 6*B*T*in_dim floats. alphas_cumprod: HOST table [steps]. */
int ss_meldiff_sample_plms(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T, int step_hi, int interval,
                           const float* alphas_cumprod, int precompute, float* hist, void* ws, int64_t ws_bytes, void* stream);
/* ProDiff teacher sampler (modules/diff/prodiff.py:205-221 with p_sample :150-153 and q_posterior_sample :141-148): the
 * denoiser predicts x0 directly; x_{t-1} = c1[t]*x0 + c2[t]*x_t + sigma[t]*z for t = n_steps-1 ... 0.
 *   x [B][T][80] in: x_T ~ N(0,1) (filled by the caller) ; out: the mel (norm/denorm are identities there, :223-227)
 *   c1, c2, sigma: HOST arrays [n_steps] (posterior_mean_coef1/2, exp(0.5*posterior_log_variance_clipped), sigma[0] = 0)
 *   noise [n_steps][B][T][80] tape or NULL -> Philox */
int ss_prodiff_sample(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T, const float* noise,
                      uint64_t seed, const uint64_t* seed_dev, int n_steps, const float* c1, const float* c2, const float* sigma,
                      int precompute_cond, void* ws, int64_t ws_bytes, void* stream);
/* q_sample + norm_spec: x = sqrt_ac*((mel-min)/(max-min)*2-1) + sqrt_1mac*z  (shallow_diffusion_tts.py:199-204,271-272) */
int ss_mel_qsample(const float* coarse_mel, const float* spec_min, const float* spec_max, float sqrt_ac, float sqrt_1mac,
                   const float* noise, uint64_t seed, const uint64_t* seed_dev, float* x, int B, int T, int M, void* stream);
/* denorm_spec (+ optional row mask): mel = (x+1)/2*(max-min)+min (shallow_diffusion_tts.py:274-275). nonfinite (optional device word, never
 * cleared here): set to 1 when a VALID frame's value is NaN / inf - the cheap, unconditional form of the fp16 modes' range check (their residual
 * stream overflows beyond 65504; the host reads the word wherever it synchronises anyway). */
int ss_mel_denorm(const float* x, const float* spec_min, const float* spec_max, float* mel, int B, int T, int M,
                  const int32_t* lens, int32_t* nonfinite, void* stream);

/* Joint Gaussian(f0)/multinomial(uv) reverse loop (GaussianMultinomialDiffusion.sample,
 * gaussian_multinomial_diffusion.py:922-942 with gaussian_p_sample :326-333, p_sample :410-413).
 *   With net->n_groups == 2 the call runs BOTH samplers at once: every per-item array below is [2B] long, items
 *   [0,B) belong to net 0 (agnostic) and [B,2B) to net 1 (specific); pass B = 2*B_utterances.
 *   f0     [B][T] in: z_f0 ~ N(0,1); out: final normalised f0
 *   uv     [B][T] int32 in: initial class (0); out: final class
 *   lo/hi  [B][T] per-frame clamp bounds (dyn_clip, stylesinger.py:275-283)
 *   noise  [steps][B][T] gaussian tape or NULL ; gumbel_u [steps][B][2][T] uniform tape or NULL */
int ss_f0diff_sample(const ss_wavenet* net, float* f0, int32_t* uv, const float* cond, const float* lo, const float* hi,
                     const int32_t* lens, int B, int T, const float* noise, const float* gumbel_u, uint64_t seed,
                     const uint64_t* seed_dev, int step_lo, int step_hi, int precompute_cond, void* ws, int64_t ws_bytes,
                     void* stream);

/* f0 post-processing of the two predictors (stylesinger.py:216-311, utils/pitch_utils.py:22-31,65-78):
 * midi -> clamp bounds ; merge ; denorm ; coarse. */
int ss_f0_bounds(const int64_t* midi, float* lo, float* hi, int n, void* stream);
int ss_pitch_post(const float* f0_a, const int32_t* uv_a, const float* f0_b, const int32_t* uv_b, const int64_t* midi,
                  const int64_t* mel2ph, float* pitch_pred /*[n][2]*/, float* f0_denorm, int64_t* pitch_coarse, int n,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * HiFi-GAN-NSF vocoder (modules/hifigan/hifigan_nsf.py:105-169, modules/parallel_wavegan/models/source.py:311-531)
 * ------------------------------------------------------------------------------------------ */
#define SS_HG_MAX_UPS 6
#define SS_HG_MAX_KERNELS 4
typedef struct ss_hifigan {
  int32_t n_ups, n_kernels, c0, sr, harmonics; /* harmonics = 8 -> 9 sine channels */
  int32_t up_rate[SS_HG_MAX_UPS], up_k[SS_HG_MAX_UPS];
  int32_t rb_k[SS_HG_MAX_KERNELS], rb_d[SS_HG_MAX_KERNELS][3];
  const float* w_pre; /* packed [c0][7*Kp(80)] */
  const float* b_pre;
  const float* w_up[SS_HG_MAX_UPS][2]; /* two polyphase groups */
  const float* b_up[SS_HG_MAX_UPS];    /* [u*Cout] repeated per phase */
  const float* w_noise[SS_HG_MAX_UPS]; /* raw [c][1][k] */
  const float* b_noise[SS_HG_MAX_UPS];
  const float* w_rb1[SS_HG_MAX_UPS][SS_HG_MAX_KERNELS][3]; /* convs1 packed */
  const float* b_rb1[SS_HG_MAX_UPS][SS_HG_MAX_KERNELS][3];
  const float* w_rb2[SS_HG_MAX_UPS][SS_HG_MAX_KERNELS][3]; /* convs2 packed */
  const float* b_rb2[SS_HG_MAX_UPS][SS_HG_MAX_KERNELS][3];
  const float* w_post; /* raw [1][c_last][7] */
  const float* b_post;
  const float* src_w; /* SourceModuleHnNSF.l_linear.weight [9] */
  const float* src_b; /* [1] */
  /* 1 = bf16-operand MFMA for conv_pre, the transposed convs and the residual blocks (conv_post and the NSF source stay fp32) */
  int32_t mfma_bf16;
  /* 1 = ResBlock convs with a Winograd pack below run as grouped F(4,3) (ss_wino43_conv; fp32 mode, C a multiple of 64, k in {3,7,11},
   * dilation in {1,3,5}); 0 / NULL pack = direct ss_conv_gemm. Same results to fp32 rounding (wav max error 1.3e-7 vs 0.9e-7 on the
   * reference's golden cases, oracle/wino_vocoder_numerics.py) */
  int32_t wino;
  const float* w_rb1_wino[SS_HG_MAX_UPS][SS_HG_MAX_KERNELS][3]; /* ss_pack_conv_weight of the [C][C][6*ceil(k/3)] transformed taps */
  const float* w_rb2_wino[SS_HG_MAX_UPS][SS_HG_MAX_KERNELS][3];
} ss_hifigan;

int64_t ss_hifigan_workspace_bytes(const ss_hifigan* hg, int B, int T);
/* mel [B][T][80] (already clipped), f0 [B][T] Hz -> wav [B][T*hop].  Noise tapes (or NULL -> Philox):
 *   rand_ini [B][9] uniform ; sine_noise [B][L][9] normal.  lens = valid frames per item. */
int ss_hifigan_forward(const ss_hifigan* hg, const float* mel, const float* f0, const int32_t* lens, int B, int T,
                       const float* rand_ini, const float* sine_noise, uint64_t seed, float* wav, float* har_source_out,
                       void* ws, int64_t ws_bytes, void* stream);

/* NSF harmonic source only (SineGen + l_linear + tanh): f0 [B][T] Hz -> har [B][T*hop] */
int ss_hifigan_source(const ss_hifigan* hg, const float* f0, int B, int T, const float* rand_ini, const float* sine_noise,
                      uint64_t seed, float* har, void* ws, int64_t ws_bytes, void* stream);

/* utility: y = clip(x, lo, hi) ; fill ; philox normal fill (for tests/bench inputs on device) */
int ss_clip(const float* x, float* y, int64_t n, float lo, float hi, void* stream);
/* reference-audio front end (utils/audios/__init__.py:36-84 librosa_wav2spec): the windowed DFT and the mel filterbank
 * are two ss_conv_gemm launches (frames = 4 taps over the waveform viewed as [L/hop][hop]); these two kernels are the
 * element-wise steps between and after them: |X| from the (re | im) column blocks, and log10(max(eps, .)). */
int ss_spec_magnitude(const float* S, float* P, int64_t rows, int lds, int ldp, int nbins, int sin_off, void* stream);
int ss_log10_floor(const float* x, float* y, int64_t n, float eps, void* stream);
/* power spectrum re^2 + im^2 from the same (re | im) layout: librosa.feature.melspectrogram(power=2.0) as the emotion encoder's
 * 40-mel front end uses it (data_gen/tts/emotion/audio.py:43-55) */
int ss_spec_power(const float* S, float* P, int64_t rows, int lds, int ldp, int nbins, int sin_off, void* stream);
/* y[b][i] = x[b][reflect(i - pad)], i < lens[b] + 2*pad, 0 beyond: numpy.pad(mode="reflect"), the centre padding of librosa.stft
 * (pad_mode="reflect", librosa 0.8.0 default) per item of a ragged batch. x [B][Lx], y [B][Ly], lens NULL = Lx. */
int ss_reflect_pad(const float* x, const int32_t* lens, float* y, int B, int Lx, int Ly, int pad, void* stream);
/* Reference-f0 conditioning (utils/pitch_utils.py:34-62 norm_f0 + norm_interp_f0 with pitch_norm='log', use_uv; called at
 * inference/StyleSinger.py:152): f0_hz [B][T] (0 = unvoiced) -> out [B][T] = log2(f0 + 1e-8) on voiced frames, np.interp
 * between voiced neighbours on unvoiced ones (flat beyond the first/last voiced frame, 0 when nothing is voiced);
 * uv [B][T] = 1.0 on unvoiced frames. Frames >= lens[b] (NULL = T) are written as 0. Inputs and outputs must not alias. */
int ss_norm_interp_f0(const float* f0_hz, const int32_t* lens, float* out, float* uv, int B, int T, void* stream);

/* f0 tracker (input producer; replaces inference/StyleSinger.py:125-127:
 *   parselmouth.Sound(wav, sr).to_pitch_ac(time_step, voicing_threshold=0.6, pitch_floor=80, pitch_ceiling=800).selected_array['frequency']
 * and the padding onto the mel frame grid of :128-135). parselmouth / Praat are un-vendored: the kernels follow the published algorithm
 * (Boersma 1993; Hanning window of three floor periods) - PARITY UNPINNED, oracle/praat_pitch.py is the CPU restatement. float64 throughout.
 * Geometry (host side: stylesinger_amd/f0track.py::geometry, the paper's / the manual's formulas): */
typedef struct ss_f0track_params {
  double sample_rate, time_step /* s */, pitch_floor, pitch_ceiling, voicing_threshold, silence_threshold, octave_cost, octave_jump_cost,
      voiced_unvoiced_cost;
  int32_t nsamp_window;     /* even: 2 * halfnsamp_window */
  int32_t halfnsamp_window; /* floor(3 / floor / dx) / 2 - 1 */
  int32_t nsamp_period;     /* floor(1 / dx / floor) */
  int32_t halfnsamp_period; /* nsamp_period / 2 + 1 */
  int32_t maximum_lag;      /* min(floor(nsamp_window / 3) + 2, nsamp_window) */
  int32_t nlag;             /* floor(nsamp_window / 2): lags 0 .. nlag are kept per frame (< 1024) */
  int32_t hop;              /* time_step in samples (must be an integer) */
  int32_t reserved_;
} ss_f0track_params;
int64_t ss_f0track_workspace_bytes(int B, int max_frames, int nlag);
/* wav [B][wav_stride] fp32 (zero beyond n_samples[b]); n_frames[b] analysis frames of item b, frame i centred between samples
 * left0[b] + i * hop and left0[b] + i * hop + 1 (0-based); window [nsamp_window] = the Hanning window, window_r [nlag + 1] = its normalised
 * autocorrelation (float64, device). f0_out [B][ld_out] fp32 <- 0 everywhere, then the selected frequency of frame i (0 = unvoiced) at
 * column lpad + i (the reference pads the contour by 2 * pad_size frames on the left and zeros on the right, :128-130). */
int ss_f0track(const float* wav, int64_t wav_stride, const int32_t* n_samples, const int32_t* n_frames, const int32_t* left0, int B, int max_frames,
               const ss_f0track_params* prm, const double* window, const double* window_r, float* f0_out, int ld_out, int lpad, void* workspace,
               int64_t workspace_bytes, void* stream);

/* audio.normalize_volume(wav, target_dbfs, increase_only=True) (data_gen/tts/emotion/audio.py:109-115, as preprocess_wav applies it) per item of a
 * zero-padded batch wav [B][L]: out = wav * 10^(change / 20) with change = target_dbfs - 10 log10(mean over the item's lens[b] samples of wav^2)
 * when change >= 0, else out = wav. In place allowed. */
int ss_normalize_volume(const float* wav, const int32_t* lens, float* out, int B, int L, float target_dbfs, void* stream);
/* y[b][t] = fp32(fp16_rne(x[b][t])) for t < min(n_out[b], n_in[b], Lx) (n_in: the item's own sample count, NULL = Lx), 0 for the rest of the ldy
 * columns: the waveform `process_audio` returns
 * (inference/StyleSinger.py:86-88: padded to n_mel * hop samples, `.astype(np.float16)`), which the reference hands resemblyzer and parselmouth. */
int ss_round_f16_rows(const float* x, int64_t ldx, int Lx, const int32_t* n_in, const int32_t* n_out, float* y, int64_t ldy, int B, void* stream);
/* trim_long_silences (data_gen/tts/emotion/audio.py:58-100; called by preprocess_wav :38) AROUND the caller's voice-activity flags: the decision
 * itself is webrtcvad's (an un-vendored C library, no published text to restate); the reference's windowing, smoothing, dilation and compaction
 * run here. wav [B][wav_stride] fp32, n_samples[b] valid samples; flags [B][flags_stride] uint8, one per window of samples_per_window samples
 * (floor(n_samples / samples_per_window) of them are read). A window is kept iff, after a moving average of avg_width windows rounded half to even
 * (more than half of the neighbourhood voiced), any window within max_silence / 2 of it (binary_dilation with ones(max_silence + 1)) is voiced.
 * out [B][out_stride] <- the kept windows back to back, zeros behind; out_lens[b] = kept samples; win_dst [B][max_windows] int32 scratch. */
int ss_vad_trim(const float* wav, int64_t wav_stride, const int32_t* n_samples, const uint8_t* flags, int flags_stride, int B, int max_windows,
                int samples_per_window, int avg_width, int max_silence, float* out, int64_t out_stride, int32_t* out_lens, int32_t* win_dst, void* stream);

/* Emotion encoder (input producer; data_gen/tts/emotion/model.py:11-78 = nn.LSTM(40, 256, 3) + Linear, inference.py:39-53,
 * 139-151). One LSTM layer's recurrence as a persistent launch (one workgroup per sequence):
 *   xproj  [P][n][H][4] = x_t . W_ih^T + b_ih + b_hh for every step, gate-interleaved (i,f,g,o per hidden unit) - one
 *          ss_conv_gemm over all P*n rows with the weight rows permuted to 4*j + gate;
 *   w_hh_packed [H (k)][H (j)][4 (gate)] = weight_hh[gate*H + j][k];
 *   h_seq  [P][n][H] every hidden state (input of the next layer) or NULL ; h_last [P][H] final hidden state or NULL. */
int ss_lstm_layer(const float* xproj, const float* w_hh_packed, float* h_seq, float* h_last, int P, int n, int H, void* stream);
/* out[C] = normalise(mean over rows of x[rows][C])  (embed_utterance, inference.py:147-151) */
int ss_mean_l2norm(const float* x, float* out, int rows, int C, void* stream);
/* y[r][:] = x[r][:] / ||x[r][:]||_2  (EmotionEncoder.forward, model.py:57-58) */
int ss_l2norm_rows(const float* x, float* y, int rows, int C, void* stream);

/* output writer (utils/audio.py:12-17 save_wav): pcm = (int16) trunc(wav * scale), scale = 32767 (or 32767 / max|wav| when
 * out_wav_norm is set); saturating. */
int ss_wav_to_pcm16(const float* wav, int16_t* pcm, int64_t n, float scale, void* stream);
int ss_fill_normal(float* x, int64_t n, uint64_t seed, const uint64_t* seed_dev, uint64_t offset, void* stream);
/* x[b][t] ~ N(0,1) for t < T with Philox counter (t/4, b): the values of the first T' <= T columns do not depend on T, so
 * padding the frame axis to a hipGraph bucket leaves the noise of the real frames unchanged. ld = row stride (floats). */
int ss_fill_normal_rows(float* x, int B, int T, int ld, uint64_t seed, const uint64_t* seed_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STYLESINGER_HIP_H */
