// Diffusion samplers: the three reverse loops of the hot path, as host-side launch sequences over
// the fp32-MFMA conv kernel plus small fused sampler-update kernels.
//
//   mel : DiffusionDecoder.forward (modules/diff/shallow_diffusion_tts.py:285-307) with DiffNet (modules/diff/net.py:81-130)
//   f0  : GaussianMultinomialDiffusion.sample (modules/diff/gaussian_multinomial_diffusion.py:922-942) with DDiffNet (net.py:215-266)
//
// Restructuring vs the reference (same arithmetic, fewer flops / launches):
//   * conditioner_projection(cond) of every layer is step-invariant -> computed ONCE per utterance
//     batch as a single [B*T, 256] x [256, L*2C] GEMM ("E"), and the dilated-conv bias is folded into it;
//   * diffusion_projection_l(mlp(sinemb(t))) depends on weights only -> a [steps][L][C] table built at
//     load time and applied as a per-channel bias in the A-operand prologue of the dilated conv;
//   * gate (sigmoid*tanh), residual/sqrt(2), skip accumulation and the DDPM posterior step are GEMM epilogues.
// Nothing here allocates or synchronises: the whole loop is hipGraph-capturable.
#include "common.h"
#include <stdlib.h>
#include "../../include/stylesinger_hip.h"

namespace {

struct WsLayout {
  float* E;   // [B*T][L*2C]
  float* X;   // [B*T][C]
  float* G;   // [B*T][C]
  float* S;   // [B*T][C]
  float* O;   // [B*T][4]   (f0 net output)
  float* GA;  // [B*T][L*C] gate outputs of ALL layers (deferred-skip mode only, else null)
  float* E16[SS_MAX_LAYERS];  // per layer: the conditioner addend in the 16x16x4 gate kernel's fetch order (null: the layer reads E)
  int mt16[SS_MAX_LAYERS];    // ... and the tiling it was laid out for (0 = none)
  float* KP;  // [ksplit][B*T][C] partial sums of the split-K skip GEMM (small launches only, else null)
  int ksplit; // K slices of the skip GEMM for this (B, T): ss_gemm16_ksplit_pick
  // bf16-in-HBM mode (net->w_dil_h set): the hidden activations travel as bf16
  uint16_t* Yh;     // [B*T][C]        x + dstep of the next layer (the dilated conv's operand)
  uint16_t* GAh;    // [B*T][L*C]      gate outputs of all layers
  uint16_t* condh;  // [B*T][cond_dim] conditioner
  // fused-layer form of the fp16x2 stack (ss_layer512, one launch per layer): the stream as hi rows (double buffered) + pairs in accumulator
  // order, and every layer's addend slab in the kernel's accumulator order; null when the stack runs as gate + projection launches
  uint16_t* H512[2];  // ss_layer512_h_elems: slot-major tiles
  void* P512;         // ss_layer512_stream_bytes
  float* E512;        // [L][ss_layer512_addend_floats]
  int64_t e512_layer; // floats per layer
  uint16_t* E512h;    // fp16sd with ss_wavenet.n_esets > 0: [set][L][ss_layer512_addend_halfs] instead of E512
  int64_t e512h_layer, e512h_set;
  bool g_compact;     // fused form whose skip GEMM runs on the many-round kernel: GAh rows hold the L*C hi terms only (no dead second plane)
  int64_t bytes;
};

// hmode: the hidden activations travel as bf16 in HBM. split (net->mfma_split, "bf16x2" precision): every bf16 operand is a (hi, mid) pair,
// pairs interleaved by 32 channels (one 128-byte line = 32 channels of both planes; rows of Yh / GAh / the weights are twice as long) - and
// the matrix cores run hi*hi + hi*mid + mid*hi (ss_gemm_bf16_args.split); the hoisted conditioner projection then runs in exact fp32 (it is
// outside the step loop). mfma_split = 2 ("fp16x2"): the same layouts with fp16 terms, weights-only split (two products), accumulators scaled
// by net->mfma_out_scale (ss_gemm_bf16_args.split = 2).
inline bool smode(const ss_wavenet* net) { return net->mfma_split != 0; }
inline bool hmode(const ss_wavenet* net) { return net->mfma_bf16 && net->w_dil_h[0] && net->w_skipall_h && (net->w_cond_h || smode(net)); }

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// "fp16sd": which of the net's noise-shaped weight sets the CURRENT network evaluation uses. The samplers set the evaluation index before every
// evaluation (0 for the first one of a loop, in an order that does not depend on how a loop is cut into calls); the stack adds set * stride to
// every 16-bit weight pointer. Launch parameters only: a captured hipGraph replays the same sequence.
thread_local int g_wset_eval = 0;
inline int64_t wset_off(const ss_wavenet* net, int64_t stride) { return net->n_wsets > 1 ? (int64_t)(g_wset_eval % net->n_wsets) * stride : 0; }

// the fp16x2 mel stack as ONE ss_layer512 launch per layer: the net carries the fragment-order packs of every layer, one weight group, C = 256,
// dilations <= 8, and the launch fills the chip (ss_layer512_ok); knob "layer512"
inline bool fused512(const ss_wavenet* net, int B, int T) {
  if (!g_ss_tuning.layer512 || net->mfma_split != 2 || !hmode(net) || net->n_groups > 1 || net->C != 256 || !net->w_skipall_h) return false;
  if (net->w_dil_q[0] && net->q_scale_gate > 0.f) return false;   // "fp16q4": its gate runs the second product on the fp4 instruction (gate128q), two launches per layer
  for (int l = 0; l < net->L; ++l)
    if (!net->w_dil_f[l] || (l + 1 < net->L && !net->w_out_f[l])) return false;
  const int dmax = 1 << ((net->L < net->dil_cycle ? net->L : net->dil_cycle) - 1);
  return ss_layer512_ok(B, T, net->C, dmax, net->L * net->C * 2) != 0;
}

WsLayout ws_layout(const ss_wavenet* net, int B, int T, void* base) {
  WsLayout w;
  const int64_t rows = (int64_t)B * T;
  int64_t off = 0;
  char* p = (char*)base;
  auto take = [&](int64_t floats) {
    float* r = (float*)(p + off);
    off = align_up(off + floats * 4, 256);
    return r;
  };
  w.E = take(rows * net->L * 2 * net->C);
  w.X = take(rows * net->C);
  w.G = take(rows * net->C);
  w.S = take(rows * net->C);
  w.O = take(rows * 4);
  const bool h = hmode(net);
  w.GA = (net->w_skipall && !h) ? take(rows * net->L * net->C) : nullptr;
  // fp32 F(4,3) loops on the 16x16x4 gate kernel: every layer's slab of E is re-laid once per forward in that kernel's fetch order
  // (ss_gate16_tile_addend), so that a wave's addend fetch is 1 KB contiguous per instruction instead of 8 lines x 32 B ("e16" knob)
  for (int l = 0; l < SS_MAX_LAYERS; ++l) {
    w.E16[l] = nullptr;
    w.mt16[l] = 0;
  }
  if (!h && !net->mfma_bf16 && !net->mfma_x3 && net->wino_m == 4 && g_ss_tuning.gate16 != 0 && g_ss_tuning.e16 != 0)
    for (int l = 0; l < net->L && l < SS_MAX_LAYERS; ++l) {
      if (!net->w_dil_wino[l] || !net->w_dil_wino16[l]) continue;
      const int d = 1 << (l % net->dil_cycle);
      int mt = g_ss_tuning.gate16 == 1 ? ss_wino43_gate16_pick(B, T, 2 * net->C, d) : g_ss_tuning.gate16;
      if (mt == 1 && !(g_ss_tuning.gate16_ks != 0 && net->C >= 64)) mt = 2;   // as ss_wino43_gate16 resolves it (MT = 1 is K-staged only)
      const int64_t fl = mt > 0 ? ss_gate16_tiled_floats(B, T, 2 * net->C, d, mt) : -1;
      if (fl <= 0 || fl * 4 >= (1ll << 31)) continue;
      w.E16[l] = take(fl);
      w.mt16[l] = mt;
    }
  w.ksplit = (net->w_skipall && !h && !net->mfma_bf16 && g_ss_tuning.skip16 != 0 && (net->C & 3) == 0) ? ss_gemm16_ksplit_pick(B, T, net->C, net->L * net->C) : 1;
  w.KP = w.ksplit > 1 ? take((int64_t)w.ksplit * rows * net->C) : nullptr;
  const int planes = smode(net) ? 2 : 1;
  w.Yh = h ? (uint16_t*)take((rows * net->C * planes + 1) / 2) : nullptr;
  // the skip GEMM's many-round kernel reads a compact operand (ss_gemm_bf16_args.a_compact; its size rule: >= 2 rounds of 256-row tiles)
  w.g_compact = fused512(net, B, T) && g_ss_tuning.gate256 != 0 && (long)ss_cdiv(T, 256) * B >= 2L * ss_n_cu() && ((net->L * net->C) % 64) == 0 && net->C <= 256;
  w.GAh = h ? (uint16_t*)take((rows * net->L * net->C * (w.g_compact ? 1 : planes) + 1) / 2) : nullptr;
  w.condh = (h && !smode(net)) ? (uint16_t*)take((rows * net->cond_dim + 1) / 2) : nullptr;
  w.H512[0] = w.H512[1] = nullptr;
  w.P512 = nullptr;
  w.E512 = nullptr;
  w.e512_layer = 0;
  w.E512h = nullptr;
  w.e512h_layer = w.e512h_set = 0;
  if (fused512(net, B, T)) {
    w.H512[0] = (uint16_t*)take((ss_layer512_h_elems(B, T) + 1) / 2);
    w.H512[1] = (uint16_t*)take((ss_layer512_h_elems(B, T) + 1) / 2);
    w.P512 = take(ss_layer512_stream_bytes(B, T) / 4);
    if (net->mfma_products == 1 && net->n_esets > 0) {   // the addend as n_esets fp16 sets (half the bytes per launch; ss_layer512_tile_addend_f16)
      w.e512h_layer = ss_layer512_addend_halfs(B, T);
      w.e512h_set = w.e512h_layer * net->L;
      w.E512h = (uint16_t*)take((w.e512h_set * net->n_esets + 1) / 2);
    } else {
      w.e512_layer = ss_layer512_addend_floats(B, T);
      w.E512 = take(w.e512_layer * net->L);
    }
  }
  w.bytes = off;
  return w;
}

inline ss_conv_gemm_args base_args(int B, int T, const int32_t* lens) {
  ss_conv_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.B = B;
  a.T = T;
  a.lens = lens;
  a.ntaps = 1;
  a.a_scale = 1.0f;
  a.a_lrelu = 1.0f;
  a.pre_scale = 1.0f;
  a.post_scale = 1.0f;
  a.mask_rows = 1;
  return a;
}

inline int round_up32(int x) { return (x + 31) / 32 * 32; }

// E = cond . Wc^T + (bc + b_dil)  for all layers at once
inline ss_gemm_bf16_args base_args_h(const ss_wavenet* net, int B, int T, const int32_t* lens) {
  ss_gemm_bf16_args a;
  memset(&a, 0, sizeof(a));
  a.B = B;
  a.T = T;
  a.lens = lens;
  a.ntaps = 1;
  a.post_scale = 1.0f;
  a.mask_rows = 1;
  if (net->n_groups > 1) a.group_size = B / net->n_groups;
  return a;
}

int precompute_cond(const ss_wavenet* net, const float* cond, const int32_t* lens, int B, int T, const WsLayout& w,
                    hipStream_t stream) {
  const int NE = net->L * 2 * net->C;
  if (hmode(net) && !smode(net)) {  // cond rounded once to bf16, then E = cond . Wc^T + (bc + b_dil) on the bf16 kernel (fp32 out)
    SS_PROPAGATE(ss_to_bf16(cond, nullptr, w.condh, B, T, net->cond_dim, net->cond_dim, net->cond_dim, nullptr, 0, 0, stream));
    ss_gemm_bf16_args h = base_args_h(net, B, T, lens);
    h.A = w.condh;
    h.lda = net->cond_dim;
    h.a_batch_stride = (int64_t)T * net->cond_dim;
    h.K = net->cond_dim;
    h.W = net->w_cond_h;
    h.w_group_stride = net->gs_w_cond_h;
    h.N = NE;
    h.Np = NE;
    h.epi = SS_HEPI_STORE;
    h.bias = net->b_cond;
    h.bias_group_stride = net->gs_b_cond;
    h.C = w.E;
    h.ldc = NE;
    h.c_batch_stride = (int64_t)T * NE;
    h.mask_rows = 0;
    return ss_gemm_bf16(&h, stream);
  }
  ss_conv_gemm_args a = base_args(B, T, lens);
  a.A = cond;
  a.lda = net->cond_dim;
  a.a_batch_stride = (int64_t)T * net->cond_dim;
  a.Cin = net->cond_dim;
  a.W = net->w_cond;
  a.N = NE;
  a.Np = NE;
  a.Kp = round_up32(net->cond_dim);
  a.epi = SS_EPI_STORE;
  a.bias = net->b_cond;
  a.C = w.E;
  a.ldc = NE;
  a.c_batch_stride = (int64_t)T * NE;
  a.mask_rows = 0;
  a.mfma_bf16 = smode(net) ? 0 : net->mfma_bf16;   // split mode: exact fp32 (a fixed rounding error of E would enter every step)
  if (net->n_groups > 1) {
    a.group_size = B / net->n_groups;
    a.w_group_stride = net->gs_w_cond;
    a.bias_group_stride = net->gs_b_cond;
  }
  SS_PROPAGATE(ss_conv_gemm(&a, stream));
  if (w.E512)   // fused-layer form: every layer's 512 addend columns in ss_layer512's accumulator order
    for (int l = 0; l < net->L; ++l)
      SS_PROPAGATE(ss_layer512_tile_addend(w.E + (int64_t)l * 2 * net->C, NE, (int64_t)T * NE, w.E512 + (int64_t)l * w.e512_layer, B, T, stream));
  if (w.E512h)  // ... as n_esets fp16 sigma-delta sets (fp16sd)
    for (int l = 0; l < net->L; ++l)
      SS_PROPAGATE(ss_layer512_tile_addend_f16(w.E + (int64_t)l * 2 * net->C, NE, (int64_t)T * NE, w.E512h + (int64_t)l * w.e512h_layer, net->n_esets, w.e512h_set, B, T,
                                               stream));
  for (int l = 0; l < net->L; ++l)
    if (w.E16[l])
      SS_PROPAGATE(ss_gate16_tile_addend(w.E + (int64_t)l * 2 * net->C, NE, (int64_t)T * NE, w.E16[l], B, T, 2 * net->C, 1 << (l % net->dil_cycle),
                                         w.mt16[l], stream));
  return SS_OK;
}

// the L residual layers + skip projection; X in/out, leaves relu(skip_projection) in G.
// Deferred-skip mode (net->w_skipall set): the per-layer output projection only runs its residual half (N = C, no skip
// read-modify-write), every layer's gate output is kept ([rows][L*C]) and the skip sum of all layers is ONE GEMM with
// K = L*C at the end of the stack - same products, summed in one accumulator chain instead of layer by layer.
// bf16-in-HBM form of the stack: Yh = bf16(x + dstep_0) on entry; per layer gate (Yh -> GAh[l], bf16) and residual projection
// (GAh[l] -> X fp32 in place, Yh = bf16(x + dstep_{l+1})); then the K = L*C skip GEMM -> S (fp32). Same operand roundings as
// oracle/restatement.py with set_matmul_rounding("bf16") - each operand is rounded once, where it is produced.
int run_residual_stack_h(const ss_wavenet* net, int step, const int32_t* lens, int B, int T, const WsLayout& w, hipStream_t stream) {
  const int C = net->C, L = net->L;
  const int NE = L * 2 * C;
  const int sp = smode(net) ? (net->mfma_split == 2 ? 2 : 1) : 0, pl = sp ? 2 : 1;   // split form; planes per 16-bit row
  const bool fused = w.H512[0] != nullptr;
  for (int l = 0; l < L && fused; ++l) {   // fused-layer form: gate + residual projection of layer l in one launch, G kept in LDS
    ss_layer512_args f;
    memset(&f, 0, sizeof(f));
    f.Hin = w.H512[l & 1];
    f.d = 1 << (l % net->dil_cycle);
    f.lens = lens;
    f.B = B;
    f.T = T;
    f.Wg = net->w_dil_f[l] + wset_off(net, net->ws_w_dil_f);
    f.n_products = net->mfma_products == 1 ? 1 : 2;
    if (w.E512h) {   // evaluation j reads addend set j % n_esets
      f.E512 = reinterpret_cast<const float*>(w.E512h + (int64_t)(g_wset_eval % net->n_esets) * w.e512h_set + (int64_t)l * w.e512h_layer);
      f.e_f16 = 1;
    } else {
      f.E512 = w.E512 + (int64_t)l * w.e512_layer;
    }
    const int gpl = w.g_compact ? 1 : pl;   // planes per G row
    f.G = w.GAh + (int64_t)l * C * gpl;
    f.g_batch_stride = (int64_t)T * L * C * gpl;
    f.ldg = L * C * gpl;
    f.g_compact = w.g_compact ? 1 : 0;
    f.mask_rows = 1;
    f.out_scale = net->mfma_out_scale;
    f.post_scale = 0.70710678118654752440f;
    if (l + 1 < L) {   // the residual stream of the last layer is never read (net.py:120-127)
      f.Hout = w.H512[(l & 1) ^ 1];
      f.P = w.P512;
      f.Wr = net->w_out_f[l] + wset_off(net, net->ws_w_out_f);
      f.bias_r = net->b_out[l];
      f.next_bias = net->dstep + ((int64_t)step * L + l + 1) * C;
      f.cur_bias = net->dstep + ((int64_t)step * L + l) * C;
    }
    SS_PROPAGATE(ss_layer512(&f, stream));
  }
  for (int l = 0; l < L && !fused; ++l) {
    const int d = 1 << (l % net->dil_cycle);
    ss_gemm_bf16_args g = base_args_h(net, B, T, lens);
    g.A = w.Yh;
    g.lda = C * pl;
    g.a_batch_stride = (int64_t)T * C * pl;
    g.split = sp;
    g.out_scale = net->mfma_out_scale;
    g.K = C;
    g.ntaps = 3;
    g.tap_off[0] = -d;
    g.tap_off[1] = 0;
    g.tap_off[2] = d;
    g.W = net->w_dil_h[l] + wset_off(net, net->ws_w_dil_h);
    g.w_group_stride = net->gs_w_dil_h;
    g.N = C;
    g.Np = 2 * C;
    g.epi = SS_HEPI_GATE;
    g.gate_mode = 0;
    g.E = w.E + (int64_t)l * 2 * C;
    g.lde = NE;
    g.e_batch_stride = (int64_t)T * NE;
    g.C = w.GAh + (int64_t)l * C * pl;
    g.ldc = L * C * pl;
    g.c_batch_stride = (int64_t)T * L * C * pl;
    bool gate_done = false;
    if (sp == 2 && net->w_dil_q[l] && net->q_scale_gate > 0.f) {   // "fp16q4": this layer's gate with its second product on the fp4 instruction, when the launch qualifies
      ss_gemm_bf16_args q = g;
      q.split = 3;
      q.W = net->w_dil_q[l];
      q.w_group_stride = net->gs_w_dil_q;
      q.q_scale = net->q_scale_gate;
      if (ss_gemm_bf16_gate128q_ok(&q)) {
        SS_PROPAGATE(ss_gemm_bf16_gate128q(&q, stream));
        gate_done = true;
      }
    }
    if (!gate_done) SS_PROPAGATE(ss_gemm_bf16(&g, stream));
    // the residual stream of the LAST layer is never read (only the skip sum leaves the stack, net.py:120-127): no projection for it
    if (l + 1 == L) break;
    ss_gemm_bf16_args o = base_args_h(net, B, T, lens);
    o.A = w.GAh + (int64_t)l * C * pl;
    o.lda = L * C * pl;
    o.a_batch_stride = (int64_t)T * L * C * pl;
    o.split = sp;
    o.out_scale = net->mfma_out_scale;
    o.K = C;
    o.W = net->w_out_h[l] + wset_off(net, net->ws_w_out_h);
    o.w_group_stride = net->gs_w_out_h;
    o.N = C;
    o.Np = C;  // the residual half = the first C packed rows
    o.epi = SS_HEPI_RESX;
    o.bias = net->b_out[l];
    o.bias_group_stride = net->gs_b_out;
    o.post_scale = 0.70710678118654752440f;
    o.next_bias = net->dstep + ((int64_t)step * L + l + 1) * C;
    o.next_bias_group_stride = net->gs_dstep;
    o.Y = w.Yh;
    o.ldy = C * pl;
    o.y_batch_stride = (int64_t)T * C * pl;
    if (sp) {   // split mode: the stream lives only as the pair Yh = x + dstep_l (16 significant bits; measured harmless, oracle/bf16x2_numerics.py)
      o.cur_bias = net->dstep + ((int64_t)step * L + l) * C;
      o.cur_bias_group_stride = net->gs_dstep;
    } else {
      o.X = w.X;
      o.ldx = C;
      o.x_batch_stride = (int64_t)T * C;
    }
    SS_PROPAGATE(ss_gemm_bf16(&o, stream));
  }
  ss_gemm_bf16_args k = base_args_h(net, B, T, lens);
  k.A = w.GAh;
  k.lda = L * C * (w.g_compact ? 1 : pl);
  k.a_batch_stride = (int64_t)T * L * C * (w.g_compact ? 1 : pl);
  k.a_compact = w.g_compact ? 1 : 0;
  k.split = sp;
  k.out_scale = net->mfma_out_scale;
  k.K = L * C;
  k.W = net->w_skipall_h + wset_off(net, net->ws_w_skipall_h);
  k.one_product = net->mfma_products == 1 ? 1 : 0;
  if (w.g_compact && net->mfma_products == 1 && net->w_skipall_c && net->n_groups <= 1) {   // the one-term sets without their zero plane (L2-resident)
    k.W = net->w_skipall_c + wset_off(net, net->ws_w_skipall_c);
    k.one_product = 2;
  }
  k.w_group_stride = net->gs_w_skipall_h;
  k.N = C;
  k.Np = round_up32(C);
  k.epi = SS_HEPI_STORE;
  k.bias = net->b_skipall;
  k.bias_group_stride = net->gs_b_skipall;
  k.C = w.S;
  k.ldc = C;
  k.c_batch_stride = (int64_t)T * C;
  if (net->skipall_folded) {   // w_skipall_h already carries skip_projection / sqrt(L): this GEMM + ReLU is the stack's output
    k.act = SS_ACT_RELU;
    k.C = w.G;
  }
  if (sp == 2 && net->w_skipall_q && net->q_scale_z > 0.f) {   // "fp16q4": the second product on the fp4 instruction, when the launch qualifies
    ss_gemm_bf16_args q = k;
    q.split = 3;
    q.W = net->w_skipall_q;
    q.w_group_stride = net->gs_w_skipall_q;
    q.q_scale = net->q_scale_z;
    if (ss_gemm_bf16_tile256q_ok(&q)) return ss_gemm_bf16_tile256q(&q, stream);
  }
  return ss_gemm_bf16(&k, stream);
}

// Yh = bf16(X + dstep[step][0]) : the first layer's conv operand (bf16-in-HBM mode)
int stack_entry_h(const ss_wavenet* net, int step, const int32_t* lens, int B, int T, const WsLayout& w, hipStream_t stream) {
  if (w.H512[0])
    return ss_layer512_entry(w.X, net->C, (int64_t)T * net->C, net->dstep + (int64_t)step * net->L * net->C, lens, w.H512[0], w.P512, B, T, stream);
  if (net->mfma_split == 2)
    return ss_split_f16(w.X, net->dstep + (int64_t)step * net->L * net->C, 1.0f, w.Yh, B, T, net->C, net->C, 2 * net->C, lens,
                        net->n_groups > 1 ? B / net->n_groups : 0, net->gs_dstep, stream);
  if (smode(net))
    return ss_split_bf16(w.X, net->dstep + (int64_t)step * net->L * net->C, w.Yh, B, T, net->C, net->C, 2 * net->C, lens,
                         net->n_groups > 1 ? B / net->n_groups : 0, net->gs_dstep, stream);
  return ss_to_bf16(w.X, net->dstep + (int64_t)step * net->L * net->C, w.Yh, B, T, net->C, net->C, net->C, lens,
                    net->n_groups > 1 ? B / net->n_groups : 0, net->gs_dstep, stream);
}

// partials_ok: the caller's next kernel (f0_tail_kernel / mel_tail_kernel) can add the split-K slices of the skip GEMM itself (skip_partials()
// tells it whether it has to): the reduction launch is then left out
inline bool skip_partials(const ss_wavenet* net, const WsLayout& w) {
  return w.ksplit > 1 && net->skipall_folded && !hmode(net) && !net->mfma_bf16 && g_ss_tuning.skip16 != 0 && !(net->mfma_x3 && net->w_skipall_x3);
}
int run_residual_stack(const ss_wavenet* net, int step, const int32_t* lens, int B, int T, const WsLayout& w,
                       hipStream_t stream, bool partials_ok = false) {
  const int C = net->C, L = net->L;
  const int NE = L * 2 * C;
  if (hmode(net)) {
    SS_PROPAGATE(stack_entry_h(net, step, lens, B, T, w, stream));
    SS_PROPAGATE(run_residual_stack_h(net, step, lens, B, T, w, stream));
    if (net->skipall_folded) return SS_OK;
  } else
  for (int l = 0; l < L; ++l) {
    const int d = 1 << (l % net->dil_cycle);
    // y = dilated_conv(x + dstep) + cond_proj ; g = sigmoid(y[:C]) * tanh(y[C:])   (net.py:66-73)
    ss_conv_gemm_args a = base_args(B, T, lens);
    a.A = w.X;
    a.lda = C;
    a.a_batch_stride = (int64_t)T * C;
    a.Cin = C;
    a.ntaps = 3;
    a.tap_off[0] = -d;
    a.tap_off[1] = 0;
    a.tap_off[2] = d;
    a.a_bias = net->dstep + ((int64_t)step * L + l) * C;
    a.W = net->w_dil[l];
    a.N = C;
    a.Np = 2 * C;
    a.Kp = round_up32(C);
    a.epi = SS_EPI_GATE;
    a.gate_mode = 0;
    a.E = w.E + (int64_t)l * 2 * C;
    a.lde = NE;
    a.e_batch_stride = (int64_t)T * NE;
    const bool defer = net->w_skipall != nullptr;  // see the note above run_residual_stack
    const int ldg = defer ? L * C : C;
    float* Gl = defer ? w.GA + (int64_t)l * C : w.G;
    a.C = Gl;
    a.ldc = ldg;
    a.c_batch_stride = (int64_t)T * ldg;
    a.mfma_bf16 = net->mfma_bf16;
    if (net->n_groups > 1) {
      a.group_size = B / net->n_groups;
      a.w_group_stride = net->gs_w_dil;
      a.a_bias_group_stride = net->gs_dstep;
    }
    if (net->w_dil_wino[l] && !net->mfma_bf16) {  // Winograd F(2,3): pairs of frames (t, t+d) from 4 products instead of 6
      a.W = net->w_dil_wino[l];
      a.w_group_stride = net->gs_w_dil_wino;
      if (net->wino_m == 4 && net->mfma_x3 && net->w_dil_x3[l]) {   // opt-in "bf16x3" mode: split operands on the bf16 matrix cores
        a.w_group_stride = net->gs_w_dil_x3;
        SS_PROPAGATE(ss_wino43_gate16x(&a, net->w_dil_x3[l], d, 0, stream));
      } else if (net->wino_m == 4) {
        const int g16 = g_ss_tuning.gate16;  // 0: 32x32x2 tiles; 1: per-launch pick; 2 / 3: 16x16x4 tiles of 16*MT quads
        if (g16 == 0) SS_PROPAGATE(ss_wino43_gate(&a, d, stream));
        else if (net->w_dil_wino16[l] && w.E16[l]) {   // addend in fetch order, laid out for exactly this tiling
          a.E = w.E16[l];
          a.e_tiled = 1;
          SS_PROPAGATE(ss_wino43_gate16w(&a, net->w_dil_wino16[l], d, w.mt16[l], stream));
        } else if (net->w_dil_wino16[l]) SS_PROPAGATE(ss_wino43_gate16w(&a, net->w_dil_wino16[l], d, g16 == 1 ? 0 : g16, stream));
        else SS_PROPAGATE(ss_wino43_gate16(&a, d, g16 == 1 ? 0 : g16, stream));
      } else {
        SS_PROPAGATE(ss_wino_gate(&a, d, stream));
      }
    } else {
      SS_PROPAGATE(ss_conv_gemm(&a, stream));
    }
    // y = output_projection(g) ; x = (x + y[:C]) / sqrt(2) ; skip += y[C:]   (net.py:75-77)
    // deferred-skip form: this launch only produces the residual stream, and the LAST layer's stream is never read (net.py:120-127)
    if (defer && l + 1 == L) break;
    ss_conv_gemm_args o = base_args(B, T, lens);
    o.A = Gl;
    o.lda = ldg;
    o.a_batch_stride = (int64_t)T * ldg;
    o.Cin = C;
    o.W = net->w_out[l];
    o.N = defer ? C : 2 * C;  // deferred skip: only the residual half (the first C packed rows) runs per layer
    if (defer) o.tile = g_ss_tuning.res_tile > 0 ? g_ss_tuning.res_tile : SS_TILE_64x64;
    o.Np = 2 * C;
    o.Kp = round_up32(C);
    o.epi = SS_EPI_RESSKIP;
    o.bias = net->b_out[l];
    o.Nh = C;
    o.R = w.X;
    o.ldr = C;
    o.r_batch_stride = (int64_t)T * C;
    o.C = w.X;
    o.ldc = C;
    o.c_batch_stride = (int64_t)T * C;
    o.post_scale = 0.70710678118654752440f;  // 1/sqrt(2.0)
    o.C2 = w.S;
    o.ldc2 = C;
    o.c2_batch_stride = (int64_t)T * C;
    o.accumulate = defer ? 0 : l > 0;
    o.mfma_bf16 = net->mfma_bf16;
    if (net->n_groups > 1) {
      o.group_size = B / net->n_groups;
      o.w_group_stride = net->gs_w_out;
      o.bias_group_stride = net->gs_b_out;
    }
    const int r16 = g_ss_tuning.res16;
    if (defer && r16 != 0 && !net->mfma_bf16 && (C == 192 || C == 256) && net->w_out16[l]) {   // weights in the kernel's fetch order
      o.w_group_stride = net->gs_w_out16;
      SS_PROPAGATE(ss_gemm16_resw(&o, net->w_out16[l], r16 == 1 ? 0 : r16, stream));
    } else if (defer && r16 != 0 && !net->mfma_bf16 && (C == 192 || C == 256)) {
      SS_PROPAGATE(ss_gemm16_res(&o, r16 == 1 ? 0 : r16, stream));   // 16x16x4 tiles, balanced single round (gemm16.hip)
    } else {
      SS_PROPAGATE(ss_conv_gemm(&o, stream));
    }
  }
  if (net->w_skipall && !hmode(net)) {  // S = sum_l skip_l = [g_0 | g_1 | ... | g_{L-1}] . [W_skip_0 ; ... ; W_skip_{L-1}]^T + sum_l b_skip_l
    ss_conv_gemm_args k = base_args(B, T, lens);
    k.A = w.GA;
    k.lda = L * C;
    k.a_batch_stride = (int64_t)T * L * C;
    k.Cin = L * C;
    k.W = net->w_skipall;
    k.N = C;
    k.Np = round_up32(C);
    k.Kp = L * C;
    k.epi = SS_EPI_STORE;
    k.bias = net->b_skipall;
    k.C = w.S;
    k.ldc = C;
    k.c_batch_stride = (int64_t)T * C;
    k.tile = g_ss_tuning.skip_tile > 0 ? g_ss_tuning.skip_tile : SS_TILE_64x64;
    k.mfma_bf16 = net->mfma_bf16;
    if (net->n_groups > 1) {
      k.group_size = B / net->n_groups;
      k.w_group_stride = net->gs_w_skipall;
      k.bias_group_stride = net->gs_b_skipall;
    }
    int s16 = g_ss_tuning.skip16;
    const bool use16 = s16 != 0 && !net->mfma_bf16 && k.Kp == k.Cin;   // 16x16x4 tiles, both operands by LDS-DMA (gemm16.hip)
    // long-K launch: when 64-row tiles also fit one round at three workgroups per CU they beat the 96-row pick (mel at C2: 240.7 vs 250.7 us)
    if (s16 == 1 && (long)((T + 63) / 64) * B * ((C + 63) / 64) <= 3L * ss_n_cu() && !(net->mfma_x3 && net->w_skipall_x3)) s16 = 4;
    if (net->skipall_folded) {  // w_skipall already carries skip_projection / sqrt(L): this GEMM + ReLU is the stack's output
      k.act = SS_ACT_RELU;
      k.C = w.G;
      if (use16 && net->mfma_x3 && net->w_skipall_x3) {   // opt-in bf16x3 mode: split operands on the bf16 matrix cores (gemm16x.hip)
        k.w_group_stride = net->gs_w_skipall_x3;
        return ss_gemm16x_store(&k, net->w_skipall_x3, s16 == 1 ? 0 : s16, stream);
      }
      if (use16 && w.ksplit > 1)   // one short utterance: split K over the idle CUs
        return partials_ok && skip_partials(net, w) ? ss_gemm16_store_partials(&k, 4, w.ksplit, w.KP, stream) : ss_gemm16_store_splitk(&k, 4, w.ksplit, w.KP, stream);
      return use16 ? ss_gemm16_store(&k, s16 == 1 ? 0 : s16, stream) : ss_conv_gemm(&k, stream);
    }
    if (use16 && w.ksplit > 1) SS_PROPAGATE(ss_gemm16_store_splitk(&k, 4, w.ksplit, w.KP, stream));
    else SS_PROPAGATE(use16 ? ss_gemm16_store(&k, s16 == 1 ? 0 : s16, stream) : ss_conv_gemm(&k, stream));
  }
  // x = relu(skip_projection(sum(skip) / sqrt(L)))   (net.py:124-127)
  ss_conv_gemm_args s = base_args(B, T, lens);
  s.A = w.S;
  s.lda = C;
  s.a_batch_stride = (int64_t)T * C;
  s.Cin = C;
  s.a_scale = 1.0f / sqrtf((float)L);
  s.W = net->w_skip;
  s.N = C;
  s.Np = round_up32(C);
  s.Kp = round_up32(C);
  s.epi = SS_EPI_STORE;
  s.bias = net->b_skip;
  s.act = SS_ACT_RELU;
  s.C = w.G;
  s.ldc = C;
  s.c_batch_stride = (int64_t)T * C;
  s.mfma_bf16 = smode(net) ? 0 : net->mfma_bf16;
  if (net->n_groups > 1) {
    s.group_size = B / net->n_groups;
    s.w_group_stride = net->gs_w_skip;
    s.bias_group_stride = net->gs_b_skip;
  }
  return ss_conv_gemm(&s, stream);
}

// ---------------------------------------------------------------------------------------------
// f0 net input: x[:, :C/2] = w*f0 + b ; x[:, C/2:] = uv_embed[uv]   (net.py:249-252)
// ---------------------------------------------------------------------------------------------
__global__ void f0_input_kernel(const float* __restrict__ f0, const int32_t* __restrict__ uv,
                                const float* __restrict__ w_in, const float* __restrict__ b_in,
                                const float* __restrict__ uv_embed, float* __restrict__ X, int B, int T, int C,
                                const int32_t* __restrict__ lens, int group_size, int64_t gs_w, int64_t gs_b, int64_t gs_e) {
  const int half = C / 2;
  const int64_t total = (int64_t)B * T * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int b = (int)(r / T), t = (int)(r % T);
    const int g = group_size > 0 ? b / group_size : 0;
    float v;
    if (c < half) v = w_in[g * gs_w + c] * f0[r] + b_in[g * gs_b + c];
    else v = uv_embed[g * gs_e + (uv[r] != 0 ? 1 : 0) * half + (c - half)];
    if (lens && t >= lens[b]) v = 0.f;
    X[i] = v;
  }
}

__device__ __forceinline__ float log_add_exp(float a, float b) {
  const float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}

// One reverse step of the joint sampler for one frame (gaussian_p_sample :326-333, p_sample :410-413,
// q_posterior :374-397, log_sample_categorical :447-452): (eps, l0, l1) = the network's three outputs for frame i = (b, t).
struct F0StepCoef {
  float recip, recipm1, c1, c2, sigma, log_alpha_t, log_1m_alpha_t, log_cp_tm1, log_1m_cp_tm1;
};
__device__ __forceinline__ void f0_update_row(float eps, float l0, float l1, int64_t i, int b, int t, int T, float& f0v, int32_t& uvv, float lo, float hi,
                                              const float* __restrict__ noise, const float* __restrict__ gumbel_u, const SsPhilox& rng, int step,
                                              const F0StepCoef& k) {
  const float LOG2 = 0.69314718055994530942f;
  const float LOG_TINY = -69.07755278982137f;  // log(1e-30) as torch computes log(clamp(onehot, 1e-30))
  float z = 0.f, u0, u1;
  if (noise) z = noise[i];
  if (gumbel_u) {
    u0 = gumbel_u[((int64_t)b * 2 + 0) * T + t];
    u1 = gumbel_u[((int64_t)b * 2 + 1) * T + t];
  }
  if (!noise || !gumbel_u) {
    uint32_t o[4];
    rng.gen((uint32_t)t, (uint32_t)b, (uint32_t)step, 0x46305556u, o);  // counter = (frame, item): invariant to T padding
    float z0, z1;
    ss_boxmuller(o[0], o[1], z0, z1);
    if (!noise) z = z0;
    if (!gumbel_u) {
      u0 = (float)(o[2] >> 8) * (1.0f / 16777216.0f);  // [0,1) like torch.rand
      u1 = (float)(o[3] >> 8) * (1.0f / 16777216.0f);
    }
  }
  // ---- Gaussian f0 ----
  const float x = f0v;
  float x0 = k.recip * x - k.recipm1 * eps;
  x0 = fminf(fmaxf(x0, lo), hi);
  const float mean = k.c1 * x0 + k.c2 * x;
  f0v = mean + k.sigma * z;
  // ---- multinomial uv ----
  const int cls = uvv != 0 ? 1 : 0;
  const float lx0 = cls == 0 ? 0.f : LOG_TINY, lx1 = cls == 1 ? 0.f : LOG_TINY;
  // log_softmax
  const float m = fmaxf(l0, l1);
  const float lse = logf(expf(l0 - m) + expf(l1 - m));
  const float p0 = (l0 - m) - lse, p1 = (l1 - m) - lse;
  float ev0, ev1;
  if (step == 0) {
    ev0 = p0;
    ev1 = p1;
  } else {
    ev0 = log_add_exp(p0 + k.log_cp_tm1, k.log_1m_cp_tm1 - LOG2);
    ev1 = log_add_exp(p1 + k.log_cp_tm1, k.log_1m_cp_tm1 - LOG2);
  }
  const float un0 = ev0 + log_add_exp(lx0 + k.log_alpha_t, k.log_1m_alpha_t - LOG2);
  const float un1 = ev1 + log_add_exp(lx1 + k.log_alpha_t, k.log_1m_alpha_t - LOG2);
  const float mm = fmaxf(un0, un1);
  const float nlse = mm + logf(expf(un0 - mm) + expf(un1 - mm));
  const float q0 = un0 - nlse, q1 = un1 - nlse;
  const float g0 = -logf(-logf(u0 + 1e-30f) + 1e-30f);
  const float g1 = -logf(-logf(u1 + 1e-30f) + 1e-30f);
  uvv = (g1 + q1) > (g0 + q0) ? 1 : 0;  // argmax, first max wins ties
}

// where a tail kernel reads the stack output g = relu(skip GEMM) from: the reduced tensor G, or the split-K slices P[s][row][C] (+ bias, ReLU,
// row mask - the arithmetic and order of splitk_reduce_kernel)
struct SkipSrc {
  const float* G;
  const float* P;
  const float* bias;
  int ksplit;
  int64_t per;   // floats per slice = B*T*C
};
__device__ __forceinline__ float4 skip_load4(const SkipSrc& s, const float* bias, int64_t row, int C, int k, bool masked) {
  if (s.ksplit <= 1) return *reinterpret_cast<const float4*>(s.G + row * C + k);
  float4 v = *reinterpret_cast<const float4*>(s.P + row * C + k);
  for (int q = 1; q < s.ksplit; ++q) {
    const float4 u = *reinterpret_cast<const float4*>(s.P + q * s.per + row * C + k);
    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
  }
  if (bias) {
    v.x += bias[k]; v.y += bias[k + 1]; v.z += bias[k + 2]; v.w += bias[k + 3];
  }
  v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
  if (masked) v = make_float4(0.f, 0.f, 0.f, 0.f);
  return v;
}

// The tail of an f0 network evaluation in ONE launch (round 4): output projection (C -> 3: a GEMV per frame, not matrix-core work - it ran on a
// 128x32 MFMA tile with 29 dead columns), the joint sampler update, and the NEXT evaluation's input row x[:, :C/2] = w*f0 + b ;
// x[:, C/2:] = uv_embed[uv] (net.py:249-252): 3 launches per step -> 1. 16 lanes per frame: lane j owns the float4s j, j + 16, ... of the row.
__global__ __launch_bounds__(256) void f0_tail_kernel(const SkipSrc src, int64_t gs_bskip, const float* __restrict__ w_final, const float* __restrict__ b_final,
                                                      float* __restrict__ f0, int32_t* __restrict__ uv, const float* __restrict__ lo,
                                                      const float* __restrict__ hi, const float* __restrict__ noise,
                                                      const float* __restrict__ gumbel_u, uint64_t seed, const uint64_t* __restrict__ seed_dev, int step,
                                                      int B, int T, int C, F0StepCoef k, const float* __restrict__ w_in, const float* __restrict__ b_in,
                                                      const float* __restrict__ uv_embed, float* __restrict__ X, const int32_t* __restrict__ lens,
                                                      int group_size, int64_t gs_wf, int64_t gs_bf, int64_t gs_w, int64_t gs_b, int64_t gs_e) {
  const int64_t n = (int64_t)B * T;
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (i >= n) return;   // whole 16-lane groups leave together
  const int j = threadIdx.x & 15;
  const int b = (int)(i / T), t = (int)(i % T);
  const int g = group_size > 0 ? b / group_size : 0;
  const float* W = w_final + g * gs_wf;   // packed rows [n][Kp = C]
  const float* bsk = src.bias ? src.bias + g * gs_bskip : nullptr;
  const bool masked = lens && t >= lens[b];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int q = j; q < C / 4; q += 16) {
    const float4 gv = skip_load4(src, bsk, i, C, 4 * q, masked);
    const float4 w0 = *reinterpret_cast<const float4*>(W + 4 * q), w1 = *reinterpret_cast<const float4*>(W + C + 4 * q),
                 w2 = *reinterpret_cast<const float4*>(W + 2 * C + 4 * q);
    a0 = fmaf(gv.w, w0.w, fmaf(gv.z, w0.z, fmaf(gv.y, w0.y, fmaf(gv.x, w0.x, a0))));
    a1 = fmaf(gv.w, w1.w, fmaf(gv.z, w1.z, fmaf(gv.y, w1.y, fmaf(gv.x, w1.x, a1))));
    a2 = fmaf(gv.w, w2.w, fmaf(gv.z, w2.z, fmaf(gv.y, w2.y, fmaf(gv.x, w2.x, a2))));
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) {   // butterfly over the 16 lanes of the frame: every lane ends with the same three sums
    a0 += __shfl_xor(a0, o, 16);
    a1 += __shfl_xor(a1, o, 16);
    a2 += __shfl_xor(a2, o, 16);
  }
  const float* bf = b_final + g * gs_bf;
  const SsPhilox rng(seed + (seed_dev ? seed_dev[0] : 0ull));
  float f0v = f0[i];
  int32_t uvv = uv[i];
  // padded frames: eps / logits = 0 as the row-masked projection GEMM wrote them before this kernel replaced it (not 0 + b_final)
  f0_update_row(masked ? 0.f : a0 + bf[0], masked ? 0.f : a1 + bf[1], masked ? 0.f : a2 + bf[2], i, b, t, T, f0v, uvv, lo[i], hi[i], noise, gumbel_u, rng,
                step, k);
  if (j == 0) {
    f0[i] = f0v;
    uv[i] = uvv;
  }
  if (X) {   // the next evaluation's input row (exactly f0_input_kernel's values)
    const int half = C / 2;
    const bool pad = lens && t >= lens[b];
    float* Xr = X + i * C;
    for (int q = j; q < C / 4; q += 16) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * q + e;
        v[e] = c < half ? w_in[g * gs_w + c] * f0v + b_in[g * gs_b + c] : uv_embed[g * gs_e + (uvv != 0 ? 1 : 0) * half + (c - half)];
        if (pad) v[e] = 0.f;
      }
      *reinterpret_cast<float4*>(Xr + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// q_sample of the normalised coarse mel
__global__ void mel_qsample_kernel(const float* __restrict__ mel, const float* __restrict__ smin,
                                   const float* __restrict__ smax, float sa, float s1, const float* __restrict__ noise,
                                   uint64_t seed, const uint64_t* __restrict__ seed_dev, float* __restrict__ x, int64_t rows,
                                   int T, int M) {
  const SsPhilox rng(seed + (seed_dev ? seed_dev[0] : 0ull));
  const int64_t n = rows * M;
  const int64_t per_item = (int64_t)T * M;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % M);
    const float xs = (mel[i] - smin[c]) / (smax[c] - smin[c]) * 2.0f - 1.0f;
    float z;
    if (noise) z = noise[i];
    else {
      uint32_t o[4];
      rng.gen((uint32_t)(i % per_item), (uint32_t)(i / per_item), 0xffffffffu, 0x4d454c44u, o);  // (element of item, item)
      float z1;
      ss_boxmuller(o[0], o[1], z, z1);
    }
    x[i] = sa * xs + s1 * z;
  }
}

__global__ void mel_denorm_kernel(const float* __restrict__ x, const float* __restrict__ smin,
                                  const float* __restrict__ smax, float* __restrict__ mel, int B, int T, int M,
                                  const int32_t* __restrict__ lens, int32_t* __restrict__ nonfinite) {
  const int64_t n = (int64_t)B * T * M;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % M);
    const int64_t r = i / M;
    const int b = (int)(r / T), t = (int)(r % T);
    float v = (x[i] + 1.0f) / 2.0f * (smax[c] - smin[c]) + smin[c];
    if (lens && t >= lens[b]) v = 0.f;
    else if (nonfinite && !isfinite(v)) atomicOr(nonfinite, 1);   // a valid frame left the number range (fp16 modes: the stream overflowed)
    mel[i] = v;
  }
}

// dyn_clip bounds from MIDI (stylesinger.py:274-283)
__global__ void f0_bounds_kernel(const int64_t* __restrict__ midi, float* __restrict__ lo, float* __restrict__ hi, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float m = (float)midi[i];
  auto nb = [](float note) {
    float hz = powf(2.0f, (note - 69.0f) / 12.0f) * 440.0f;
    float x = log2f(hz);
    if (x > 10.0f) x = 10.0f;
    float v = (x - 6.0f) / (10.0f - 6.0f) * 2.0f - 1.0f;
    return fminf(fmaxf(v, -1.0f), 1.0f);
  };
  lo[i] = nb(m - 3.0f);
  hi[i] = nb(m + 3.0f);
}

// merge of the two predictors + denorm + coarse (stylesinger.py:230-246,286-311; pitch_utils.py:22-31,65-78)
__global__ void pitch_post_kernel(const float* __restrict__ f0_a, const int32_t* __restrict__ uv_a,
                                  const float* __restrict__ f0_b, const int32_t* __restrict__ uv_b,
                                  const int64_t* __restrict__ midi, const int64_t* __restrict__ mel2ph,
                                  float* __restrict__ pitch_pred, float* __restrict__ f0_denorm,
                                  int64_t* __restrict__ coarse, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool rest = midi[i] == 0;
  const float ua = rest ? 1.0f : (float)(uv_a[i] != 0), ub = rest ? 1.0f : (float)(uv_b[i] != 0);
  const float fa = (f0_a[i] + 1.0f) / 2.0f * (10.0f - 6.0f) + 6.0f;
  const float fb = (f0_b[i] + 1.0f) / 2.0f * (10.0f - 6.0f) + 6.0f;
  const float f = fb / 2.0f + fa / 2.0f;  // pitch_domain_specific/2 + pitch_domain_agnostic/2
  const float u = ub / 2.0f + ua / 2.0f;
  pitch_pred[(int64_t)i * 2 + 0] = f;
  pitch_pred[(int64_t)i * 2 + 1] = u;
  float hz = exp2f(f);
  if (u > 0.0f) hz = 0.0f;
  if (mel2ph[i] == 0) hz = 0.0f;
  f0_denorm[i] = hz;
  // f0_to_coarse
  // utils/pitch_utils.py:17-18: numpy float64 constants, cast to fp32 when they meet the tensor
  const float f0_mel_min = (float)77.75496616579426;          // 1127*ln(1+50/700)
  const float f0_mel_span = (float)986.6532669978451;         // 1127*ln(1+1100/700) - f0_mel_min
  float mel = 1127.0f * logf(1.0f + hz / 700.0f);
  if (mel > 0.0f) mel = (mel - f0_mel_min) * 254.0f / f0_mel_span + 1.0f;
  if (mel <= 1.0f) mel = 1.0f;
  if (mel > 255.0f) mel = 255.0f;
  coarse[i] = (int64_t)(mel + 0.5f);
}

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int64_t ss_wavenet_workspace_bytes(const ss_wavenet* net, int B, int T) {
  if (!net || B <= 0 || T <= 0) return -1;
  return ws_layout(net, B, T, nullptr).bytes;
}

// x_in -> relu(input_projection) -> residual stack; leaves relu(skip_projection(.)) in w.G  (net.py:114-127)
static int mel_input_proj(const ss_wavenet* net, const float* x_in, const int32_t* lens, int B, int T, const WsLayout& w, hipStream_t stream);
static int mel_net_body(const ss_wavenet* net, const float* x_in, const int32_t* lens, int B, int T, const WsLayout& w, int t,
                        hipStream_t stream) {
  SS_PROPAGATE(mel_input_proj(net, x_in, lens, B, T, w, stream));
  return run_residual_stack(net, t, lens, B, T, w, stream);
}

// x_in -> w.X = relu(input_projection(x_in))  (net.py:114-116)
static int mel_input_proj(const ss_wavenet* net, const float* x_in, const int32_t* lens, int B, int T, const WsLayout& w, hipStream_t stream) {
  const int C = net->C, M = net->in_dim;
  ss_conv_gemm_args a = base_args(B, T, lens);
  a.A = x_in;
  a.lda = M;
  a.a_batch_stride = (int64_t)T * M;
  a.Cin = M;
  a.W = net->w_in;
  a.N = C;
  a.Np = round_up32(C);
  a.Kp = round_up32(M);
  a.epi = SS_EPI_STORE;
  a.bias = net->b_in;
  a.act = SS_ACT_RELU;
  a.C = w.X;
  a.ldc = C;
  a.c_batch_stride = (int64_t)T * C;
  return ss_conv_gemm(&a, stream);
}

// The tail of a mel network evaluation for SMALL launches (one short utterance: the B = 1 latency shape) in ONE launch: output projection
// (C -> M), the DDPM posterior step on x (shallow_diffusion_tts.py:130-162; same tape / Philox counters as the SS_EPI_DDPM epilogue) and the
// NEXT evaluation's input projection relu(W_in x + b) - two 16-20 us MFMA launches of 24-48 workgroups become one ~8 us VALU launch. Exact
// fp32 FMAs; only the summation order over K differs from the matrix-core form. MTR = 2 frames per workgroup.
constexpr int MTR = 2;
__global__ __launch_bounds__(256) void mel_tail_kernel(const SkipSrc src, const float* __restrict__ Wf, const float* __restrict__ bf,
                                                       float* __restrict__ x, const float* __restrict__ noise, uint64_t seed,
                                                       const uint64_t* __restrict__ seed_dev, uint32_t step, float recip, float recipm1, float c1,
                                                       float c2, float sigma, const float* __restrict__ Win, const float* __restrict__ bin,
                                                       float* __restrict__ X, const int32_t* __restrict__ lens, int B, int T, int C, int M, int Kp_in) {
  extern __shared__ __attribute__((aligned(16))) float smem_mt[];
  float* gs = smem_mt;            // [MTR][C]
  float* xs = smem_mt + MTR * C;    // [MTR][Kp_in] (zero padded beyond M)
  const int64_t n_rows = (int64_t)B * T, r0 = (int64_t)blockIdx.x * MTR;
  const int tid = threadIdx.x, C4 = C / 4;
  for (int idx = tid; idx < MTR * C4; idx += 256) {
    const int row = idx / C4, q = idx - row * C4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + row < n_rows) {
      const int64_t i = r0 + row;
      v = skip_load4(src, src.bias, i, C, 4 * q, lens && (int)(i % T) >= lens[i / T]);
    }
    *reinterpret_cast<float4*>(gs + row * C + 4 * q) = v;
  }
  for (int idx = tid; idx < MTR * Kp_in; idx += 256) xs[idx] = 0.f;
  __syncthreads();
  const SsPhilox rng(seed + (seed_dev ? seed_dev[0] : 0ull));
  for (int idx = tid; idx < MTR * M; idx += 256) {
    const int row = idx / M, n = idx - row * M;
    const int64_t i = r0 + row;
    if (i >= n_rows) continue;
    const float* wr = Wf + (int64_t)n * C;
    const float* gr = gs + row * C;
    float acc = 0.f;
#pragma unroll 8
    for (int q = 0; q < C4; ++q) {
      const float4 wv = *reinterpret_cast<const float4*>(wr + 4 * q), gv = *reinterpret_cast<const float4*>(gr + 4 * q);
      acc = fmaf(gv.w, wv.w, fmaf(gv.z, wv.z, fmaf(gv.y, wv.y, fmaf(gv.x, wv.x, acc))));
    }
    const float eps = acc + bf[n];
    const int b = (int)(i / T), t = (int)(i % T);
    const float xv = x[i * M + n];
    float x0 = recip * xv - recipm1 * eps;
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
    const float mean = c1 * x0 + c2 * xv;
    float z = 0.f;
    if (sigma != 0.f) {
      if (noise) z = noise[i * M + n];
      else {
        float z4[4];   // exactly the SS_EPI_DDPM epilogue's draw: output t & 3 of the block of frames 4 (t >> 2) .. + 3 of this bin
        ss_mel_draw4(rng, (uint32_t)(t >> 2), (uint32_t)M, (uint32_t)n, (uint32_t)b, step, z4);
        z = z4[t & 3];
      }
    }
    float xn = mean + sigma * z;
    if (lens && t >= lens[b]) xn = 0.f;
    x[i * M + n] = xn;
    xs[row * Kp_in + n] = xn;
  }
  if (!X) return;   // block-uniform
  __syncthreads();
  const int K4 = Kp_in / 4;   // W_in is packed [C][Kp_in], zero filled beyond M
  for (int c = tid; c < C; c += 256) {
    float acc[MTR];
#pragma unroll
    for (int r = 0; r < MTR; ++r) acc[r] = 0.f;
    const float* wr = Win + (int64_t)c * Kp_in;
#pragma unroll 8
    for (int q = 0; q < K4; ++q) {
      const float4 wv = *reinterpret_cast<const float4*>(wr + 4 * q);
#pragma unroll
      for (int r = 0; r < MTR; ++r) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + r * Kp_in + 4 * q);
        acc[r] = fmaf(xv.w, wv.w, fmaf(xv.z, wv.z, fmaf(xv.y, wv.y, fmaf(xv.x, wv.x, acc[r]))));
      }
    }
    const float bc = bin[c];
#pragma unroll
    for (int r = 0; r < MTR; ++r) {
      const int64_t i = r0 + r;
      if (i >= n_rows) continue;
      const int b = (int)(i / T), t = (int)(i % T);
      float v = fmaxf(acc[r] + bc, 0.f);
      if (lens && t >= lens[b]) v = 0.f;
      X[i * C + c] = v;
    }
  }
}

// eps_out [B][T][M] = DiffNet(x_in, t, cond)  (plain output projection, for samplers that keep a history of eps)
static int mel_eps(const ss_wavenet* net, const float* x_in, float* eps_out, const int32_t* lens, int B, int T, const WsLayout& w,
                   int t, hipStream_t stream) {
  const int C = net->C, M = net->in_dim;
  SS_PROPAGATE(mel_net_body(net, x_in, lens, B, T, w, t, stream));
  ss_conv_gemm_args f = base_args(B, T, lens);
  f.A = w.G;
  f.lda = C;
  f.a_batch_stride = (int64_t)T * C;
  f.Cin = C;
  f.W = net->w_final;
  f.N = M;
  f.Np = round_up32(M);
  f.Kp = round_up32(C);
  f.epi = SS_EPI_STORE;
  f.bias = net->b_final;
  f.C = eps_out;
  f.ldc = M;
  f.c_batch_stride = (int64_t)T * M;
  f.mask_rows = 0;
  return ss_conv_gemm(&f, stream);
}

// One reverse step of the mel net at network time `t` with explicit update coefficients
//   x0 = clamp(recip*x - recipm1*eps, -1, 1) ; x <- c1*x0 + c2*x + sigma*z
static int mel_step(const ss_wavenet* net, float* x, const int32_t* lens, int B, int T, const WsLayout& w, int t, float recip,
                    float recipm1, float c1, float c2, float sigma, const float* noise_t, uint64_t seed,
                    const uint64_t* seed_dev, uint32_t step_id, hipStream_t stream, int x0_pred = 0) {
  const int C = net->C, M = net->in_dim;
  SS_PROPAGATE(mel_net_body(net, x, lens, B, T, w, t, stream));
  // eps = output_projection(.) fused with the posterior step (shallow_diffusion_tts.py:130-162)
  ss_conv_gemm_args f = base_args(B, T, lens);
  f.A = w.G;
  f.lda = C;
  f.a_batch_stride = (int64_t)T * C;
  f.Cin = C;
  f.W = net->w_final;
  f.N = M;
  f.Np = round_up32(M);
  f.Kp = round_up32(C);
  f.epi = SS_EPI_DDPM;
  f.bias = net->b_final;
  f.C = x;
  f.ldc = M;
  f.c_batch_stride = (int64_t)T * M;
  f.ddpm_recip = recip;
  f.ddpm_recipm1 = recipm1;
  f.ddpm_c1 = c1;
  f.ddpm_c2 = c2;
  f.ddpm_sigma = sigma;
  f.noise = noise_t;
  f.seed = seed;
  f.seed_dev = seed_dev;
  f.step = step_id;
  f.ddpm_x0_pred = x0_pred;
  return ss_conv_gemm(&f, stream);
}

extern "C" int ss_meldiff_sample(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T,
                                 const float* noise, uint64_t seed, const uint64_t* seed_dev, int step_lo, int step_hi,
                                 int do_precompute, void* ws, int64_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(net && x && cond && ws, "ss_meldiff_sample: null pointer");
  SS_CHECK_ARG(net->L > 0 && net->L <= SS_MAX_LAYERS && (net->C % 32) == 0, "ss_meldiff_sample: bad net C=%d L=%d", net->C, net->L);
  SS_CHECK_ARG(0 <= step_lo && step_lo <= step_hi && step_hi <= net->steps, "ss_meldiff_sample: bad step range");
  SS_CHECK_ARG(net->n_groups <= 1, "ss_meldiff_sample: grouped nets are only supported by the f0 sampler");
  const WsLayout w = ws_layout(net, B, T, ws);
  SS_CHECK_ARG(ws_bytes >= w.bytes, "ss_meldiff_sample: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)w.bytes);
  const int M = net->in_dim, C = net->C;
  if (do_precompute) SS_PROPAGATE(precompute_cond(net, cond, lens, B, T, w, stream));
  // launches that leave most CUs idle (one short utterance): the two small projections around the sampler update run as ONE VALU launch
  const int Kp_in = round_up32(M);
  const bool tail = g_ss_tuning.mel_tail != 0 && !net->mfma_bf16 && (int64_t)B * T <= 8L * ss_n_cu() && (C % 4) == 0 && (M % 4) == 0 &&
                    (size_t)(MTR * C + MTR * Kp_in) * 4 <= 48 * 1024;
  for (int t = step_hi - 1; t >= step_lo; --t) {
    g_wset_eval = net->steps - 1 - t;   // evaluation index of the whole loop, whatever [step_lo, step_hi) this call covers
    const float sigma = t > 0 ? expf(0.5f * net->post_logvar[t]) : 0.0f;
    const float* noise_t = noise ? noise + (int64_t)t * B * T * M : nullptr;
    if (!tail) {
      SS_PROPAGATE(mel_step(net, x, lens, B, T, w, t, net->sqrt_recip_ac[t], net->sqrt_recipm1_ac[t], net->post_c1[t], net->post_c2[t], sigma, noise_t,
                            seed, seed_dev, (uint32_t)t, stream));
      continue;
    }
    if (t == step_hi - 1) SS_PROPAGATE(mel_input_proj(net, x, lens, B, T, w, stream));   // later evaluations get their input from the tail kernel
    SS_PROPAGATE(run_residual_stack(net, t, lens, B, T, w, stream, true));
    const bool parts = skip_partials(net, w);
    const SkipSrc src = {w.G, w.KP, parts ? net->b_skipall : nullptr, parts ? w.ksplit : 1, (int64_t)B * T * C};
    hipLaunchKernelGGL(mel_tail_kernel, dim3((unsigned)(((int64_t)B * T + MTR - 1) / MTR)), dim3(256), (size_t)(MTR * C + MTR * Kp_in) * 4, stream, src, net->w_final,
                       net->b_final, x, noise_t, seed, seed_dev, (uint32_t)t, net->sqrt_recip_ac[t], net->sqrt_recipm1_ac[t], net->post_c1[t],
                       net->post_c2[t], sigma, net->w_in, net->b_in, t > step_lo ? w.X : nullptr, lens, B, T, C, M, Kp_in);
    SS_CHECK_LAUNCH("mel_tail_kernel");
  }
  return SS_OK;
}

// ProDiff teacher sampler: same launches as ss_meldiff_sample, the final GEMM's epilogue takes the network output as x0.
extern "C" int ss_prodiff_sample(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T,
                                 const float* noise, uint64_t seed, const uint64_t* seed_dev, int n_steps, const float* c1,
                                 const float* c2, const float* sigma, int do_precompute, void* ws, int64_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(net && x && cond && ws && c1 && c2 && sigma, "ss_prodiff_sample: null pointer");
  SS_CHECK_ARG(net->n_groups <= 1 && net->L > 0 && net->L <= SS_MAX_LAYERS && (net->C % 32) == 0, "ss_prodiff_sample: bad net");
  SS_CHECK_ARG(n_steps >= 1 && n_steps <= net->steps, "ss_prodiff_sample: n_steps=%d outside [1, %d]", n_steps, net->steps);
  const WsLayout w = ws_layout(net, B, T, ws);
  SS_CHECK_ARG(ws_bytes >= w.bytes, "ss_prodiff_sample: workspace too small");
  const int M = net->in_dim;
  if (do_precompute) SS_PROPAGATE(precompute_cond(net, cond, lens, B, T, w, stream));
  for (int t = n_steps - 1; t >= 0; --t) {
    g_wset_eval = n_steps - 1 - t;
    SS_PROPAGATE(mel_step(net, x, lens, B, T, w, t, 0.0f, 0.0f, c1[t], c2[t], t > 0 ? sigma[t] : 0.0f,
                          noise ? noise + (int64_t)t * B * T * M : nullptr, seed, seed_dev, (uint32_t)t, stream, 1));
  }
  return SS_OK;
}

// Strided sampler (DDIM family, Song et al. 2021 eq. 12 / 16) over the SAME denoiser: visits network times ts[0] > ts[1] > ... and jumps
// x_{ts[i]} -> x_{ts[i+1]} (-> x_0 after the last).  With x0 = clamp(...), eps' = (x - sqrt(ac_t) x0)/sqrt(1-ac_t) and
//   sigma = eta * sqrt((1-ac_prev)/(1-ac_t)) * sqrt(1 - ac_t/ac_prev):
//   x_prev = sqrt(ac_prev) x0 + sqrt(1-ac_prev-sigma^2) eps' + sigma z = c1 x0 + c2 x + sigma z,
//   c2 = sqrt((1-ac_prev-sigma^2)/(1-ac_t)), c1 = sqrt(ac_prev) - c2 sqrt(ac_t)
// i.e. the DDPM epilogue with other coefficients.  eta = 0 is the deterministic sampler of BASELINE config 5 (the reference has none);
// eta = 1 with ts = K-1 ... 0 IS the reference's ancestral p_sample (shallow_diffusion_tts.py:136-162: c1, c2 become
// posterior_mean_coef1/2 and sigma^2 the posterior variance), which is how this entry point is pinned to the reference
// (golden acoustic_t64_s100).  The schedule comes in DOUBLE precision (1 - ac_prev loses its digits in a float table at small t).
extern "C" int ss_meldiff_sample_ddim(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T,
                                      const int32_t* ts, int n_ts, const double* alphas_cumprod, float eta, const float* noise,
                                      uint64_t seed, const uint64_t* seed_dev, int do_precompute, void* ws, int64_t ws_bytes,
                                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(net && x && cond && ws && ts && alphas_cumprod && n_ts > 0, "ss_meldiff_sample_ddim: null pointer");
  SS_CHECK_ARG(net->n_groups <= 1 && net->L > 0 && net->L <= SS_MAX_LAYERS, "ss_meldiff_sample_ddim: bad net");
  SS_CHECK_ARG(eta >= 0.0f && eta <= 1.0f, "ss_meldiff_sample_ddim: eta=%g outside [0, 1]", (double)eta);
  const WsLayout w = ws_layout(net, B, T, ws);
  SS_CHECK_ARG(ws_bytes >= w.bytes, "ss_meldiff_sample_ddim: workspace too small");
  for (int i = 0; i < n_ts; ++i)
    SS_CHECK_ARG(ts[i] >= 0 && ts[i] < net->steps && (i == 0 || ts[i] < ts[i - 1]), "ss_meldiff_sample_ddim: ts must be strictly decreasing in [0,steps)");
  const int M = net->in_dim;
  if (do_precompute) SS_PROPAGATE(precompute_cond(net, cond, lens, B, T, w, stream));
  for (int i = 0; i < n_ts; ++i) {
    g_wset_eval = i;
    const int t = ts[i];
    const double ac_t = alphas_cumprod[t];
    const double ac_p = (i + 1 < n_ts) ? alphas_cumprod[ts[i + 1]] : 1.0;
    const double sig = (double)eta * sqrt((1.0 - ac_p) / (1.0 - ac_t)) * sqrt(fmax(0.0, 1.0 - ac_t / ac_p));
    const double c2 = sqrt(fmax(0.0, 1.0 - ac_p - sig * sig) / (1.0 - ac_t));
    const double c1 = sqrt(ac_p) - c2 * sqrt(ac_t);
    const bool draw = eta > 0.0f;  // like p_sample, a stochastic run draws at every step (sigma = 0 on the last)
    SS_PROPAGATE(mel_step(net, x, lens, B, T, w, t, net->sqrt_recip_ac[t], net->sqrt_recipm1_ac[t], (float)c1, (float)c2, (float)sig,
                          (draw && noise) ? noise + (int64_t)t * B * T * M : nullptr, seed, draw ? seed_dev : nullptr, (uint32_t)t, stream));
  }
  return SS_OK;
}

// ---------------------------------------------------------------------------------------------
// PLMS / "pndm_speedup" sampler of the reference (shallow_diffusion_tts.py:165-197 p_sample_plms, loop :254-260):
//   x_prev = x + (a_prev - a_t) * (kx * x - ke * eps'),  eps' = linear multistep combination of the eps history.
// The reference runs it for one utterance at a time (its `max(t - interval, 0)` on a tensor only works for B = 1).
// ---------------------------------------------------------------------------------------------
namespace {
__global__ void plms_update_kernel(const float* __restrict__ x, float* __restrict__ x_out, const float* __restrict__ e0,
                                   const float* __restrict__ e1, const float* __restrict__ e2, const float* __restrict__ e3,
                                   int order, float d, float kx, float ke, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float n0 = e0[i];
    float prime;
    switch (order) {  // same operation order as the reference expressions (:186-193)
      case 0: prime = n0; break;
      case 1: prime = (n0 + e1[i]) / 2.0f; break;
      case 2: prime = (3.0f * n0 - e1[i]) / 2.0f; break;
      case 3: prime = (23.0f * n0 - 16.0f * e1[i] + 5.0f * e2[i]) / 12.0f; break;
      default: prime = (55.0f * n0 - 59.0f * e1[i] + 37.0f * e2[i] - 9.0f * e3[i]) / 24.0f; break;
    }
    const float xv = x[i];
    x_out[i] = xv + d * (kx * xv - ke * prime);
  }
}
}  // namespace

extern "C" int ss_meldiff_sample_plms(const ss_wavenet* net, float* x, const float* cond, const int32_t* lens, int B, int T,
                                      int step_hi, int interval, const float* alphas_cumprod, int do_precompute, float* hist,
                                      void* ws, int64_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(net && x && cond && ws && hist && alphas_cumprod, "ss_meldiff_sample_plms: null pointer");
  SS_CHECK_ARG(net->n_groups <= 1 && net->L > 0 && net->L <= SS_MAX_LAYERS, "ss_meldiff_sample_plms: bad net");
  SS_CHECK_ARG(step_hi >= 1 && step_hi <= net->steps, "ss_meldiff_sample_plms: step_hi=%d must be in [1, steps]", step_hi);
  SS_CHECK_ARG(interval >= 1 && interval < step_hi, "ss_meldiff_sample_plms: interval=%d must be in [1, step_hi)", interval);
  const WsLayout w = ws_layout(net, B, T, ws);
  SS_CHECK_ARG(ws_bytes >= w.bytes, "ss_meldiff_sample_plms: workspace too small");
  const int64_t n = (int64_t)B * T * net->in_dim;
  float* slot[5];
  for (int i = 0; i < 5; ++i) slot[i] = hist + i * n;  // ring of eps(x_t, t): the current one + the last 4
  float* x_pred = hist + 5 * n;
  if (do_precompute) SS_PROPAGATE(precompute_cond(net, cond, lens, B, T, w, stream));
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  int n_hist = 0, cur = 0;
  g_wset_eval = 0;   // PLMS: evaluations counted as they come (the first step evaluates twice)
  const int t_first = (step_hi - 1) / interval * interval;  // reversed(range(0, K_step, interval)), shallow_diffusion_tts.py:254-260
  for (int t = t_first; t >= 0; t -= interval) {
    const int tp = t - interval > 0 ? t - interval : 0;
    const float a_t = alphas_cumprod[t], a_p = alphas_cumprod[tp];
    const float a_t_sq = sqrtf(a_t), a_p_sq = sqrtf(a_p);
    const float d = a_p - a_t;
    const float kx = 1.0f / (a_t_sq * (a_t_sq + a_p_sq));
    const float ke = 1.0f / (a_t_sq * (sqrtf((1.0f - a_p) * a_t) + sqrtf((1.0f - a_t) * a_p)));
    float* e0 = slot[cur];
    SS_PROPAGATE(mel_eps(net, x, e0, lens, B, T, w, t, stream));
    ++g_wset_eval;
    auto h = [&](int back) { return slot[(cur + 5 - back) % 5]; };
    if (n_hist == 0) {
      hipLaunchKernelGGL(plms_update_kernel, dim3(blocks), dim3(256), 0, stream, x, x_pred, e0, e0, e0, e0, 0, d, kx, ke, n);
      float* e_prev = slot[(cur + 1) % 5];  // scratch: overwritten by the next step's eps
      SS_PROPAGATE(mel_eps(net, x_pred, e_prev, lens, B, T, w, tp, stream));
      ++g_wset_eval;
      hipLaunchKernelGGL(plms_update_kernel, dim3(blocks), dim3(256), 0, stream, x, x, e0, e_prev, e0, e0, 1, d, kx, ke, n);
    } else {
      const int order = n_hist == 1 ? 2 : n_hist == 2 ? 3 : 4;
      hipLaunchKernelGGL(plms_update_kernel, dim3(blocks), dim3(256), 0, stream, x, x, e0, h(1), h(2), h(3), order, d, kx, ke, n);
    }
    SS_CHECK_LAUNCH("ss_meldiff_sample_plms");
    if (n_hist < 4) ++n_hist;
    cur = (cur + 1) % 5;
  }
  return SS_OK;
}

extern "C" int ss_f0diff_sample(const ss_wavenet* net, float* f0, int32_t* uv, const float* cond, const float* lo,
                                const float* hi, const int32_t* lens, int B, int T, const float* noise,
                                const float* gumbel_u, uint64_t seed, const uint64_t* seed_dev, int step_lo, int step_hi,
                                int do_precompute, void* ws, int64_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SS_CHECK_ARG(net && f0 && uv && cond && lo && hi && ws, "ss_f0diff_sample: null pointer");
  SS_CHECK_ARG(net->L > 0 && net->L <= SS_MAX_LAYERS && (net->C % 64) == 0, "ss_f0diff_sample: bad net C=%d L=%d", net->C, net->L);
  SS_CHECK_ARG(0 <= step_lo && step_lo <= step_hi && step_hi <= net->steps, "ss_f0diff_sample: bad step range");
  SS_CHECK_ARG(net->out_dim == 3 && net->in_dim == 1, "ss_f0diff_sample: net must be the 1->3 DDiffNet");
  SS_CHECK_ARG(net->n_groups <= 1 || (net->n_groups == 2 && B % 2 == 0), "ss_f0diff_sample: paired nets need an even item count");
  const WsLayout w = ws_layout(net, B, T, ws);
  SS_CHECK_ARG(ws_bytes >= w.bytes, "ss_f0diff_sample: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)w.bytes);
  const int C = net->C;
  const int64_t n = (int64_t)B * T;
  if (do_precompute) SS_PROPAGATE(precompute_cond(net, cond, lens, B, T, w, stream));
  const int gsz = net->n_groups > 1 ? B / net->n_groups : 0;
  for (int t = step_hi - 1; t >= step_lo; --t) {
    if (t == step_hi - 1) {   // the first evaluation's input; every later one is written by the previous step's tail kernel
      hipLaunchKernelGGL(f0_input_kernel, dim3(grid_for(n * C)), dim3(256), 0, stream, f0, uv, net->w_in, net->b_in, net->uv_embed, w.X, B, T, C,
                         lens, gsz, net->gs_w_in, net->gs_b_in, net->gs_uv_embed);
      SS_CHECK_LAUNCH("f0_input_kernel");
    }
    SS_PROPAGATE(run_residual_stack(net, t, lens, B, T, w, stream, true));
    const bool parts = skip_partials(net, w);   // the skip GEMM left its split-K slices in w.KP: the tail kernel adds them
    const SkipSrc src = {w.G, w.KP, parts ? net->b_skipall : nullptr, parts ? w.ksplit : 1, n * C};
    const int tm1 = t > 0 ? t - 1 : 0;
    const F0StepCoef k = {net->sqrt_recip_ac[t], net->sqrt_recipm1_ac[t], net->post_c1[t], net->post_c2[t],
                          t > 0 ? expf(0.5f * net->post_logvar[t]) : 0.0f, net->log_alpha[t], net->log_1m_alpha[t],
                          net->log_cumprod_alpha[tm1], net->log_1m_cumprod_alpha[tm1]};
    // output projection (C -> 3) + joint sampler update + the next evaluation's input row, one launch (f0_tail_kernel)
    hipLaunchKernelGGL(f0_tail_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream, src, net->gs_b_skipall, net->w_final, net->b_final, f0, uv, lo, hi,
                       noise ? noise + (int64_t)t * n : nullptr, gumbel_u ? gumbel_u + (int64_t)t * n * 2 : nullptr, seed, seed_dev, t, B, T, C, k,
                       net->w_in, net->b_in, net->uv_embed, t > step_lo ? w.X : nullptr, lens, gsz, net->gs_w_final, net->gs_b_final, net->gs_w_in,
                       net->gs_b_in, net->gs_uv_embed);
    SS_CHECK_LAUNCH("f0_tail_kernel");
  }
  return SS_OK;
}

extern "C" int ss_mel_qsample(const float* coarse_mel, const float* spec_min, const float* spec_max, float sqrt_ac,
                              float sqrt_1mac, const float* noise, uint64_t seed, const uint64_t* seed_dev, float* x, int B,
                              int T, int M, void* stream) {
  SS_CHECK_ARG(coarse_mel && spec_min && spec_max && x, "ss_mel_qsample: null pointer");
  hipLaunchKernelGGL(mel_qsample_kernel, dim3(grid_for((int64_t)B * T * M)), dim3(256), 0, (hipStream_t)stream, coarse_mel,
                     spec_min, spec_max, sqrt_ac, sqrt_1mac, noise, seed, seed_dev, x, (int64_t)B * T, T, M);
  SS_CHECK_LAUNCH("ss_mel_qsample");
  return SS_OK;
}

extern "C" int ss_mel_denorm(const float* x, const float* spec_min, const float* spec_max, float* mel, int B, int T, int M,
                             const int32_t* lens, int32_t* nonfinite, void* stream) {
  SS_CHECK_ARG(x && spec_min && spec_max && mel, "ss_mel_denorm: null pointer");
  hipLaunchKernelGGL(mel_denorm_kernel, dim3(grid_for((int64_t)B * T * M)), dim3(256), 0, (hipStream_t)stream, x, spec_min,
                     spec_max, mel, B, T, M, lens, nonfinite);
  SS_CHECK_LAUNCH("ss_mel_denorm");
  return SS_OK;
}

extern "C" int ss_f0_bounds(const int64_t* midi, float* lo, float* hi, int n, void* stream) {
  SS_CHECK_ARG(midi && lo && hi && n > 0, "ss_f0_bounds: bad args");
  hipLaunchKernelGGL(f0_bounds_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, midi, lo, hi, n);
  SS_CHECK_LAUNCH("ss_f0_bounds");
  return SS_OK;
}

extern "C" int ss_pitch_post(const float* f0_a, const int32_t* uv_a, const float* f0_b, const int32_t* uv_b,
                             const int64_t* midi, const int64_t* mel2ph, float* pitch_pred, float* f0_denorm,
                             int64_t* pitch_coarse, int n, void* stream) {
  SS_CHECK_ARG(f0_a && uv_a && f0_b && uv_b && midi && mel2ph && pitch_pred && f0_denorm && pitch_coarse && n > 0,
               "ss_pitch_post: bad args");
  hipLaunchKernelGGL(pitch_post_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, f0_a, uv_a, f0_b, uv_b,
                     midi, mel2ph, pitch_pred, f0_denorm, pitch_coarse, n);
  SS_CHECK_LAUNCH("ss_pitch_post");
  return SS_OK;
}
