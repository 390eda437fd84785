// gfv_rowtile_chain: the entry point of the fused GEMM chain (contract: include/gfv.h) - argument checks, the library's launch
// context (product form, hidden size), the profiler's algorithmic-work bookkeeping and the choice of the kernel family:
// the lean single-layer kernel (lin1.hip), the column-owner persistent backward (colchain.hip), the register-resident
// row-owner chain and its ragged-shape instantiation (tchain.hip / tchain_fwd.hip).
//
// Rounds 1 - 3 also kept the first implementation here - a generic kernel whose 64-row tile lived in LDS, fp32 MFMA only - as
// the fallback for shapes none of the families takes (GFV_TCHAIN=0 forced it).  Over the whole GPU test suite not one launch
// needed it (GFV_ROWTILE_DEBUG, round 4: 0 of ~10^5 launches), so it is gone: such a shape is an argument error now.
#include <atomic>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include "gfv_common.h"
#include "gfv_limits.h"
#include "gfv_prof.h"
#include "../../include/gfv.h"

namespace {
constexpr int BM = 64;   // rows per tile of the chain kernels (= rows of ln_partial per tile)
}  // namespace

int gfv_internal_tchain_launch(const gfv_rowtile_args_t* args, int ragged, int f16, hipStream_t stream);  // tchain.hip

// GFV_F16SPLIT (or gfv_set_f16split): 0 = every GEMM product on the fp32 MFMA, even when a launch carries split-fp16
// weight images; 1 (default) = split-fp16 products; 2 = the reduced-precision form, ONE fp16 x fp16 product with fp32
// accumulation (the high parts only); 3 = the same with bf16 operands (v_mfma_f32_16x16x32_bf16: BASELINE config 3's wording;
// weight images must be built in the form they are used in - gfv_weight_images reads it); shared with dw.hip
// State: a PROCESS-WIDE default (atomic; gfv_set_f16split) - PyTorch runs the backward of an autograd node on its device
// worker thread, and a form chosen on the user's thread must reach the launches issued there - and a per-thread OVERRIDE
// (gfv_set_f16split_thread; -1 = none) for two host threads driving two models in different forms; a launch may also carry its
// own, gfv_rowtile_args_t.product_form / .hidden
static std::atomic<int> g_f16_default{-1};
static thread_local int g_f16split = -1;   // this thread's override
static int f16_default() {
  int d = g_f16_default.load(std::memory_order_relaxed);
  if (d < 0) {
    const char* e = getenv("GFV_F16SPLIT");
    d = e ? ((atoi(e) == 2 || atoi(e) == 3) ? atoi(e) : (atoi(e) != 0)) : 1;
    int expect = -1;
    if (!g_f16_default.compare_exchange_strong(expect, d)) d = expect;
  }
  return d;
}
extern "C" int gfv_f16split_enabled(void) { return g_f16split >= 0 ? g_f16split : f16_default(); }
extern "C" int gfv_set_f16split(int32_t on) {
  g_f16_default.store((on == 2 || on == 3) ? on : (on ? 1 : 0));
  g_f16split = -1;   // (the caller asked for the process-wide form: its own override, if any, would hide it)
  return GFV_OK;
}
extern "C" int gfv_set_f16split_thread(int32_t on) {
  if (on < -1 || on > 3) return GFV_ERR_ARG;
  g_f16split = on;
  return GFV_OK;
}
static int f16_mode() { return gfv_f16split_enabled(); }

// hidden_size of the model the following launches belong to (utils/get_param.py:69; default 128).  A model of h < 128 runs
// zero-padded to 128 columns (FVMmodel/padding.py): what changes in the kernels is the LayerNorm width (statistics over the h
// real columns) and the attention scale (dim_head = h / 8).  Host-side state read by the launchers (chain: passed in the
// kernel arguments, also of the generic row-tile kernel; weight gradient: DwLaunch; slice attention: its argument struct) -
// set it before the launches of a model, gfv.engine.Engine does on every forward / backward.
static thread_local int g_hidden = 128;
extern "C" int gfv_hidden_size(void) { return g_hidden; }
extern "C" int gfv_set_hidden_size(int32_t h) {
  if (h < 16 || h > 128 || (h & 15)) return GFV_ERR_ARG;
  g_hidden = h;   // (host state only: safe inside a stream capture)
  return GFV_OK;
}

static thread_local int g_last_path = -1;
extern "C" int gfv_rowtile_last_path(void) { return g_last_path; }
// ---- dispatch limits (gfv_limits.h) ----
namespace {
struct LimitDef { const char* env; int dflt; };
const LimitDef k_limits[GFV_LIM_COUNT] = {
    {"GFV_CBWD", 1},        {"GFV_CBWD_MAX_M", 25000},      {"GFV_CFWD", 1},          {"GFV_CFWD_MAX_M", 250000},
    {"GFV_CTRANS", 1},         {"GFV_CTRANS_MAX_M", 16384},
    {"GFV_LIN1S", 1},       {"GFV_LIN1S_MAX_M", 16384}};
int g_limit[GFV_LIM_COUNT];
bool g_limit_set[GFV_LIM_COUNT];
std::once_flag g_limit_once;
void limits_init() {
  for (int i = 0; i < GFV_LIM_COUNT; ++i) {
    const char* e = getenv(k_limits[i].env);
    g_limit[i] = e ? atoi(e) : k_limits[i].dflt;
    g_limit_set[i] = false;
  }
}
}  // namespace
int gfv_internal_limit(int which) {
  std::call_once(g_limit_once, limits_init);
  return (which >= 0 && which < GFV_LIM_COUNT) ? g_limit[which] : 0;
}
extern "C" int gfv_get_limit(int32_t which) { return (which >= 0 && which < GFV_LIM_COUNT) ? gfv_internal_limit(which) : -1; }
extern "C" int gfv_set_limit(int32_t which, int32_t value) {
  if (which < 0 || which >= GFV_LIM_COUNT) return GFV_ERR_ARG;
  std::call_once(g_limit_once, limits_init);
  if (value < 0) {   // back to the environment's / the built-in value
    const char* e = getenv(k_limits[which].env);
    g_limit[which] = e ? atoi(e) : k_limits[which].dflt;
  } else {
    g_limit[which] = value;
  }
  return GFV_OK;
}
extern "C" const char* gfv_limit_name(int32_t which) { return (which >= 0 && which < GFV_LIM_COUNT) ? k_limits[which].env : nullptr; }

static thread_local int g_last_ln_rows = 0;
extern "C" int gfv_rowtile_last_ln_rows(void) { return g_last_ln_rows; }

extern "C" int gfv_rowtile_tiles(int32_t M) { return (M + BM - 1) / BM; }

static int rowtile_chain_impl(const gfv_rowtile_args_t* args, void* stream);
extern "C" int gfv_rowtile_chain(const gfv_rowtile_args_t* args, void* stream) {
  if (!args) return GFV_ERR_ARG;
  // the launch's own product form / hidden size (0: the calling thread's context)
  if (args->product_form < 0 || args->product_form > 4 || (args->hidden != 0 && (args->hidden < 16 || args->hidden > 128 || (args->hidden & 15))))
    return GFV_ERR_ARG;
  if (args->product_form == 0 && args->hidden == 0) return rowtile_chain_impl(args, stream);
  const int f0 = g_f16split, h0 = g_hidden;   // (f0: this thread's override or -1; restored below)
  if (args->product_form) g_f16split = args->product_form - 1;   // 1 fp32 MFMA, 2 split-fp16, 3 / 4 reduced precision (fp16 / bf16)
  if (args->hidden) g_hidden = args->hidden;
  const int rc = rowtile_chain_impl(args, stream);
  g_f16split = f0;
  g_hidden = h0;
  return rc;
}
int gfv_internal_lin1_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry);   // lin1.hip
int gfv_internal_cfwd_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry);   // cfwd.hip
bool gfv_internal_wimg_form_ok(const float* wmax);                                               // wimg.hip
int gfv_internal_cbwd_try(const gfv_rowtile_args_t* a, int lowp, hipStream_t stream, int dry);   // cbwd.hip
// the column-owner small-tile forward (cfwd.hip) reads the LayerNorm width from its arguments
static int cfwd_try(const gfv_rowtile_args_t* args, int lowp, hipStream_t stream, int dry) {
  if (args->nlayers != 3 || (args->fin_op != GFV_FIN_LN && args->fin_op != GFV_FIN_PLAIN)) return 0;
  gfv_rowtile_args_t local = *args;
  local.hidden = g_hidden;
  return gfv_internal_cfwd_try(&local, lowp, stream, dry);
}
// the column-owner small-tile backward (cbwd.hip): the dX chain behind a LayerNorm backward, no fused weight gradients
static int cbwd_try(const gfv_rowtile_args_t* args, int lowp, hipStream_t stream, int dry) {
  if (args->in_op != GFV_IN_LNBWD || args->dw_partial || !args->in_stats) return 0;
  gfv_rowtile_args_t local = *args;
  local.hidden = g_hidden;
  return gfv_internal_cbwd_try(&local, lowp, stream, dry);
}
static int rowtile_chain_impl(const gfv_rowtile_args_t* args, void* stream) {
  if (!args || args->M < 0 || args->nlayers < 1 || args->nlayers > 3 || args->nseg < 1 || args->nseg > 3) return GFV_ERR_ARG;
  if (args->M == 0) {
    g_last_ln_rows = 0;   // (an empty batch fills no ln_partial row: a caller that sums gfv_rowtile_last_ln_rows() rows sums none)
    return GFV_OK;
  }
  for (int i = 0; i < args->nseg; ++i)
    if (args->seg[i].width < 1 || args->seg[i].width > 128) return GFV_ERR_ARG;
  for (int l = 0; l + 1 < args->nlayers; ++l)
    if (args->layer[l].N != 128 || args->layer[l + 1].K != 128) return GFV_ERR_ARG;
  int k0 = 0;
  for (int i = 0; i < args->nseg; ++i) k0 += args->seg[i].width;
  if (k0 != args->layer[0].K) return GFV_ERR_ARG;
  const gfv_layer_t& last = args->layer[args->nlayers - 1];
  if (last.N > 384 || last.N < 1) return GFV_ERR_ARG;
  if ((args->rc_Wh[0] || args->rc_Wh[1]) && !args->dw_partial) return GFV_ERR_ARG;   // (recompute: the fused column-owner backward only)
  if (args->padd && (args->nlayers < 2 || args->padd_ld < 256 || (args->padd_ld & 3) || !args->padd_s || !args->padd_r)) return GFV_ERR_ARG;
  for (int l = 0; l < args->nlayers; ++l)
    if (args->layer[l].ldw != 0 && args->layer[l].ldw < args->layer[l].K) return GFV_ERR_ARG;
  if (last.N > 128 && (last.N & 15)) return GFV_ERR_ARG;
  if ((args->fin_op != GFV_FIN_PLAIN || args->in_op == GFV_IN_LN || args->in_op == GFV_IN_LNBWD) &&
      (args->fin_op != GFV_FIN_PLAIN ? last.N != 128 : false))
    return GFV_ERR_ARG;
  if ((args->in_op == GFV_IN_LN || args->in_op == GFV_IN_LNBWD) && (args->nseg != 1 || args->seg[0].width != 128))
    return GFV_ERR_ARG;
  // the register-resident chain: segments 32-multiples wide, a last layer whose final 128-chunk may be 64 wide (NodeBlock dX: 128 + 64)
  bool fast_t = true;
  for (int i = 0; i < args->nseg; ++i) fast_t = fast_t && (args->seg[i].width % 32 == 0) && (args->seg[i].ld % 4 == 0);
  for (int l = 0; l < args->nlayers; ++l) {
    const bool lastl = (l == args->nlayers - 1);
    fast_t = fast_t && (args->layer[l].K % 4 == 0) && (args->layer[l].ldw % 4 == 0) && (args->layer[l].N % (lastl ? 64 : 128) == 0);
  }
  if (args->fin_op != GFV_FIN_PLAIN) fast_t = fast_t && last.N == 128;
  for (int c = 0; c < 3; ++c) {
    if (args->out[c]) fast_t = fast_t && (args->out_ld[c] % 4 == 0);
    if (args->res[c]) fast_t = fast_t && (args->res_ld[c] % 4 == 0);
  }
  // ragged shapes the register-resident chain also takes (its RAG instantiation): any segment width / row stride, any
  // first-layer K, a last layer of any width <= 128 (decoder N = 3); no LayerNorm backward, no DGELU on a ragged last
  // layer, inner layers 128 wide
  bool rag_t = !fast_t && args->in_op != GFV_IN_LNBWD && args->fin_op != GFV_FIN_LNBWD;
  for (int l = 0; l + 1 < args->nlayers; ++l) rag_t = rag_t && args->layer[l].N == 128;
  if (last.N > 128) rag_t = rag_t && (last.N % 64 == 0);
  if (last.N % 16) rag_t = rag_t && last.op != GFV_OP_MUL_DGELU && args->fin_op == GFV_FIN_PLAIN && !args->out_nores;
  if (args->in_op == GFV_IN_LN) rag_t = rag_t && args->seg[0].width == 128 && (args->seg[0].ld % 4 == 0);
  if (args->gadd) rag_t = rag_t && args->seg[0].width == 128 && (args->seg[0].ld % 4 == 0);
  for (int l = 1; l < args->nlayers; ++l) rag_t = rag_t && (args->layer[l].K == 128);
  // segmented-sum segments / per-segment saves: the plain instantiation of the register-resident chain only
  {
    bool csr = false;
    for (int i = 0; i < args->nseg; ++i) {
      const gfv_seg_t& sg = args->seg[i];
      csr = csr || sg.csr_rowptr != nullptr || sg.save != nullptr;
      if (sg.csr_rowptr && (!sg.idx || (i == 0 && args->in_add))) return GFV_ERR_ARG;
    }
    if (csr && (!fast_t || args->in_op == GFV_IN_LNBWD || args->fin_op == GFV_FIN_LNBWD))
      return GFV_ERR_ARG;
  }
  // split-fp16 form: every layer carries a weight image; first-layer segments start at 32-k slice boundaries
  bool f16 = (fast_t || rag_t) && f16_mode() != 0 && args->wmax != nullptr;
  for (int l = 0; l < args->nlayers; ++l) f16 = f16 && args->layer[l].Wh != nullptr;
  for (int i = 0; i + 1 < args->nseg; ++i) f16 = f16 && (args->seg[i].width % 32 == 0);
  // images built in the other class of product form (fp16 parts against bf16 high parts: include/gfv.h) are refused, not multiplied
  if (f16 && !gfv_internal_wimg_form_ok(args->wmax)) return GFV_ERR_ARG;
  // a layer may come without fp32 weights (a row-stacked virtual layer that exists as an image only): split form or nothing
  for (int l = 0; l < args->nlayers; ++l)
    if ((!args->layer[l].W || args->layer[l].bias2) && !f16) return GFV_ERR_ARG;
  if (!fast_t && !rag_t) {
    // no kernel family takes this shape (rounds 1 - 3 sent it to a generic LDS kernel; no launch of the test suite ever did)
    static const bool dbg = getenv("GFV_ROWTILE_DEBUG") != nullptr;
    if (dbg)
      fprintf(stderr, "[gfv] gfv_rowtile_chain: unsupported shape M=%d nseg=%d widths=%d,%d,%d ld0=%d nlayers=%d K0=%d Nlast=%d out_ld=%d in_op=%d fin_op=%d\n",
              args->M, args->nseg, args->seg[0].width, args->nseg > 1 ? args->seg[1].width : 0,
              args->nseg > 2 ? args->seg[2].width : 0, args->seg[0].ld, args->nlayers, args->layer[0].K, last.N,
              args->out_ld[0], args->in_op, args->fin_op);
    return GFV_ERR_ARG;
  }
  void* tok = nullptr;
  if (gfv_prof_enabled()) {
    // algorithmic work: 2*M*K*N flops per layer; bytes: every input row read once, every output/saved row written once
    double fl = 0, by = 0;
    for (int l = 0; l < args->nlayers; ++l) {
      fl += 2.0 * args->M * (double)args->layer[l].K * args->layer[l].N;
      by += 4.0 * ((double)args->layer[l].K * args->layer[l].N + args->layer[l].N);
      if (args->layer[l].save) by += 4.0 * args->M * args->layer[l].N;
      if (args->layer[l].aux) by += 4.0 * args->M * args->layer[l].N;
    }
    by += 4.0 * args->M * (double)args->layer[0].K + 4.0 * args->M * (double)last.N;
    for (int i = 0; i < args->nseg; ++i) if (args->seg[i].idx) by += 4.0 * args->M;
    if (args->in_aux || args->fin_aux) by += 4.0 * args->M * 128.0;
    if (args->in_save) by += 4.0 * args->M * 128.0;
    if (args->fin_presave) by += 4.0 * args->M * 128.0;
    if (args->out_nores) by += 4.0 * args->M * 128.0;
    if (args->padd) by += 4.0 * args->M * 258.0;
    for (int c = 0; c < 3; ++c) if (args->res[c]) by += 4.0 * args->M * 128.0;
    const int lnm = args->in_op == GFV_IN_LNBWD ? 1 : (args->fin_op == GFV_FIN_LNBWD ? 2 : 0);
    int kind = fast_t ? GFV_K_TCHAIN0 + lnm : GFV_K_TCHAIN_RAG;
    for (int i = 0; i < args->nseg; ++i)
      if (args->seg[i].csr_rowptr || args->seg[i].save) kind = GFV_K_TCHAIN_CSR;
    if (fast_t && f16 && args->nlayers == 1 && gfv_internal_lin1_try(args, f16_mode() >= 2 ? f16_mode() - 1 : 0, (hipStream_t)stream, 1))
      kind = GFV_K_LIN1;   // the lean single-layer kernel (same algorithmic work as the chain launch it stands in for)
    if (f16 && cfwd_try(args, f16_mode() >= 2 ? f16_mode() - 1 : 0, (hipStream_t)stream, 1))
      kind = GFV_K_COLCHAIN_FWD;   // the column-owner small-tile forward (same algorithmic work)
    if (args->dw_partial) {
      // dX chain with fused weight gradients (column-owner backward family): + the two (three) weight-gradient GEMMs, the
      // forward's row statistics read, the first Linear's input rows read, the per-workgroup blocks written
      kind = GFV_K_COLCHAIN_BWD;
      const int nfused = 2 + (args->rc_Wh[0] ? 2 : 0);   // (+ the two recomputed forward layers)
      fl += nfused * 2.0 * args->M * 128.0 * 128.0;
      by += 8.0 * args->M + 4.0 * (double)gfv_rowtile_dw_partials_m(args->M) * (double)args->dw_partial_stride;
    }
    tok = gfv_prof_begin(kind, fl, by, (hipStream_t)stream);
  }
  g_last_path = (fast_t ? 1 : 2) + (f16 ? 4 : 0);
  g_last_ln_rows = (args->M + BM - 1) / BM;
  if (args->dw_partial && !(fast_t && f16)) return GFV_ERR_ARG;   // (fused weight gradients: ask gfv_rowtile_fuses_dw first)
  if (fast_t && f16 && args->nlayers == 1 && gfv_internal_lin1_try(args, f16_mode() >= 2 ? f16_mode() - 1 : 0, (hipStream_t)stream, 0)) {
    g_last_path += 32;   // the lean single-layer kernel (lin1.hip)
  } else if (f16 && !args->dw_partial && cfwd_try(args, f16_mode() >= 2 ? f16_mode() - 1 : 0, (hipStream_t)stream, 0)) {
    g_last_path += 64;   // the column-owner small-tile forward (cfwd.hip)
  } else if (fast_t && f16 && cbwd_try(args, f16_mode() >= 2 ? f16_mode() - 1 : 0, (hipStream_t)stream, 0)) {
    g_last_path += 128;   // the column-owner small-tile backward (cbwd.hip): ln_partial holds one row per 32 rows
    g_last_ln_rows = (args->M + 31) / 32;
  } else if (fast_t) {
    const int took = gfv_internal_tchain_launch(args, 0, f16 ? 1 : 0, (hipStream_t)stream);   // 1: the column-owner family, 2: with fused dW
    if (args->dw_partial && took != 2) return GFV_ERR_ARG;
    g_last_path += 8 * took;
  }
  else
    gfv_internal_tchain_launch(args, 1, f16 ? 1 : 0, (hipStream_t)stream);
  gfv_prof_end(tok, (hipStream_t)stream);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}
