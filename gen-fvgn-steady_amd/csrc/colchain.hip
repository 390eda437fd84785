// gfv-build-flags: -fno-slp-vectorize
// Launcher of the column-owner persistent chain (kernel: colchain_kernel.h).  gfv_rowtile_chain (rowtile.hip) -> the
// register-resident chain's launcher (tchain.hip) asks here first: a launch in the split-fp16 form whose shape this family
// covers and that is big enough to give every CU a few groups of rows takes it, everything else stays where it was.
// GFV_COLCHAIN=0 switches the family off, GFV_COLCHAIN_MIN_M moves the size threshold.
#include "colchain_kernel.h"

int* gfv_internal_status_ptr();   // dw.hip: device address of the status word

static int cc_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static bool al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

static int cc_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount;
    if (n <= 0) n = 256;
    // (experiment: fewer persistent workgroups than CUs leaves CUs to the weight-gradient queue while a launch runs)
    const int want = cc_env("GFV_COLCHAIN_WGS", 0);
    if (want > 0 && want <= n) n = want;
  }
  return n;
}

// the backward form with fused weight gradients (colchain_bwd_kernel): does the launch qualify?
static bool cc_bwd_ok(const gfv_rowtile_args_t* a) {
  static const int on = cc_env("GFV_COLCHAIN_BWD", 1);
  static const int min_m = cc_env("GFV_COLCHAIN_BWD_MIN_M", 16384);
  if (a->flags & GFV_CHAIN_ROW_OWNER) return false;
  if (!(a->flags & GFV_CHAIN_COLUMN_OWNER) && (!on || a->M < min_m)) return false;
  if (!a->dw_partial || a->dw_partial_stride < (a->dw_in ? GFV_DW_FUSED_FLOATS_IN : GFV_DW_FUSED_FLOATS) || !a->in_stats || !a->wmax) return false;
  if (a->dw_in && (!al16(a->dw_in) || (a->dw_in_ld & 3) || a->dw_in_ld < 128)) return false;
  // (two layers: the input needs no gradient - the encoders; out[0] then receives gz1, the last layer's op is the second DGELU)
  const bool noout = a->nlayers == 2;
  if ((a->nlayers != 3 && !noout) || a->in_op != GFV_IN_LNBWD || a->fin_op != GFV_FIN_PLAIN || a->nseg != 1) return false;
  if (noout && (a->layer[1].op != GFV_OP_MUL_DGELU || a->layer[1].save || a->res[0] || a->gadd || a->dw_in || !a->out[0])) return false;
  // (every [M, 128] array of the launch shares one byte offset per row: row stride 128 everywhere, at most 2^22 rows)
  if (a->seg[0].width != 128 || a->seg[0].idx || a->seg[0].csr_rowptr || a->seg[0].save || a->seg[0].ld != 128 || !al16(a->seg[0].ptr)) return false;
  if (a->M > (1 << 22) || a->out_ld[0] != 128 || (a->res[0] && a->res_ld[0] != 128) || (a->dw_in && a->dw_in_ld != 128)) return false;
  if (a->layer[0].op != GFV_OP_MUL_DGELU || a->layer[1].op != GFV_OP_MUL_DGELU || (!noout && a->layer[2].op != GFV_OP_NONE)) return false;
  for (int l = 0; l < a->nlayers; ++l) {
    const gfv_layer_t& L = a->layer[l];
    if (!L.Wh || (L.N != 128 && !(l == 2 && L.N == 192)) || L.K != 128 || L.bias || L.bias2) return false;
    if (l < 2 && (!L.aux || !al16(L.aux))) return false;
    if (L.save && !al16(L.save)) return false;
  }
  if (!a->in_aux || !al16(a->in_aux) || !a->in_gamma || !al16(a->in_gamma) || a->ln_partial || a->padd) return false;
  const bool out2 = !noout && a->layer[2].N == 192;   // [x part 128 (+ residual) | neighbour-mean part 64]
  if (!a->out[0] || (out2 ? (!a->out[1] || a->out_ld[1] != 64 || !al16(a->out[1]) || a->dw_in) : a->out[1] != nullptr) || a->out[2] ||
      a->res[1] || a->res[2] || a->out_nores)
    return false;
  if ((a->out_ld[0] & 3) || !al16(a->out[0]) || (a->res[0] && ((a->res_ld[0] & 3) || !al16(a->res[0])))) return false;
  if (a->in_add && !al16(a->in_add)) return false;
  if (a->gadd && (!al16(a->gadd) || !a->gadd_s || !a->gadd_r)) return false;
  if (a->in_save && !al16(a->in_save)) return false;
  if ((reinterpret_cast<size_t>(a->in_stats) & 7) || !al16(a->dw_partial) || (a->dw_partial_stride & 3)) return false;
  return true;
}
extern "C" int gfv_hidden_size(void);
extern "C" int gfv_f16split_enabled(void);
extern "C" int gfv_rowtile_dw_partials(void) { return cc_cus(); }
extern "C" int gfv_rowtile_fuses_dw(const gfv_rowtile_args_t* a) {
  if (!a || gfv_hidden_size() != 128 || gfv_f16split_enabled() == 0) return 0;
  gfv_rowtile_args_t t = *a;
  t.hidden = 128;
  return cc_bwd_ok(&t) ? 1 : 0;
}

// 1: launched; 2: launched with fused weight gradients; 0: not this family's launch
int gfv_internal_colchain_try(const gfv_rowtile_args_t* a, hipStream_t stream) {
  if (a->hidden == 128 && cc_bwd_ok(a)) {
    int* st = gfv_internal_status_ptr();
    if (!st) return 0;
    const dim3 grid(cc_cus()), blk(64 * CC_W);
#define CB_LAUNCH(LOWP)                                                                                                          \
  do {                                                                                                                           \
    if (a->nlayers == 2) hipLaunchKernelGGL((colchain_bwd_kernel<LOWP, false, false, false, true>), grid, blk, 0, stream, *a, st);           \
    else if (a->layer[2].N == 192 && a->gadd) hipLaunchKernelGGL((colchain_bwd_kernel<LOWP, true, false, true>), grid, blk, 0, stream, *a, st);   \
    else if (a->layer[2].N == 192) hipLaunchKernelGGL((colchain_bwd_kernel<LOWP, false, false, true>), grid, blk, 0, stream, *a, st);        \
    else if (a->gadd && a->dw_in) hipLaunchKernelGGL((colchain_bwd_kernel<LOWP, true, true>), grid, blk, 0, stream, *a, st);      \
    else if (a->gadd) hipLaunchKernelGGL((colchain_bwd_kernel<LOWP, true, false>), grid, blk, 0, stream, *a, st);                 \
    else if (a->dw_in) hipLaunchKernelGGL((colchain_bwd_kernel<LOWP, false, true>), grid, blk, 0, stream, *a, st);                \
    else hipLaunchKernelGGL((colchain_bwd_kernel<LOWP, false, false>), grid, blk, 0, stream, *a, st);                             \
  } while (0)
    if (a->product_form != 0) CB_LAUNCH(true);
    else CB_LAUNCH(false);
#undef CB_LAUNCH
    return 2;
  }
  if (a->dw_partial) return 0;   // (the caller asked gfv_rowtile_fuses_dw first; anything else is an argument error upstream)
  static const int on = cc_env("GFV_COLCHAIN", 0);
  static const int min_m = cc_env("GFV_COLCHAIN_MIN_M", 16384);
  static const int max_m = cc_env("GFV_COLCHAIN_MAX_M", 1 << 30);
  if (a->flags & GFV_CHAIN_ROW_OWNER) return 0;
  if (!(a->flags & GFV_CHAIN_COLUMN_OWNER) && (!on || a->M < min_m || a->M > max_m)) return 0;
  if (a->nlayers != 3 || a->hidden != 128 || !a->wmax) return 0;
  for (int l = 0; l < 3; ++l) {
    const gfv_layer_t& L = a->layer[l];
    if (!L.Wh || L.N != 128 || L.bias2 || (l > 0 && L.K != 128)) return 0;
    if (L.bias && !al16(L.bias)) return 0;
    if (L.save && !al16(L.save)) return 0;
  }
  int k0 = 0;
  for (int i = 0; i < a->nseg; ++i) {
    const gfv_seg_t& s = a->seg[i];
    if (s.csr_rowptr || s.save || (s.width & 31) || (s.ld & 3) || !al16(s.ptr)) return 0;
    k0 += s.width;
  }
  if (k0 != a->layer[0].K) return 0;
  const bool edge = a->nseg == 1 && k0 == 128, node = a->nseg == 2 && a->seg[0].width == 64 && a->seg[1].width == 128;
  if (!edge && !node) return 0;
  if (!a->out[0] || a->out[1] || a->out[2] || a->res[1] || a->res[2]) return 0;
  if ((a->out_ld[0] & 3) || !al16(a->out[0]) || (a->res[0] && ((a->res_ld[0] & 3) || !al16(a->res[0])))) return 0;
  if (a->padd && (!al16(a->padd) || (a->padd_ld & 3))) return 0;
  if (a->out_nores && !al16(a->out_nores)) return 0;
  int* status = gfv_internal_status_ptr();
  if (!status) return 0;
  const bool fwd = a->in_op == GFV_IN_NONE && a->layer[0].op == GFV_OP_BIAS_GELU && a->layer[1].op == GFV_OP_BIAS_GELU &&
                   a->layer[2].op == GFV_OP_NONE && a->fin_op == GFV_FIN_LN && !a->in_add &&
                   !a->gadd && !a->in_save && !a->ln_partial && !a->gscale;
  if (!fwd) return 0;
  if (!al16(a->fin_gamma) || !al16(a->fin_beta) || (a->fin_presave && !al16(a->fin_presave))) return 0;
  if (node && a->padd) return 0;
  // second generation (colchain_fwd2_kernel): the edge shape (one 128-wide segment), every [M,128] array with row stride 128
  static const int gen2 = cc_env("GFV_COLCHAIN_GEN2", 1);   // 0: first generation; 2 / 4: waves per SIMD the allocation aims at
  if (gen2 && edge && a->M <= (1 << 22) && !a->seg[0].idx && a->seg[0].ld == 128 && a->out_ld[0] == 128 &&
      (!a->res[0] || a->res_ld[0] == 128) && (!a->padd || a->padd_ld <= 4096)) {
    const dim3 blk2(64 * CC_W);
    const bool lp = a->product_form != 0;
#define CF2(PADD, LP, WPS) hipLaunchKernelGGL((colchain_fwd2_kernel<PADD, LP, WPS>), dim3(cc_cus() * (WPS == 4 ? 2 : 1)), blk2, 0, stream, *a, status)
    if (gen2 == 4) {
      if (a->padd) { if (lp) CF2(true, true, 4); else CF2(true, false, 4); }
      else { if (lp) CF2(false, true, 4); else CF2(false, false, 4); }
    } else {
      if (a->padd) { if (lp) CF2(true, true, 2); else CF2(true, false, 2); }
      else { if (lp) CF2(false, true, 2); else CF2(false, false, 2); }
    }
#undef CF2
    return 1;
  }
  static const int lite = cc_env("GFV_COLCHAIN_LITE", 1);   // 1: the high-occupancy form (2 workgroups per CU), 0: resident weights
  const dim3 blk(64 * CC_W);
  const bool lowp = a->product_form != 0;
#define CC_LAUNCH(KT0, N0, TG, PADD, LITE)                                                                                    \
  do {                                                                                                                        \
    const dim3 grid(cc_cus() * (LITE ? 2 : 1));                                                                               \
    if (lowp) hipLaunchKernelGGL((colchain_fwd_kernel<KT0, N0, TG, PADD, true, LITE>), grid, blk, 0, stream, *a, status);     \
    else hipLaunchKernelGGL((colchain_fwd_kernel<KT0, N0, TG, PADD, false, LITE>), grid, blk, 0, stream, *a, status);         \
  } while (0)
  if (lite) {
    if (edge && a->padd) CC_LAUNCH(4, 8, 4, true, true);
    else if (edge) CC_LAUNCH(4, 8, 4, false, true);
    else CC_LAUNCH(6, 4, 2, false, true);
  } else {
    if (edge && a->padd) CC_LAUNCH(4, 8, 8, true, false);
    else if (edge) CC_LAUNCH(4, 8, 8, false, false);
    else CC_LAUNCH(6, 4, 6, false, false);
  }
#undef CC_LAUNCH
  return 1;
}
