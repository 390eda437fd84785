// gfv-build-flags: -fno-slp-vectorize
// Launcher of the column-owner persistent backward chain (kernel: colchain_kernel.h).  gfv_rowtile_chain (rowtile.hip) -> the
// register-resident chain's launcher (tchain.hip) asks here first: a backward launch with fused weight gradients
// (gfv_rowtile_args_t.dw_partial) in the split-fp16 form whose shape this family covers takes it, everything else stays where
// it was.  GFV_COLCHAIN_BWD=0 switches the family off, GFV_COLCHAIN_BWD_MIN_M moves the size threshold.
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
  static const int min_m = cc_env("GFV_COLCHAIN_BWD_MIN_M", 2048);
  if (a->flags & GFV_CHAIN_ROW_OWNER) return false;
  if (!(a->flags & GFV_CHAIN_COLUMN_OWNER) && (!on || a->M < min_m)) return false;
  if (!a->dw_partial || a->dw_partial_stride < GFV_DW_FUSED_FLOATS || !a->in_stats || !a->wmax) return false;
  if (a->dw_in) return false;   // (the first Linear's weight gradient fused as well: removed in round 6, include/gfv.h)
  // (two layers: the input needs no gradient - the encoders; out[0] then receives gz1, the last layer's op is the second DGELU)
  const bool noout = a->nlayers == 2;
  // recompute form: the forward images of the second and third Linear are given, z2 (layer[0].aux) and y (in_aux) are not read
  const bool rc = a->rc_Wh[0] != nullptr;
  if (rc && (!a->rc_Wh[1] || a->dw_in || !al16(a->rc_Wh[0]) || !al16(a->rc_Wh[1]) || (a->rc_bias[0] && !al16(a->rc_bias[0])) ||
             (a->rc_bias[1] && !al16(a->rc_bias[1]))))
    return false;
  if (!rc && a->rc_Wh[1]) return false;
  if ((a->nlayers != 3 && !noout) || a->in_op != GFV_IN_LNBWD || a->fin_op != GFV_FIN_PLAIN || a->nseg != 1) return false;
  if (noout && (a->layer[1].op != GFV_OP_MUL_DGELU || a->layer[1].save || a->res[0] || a->gadd || a->dw_in || !a->out[0])) return false;
  // (every [M, 128] array of the launch shares one byte offset per row: row stride 128 everywhere, at most 2^22 rows)
  if (a->seg[0].width != 128 || a->seg[0].idx || a->seg[0].csr_rowptr || a->seg[0].save || a->seg[0].ld != 128 || !al16(a->seg[0].ptr)) return false;
  if (a->M > (1 << 22) || a->out_ld[0] != 128 || (a->res[0] && a->res_ld[0] != 128) || (a->dw_in && a->dw_in_ld != 128)) return false;
  if (a->layer[0].op != GFV_OP_MUL_DGELU || a->layer[1].op != GFV_OP_MUL_DGELU || (!noout && a->layer[2].op != GFV_OP_NONE)) return false;
  for (int l = 0; l < a->nlayers; ++l) {
    const gfv_layer_t& L = a->layer[l];
    if (!L.Wh || (L.N != 128 && !(l == 2 && L.N == 192)) || L.K != 128 || L.bias || L.bias2) return false;
    if (l < 2 && !(rc && l == 0) && (!L.aux || !al16(L.aux))) return false;   // (RC: z2 is recomputed)
    if (L.save && !al16(L.save)) return false;
  }
  if ((!rc && (!a->in_aux || !al16(a->in_aux))) || !a->in_gamma || !al16(a->in_gamma) || a->ln_partial || a->padd) return false;
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
// a launch over M rows runs min(CUs, 64-row tiles) workgroups: on a short launch the blocks of idle workgroups were 34 MB written
// and read again for nothing (round 5: 21 - 26 us per launch at 1 - 5 k rows, profiles/r05_latency_floor.txt)
extern "C" int gfv_rowtile_dw_partials_m(int32_t M) {
  const int tiles = (M + 63) / 64;
  return tiles < cc_cus() ? (tiles > 0 ? tiles : 1) : cc_cus();
}
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
    const dim3 grid(gfv_rowtile_dw_partials_m(a->M)), blk(64 * CC_W);
#define CB_K(...) GFV_LAUNCH((colchain_bwd_kernel<__VA_ARGS__>), grid, blk, 0, stream, *a, st)
#define CB_LAUNCH(LOWP, RC)                                                                                                      \
  do {                                                                                                                           \
    if (a->nlayers == 2) CB_K(LOWP, false, false, true, RC);                                                                     \
    else if (a->layer[2].N == 192 && a->gadd) CB_K(LOWP, true, true, false, RC);                                                 \
    else if (a->layer[2].N == 192) CB_K(LOWP, false, true, false, RC);                                                           \
    else if (a->gadd) CB_K(LOWP, true, false, false, RC);                                                                        \
    else CB_K(LOWP, false, false, false, RC);                                                                                    \
  } while (0)
    const bool rc = a->rc_Wh[0] != nullptr;
    // (product_form as gfv_internal_tchain_launch left it: 0 three products, 1 / 2 the single-product forms in fp16 / bf16)
    if (a->product_form == 2) { if (rc) CB_LAUNCH(2, true); else CB_LAUNCH(2, false); }
    else if (a->product_form != 0) { if (rc) CB_LAUNCH(1, true); else CB_LAUNCH(1, false); }
    else { if (rc) CB_LAUNCH(0, true); else CB_LAUNCH(0, false); }
#undef CB_K
#undef CB_LAUNCH
    return 2;
  }
  // (the column-owner FORWARD family of round 3 - two generations, parity-green, never faster than the row-owner chain:
  // DESIGN.md 5 - is gone; GFV_CHAIN_COLUMN_OWNER on a forward launch falls through to the row-owner chain)
  return 0;
}
