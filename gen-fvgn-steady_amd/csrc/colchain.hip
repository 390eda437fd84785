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
  }
  return n;
}

// 1: launched; 0: not this family's launch
int gfv_internal_colchain_try(const gfv_rowtile_args_t* a, hipStream_t stream) {
  static const int on = cc_env("GFV_COLCHAIN", 1);
  static const int min_m = cc_env("GFV_COLCHAIN_MIN_M", 16384);
  if (a->flags & GFV_CHAIN_ROW_OWNER) return 0;
  if (!(a->flags & GFV_CHAIN_COLUMN_OWNER) && (!on || a->M < min_m)) return 0;
  if (a->nlayers != 3 || a->pad_ != 128 || !a->wmax) return 0;
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
  const dim3 grid(cc_cus()), blk(64 * CC_W);
  const bool lowp = a->pad3_ != 0;
#define CC_LAUNCH(KT0, N0, TG, PADD)                                                                                          \
  do {                                                                                                                        \
    if (lowp) hipLaunchKernelGGL((colchain_fwd_kernel<KT0, N0, TG, PADD, true>), grid, blk, 0, stream, *a, status);           \
    else hipLaunchKernelGGL((colchain_fwd_kernel<KT0, N0, TG, PADD, false>), grid, blk, 0, stream, *a, status);               \
  } while (0)
  if (edge && a->padd) CC_LAUNCH(4, 8, 8, true);
  else if (edge) CC_LAUNCH(4, 8, 8, false);
  else CC_LAUNCH(6, 4, 6, false);
#undef CC_LAUNCH
  return 1;
}
