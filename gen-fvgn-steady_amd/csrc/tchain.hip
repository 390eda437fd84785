// Launchers of the register-resident chain (kernel template: tchain_kernel.h).  This translation unit holds every
// instantiation except the plain ones with compile-time element ops, which tchain_fwd.hip compiles without the SLP vectoriser.
#include "tchain_kernel.h"

void gfv_internal_tchain_fwd_plain(const gfv_rowtile_args_t* args, int f16, int iop, hipStream_t stream);   // tchain_fwd.hip
int gfv_internal_colchain_try(const gfv_rowtile_args_t* args, hipStream_t stream);                              // colchain.hip
void gfv_internal_tchain_launch_bf16(const gfv_rowtile_args_t* args, int ragged, int lnm, hipStream_t stream);  // tchain_bf16.hip

// 1 / 2: every layer before the last has GFV_OP_BIAS_GELU / GFV_OP_MUL_DGELU and the prologue op is none or the LayerNorm
// backward (the IOP instantiations, tchain_kernel.h); 0: anything else - the instantiation that reads the ops at run time
static int chain_iop(const gfv_rowtile_args_t* a) {
  if (a->in_op != GFV_IN_NONE && a->in_op != GFV_IN_LNBWD) return 0;
  int iop = 0;
  for (int l = 0; l + 1 < a->nlayers; ++l) {
    const int op = a->layer[l].op;
    if (op != GFV_OP_BIAS_GELU && op != GFV_OP_MUL_DGELU) return 0;
    if (iop != 0 && iop != op) return 0;
    iop = op;
  }
  return iop ? iop : GFV_OP_BIAS_GELU;   // (a single layer has no inner epilogue: either form)
}

template <int NW>
static void launch_h(const gfv_rowtile_args_t* args, int ragged, int lnm, hipStream_t stream) {
  const int tiles = (args->M + 16 * NW - 1) / (16 * NW);
  const dim3 wgs(tiles), blk(64 * NW);
  const int iop = chain_iop(args);
  bool csr = false;
  for (int i = 0; i < args->nseg; ++i) csr = csr || args->seg[i].csr_rowptr != nullptr || args->seg[i].save != nullptr;
  if (csr) {   // (the caller checked: plain instantiation only)
    if (iop == GFV_OP_BIAS_GELU)
      GFV_LAUNCH((tchain_kernel<1, 0, false, true, NW, true, 1>), dim3(gfv_xcd_grid(tiles)), blk, 0, stream, *args);
    else
      GFV_LAUNCH((tchain_kernel<1, 0, false, true, NW, true>), dim3(gfv_xcd_grid(tiles)), blk, 0, stream, *args);
    return;
  }
  if (ragged) GFV_LAUNCH((tchain_kernel<1, 0, true, true, NW>), wgs, blk, 0, stream, *args);
  else if (lnm == 0 && iop != 0) gfv_internal_tchain_fwd_plain(args, 1, iop, stream);
  else if (lnm == 0) GFV_LAUNCH((tchain_kernel<1, 0, false, true, NW>), wgs, blk, 0, stream, *args);
  else if (lnm == 1 && iop == GFV_OP_MUL_DGELU) GFV_LAUNCH((tchain_kernel<1, 1, false, true, NW, false, 2>), wgs, blk, 0, stream, *args);
  else if (lnm == 1) GFV_LAUNCH((tchain_kernel<1, 1, false, true, NW>), wgs, blk, 0, stream, *args);
  else GFV_LAUNCH((tchain_kernel<1, 2, false, true, NW>), wgs, blk, 0, stream, *args);
}

// fast-path launcher used by gfv_rowtile_chain (rowtile.hip); ragged: the instantiation that also takes ragged shapes
// (only without LayerNorm backward); f16: the split-fp16 form (every layer has a weight image)
extern "C" int gfv_f16split_enabled(void);   // rowtile.hip: 0 fp32 MFMA, 1 split-fp16, 2 / 3 reduced precision (fp16 / bf16)
extern "C" int gfv_hidden_size(void);         // rowtile.hip

int gfv_internal_tchain_launch(const gfv_rowtile_args_t* args_in, int ragged, int f16, hipStream_t stream) {
  gfv_rowtile_args_t local = *args_in;
  local.product_form = f16 ? (gfv_f16split_enabled() == 2 ? 1 : (gfv_f16split_enabled() == 3 ? 2 : 0)) : 0;   // 1 / 2: the single-product forms (fp16 / bf16)
  local.hidden = gfv_hidden_size();   // LayerNorm width (gfv_set_hidden_size)
  const gfv_rowtile_args_t* args = &local;
  const int lnm = args->in_op == GFV_IN_LNBWD ? 1 : (args->fin_op == GFV_FIN_LNBWD ? 2 : 0);
  const dim3 wgs((args->M + 63) / 64), blk(256);
  if (f16 && !ragged) {   // the column-owner persistent family
    const int took = gfv_internal_colchain_try(args, stream);
    if (took) return took;
  }
  if (f16 && local.product_form == 2) {   // the bf16 form: the run-time-op instantiations of its own translation unit
    gfv_internal_tchain_launch_bf16(args, ragged, lnm, stream);
    return 0;
  }
  if (f16) {
    // (an 8-wave workgroup sharing one weight stream over 128 rows - launch_h<8> - was measured in round 2: 5.28 ms / step
    // against 4.96 with it on the 75 k-row launches, 5.29 with it everywhere; the 4-wave form stays)
    launch_h<4>(args, ragged, lnm, stream);
    return 0;
  }
  // fp32-MFMA form: the run-time-op instantiations only (it is the reference form of the tests, not the product path)
  bool csr = false;
  for (int i = 0; i < args->nseg; ++i) csr = csr || args->seg[i].csr_rowptr != nullptr || args->seg[i].save != nullptr;
  if (csr) GFV_LAUNCH((tchain_kernel<1, 0, false, false, 4, true>), dim3(gfv_xcd_grid((args->M + 63) / 64)), blk, 0, stream, *args);
  else if (ragged) GFV_LAUNCH((tchain_kernel<1, 0, true, false>), wgs, blk, 0, stream, *args);
  else if (lnm == 0) gfv_internal_tchain_fwd_plain(args, 0, 0, stream);
  else if (lnm == 1) GFV_LAUNCH((tchain_kernel<1, 1, false, false>), wgs, blk, 0, stream, *args);
  else GFV_LAUNCH((tchain_kernel<1, 2, false, false>), wgs, blk, 0, stream, *args);
  return 0;
}
