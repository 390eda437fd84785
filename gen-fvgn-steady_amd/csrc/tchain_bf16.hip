// The register-resident chain in the bf16 single-product form (gfv_set_f16split(3): v_mfma_f32_16x16x32_bf16 on bf16-rounded
// operands; include/gfv.h).  Its own instantiations - the ones that read the element ops at run time - in its own translation
// unit, so that the default form's kernels keep their code and registers.
#include "tchain_kernel.h"

void gfv_internal_tchain_launch_bf16(const gfv_rowtile_args_t* args, int ragged, int lnm, hipStream_t stream) {
  const int tiles = (args->M + 63) / 64;
  const dim3 wgs(tiles), blk(256);
  bool csr = false;
  for (int i = 0; i < args->nseg; ++i) csr = csr || args->seg[i].csr_rowptr != nullptr || args->seg[i].save != nullptr;
  if (csr) GFV_LAUNCH((tchain_kernel<1, 0, false, true, 4, true, 0, true>), dim3(gfv_xcd_grid(tiles)), blk, 0, stream, *args);
  else if (ragged) GFV_LAUNCH((tchain_kernel<1, 0, true, true, 4, false, 0, true>), wgs, blk, 0, stream, *args);
  else if (lnm == 0) GFV_LAUNCH((tchain_kernel<1, 0, false, true, 4, false, 0, true>), wgs, blk, 0, stream, *args);
  else if (lnm == 1) GFV_LAUNCH((tchain_kernel<1, 1, false, true, 4, false, 0, true>), wgs, blk, 0, stream, *args);
  else GFV_LAUNCH((tchain_kernel<1, 2, false, true, 4, false, 0, true>), wgs, blk, 0, stream, *args);
}
