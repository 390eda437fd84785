// gfv-build-flags: -fno-slp-vectorize
// The plain instantiations of the register-resident chain (no LayerNorm backward, no ragged shapes, no segmented-sum
// segments; element ops known at compile time), compiled WITHOUT the SLP vectoriser.  hipcc's SLP pass packs neighbouring scalar fp32 operations into
// v_pk_* instructions: a quarter fewer VALU instructions, but the operand pairs need aligned register pairs and shuffling
// moves, and the kernel lands at 205 (f16 form) / 201 (fp32 form) VGPRs = 2 waves per SIMD.  Without the pass the same
// source needs 164 / 153 registers = 3 waves per SIMD, and that wins for this instantiation: 0.941 -> 0.881 ms per step
// over its 20 launches (A/B on one box, profiles/tools/abn.sh); the LayerNorm-backward instantiation stays at 256
// registers either way and loses 4 % to the extra instructions, the segmented-sum one (183) gains nothing: they keep the
// default in tchain.hip.
#include "tchain_kernel.h"

void gfv_internal_tchain_fwd_plain(const gfv_rowtile_args_t* args, int f16, int iop, hipStream_t stream) {
  const dim3 wgs((args->M + 63) / 64), blk(256);
  if (!f16) GFV_LAUNCH((tchain_kernel<1, 0, false, false>), wgs, blk, 0, stream, *args);
  else if (iop == GFV_OP_BIAS_GELU) GFV_LAUNCH((tchain_kernel<1, 0, false, true, 4, false, 1>), wgs, blk, 0, stream, *args);
  else GFV_LAUNCH((tchain_kernel<1, 0, false, true, 4, false, 2>), wgs, blk, 0, stream, *args);
}
