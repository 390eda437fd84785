#pragma once
#include <hip/hip_runtime.h>
enum { GFV_K_ROWTILE = 1, GFV_K_DW = 2, GFV_K_SEG = 3, GFV_K_SLICE = 4, GFV_K_FVM = 5, GFV_K_MISC = 6,
       GFV_K_TCHAIN0 = 7, GFV_K_TCHAIN1 = 8, GFV_K_TCHAIN2 = 9,  // tchain_kernel<1, LNM, false>
       GFV_K_TCHAIN_RAG = 10,                                    // tchain_kernel<1, 0, true>
       GFV_K_REDUCE = 11,                                        // reduce_partials_* (deterministic second stages)
       GFV_K_WIMG = 12,                                          // per-step weight images + transposed copies
       GFV_K_TCHAIN_CSR = 13,                                    // tchain_kernel<1, 0, false, H, 4, true>: segmented-sum segments
       GFV_K_COLCHAIN_BWD = 14,                                  // colchain_bwd_kernel: dX chain with fused weight gradients
       GFV_K_COLCHAIN_FWD = 15,                                  // colchain_fwd*_kernel (column-owner forward family)
       GFV_K_LIN1 = 16,                                          // lin1_*_kernel: single-layer launches of gfv_rowtile_chain on the lean kernel
       GFV_K_COUNT = 17 };
bool gfv_prof_enabled();
void* gfv_prof_begin(int kind, double flops, double bytes, hipStream_t st);
void gfv_prof_end(void* tok, hipStream_t st);

// sizes the entry points cannot see from their arguments (stencil entries S, (cell, face) incidences Sigma): set by the
// caller of the roofline leg so that the algorithmic bytes of the finite-volume kernels can be priced
double gfv_prof_size_S();
double gfv_prof_size_Sigma();

// prices everything an entry point launches as ONE record of `kind` (HIP events on the launch stream)
struct GfvProfScope {
  void* tok;
  hipStream_t st;
  GfvProfScope(int kind, double flops, double bytes, void* stream) : tok(nullptr), st((hipStream_t)stream) {
    if (gfv_prof_enabled()) tok = gfv_prof_begin(kind, flops, bytes, st);
  }
  ~GfvProfScope() { gfv_prof_end(tok, st); }
};
