#pragma once
#include <hip/hip_runtime.h>
enum { GFV_K_ROWTILE = 1, GFV_K_DW = 2, GFV_K_SEG = 3, GFV_K_SLICE = 4, GFV_K_FVM = 5, GFV_K_MISC = 6,
       GFV_K_TCHAIN0 = 7, GFV_K_TCHAIN1 = 8, GFV_K_TCHAIN2 = 9,  // tchain_kernel<1, LNM, false>
       GFV_K_TCHAIN_RAG = 10 };                                  // tchain_kernel<1, 0, true>
bool gfv_prof_enabled();
void* gfv_prof_begin(int kind, double flops, double bytes, hipStream_t st);
void gfv_prof_end(void* tok, hipStream_t st);
