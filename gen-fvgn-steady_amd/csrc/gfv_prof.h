#pragma once
#include <hip/hip_runtime.h>
enum { GFV_K_ROWTILE = 1, GFV_K_DW = 2, GFV_K_SEG = 3, GFV_K_SLICE = 4, GFV_K_FVM = 5, GFV_K_MISC = 6 };
bool gfv_prof_enabled();
void* gfv_prof_begin(int kind, double flops, double bytes, hipStream_t st);
void gfv_prof_end(void* tok, hipStream_t st);
