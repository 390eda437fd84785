// VERDICT r4 item 2, tested at the level of ONE phase before anybody writes the kernel: would a persistent backward of 4 waves x 32
// columns at 512 registers run its phases faster than the one that is built (8 waves x 16 columns, two waves per SIMD in lockstep)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I gen-fvgn-steady_amd/csrc \
//         -o /tmp/phase4 profiles/tools/phase4/phase4.hip && /tmp/phase4
// The two heaviest phase kinds of colchain_bwd_kernel (colchain_kernel.h), with the product's own device functions where the shape
// allows, in a persistent loop over 64-row tiles (one workgroup per CU, 40 tiles each, every tile reads its own z rows and writes
// its own gz rows: the memory side of the phase is there too):
//   CHAIN  a chain layer with the GELU' epilogue (P3 / P2): fragments from LDS x the layer's weight slice (fetched per tile, L2),
//          v = acc x gelu'(z), a = gelu(z), both split into the next fragments, v stored
//   DW     a fused weight-gradient phase: transposed operand reads (ds_read_b64_tr_b16) x 3 products per output tile
// Forms:
//   A   as built: 8 waves, a wave owns ONE n-tile (16 columns) of the chain and a 2 x 4 block of the weight gradient's 8 x 8 output tiles
//   B   4 waves, a wave owns TWO n-tiles (32 columns: every fragment read feeds two MFMA columns, a 16-byte fragment write per lane and
//       part) and a 4 x 4 block of the weight gradient (8 transposed reads per 16 output tiles instead of 6 per 8); launch bounds 256 x 1:
//       up to 512 registers
//   Bw  B with the second pair's matrix instructions woven between the first pair's epilogue arithmetic (sched_group_barrier: one
//       MFMA per ~10 vector instructions) - the instruction-level overlap round 4 could not afford in 256 registers
//   C   16 waves, four per SIMD: 8 column owners x 2 row pairs, <= 128 registers; a wave owns a 2 x 2 block of the weight gradient (16
//       accumulator registers per fused gradient) - more waves to hide each other's latencies instead of fewer
//   D   16 waves of TWO KINDS on 32-row tiles: 8 chain waves on tile t beside 8 weight-gradient waves on tile t - 1 (fragments handed over
//       in LDS, two generations), one barrier per 32-row tile: every SIMD holds two waves of each kind (cycles are per 64 ROWS = two of its tiles)
// Each with the tiles' rows streaming from / to HBM (1 KB per row and chain phase: what a phase of the real kernel moves) and with every
// tile of a workgroup on the same 64 rows (cache hits: the compute side alone).
// Output: cycles per tile and phase kind (wall time x clock / tiles per workgroup), and the registers each form took.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "colchain_kernel.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

namespace {

constexpr int TGX = 4;   // groups of 16 rows per tile

struct Args {
  const float* z;      // [M, 128] saved pre-activations
  float* v;            // [M, 128] the phase's stored product
  const void* wimg;    // one 128 x 128 image (64 KB)
  float* sink;         // [workgroups, 128, 128] weight-gradient partials
  int M, tiles_per_wg, do_chain, do_dw, alias;   // alias: every tile of a workgroup uses the rows of its first one (cache hits)
};

// ---- form A: the product's own device functions ----
__global__ __launch_bounds__(512, 2) void phase_a(const Args A, int* status) {
  __shared__ __attribute__((aligned(16))) char lds[3 * TGX * 8192];
  char* b0 = lds;
  char* b1 = lds + TGX * 8192;
  char* b2 = lds + 2 * TGX * 8192;
  CcCtx c;
  c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 16 * c.w + 4 * c.g;
  c.M = A.M;
  c.mabs = 0.f;
  c.invw = 1.0f;
  c.ngt = TGX;
  // some fragments to multiply: every lane fills its share of b0 and b2 once
  for (int q = 0; q < TGX; ++q) {
    const float a[4] = {0.01f * c.lane, 0.02f * c.w, 0.5f, -0.25f + q};
    cc_put_frag<false>(b0, q, c, a, 64.0f);
    cc_put_frag<false>(b2, q, c, a, 16.0f);
  }
  const size_t rows128 = (size_t)A.M * 512;
  const cb_rsrc bz = cb_buf(A.z, rows128), bv = cb_buf(A.v, rows128), w0 = cb_buf(A.wimg, 65536);
  const int woff = (c.w * 128 + c.lane) * 16;
  floatx4 dw3[8], db3 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) dw3[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int tile0 = blockIdx.x * A.tiles_per_wg;
  float4 zq[TGX];
#pragma unroll
  for (int q = 0; q < TGX; ++q) zq[q] = cb_ld4(bz, (tile0 * 64 + 16 * q + c.j) * 512 + c.col0 * 4);
  const bool CB_LATE = c.w >= 4;
  cc_barrier();
  for (int t = 0; t < A.tiles_per_wg; ++t) {
    c.row0 = (tile0 + (A.alias ? 0 : t)) * 64;
    if (A.do_chain) {
      gfv_f16x8 wh[4], wl[4];
      for (int T = 0; T < 4; ++T) {
        wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384, 0, 0));
        wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384 + 1024, 0, 0));
      }
      floatx4 a0, a1;
      cc_mma_pair<4, 0, true>(b0, 0, wh, wl, c.lane, a0, a1);
#pragma unroll
      for (int p = 0; p < TGX / 2; ++p) {
        floatx4 n0 = a0, n1 = a1;
        float v0[4], v1[4];
        if (!CB_LATE && p + 1 < TGX / 2) cc_mma_pair<4, 0, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
        cb_hidden_bwd<false>(c, 2 * p, a0, 1.0f / 64.0f, zq[2 * p], 2.0f, b1, b2, v0);
        cb_hidden_bwd<false>(c, 2 * p + 1, a1, 1.0f / 64.0f, zq[2 * p + 1], 2.0f, b1, b2, v1);
        cb_st4(bv, (c.row0 + 32 * p + c.j) * 512 + c.col0 * 4, v0);
        cb_st4(bv, (c.row0 + 32 * p + 16 + c.j) * 512 + c.col0 * 4, v1);
        if (CB_LATE && p + 1 < TGX / 2) cc_mma_pair<4, 0, true>(b0, p + 1, wh, wl, c.lane, n0, n1);
        a0 = n0; a1 = n1;
      }
      // the next tile's rows: in flight through the barrier and the weight-gradient phase
      const int nr = (t + 1 < A.tiles_per_wg && !A.alias) ? c.row0 + 64 : c.row0;
#pragma unroll
      for (int q = 0; q < TGX; ++q) zq[q] = cb_ld4(bz, (nr + 16 * q + c.j) * 512 + c.col0 * 4);
      cc_barrier();
    }
    if (A.do_dw) {
      cb_dw_tile<0>(b0, b2, TGX / 2, c.w, c.lane, dw3, db3);
      cc_barrier();
    }
  }
  float* blk = A.sink + (size_t)blockIdx.x * 16384;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      blk[(16 * cb_dw_ntile(c.w, kt) + 4 * c.g + r) * 128 + 16 * cb_dw_ktile(c.w, kt) + c.j] = dw3[kt][r] + db3[r];
  if (c.mabs > 3.0e38f) atomicOr(status, 2);
}

// ---- form B: 4 waves x 32 columns ----
struct CtxB { int w, lane, j, g, col0; float mabs; };

// this lane's 8 values of one row (two n-tiles x 4) -> its 16 bytes of the next layer's fragments: k-group w, all eight slots
__device__ __forceinline__ void put_frag8(char* xbuf, int q, const CtxB& c, const float (&a)[8], float scale) {
  float e[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = a[k] * scale;
  gfv_uint4 hi, lo;
  gfv_split8_t<false>(e, hi, lo);
  gfv_uint4* dst = reinterpret_cast<gfv_uint4*>(xbuf + (size_t)((q * 4 + c.w) * 2) * 1024) + c.lane;
  dst[0] = hi;
  dst[64] = lo;
}
// one pair of groups against the wave's two resident n-tiles
__device__ __forceinline__ void mma_pair2(const char* xbuf, int pair, const gfv_f16x8 (&wh)[2][4], const gfv_f16x8 (&wl)[2][4], int lane,
                                          floatx4 (&a)[2][2]) {   // a[group][n]
  const gfv_f16x8* f0 = reinterpret_cast<const gfv_f16x8*>(xbuf + (size_t)(2 * pair) * 4 * 2048) + lane;
  const gfv_f16x8* f1 = f0 + 4 * 128;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int n = 0; n < 2; ++n) a[h][n] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    const gfv_f16x8 xh0 = f0[(2 * T) * 64], xh1 = f1[(2 * T) * 64];
    const gfv_f16x8 xl0 = f0[(2 * T + 1) * 64], xl1 = f1[(2 * T + 1) * 64];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      a[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[n][T], xh0, a[0][n], 0, 0, 0);
      a[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[n][T], xh1, a[1][n], 0, 0, 0);
      a[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n][T], xl0, a[0][n], 0, 0, 0);
      a[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n][T], xl1, a[1][n], 0, 0, 0);
      a[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n][T], xh0, a[0][n], 0, 0, 0);
      a[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[n][T], xh1, a[1][n], 0, 0, 0);
    }
  }
}
// GELU' epilogue of one group: both n-tiles
__device__ __forceinline__ void hidden_bwd2(CtxB& c, int q, const floatx4 (&acc)[2], float inv_in, const float4 (&z)[2], float sg, char* gout,
                                            char* aout, float (&v)[8]) {
  float a[8];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const gfv_f2 z01 = {z[n].x, z[n].y}, z23 = {z[n].z, z[n].w};
    gfv_f2 a01, a23, d01, d23;
    gfv_gelu_dgelu2(z01, a01, d01);
    gfv_gelu_dgelu2(z23, a23, d23);
    const gfv_f2 ki = gfv_splat2(inv_in);
    const gfv_f2 v01 = (gfv_f2{acc[n][0], acc[n][1]} * ki) * d01, v23 = (gfv_f2{acc[n][2], acc[n][3]} * ki) * d23;
    v[4 * n] = v01.x; v[4 * n + 1] = v01.y; v[4 * n + 2] = v23.x; v[4 * n + 3] = v23.y;
    a[4 * n] = a01.x; a[4 * n + 1] = a01.y; a[4 * n + 2] = a23.x; a[4 * n + 3] = a23.y;
  }
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k += 2) m = max3_abs(m, v[k] * sg, v[k + 1] * sg);
  c.mabs = fmaxf(c.mabs, m);
  put_frag8(gout, q, c, v, sg);
  put_frag8(aout, q, c, a, CC_SH);
}
// weight gradient: a wave owns the 4 x 4 block n-tiles 4 (w >> 1) + nn, k-tiles 4 (w & 1) + kk
__device__ __forceinline__ void dw_tile4(const char* gbuf, const char* abuf, int npairs, int w, int lane, floatx4 (&acc)[16], floatx4& accb) {
  const gfv_f16x8 ones = gfv_frag_ones<false>();
  const int nt0 = 4 * (w >> 1), kt0 = 4 * (w & 1);
  for (int pr = 0; pr < npairs; ++pr) {
    gfv_f16x8 gh[4], gl[4];
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) cb_tr_operand(gbuf, 2 * pr, nt0 + nn, lane, gh[nn], gl[nn]);
    {
      const bool odd = (w & 1) != 0;   // bias gradient of n-tile 4 (w >> 1) + 2 (w & 1) (wave-uniform select, not an index)
      const gfv_f16x8 bh = odd ? gh[2] : gh[0], bl = odd ? gl[2] : gl[0];
      accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ones, accb, 0, 0, 0);
      accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ones, accb, 0, 0, 0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      gfv_f16x8 ah, al;
      cb_tr_operand(abuf, 2 * pr, kt0 + kk, lane, ah, al);
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        acc[4 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[nn], ah, acc[4 * nn + kk], 0, 0, 0);
        acc[4 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[nn], al, acc[4 * nn + kk], 0, 0, 0);
        acc[4 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[nn], ah, acc[4 * nn + kk], 0, 0, 0);
      }
    }
  }
}

template <bool WEAVE>
__global__ __launch_bounds__(256, 1) void phase_b(const Args A, int* status) {
  __shared__ __attribute__((aligned(16))) char lds[3 * TGX * 8192];
  char* b0 = lds;
  char* b1 = lds + TGX * 8192;
  char* b2 = lds + 2 * TGX * 8192;
  CtxB c;
  c.w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 32 * c.w + 4 * c.g;
  c.mabs = 0.f;
  for (int q = 0; q < TGX; ++q) {
    const float a[8] = {0.01f * c.lane, 0.02f * c.w, 0.5f, -0.25f + q, 0.125f, -0.5f, 0.75f, 1.0f};
    put_frag8(b0, q, c, a, 64.0f);
    put_frag8(b2, q, c, a, 16.0f);
  }
  const size_t rows128 = (size_t)A.M * 512;
  const cb_rsrc bz = cb_buf(A.z, rows128), bv = cb_buf(A.v, rows128), w0 = cb_buf(A.wimg, 65536);
  floatx4 dw3[16], db3 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 16; ++k) dw3[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int tile0 = blockIdx.x * A.tiles_per_wg;
  float4 zq[TGX][2];
#pragma unroll
  for (int q = 0; q < TGX; ++q)
#pragma unroll
    for (int n = 0; n < 2; ++n) zq[q][n] = cb_ld4(bz, (tile0 * 64 + 16 * q + c.j) * 512 + (c.col0 + 16 * n) * 4);
  cc_barrier();
  for (int t = 0; t < A.tiles_per_wg; ++t) {
    const int row0 = (tile0 + (A.alias ? 0 : t)) * 64;
    if (A.do_chain) {
      gfv_f16x8 wh[2][4], wl[2][4];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          const int woff = ((2 * c.w + n) * 128 + c.lane) * 16 + T * 16384;
          wh[n][T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff, 0, 0));
          wl[n][T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + 1024, 0, 0));
        }
      floatx4 a[2][2], nx[2][2];
      mma_pair2(b0, 0, wh, wl, c.lane, a);
      // pair 1's products, then pair 0's epilogue: independent instruction streams the scheduler may (WEAVE: must) interleave
      mma_pair2(b0, 1, wh, wl, c.lane, nx);
      float v0[8], v1[8];
      hidden_bwd2(c, 0, a[0], 1.0f / 64.0f, zq[0], 2.0f, b1, b2, v0);
      hidden_bwd2(c, 1, a[1], 1.0f / 64.0f, zq[1], 2.0f, b1, b2, v1);
      if (WEAVE) {
        // 48 MFMAs of pair 1 between ~500 vector instructions of pair 0's two epilogues
#pragma unroll
        for (int k = 0; k < 48; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);   // ten VALU
        }
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        cb_st4v(bv, (row0 + c.j) * 512 + (c.col0 + 16 * n) * 4, floatx4{v0[4 * n], v0[4 * n + 1], v0[4 * n + 2], v0[4 * n + 3]});
        cb_st4v(bv, (row0 + 16 + c.j) * 512 + (c.col0 + 16 * n) * 4, floatx4{v1[4 * n], v1[4 * n + 1], v1[4 * n + 2], v1[4 * n + 3]});
      }
      hidden_bwd2(c, 2, nx[0], 1.0f / 64.0f, zq[2], 2.0f, b1, b2, v0);
      hidden_bwd2(c, 3, nx[1], 1.0f / 64.0f, zq[3], 2.0f, b1, b2, v1);
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        cb_st4v(bv, (row0 + 32 + c.j) * 512 + (c.col0 + 16 * n) * 4, floatx4{v0[4 * n], v0[4 * n + 1], v0[4 * n + 2], v0[4 * n + 3]});
        cb_st4v(bv, (row0 + 48 + c.j) * 512 + (c.col0 + 16 * n) * 4, floatx4{v1[4 * n], v1[4 * n + 1], v1[4 * n + 2], v1[4 * n + 3]});
      }
      const int nr = (t + 1 < A.tiles_per_wg && !A.alias) ? row0 + 64 : row0;
#pragma unroll
      for (int q = 0; q < TGX; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n) zq[q][n] = cb_ld4(bz, (nr + 16 * q + c.j) * 512 + (c.col0 + 16 * n) * 4);
      cc_barrier();
    }
    if (A.do_dw) {
      dw_tile4(b0, b2, TGX / 2, c.w, c.lane, dw3, db3);
      cc_barrier();
    }
  }
  float* blk = A.sink + (size_t)blockIdx.x * 16384;
#pragma unroll
  for (int k = 0; k < 16; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      blk[(16 * (4 * (c.w >> 1) + (k >> 2)) + 4 * c.g + r) * 128 + 16 * (4 * (c.w & 1) + (k & 3)) + c.j] = dw3[k][r] + db3[r];
  if (c.mabs > 3.0e38f) atomicOr(status, 2);
}

// ---- form C: 16 waves (four per SIMD): 8 column owners x 2 row pairs; <= 128 registers; a wave owns a 2 x 2 block of the weight
// gradient's output tiles (16 accumulator registers per fused gradient instead of 32) ----
__global__ __launch_bounds__(1024, 4) void phase_c(const Args A, int* status) {
  __shared__ __attribute__((aligned(16))) char lds[3 * TGX * 8192];
  char* b0 = lds;
  char* b1 = lds + TGX * 8192;
  char* b2 = lds + 2 * TGX * 8192;
  const int w16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hp = w16 >> 3;   // this wave's pair of groups
  CcCtx c;
  c.w = w16 & 7;
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 16 * c.w + 4 * c.g;
  c.M = A.M;
  c.mabs = 0.f;
  c.invw = 1.0f;
  c.ngt = TGX;
  for (int q = 2 * hp; q < 2 * hp + 2; ++q) {
    const float a[4] = {0.01f * c.lane, 0.02f * c.w, 0.5f, -0.25f + q};
    cc_put_frag<false>(b0, q, c, a, 64.0f);
    cc_put_frag<false>(b2, q, c, a, 16.0f);
  }
  const size_t rows128 = (size_t)A.M * 512;
  const cb_rsrc bz = cb_buf(A.z, rows128), bv = cb_buf(A.v, rows128), w0 = cb_buf(A.wimg, 65536);
  const int woff = (c.w * 128 + c.lane) * 16;
  floatx4 dw3[4], db3 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) dw3[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int nt0 = 2 * (w16 >> 2), kt0 = 2 * (w16 & 3);
  const int tile0 = blockIdx.x * A.tiles_per_wg;
  float4 zq[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) zq[h] = cb_ld4(bz, (tile0 * 64 + 32 * hp + 16 * h + c.j) * 512 + c.col0 * 4);
  cc_barrier();
  for (int t = 0; t < A.tiles_per_wg; ++t) {
    c.row0 = (tile0 + (A.alias ? 0 : t)) * 64;
    if (A.do_chain) {
      gfv_f16x8 wh[4], wl[4];
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384, 0, 0));
        wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384 + 1024, 0, 0));
      }
      floatx4 a0, a1;
      cc_mma_pair<4, 0, true>(b0, hp, wh, wl, c.lane, a0, a1);
      float v0[4], v1[4];
      cb_hidden_bwd<false>(c, 2 * hp, a0, 1.0f / 64.0f, zq[0], 2.0f, b1, b2, v0);
      cb_hidden_bwd<false>(c, 2 * hp + 1, a1, 1.0f / 64.0f, zq[1], 2.0f, b1, b2, v1);
      cb_st4(bv, (c.row0 + 32 * hp + c.j) * 512 + c.col0 * 4, v0);
      cb_st4(bv, (c.row0 + 32 * hp + 16 + c.j) * 512 + c.col0 * 4, v1);
      const int nr = (t + 1 < A.tiles_per_wg && !A.alias) ? c.row0 + 64 : c.row0;
#pragma unroll
      for (int h = 0; h < 2; ++h) zq[h] = cb_ld4(bz, (nr + 32 * hp + 16 * h + c.j) * 512 + c.col0 * 4);
      cc_barrier();
    }
    if (A.do_dw) {
      const gfv_f16x8 ones = gfv_frag_ones<false>();
      for (int pr = 0; pr < TGX / 2; ++pr) {
        gfv_f16x8 gh[2], gl[2], ah[2], al[2];
        cb_tr_operand(b0, 2 * pr, nt0, c.lane, gh[0], gl[0]);
        cb_tr_operand(b0, 2 * pr, nt0 + 1, c.lane, gh[1], gl[1]);
        cb_tr_operand(b2, 2 * pr, kt0, c.lane, ah[0], al[0]);
        cb_tr_operand(b2, 2 * pr, kt0 + 1, c.lane, ah[1], al[1]);
        if ((w16 & 3) < 2) {   // (wave-uniform) the bias gradient of n-tile nt0 + (w16 & 1): one wave per n-tile
          const bool odd = (w16 & 1) != 0;
          const gfv_f16x8 bh = odd ? gh[1] : gh[0], bl = odd ? gl[1] : gl[0];
          if ((w16 & 3) == (odd ? 1 : 0)) {
            db3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ones, db3, 0, 0, 0);
            db3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ones, db3, 0, 0, 0);
          }
        }
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            dw3[2 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[nn], ah[kk], dw3[2 * nn + kk], 0, 0, 0);
            dw3[2 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[nn], al[kk], dw3[2 * nn + kk], 0, 0, 0);
            dw3[2 * nn + kk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[nn], ah[kk], dw3[2 * nn + kk], 0, 0, 0);
          }
      }
      cc_barrier();
    }
  }
  float* blk = A.sink + (size_t)blockIdx.x * 16384;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) blk[(16 * (nt0 + (k >> 1)) + 4 * c.g + r) * 128 + 16 * (kt0 + (k & 1)) + c.j] = dw3[k][r] + db3[r];
  if (c.mabs > 3.0e38f) atomicOr(status, 2);
}

// ---- form D: 16 waves of TWO KINDS on 32-row tiles: waves 0 .. 7 run the chain phase of tile t (8 x 16 columns, one pair of groups)
// while waves 8 .. 15 run the weight-gradient phase of tile t - 1 out of the fragments the chain waves left in LDS (two generations of
// buffers); one barrier per tile.  Every SIMD then holds two chain waves and two weight-gradient waves: different phase kinds side by
// side, which is what the lockstep forms cannot have.  <= 128 registers (chain waves carry no accumulators, weight-gradient waves no
// chain state) ----
__global__ __launch_bounds__(1024, 4) void phase_d(const Args A, int* status) {
  __shared__ __attribute__((aligned(16))) char lds[5 * 2 * 8192];   // input pair | g generation 0, 1 | a generation 0, 1
  char* bin = lds;
  const int w16 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool dwk = w16 >= 8;   // (wave-uniform) kind
  CcCtx c;
  c.w = w16 & 7;
  c.lane = threadIdx.x & 63;
  c.j = c.lane & 15;
  c.g = c.lane >> 4;
  c.col0 = 16 * c.w + 4 * c.g;
  c.M = A.M;
  c.mabs = 0.f;
  c.invw = 1.0f;
  c.ngt = 2;
  if (!dwk) {
    for (int q = 0; q < 2; ++q) {
      const float a[4] = {0.01f * c.lane, 0.02f * c.w, 0.5f, -0.25f + q};
      cc_put_frag<false>(bin, q, c, a, 64.0f);
      for (int gen = 0; gen < 2; ++gen) {
        cc_put_frag<false>(lds + (1 + gen) * 16384, q, c, a, 64.0f);
        cc_put_frag<false>(lds + (3 + gen) * 16384, q, c, a, 16.0f);
      }
    }
  }
  const size_t rows128 = (size_t)A.M * 512;
  const cb_rsrc bz = cb_buf(A.z, rows128), bv = cb_buf(A.v, rows128), w0 = cb_buf(A.wimg, 65536);
  const int woff = (c.w * 128 + c.lane) * 16;
  floatx4 dw3[8], db3 = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) dw3[k] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int ntile = 2 * A.tiles_per_wg;   // 32-row tiles
  const int tile0 = blockIdx.x * ntile;
  float4 zq[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) zq[h] = cb_ld4(bz, (tile0 * 32 + 16 * h + c.j) * 512 + c.col0 * 4);
  cc_barrier();
  for (int t = 0; t <= ntile; ++t) {
    char* gcur = lds + (1 + (t & 1)) * 16384;
    char* acur = lds + (3 + (t & 1)) * 16384;
    const char* gprev = lds + (1 + ((t & 1) ^ 1)) * 16384;
    const char* aprev = lds + (3 + ((t & 1) ^ 1)) * 16384;
    if (!dwk) {
      if (A.do_chain && t < ntile) {
        c.row0 = (tile0 + (A.alias ? 0 : t)) * 32;
        gfv_f16x8 wh[4], wl[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) {
          wh[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384, 0, 0));
          wl[T] = __builtin_bit_cast(gfv_f16x8, __builtin_amdgcn_raw_buffer_load_b128(w0, woff + T * 16384 + 1024, 0, 0));
        }
        floatx4 a0, a1;
        cc_mma_pair<4, 0, true>(bin, 0, wh, wl, c.lane, a0, a1);
        float v0[4], v1[4];
        cb_hidden_bwd<false>(c, 0, a0, 1.0f / 64.0f, zq[0], 2.0f, gcur, acur, v0);
        cb_hidden_bwd<false>(c, 1, a1, 1.0f / 64.0f, zq[1], 2.0f, gcur, acur, v1);
        cb_st4(bv, (c.row0 + c.j) * 512 + c.col0 * 4, v0);
        cb_st4(bv, (c.row0 + 16 + c.j) * 512 + c.col0 * 4, v1);
        const int nr = (t + 1 < ntile && !A.alias) ? c.row0 + 32 : c.row0;
#pragma unroll
        for (int h = 0; h < 2; ++h) zq[h] = cb_ld4(bz, (nr + 16 * h + c.j) * 512 + c.col0 * 4);
      }
    } else if (A.do_dw && t > 0) {
      cb_dw_tile<0>(gprev, aprev, 1, c.w, c.lane, dw3, db3);
    }
    cc_barrier();
  }
  if (dwk) {
    float* blk = A.sink + (size_t)blockIdx.x * 16384;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        blk[(16 * cb_dw_ntile(c.w, kt) + 4 * c.g + r) * 128 + 16 * cb_dw_ktile(c.w, kt) + c.j] = dw3[kt][r] + db3[r];
  }
  if (c.mabs > 3.0e38f) atomicOr(status, 2);
}

}  // namespace

int main() {
  int dev = 0;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, dev));
  const int nwg = pr.multiProcessorCount, tiles = 40;
  const int M = nwg * tiles * 64;
  printf("device %s, %d CUs, clock %d kHz; %d workgroups x %d tiles of 64 rows (M = %d)\n", pr.name, nwg, pr.clockRate, nwg, tiles, M);
  float *z, *v, *sink;
  void* wimg;
  int* status;
  CK(hipMalloc(&z, (size_t)M * 512));
  CK(hipMalloc(&v, (size_t)M * 512));
  CK(hipMalloc(&sink, (size_t)nwg * 65536));
  CK(hipMalloc(&wimg, 65536));
  CK(hipMalloc(&status, 4));
  CK(hipMemset(status, 0, 4));
  {
    std::vector<float> h((size_t)M * 128);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) * 0.002f - 2.0f;
    CK(hipMemcpy(z, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> w(32768);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (unsigned short)(0x2c00 + (i * 40503u) % 1024);   // fp16 values around 2^-4
    CK(hipMemcpy(wimg, w.data(), 65536, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto attrs = [&](const void* f, const char* name) {
    hipFuncAttributes a;
    CK(hipFuncGetAttributes(&a, f));
    printf("  %-10s %3d registers, %6zu B scratch, %6zu B LDS\n", name, a.numRegs, (size_t)a.localSizeBytes, (size_t)a.sharedSizeBytes);
  };
  attrs((const void*)phase_a, "A");
  attrs((const void*)phase_b<false>, "B");
  attrs((const void*)phase_b<true>, "Bw");
  attrs((const void*)phase_c, "C");
  attrs((const void*)phase_d, "D");
  const char* kinds[3] = {"CHAIN", "DW", "CHAIN+DW"};
  for (int alias = 0; alias < 2; ++alias)
  for (int kind = 0; kind < 3; ++kind) {
    if (alias && kind == 1) continue;
    Args a{z, v, wimg, sink, M, tiles, kind != 1, kind != 0, alias};
    for (int form = 0; form < 5; ++form) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (form == 0) hipLaunchKernelGGL(phase_a, dim3(nwg), dim3(512), 0, 0, a, status);
        else if (form == 1) hipLaunchKernelGGL(phase_b<false>, dim3(nwg), dim3(256), 0, 0, a, status);
        else if (form == 2) hipLaunchKernelGGL(phase_b<true>, dim3(nwg), dim3(256), 0, 0, a, status);
        else if (form == 3) hipLaunchKernelGGL(phase_c, dim3(nwg), dim3(1024), 0, 0, a, status);
        else hipLaunchKernelGGL(phase_d, dim3(nwg), dim3(1024), 0, 0, a, status);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double cyc = (double)best * 1e-3 * (double)pr.clockRate * 1e3 / tiles;
      printf("%-9s %s form %-2s : %8.1f us per launch, %7.0f cycles per tile\n", kinds[kind], alias ? "(rows in cache)" : "(rows from HBM)", form == 0 ? "A" : (form == 1 ? "B" : (form == 2 ? "Bw" : (form == 3 ? "C" : "D"))), best * 1e3, cyc);
    }
  }
  return 0;
}
