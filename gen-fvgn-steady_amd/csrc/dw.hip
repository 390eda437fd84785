// Weight gradients of Linear layers on fp32 MFMA (gfx950):  dW[n,k] = sum_m G[m,n] * A[m,k],  db[n] = sum_m G[m,n].
// Contract: include/gfv.h (gfv_dw_multi / gfv_linear_dw).
//
// A launch carries up to 6 independent 128 x (<=128) output tiles (all weight gradients of one fused MLP: the
// input segments of layer 1, layer 2, layer 3) so small row counts still fill the chip; blockIdx.y selects the tile,
// blockIdx.x a slab of rows.  Both operands have the contraction index m as their row index in memory, so slabs of
// 32 rows of G and A are staged row-major in LDS ([m][128], row stride 144 floats: the four k-groups of an MFMA
// operand fetch land on disjoint bank quarters), double buffered, and every MFMA operand is one ds_read_b32 per
// lane.  The 4 waves own the 64x64 quadrants of the tile (16 accumulator tiles each).  Slab partials are written
// to a workspace laid out like the parameter block ([slab][W1|b1|W2|b2|W3|b3]) and summed in a fixed order by
// gfv_reduce_partials: no float atomics, deterministic.
#include <stdlib.h>
#include "gfv_common.h"
#include "gfv_prof.h"
#include "gfv_split.h"
#include "../../include/gfv.h"

__device__ int g_gfv_status_flags = 0;

extern "C" int gfv_status_flags(int32_t* flags_out) {
  int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gfv_status_flags), sizeof(int)) != hipSuccess) return GFV_ERR_LAUNCH;
  if (v) {
    const int zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gfv_status_flags), &zero, sizeof(int)) != hipSuccess) return GFV_ERR_LAUNCH;
  }
  if (flags_out) *flags_out = v;
  return GFV_OK;
}

// device address of the status word for kernels of other translation units (the chain families raise their range flags there)
int* gfv_internal_status_ptr() {
  static int* p = nullptr;
  if (!p) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_gfv_status_flags)) == hipSuccess) p = static_cast<int*>(q);
  }
  return p;
}

// Asynchronous mirror of the status word (include/gfv.h gfv_status_mirror): one pinned, device-mapped int32 per process.  Kernels
// that end a step copy a non-zero status word into it (a plain system-scope store: the host only ever clears it after it has read a
// non-zero value and cleared the device word, so nothing is lost to a race - a flag raised in between is published by the next step).
namespace {
int32_t* g_status_host = nullptr;       // host address
int32_t* g_status_host_dev = nullptr;   // the same word as the device addresses it
__global__ void status_publish_kernel(const int* __restrict__ dev_word, int* host_word) {
  const int v = *reinterpret_cast<const volatile int*>(dev_word);
  if (v) __hip_atomic_store(host_word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
int32_t* gfv_internal_status_mirror() { return g_status_host_dev; }   // nullptr until a host asked for the mirror
extern "C" int gfv_status_mirror(int32_t** host_word) {
  if (!host_word) return GFV_ERR_ARG;
  if (!g_status_host) {
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, sizeof(int32_t), hipHostMallocMapped) != hipSuccess) return GFV_ERR_LAUNCH;
    *static_cast<int32_t*>(h) = 0;
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      hipHostFree(h);
      return GFV_ERR_LAUNCH;
    }
    g_status_host = static_cast<int32_t*>(h);
    g_status_host_dev = static_cast<int32_t*>(d);
  }
  *host_word = g_status_host;
  return GFV_OK;
}
extern "C" int gfv_status_publish(void* stream) {
  int* dev = gfv_internal_status_ptr();
  if (!dev || !g_status_host_dev) return GFV_ERR_ARG;   // (gfv_status_mirror first)
  GFV_LAUNCH(status_publish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const int*)dev, (int*)g_status_host_dev);
  GFV_CHECK_LAUNCH();
  return GFV_OK;
}

namespace {

constexpr int SUB = 32;   // rows per staged sub-tile
constexpr int LDT = 144;

struct DwLaunch {
  gfv_dw_tile_t tile[6];
  int ntiles;
  int M;
  int rows_per_slab;  // multiple of 32
  long ws_stride;     // floats per slab in the workspace
  float* ws;
  int lowp;           // split form: 1 / 2 = reduced precision (gfv_set_f16split(2) / (3)): the hi x hi products only, fp16 / bf16 operands
  float ln_inv_n, ln_npad;   // LayerNorm width of a_op = 2 (gfv_set_hidden_size): 1 / h, 128 - h
};

__device__ __forceinline__ float4 ld4(const float* base, size_t row, int ld, int col, int width, bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* p = base + row * (size_t)ld + col;
  if (vec) {
    if (col < width) v = *reinterpret_cast<const float4*>(p);
  } else {
    if (col + 0 < width) v.x = p[0];
    if (col + 1 < width) v.y = p[1];
    if (col + 2 < width) v.z = p[2];
    if (col + 3 < width) v.w = p[3];
  }
  return v;
}

template <bool FULL>
__device__ __forceinline__ void dw_body(const DwLaunch& A, const gfv_dw_tile_t& Tin, float* lds) {
  gfv_dw_tile_t T = Tin;
  T.a_op &= 7;   // (GFV_DW_COLSCALE concerns the split-fp16 form only)
  float* Gs0 = lds;
  float* As0 = lds + SUB * LDT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nl = lane & 15, q = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int slab = blockIdx.x;
  const int kpad = FULL ? 128 : ((T.width + 15) & ~15);
  const int npad = FULL ? 128 : ((T.n_out + 15) & ~15);
  const bool gvec = FULL || (((T.ldg & 3) == 0) && ((T.n_out & 3) == 0));
  const bool avec = FULL || (((T.ld & 3) == 0) && ((T.width & 3) == 0));
  const int c4 = tid & 31, rg = tid >> 5;  // staging: row group 0..7, float4 column
  const int col = 4 * c4;
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (T.a_op == 2) {
    gam = *reinterpret_cast<const float4*>(T.a_gamma + col);
    bet = *reinterpret_cast<const float4*>(T.a_beta + col);
  }

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);

  const int m_beg = slab * A.rows_per_slab;
  const int m_end = min(m_beg + A.rows_per_slab, A.M);
  float4 greg[4], areg[4];

  // loads only (rows clamped, no use of the data): the global latency overlaps the MFMAs of the current sub-tile;
  // element ops (in_add, GELU / LayerNorm of the saved pre-activation, dead-row zeroing) happen at LDS-store time
  float4 breg[4];
  // gather indices run one sub-tile ahead of the row loads, so the row loads never wait for an index round trip
  int nidx[4];
  auto load_idx = [&](int m0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(m0 + rg + 8 * p, m_end - 1);
      nidx[p] = T.idx ? T.idx[m] : m;
    }
  };
  auto load_sub = [&](int m0) {
    size_t srow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) srow[p] = (size_t)nidx[p];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(m0 + rg + 8 * p, m_end - 1);
      greg[p] = ld4(T.G, (size_t)m, T.ldg, col, FULL ? 128 : T.n_out, gvec);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      areg[p] = ld4(T.A, srow[p], T.ld, col, FULL ? 128 : T.width, avec);
      if (T.in_add) breg[p] = ld4(T.in_add, srow[p], T.ld, col, FULL ? 128 : T.width, avec);
    }
    load_idx(m0 + SUB);
  };
  auto store_sub = [&](int buf, int m0) {
    float* Gs = Gs0 + buf * 2 * SUB * LDT;
    float* As = As0 + buf * 2 * SUB * LDT;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool live = (m0 + rg + 8 * p) < m_end;
      float4 g = greg[p], a = areg[p];
      if (T.in_add) { a.x += breg[p].x; a.y += breg[p].y; a.z += breg[p].z; a.w += breg[p].w; }
      if (T.a_op == 1) {
        a = make_float4(gfv_gelu(a.x), gfv_gelu(a.y), gfv_gelu(a.z), gfv_gelu(a.w));
      } else if (T.a_op == 2) {
        // row LayerNorm (eps 1e-5); the 32 lanes that share a row are an aligned half wave
        const float mean = gfv_half_sum((a.x + a.y) + (a.z + a.w)) * A.ln_inv_n;
        const float dx = a.x - mean, dy = a.y - mean, dz = a.z - mean, dw = a.w - mean;
        const float var = (gfv_half_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) - A.ln_npad * (mean * mean)) * A.ln_inv_n;
        const float rstd = rsqrtf(var + 1e-5f);
        a = make_float4(dx * rstd * gam.x + bet.x, dy * rstd * gam.y + bet.y, dz * rstd * gam.z + bet.z,
                        dw * rstd * gam.w + bet.w);
      }
      if (!live) { g = zero; a = zero; }
      *reinterpret_cast<float4*>(&Gs[(rg + 8 * p) * LDT + col]) = g;
      *reinterpret_cast<float4*>(&As[(rg + 8 * p) * LDT + col]) = a;
      dbacc.x += g.x; dbacc.y += g.y; dbacc.z += g.z; dbacc.w += g.w;
    }
  };

  int buf = 0;
  if (m_beg < m_end) {
    load_idx(m_beg);
    load_sub(m_beg);
    store_sub(0, m_beg);
  }
  __syncthreads();
  for (int m0 = m_beg; m0 < m_end; m0 += SUB) {
    const bool more = m0 + SUB < m_end;
    if (more) load_sub(m0 + SUB);
    const float* Gs = Gs0 + buf * 2 * SUB * LDT;
    const float* As = As0 + buf * 2 * SUB * LDT;
#pragma unroll
    for (int ms = 0; ms < SUB / 4; ++ms) {
      const float* grow = &Gs[(4 * ms + q) * LDT + 64 * wn + nl];
      const float* arow = &As[(4 * ms + q) * LDT + 64 * wk + nl];
      float g[4], a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = grow[16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = arow[16 * j];
      if (FULL) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[i], a[j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (64 * wn + 16 * i < npad && 64 * wk + 16 * j < kpad)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[i], a[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_sub(buf ^ 1, m0 + SUB);
    __syncthreads();
    buf ^= 1;
  }

  float* ws = A.ws + (size_t)slab * A.ws_stride + T.out_off;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int n = 64 * wn + 16 * i + 4 * q + reg;
        const int k = 64 * wk + 16 * j + nl;
        if (FULL || (n < T.n_out && k < T.width)) ws[(size_t)n * T.ld_out + k] = acc[i][j][reg];
      }

  if (T.db_off >= 0) {
    float* red = lds;  // safe: the loop ended with a barrier
    *reinterpret_cast<float4*>(&red[rg * LDT + col]) = dbacc;
    __syncthreads();
    if (tid < T.n_out) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r * LDT + tid];
      A.ws[(size_t)slab * A.ws_stride + T.db_off + tid] = s;
    }
  }
}


// ---- split-fp16 form (GFV_F16SPLIT, default) -------------------------------------------------------------------------
// Same tiling, products on the f16 MFMA pipe (v_mfma_f32_16x16x32_f16: one MFMA contracts a whole 32-row sub-tile):
// G and A are split into (hi, lo) fp16 parts while they are staged, dW += G_lo^T A_hi + G_hi^T A_lo + G_hi^T A_hi with
// fp32 accumulation - 3 MFMAs of 16 cycles per 16x16 output tile and sub-tile instead of 8 of 32.  The contraction
// index is the row m, so a scale must be constant over the rows of a sub-tile: the gradient rows are scaled by ONE exact
// power of two per slab: the minimum of the per-16-row scales the chain launch that produced G left behind (tile.gscale;
// no extra pass), or, for gradient rows that come from elsewhere, from max|G| over a first pass over the slab's rows.
// On the activation side the free index is the COLUMN: with GFV_DW_COLSCALE a tile of raw inputs gets one exact power of two
// per column and slab from a pass over the slab's A rows (re-read from L2 right after) - the encoders' first layers, whose
// inputs hold geometric columns at mesh-spacing scale (1e-2 ... 1e-4) next to O(1) features.  Without it (latent rows:
// LayerNorm outputs and sums of them; GELU / LayerNorm outputs of a_op 1 / 2: O(1) by construction) the activations are split
// unscaled, and a value beyond the fp16 range raises GFV_FLAG_DW_RANGE instead of being clamped.  (The pass costs one more
// read of A: 2.5 % of the step when applied to every raw-input tile, profiles/tools/ab.sh - hence per tile.)
// LDS image per operand and sub-tile: [column tile 8][part 2] blocks of 64 lanes x 16 B in MFMA-fragment order (lane
// (i, g) holds rows m = 8g..8g+7 of column 16 ct + i); column i of tile ct sits in lane slot i ^ (ct & 3), which spreads
// the staging writes (a thread owns 4 consecutive rows of 4 columns: one 8-B piece per column and part) over the banks.
constexpr int HBLK = 1024;                 // bytes per (column tile, part) block
constexpr int HOP = 16 * HBLK;             // bytes per operand and sub-tile
constexpr int HBUF = 2 * HOP;              // bytes per buffer (G, A)
#ifdef GFV_DW_SINGLEBUF
constexpr int HSTAGE = HBUF;               // experiment build: one staging buffer
#else
constexpr int HSTAGE = 2 * HBUF;           // two staging buffers
#endif

template <bool FULL, bool BF>   // BF: the bf16 single-product form (its own instantiation: the default form's registers stay as they are)
__device__ __forceinline__ void dw_body_h(const DwLaunch& A, const gfv_dw_tile_t& Tin, unsigned char* lds) {
  gfv_dw_tile_t T = Tin;
  const bool colscale = (T.a_op & GFV_DW_COLSCALE) != 0;
  T.a_op &= 7;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nl = lane & 15, q = lane >> 4;
  const int wn = wave >> 1, wk = wave & 1;
  const int slab = blockIdx.x;
  const int kpad = FULL ? 128 : ((T.width + 15) & ~15);
  const int npad = FULL ? 128 : ((T.n_out + 15) & ~15);
  const bool gvec = FULL || (((T.ldg & 3) == 0) && ((T.n_out & 3) == 0));
  const bool avec = FULL || (((T.ld & 3) == 0) && ((T.width & 3) == 0));
  const int c4 = tid & 31, r4 = tid >> 5;  // staging: float4 column, group of 4 consecutive rows (0..7)
  const int col = 4 * c4;
  float4 gam = make_float4(1.f, 1.f, 1.f, 1.f), bet = make_float4(0.f, 0.f, 0.f, 0.f);
  if (T.a_op == 2) {
    gam = *reinterpret_cast<const float4*>(T.a_gamma + col);
    bet = *reinterpret_cast<const float4*>(T.a_beta + col);
  }
  const int m_beg = slab * A.rows_per_slab;
  const int m_end = min(m_beg + A.rows_per_slab, A.M);

  // ---- slab scale of the gradient rows ----
  float* wred = reinterpret_cast<float*>(lds);
  float sg;
  if (T.gscale) {
    // the producer left one power of two per group of 16 rows: the slab takes the smallest (slabs start at multiples of 32)
    float smin = 8.5070592e37f;
    for (int q = (m_beg >> 4) + tid; q < ((m_end + 15) >> 4); q += 256) smin = fminf(smin, T.gscale[q]);
    smin = gfv_wave_min(smin);
    if (lane == 0) wred[wave] = smin;
    __syncthreads();
    sg = fminf(fminf(wred[0], wred[1]), fminf(wred[2], wred[3]));
    __syncthreads();
  } else {
    float gm = 0.f;
    // 8 independent row loads in flight per thread (rows clamped to the slab: a repeated row does not change a maximum)
    for (int m = m_beg + r4; m < m_end; m += 64) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(T.G, (size_t)min(m + 8 * u, m_end - 1), T.ldg, col, FULL ? 128 : T.n_out, gvec);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        gm = fmaxf(fmaxf(gm, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    gm = gfv_wave_max(gm);
    if (lane == 0) wred[wave] = gm;
    __syncthreads();
    sg = gfv_pow2_scale(fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3])));
    __syncthreads();
  }

  // ---- column scales of raw-input activations ----
  float4 sa = make_float4(1.f, 1.f, 1.f, 1.f);   // this thread's 4 staging columns
  if (colscale && m_beg < m_end) {
    float4 cm = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int m = m_beg + r4; m < m_end; m += 64) {
      float4 v[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int mm = min(m + 8 * u, m_end - 1);
        const size_t row = T.idx ? (size_t)T.idx[mm] : (size_t)mm;
        v[u] = ld4(T.A, row, T.ld, col, FULL ? 128 : T.width, avec);
        b[u] = T.in_add ? ld4(T.in_add, row, T.ld, col, FULL ? 128 : T.width, avec) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        cm.x = fmaxf(cm.x, fabsf(v[u].x + b[u].x)); cm.y = fmaxf(cm.y, fabsf(v[u].y + b[u].y));
        cm.z = fmaxf(cm.z, fabsf(v[u].z + b[u].z)); cm.w = fmaxf(cm.w, fabsf(v[u].w + b[u].w));
      }
    }
    *reinterpret_cast<float4*>(&wred[r4 * 128 + col]) = cm;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 o = *reinterpret_cast<const float4*>(&wred[r * 128 + col]);
      cm.x = fmaxf(cm.x, o.x); cm.y = fmaxf(cm.y, o.y); cm.z = fmaxf(cm.z, o.z); cm.w = fmaxf(cm.w, o.w);
    }
    sa = make_float4(gfv_pow2_scale(cm.x), gfv_pow2_scale(cm.y), gfv_pow2_scale(cm.z), gfv_pow2_scale(cm.w));
    __syncthreads();
  }
  // the inverse column scales wait in LDS (behind the staging buffers) for the output stage
  float* inv_sa = reinterpret_cast<float*>(lds + HSTAGE);
  if (r4 == 0) *reinterpret_cast<float4*>(&inv_sa[col]) = make_float4(1.0f / sa.x, 1.0f / sa.y, 1.0f / sa.z, 1.0f / sa.w);
  float amax = 0.f;   // range check of what is split (unscaled activations; scaled ones are < 2^14 by construction)

  floatx4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 greg[4], areg[4], breg[4];
  int nidx[4];
  auto load_idx = [&](int m0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(m0 + 4 * r4 + p, m_end - 1);
      nidx[p] = T.idx ? T.idx[m] : m;
    }
  };
  auto load_sub = [&](int m0) {
    size_t srow[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) srow[p] = (size_t)nidx[p];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int m = min(m0 + 4 * r4 + p, m_end - 1);
      greg[p] = ld4(T.G, (size_t)m, T.ldg, col, FULL ? 128 : T.n_out, gvec);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      areg[p] = ld4(T.A, srow[p], T.ld, col, FULL ? 128 : T.width, avec);
      if (T.in_add) breg[p] = ld4(T.in_add, srow[p], T.ld, col, FULL ? 128 : T.width, avec);
    }
    load_idx(m0 + SUB);
  };
  // this thread's 8-B slot inside a (column tile, part) block: lane (i = column & 15, g = r4 >> 1), rows e = 4 (r4 & 1)..+3
  const int slot = ((r4 >> 1) * 16) * 16 + 8 * (r4 & 1);
  auto put4 = [&](unsigned char* op, int c, float v0, float v1, float v2, float v3) {
    unsigned h0, h1, l0, l1;
    gfv_split_pair_t<BF>(v0, v1, h0, l0);   // (BF: high parts in bf16, no low parts)
    gfv_split_pair_t<BF>(v2, v3, h1, l1);
    unsigned char* b = op + ((c >> 4) * 2) * HBLK + ((c & 15) ^ ((c >> 4) & 3)) * 16 + slot;
    *reinterpret_cast<uint2*>(b) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(b + HBLK) = make_uint2(l0, l1);
  };
  auto store_sub = [&](int buf, int m0) {
    unsigned char* Gs = lds + buf * HBUF;
    unsigned char* As = Gs + HOP;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g[4], a[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const bool live = (m0 + 4 * r4 + p) < m_end;
      g[p] = greg[p];
      a[p] = areg[p];
      if (T.in_add) { a[p].x += breg[p].x; a[p].y += breg[p].y; a[p].z += breg[p].z; a[p].w += breg[p].w; }
      if (T.a_op == 1) {
        a[p] = make_float4(gfv_gelu(a[p].x), gfv_gelu(a[p].y), gfv_gelu(a[p].z), gfv_gelu(a[p].w));
      } else if (T.a_op == 2) {
        const float mean = gfv_half_sum((a[p].x + a[p].y) + (a[p].z + a[p].w)) * A.ln_inv_n;
        const float dx = a[p].x - mean, dy = a[p].y - mean, dz = a[p].z - mean, dw = a[p].w - mean;
        const float var = (gfv_half_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) - A.ln_npad * (mean * mean)) * A.ln_inv_n;
        const float rstd = rsqrtf(var + 1e-5f);
        a[p] = make_float4(dx * rstd * gam.x + bet.x, dy * rstd * gam.y + bet.y, dz * rstd * gam.z + bet.z,
                           dw * rstd * gam.w + bet.w);
      }
      if (!live) { g[p] = zero; a[p] = zero; }
      dbacc.x += g[p].x; dbacc.y += g[p].y; dbacc.z += g[p].z; dbacc.w += g[p].w;
      a[p].x *= sa.x; a[p].y *= sa.y; a[p].z *= sa.z; a[p].w *= sa.w;   // (1 without GFV_DW_COLSCALE)
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(a[p].x), fabsf(a[p].y))), fmaxf(fabsf(a[p].z), fabsf(a[p].w)));
    }
    put4(Gs, col + 0, g[0].x * sg, g[1].x * sg, g[2].x * sg, g[3].x * sg);
    put4(Gs, col + 1, g[0].y * sg, g[1].y * sg, g[2].y * sg, g[3].y * sg);
    put4(Gs, col + 2, g[0].z * sg, g[1].z * sg, g[2].z * sg, g[3].z * sg);
    put4(Gs, col + 3, g[0].w * sg, g[1].w * sg, g[2].w * sg, g[3].w * sg);
    put4(As, col + 0, a[0].x, a[1].x, a[2].x, a[3].x);
    put4(As, col + 1, a[0].y, a[1].y, a[2].y, a[3].y);
    put4(As, col + 2, a[0].z, a[1].z, a[2].z, a[3].z);
    put4(As, col + 3, a[0].w, a[1].w, a[2].w, a[3].w);
  };

  int buf = 0;
  if (m_beg < m_end) {
    load_idx(m_beg);
    load_sub(m_beg);
    store_sub(0, m_beg);
  }
  __syncthreads();
  for (int m0 = m_beg; m0 < m_end; m0 += SUB) {
    const bool more = m0 + SUB < m_end;
    if (more) load_sub(m0 + SUB);
#ifdef GFV_DW_SINGLEBUF
    const unsigned char* Gs = lds;
#else
    const unsigned char* Gs = lds + buf * HBUF;
#endif
    const unsigned char* As = Gs + HOP;
    gfv_f16x8 gh[4], gl[4], ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // tile ct = 4 wn + i (4 wk + i): ct & 3 = i
      gh[i] = *reinterpret_cast<const gfv_f16x8*>(Gs + ((4 * wn + i) * 2 + 0) * HBLK + (lane ^ i) * 16);
      gl[i] = *reinterpret_cast<const gfv_f16x8*>(Gs + ((4 * wn + i) * 2 + 1) * HBLK + (lane ^ i) * 16);
      ah[i] = *reinterpret_cast<const gfv_f16x8*>(As + ((4 * wk + i) * 2 + 0) * HBLK + (lane ^ i) * 16);
      al[i] = *reinterpret_cast<const gfv_f16x8*>(As + ((4 * wk + i) * 2 + 1) * HBLK + (lane ^ i) * 16);
    }
    if constexpr (BF) {   // bf16 operands: v_mfma_f32_16x16x32_bf16
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (FULL || (64 * wn + 16 * i < npad && 64 * wk + 16 * j < kpad)) acc[i][j] = gfv_mma_hh<true>(gh[i], ah[j], acc[i][j]);
    } else {
#pragma unroll
      for (int term = 0; term < 3; ++term)
        if (term == 2 || !A.lowp)   // (uniform)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (FULL || (64 * wn + 16 * i < npad && 64 * wk + 16 * j < kpad))
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(term == 0 ? gl[i] : gh[i], term == 1 ? al[j] : ah[j],
                                                                   acc[i][j], 0, 0, 0);
    }
#ifdef GFV_DW_SINGLEBUF
    // experiment (profiles/tools/ab.sh): ONE 32 KB staging buffer - every wave holds its fragments after the reads above,
    // so the buffer may be refilled after a barrier; a second barrier publishes it
    __syncthreads();
    if (more) store_sub(0, m0 + SUB);
    __syncthreads();
#else
    if (more) store_sub(buf ^ 1, m0 + SUB);
    __syncthreads();
    buf ^= 1;
#endif
  }

  if (!(amax <= 65504.f)) atomicOr(&g_gfv_status_flags, GFV_FLAG_DW_RANGE);   // (integer atomic; also catches NaN)
  const float inv = 1.0f / sg;
  float* ws = A.ws + (size_t)slab * A.ws_stride + T.out_off;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 64 * wk + 16 * j + nl;
    const float inva = inv_sa[k];       // both scales are powers of two: undone one after the other, exactly
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int n = 64 * wn + 16 * i + 4 * q + reg;
        if (FULL || (n < T.n_out && k < T.width)) ws[(size_t)n * T.ld_out + k] = (acc[i][j][reg] * inv) * inva;
      }
  }

  if (T.db_off >= 0) {
    float* red = reinterpret_cast<float*>(lds);  // safe: the loop ended with a barrier
    *reinterpret_cast<float4*>(&red[r4 * LDT + col]) = dbacc;
    __syncthreads();
    if (tid < T.n_out) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r * LDT + tid];
      A.ws[(size_t)slab * A.ws_stride + T.db_off + tid] = s;
    }
  }
}

__global__ __launch_bounds__(256, 2) void dw_multi_kernel(const DwLaunch A) {
  __shared__ __attribute__((aligned(16))) float lds[4 * SUB * LDT];  // 2 buffers x (G, A)
  const gfv_dw_tile_t& T = A.tile[blockIdx.y];
  const bool full = (T.n_out == 128) && (T.width == 128) && ((T.ldg & 3) == 0) && ((T.ld & 3) == 0);
  if (full) dw_body<true>(A, T, lds);
  else dw_body<false>(A, T, lds);
}

template <bool BF>
__global__ __launch_bounds__(256, 2) void dw_multi_h_kernel(const DwLaunch A) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[HSTAGE + 512];   // + the inverse column scales
  static_assert(HSTAGE >= 8 * LDT * 4, "the bias-gradient fold reuses the staging buffers");
  const gfv_dw_tile_t& T = A.tile[blockIdx.y];
  const bool full = (T.n_out == 128) && (T.width == 128) && ((T.ldg & 3) == 0) && ((T.ld & 3) == 0);
  if (full) dw_body_h<true, BF>(A, T, lds);
  else dw_body_h<false, BF>(A, T, lds);
}

// Weight gradient against a NARROW raw input (the encoders' first Linear: edge_attr [E,15], node inputs [N,12]; EPD.py:92-119) -
// dW[n][k] = sum_m G[m][n] x[m][k], k < width <= 16, and db[n] = sum_m G[m][n] - as plain fp32 FMAs: the 128 x 128 MFMA tiles of
// dw_multi_h_kernel spend a 128-wide tile's time (and a pass for the column scales) on 15 columns, 60 us for 75 k rows where the
// rows are 38 MB; and these two launches END the backward on the main queue, with nothing beside them.
// One slab of rows per workgroup (the slab partition of gfv_dw_slabs), 1024 threads: thread (n = column of G, part = 0 .. 7) walks
// rows part, part + 8, ... of 256-row chunks whose input rows sit in LDS (broadcast reads) - its G values coalesced over n, eight
// in flight -, the 8 parts are added pairwise through LDS (fixed order), one partial block per slab as dw_multi_h_kernel leaves it.
__global__ __launch_bounds__(1024) void dw_narrow_kernel(const gfv_dw_tile_t T, int M, int rows_per_slab, float* __restrict__ ws,
                                                         long ws_stride) {
  __shared__ float red[4][128][17];
  __shared__ __attribute__((aligned(16))) float xs[256][16];   // a chunk of the slab's input rows (zero beyond `width`)
  const int tid = threadIdx.x, n = tid & 127, part = tid >> 7;
  const int r0 = blockIdx.x * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
  const int width = T.width;
  float acc[16], accb = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  for (int c0 = r0; c0 < r1; c0 += 256) {
    const int cn = min(256, r1 - c0);
    {
      const int rr = tid >> 2, k4 = (tid & 3) * 4;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (rr < cn && k4 + e < width) ? T.A[(size_t)(c0 + rr) * T.ld + k4 + e] : 0.f;
      *reinterpret_cast<float4*>(&xs[rr][k4]) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    // rows part, part + 8, ...: eight of this thread's G values in flight
    for (int i0 = part; i0 < cn; i0 += 64) {
      float g[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 8 * u;
        g[u] = i < cn ? T.G[(size_t)(c0 + i) * T.ldg + n] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + 8 * u, 255);     // (rows beyond cn carry g = 0)
        const float4* xr = reinterpret_cast<const float4*>(&xs[i][0]);   // the same address in every lane: a broadcast read
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 x4 = xr[q];
          acc[4 * q + 0] = __builtin_fmaf(g[u], x4.x, acc[4 * q + 0]);
          acc[4 * q + 1] = __builtin_fmaf(g[u], x4.y, acc[4 * q + 1]);
          acc[4 * q + 2] = __builtin_fmaf(g[u], x4.z, acc[4 * q + 2]);
          acc[4 * q + 3] = __builtin_fmaf(g[u], x4.w, acc[4 * q + 3]);
        }
        accb += g[u];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int s = 4; s >= 1; s >>= 1) {
    if (part >= s && part < 2 * s) {
#pragma unroll
      for (int k = 0; k < 16; ++k) red[part - s][n][k] = acc[k];
      red[part - s][n][16] = accb;
    }
    __syncthreads();
    if (part < s) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] += red[part][n][k];
      accb += red[part][n][16];
    }
    __syncthreads();
  }
  if (part == 0) {
#pragma unroll
    for (int k = 0; k < 16; ++k) red[0][n][k] = acc[k];
    red[0][n][16] = accb;
  }
  __syncthreads();
  float* blk = ws + (size_t)blockIdx.x * ws_stride;
  for (int i = tid; i < 128 * width; i += 1024) {
    const int nn = i / width, k = i - nn * width;
    blk[T.out_off + (long)nn * T.ld_out + k] = red[0][nn][k];
  }
  if (T.db_off >= 0 && tid < 128) blk[T.db_off + tid] = red[0][tid][16];
}

}  // namespace

extern "C" int gfv_reduce_partials(const float*, int32_t, int32_t, float*, int32_t, void*);

extern "C" int gfv_dw_slabs(int32_t M, int32_t ntiles, int32_t* rows_per_slab) {
  // aim at ~256 workgroups per launch, slabs of a multiple of 32 rows, at least 64 rows.  (Rounds 1 - 2: 512, one full round
  // at 2 per CU, best of 384 / 512 / 768 / 1024.  Since the dX chains accumulate two of every MLP's three weight gradients
  // themselves the edge-level launches of this kernel are one tile wide and run beside a persistent backward launch: fewer,
  // longer slabs = half the partial-sum traffic; step time with 512 / 384 / 320 / 256: 3.823 / 3.821 / 3.806 / 3.797 ms,
  // profiles/r03_ab_dw_wgs.txt)
  if (ntiles < 1) ntiles = 1;
  static int wgs = 0, wgs_small = 0;
  if (wgs == 0) {
    const char* e = getenv("GFV_DW_WGS");
    wgs = e ? atoi(e) : 0;
    if (wgs < 1) wgs = 0;   // 0: by size (below)
    // launches of < 40 000 rows (node-level MLPs): half as many, longer slabs - half the partial-sum traffic and one
    // workgroup per CU beside the dX chain of the next block (A/B on one box, profiles/tools/ab_env.sh, step time with
    // 128 / 192 / 256 / 320 / 512: 4.61 / 4.38 / 4.37 / 4.42 / 4.46 ms)
    e = getenv("GFV_DW_WGS_SMALL");
    wgs_small = e ? atoi(e) : 256;
    if (wgs_small < 1) wgs_small = 256;
  }
  // one 50 k-cell mesh (75 k edge rows): 256; batches (8 meshes: 604 k rows) keep 512 - there 256 costs 1.8 % (23.13 against 22.73 ms)
  const int wgs_big = wgs ? wgs : (M < 200000 ? 256 : 512);
  long target = (M < 40000 ? wgs_small : wgs_big) / ntiles;
  if (target < 1) target = 1;
  long rows = (M + target - 1) / target;
  rows = ((rows + 31) / 32) * 32;
  if (rows < 64) rows = 64;
  if (rows_per_slab) *rows_per_slab = (int)rows;
  return (int)((M + rows - 1) / rows);
}

extern "C" int gfv_dw_multi(const gfv_dw_tile_t* tiles, int32_t ntiles, int32_t M, int64_t block_floats, float* workspace,
                            float* grad_block, int32_t accumulate, void* stream) {
  if (ntiles < 1 || ntiles > 6 || M < 0 || block_floats <= 0) return GFV_ERR_ARG;
  if (M == 0) return GFV_OK;
  DwLaunch a;
  double fl = 0, by = 0;
  for (int i = 0; i < ntiles; ++i) {
    const gfv_dw_tile_t& t = tiles[i];
    if (t.n_out < 1 || t.n_out > 128 || t.width < 1 || t.width > 128) return GFV_ERR_ARG;
    if ((t.a_op & 7) > 2 || (t.a_op & ~(7 | GFV_DW_COLSCALE))) return GFV_ERR_ARG;
    if ((t.a_op & 7) == 2 && t.width != 128) return GFV_ERR_ARG;
    if ((t.a_op & GFV_DW_COLSCALE) && (t.a_op & 7) != 0) return GFV_ERR_ARG;   // column scales: raw inputs only
    a.tile[i] = t;
    fl += 2.0 * M * (double)t.n_out * t.width;
    by += 4.0 * M * ((double)t.n_out + t.width);
  }
  a.ntiles = ntiles;
  a.lowp = gfv_f16split_enabled() == 2 ? 1 : (gfv_f16split_enabled() == 3 ? 2 : 0);
  a.ln_inv_n = 1.0f / (float)gfv_hidden_size();
  a.ln_npad = (float)(128 - gfv_hidden_size());
  a.M = M;
  int rows = 0;
  const int slabs = gfv_dw_slabs(M, ntiles, &rows);
  a.rows_per_slab = rows;
  a.ws_stride = block_floats;
  a.ws = workspace;
  // slots of the block that no tile writes (alignment padding) keep whatever the workspace held: callers hand in a
  // zero-initialised workspace, so padding entries of the gradient block stay finite and are never read.
  void* tok = gfv_prof_begin(GFV_K_DW, fl, by + 8.0 * (double)slabs * block_floats, (hipStream_t)stream);
  static const int narrow_on = getenv("GFV_DW_NARROW") ? atoi(getenv("GFV_DW_NARROW")) : 1;
  const gfv_dw_tile_t& t0 = tiles[0];
  if (narrow_on && ntiles == 1 && t0.width <= 16 && t0.n_out == 128 && !t0.idx && !t0.in_add && (t0.a_op & 7) == 0 && t0.ld >= t0.width) {
    // a narrow raw input (the encoders' first Linear): plain fp32 FMAs whatever the product form (dw_narrow_kernel)
    GFV_LAUNCH(dw_narrow_kernel, dim3(slabs), dim3(1024), 0, (hipStream_t)stream, t0, M, rows, workspace, (long)block_floats);
  } else
  if (a.lowp == 2) GFV_LAUNCH(dw_multi_h_kernel<true>, dim3(slabs, ntiles), dim3(256), 0, (hipStream_t)stream, a);
  else if (gfv_f16split_enabled()) GFV_LAUNCH(dw_multi_h_kernel<false>, dim3(slabs, ntiles), dim3(256), 0, (hipStream_t)stream, a);
  else GFV_LAUNCH(dw_multi_kernel, dim3(slabs, ntiles), dim3(256), 0, (hipStream_t)stream, a);
  gfv_prof_end(tok, (hipStream_t)stream);
  GFV_CHECK_LAUNCH();
  if (!grad_block) return GFV_OK;  // the caller reduces the slab workspace itself (gfv_reduce_partials_2d)
  return gfv_reduce_partials(workspace, slabs, (int32_t)block_floats, grad_block, accumulate, stream);
}

extern "C" size_t gfv_dw_multi_workspace_floats(int32_t M, int32_t ntiles, int64_t block_floats) {
  return (size_t)gfv_dw_slabs(M, ntiles, nullptr) * (size_t)block_floats;
}

// ---- single Linear layer (kept for callers that hold separate dW / db tensors) ------------------------------------
extern "C" int gfv_dw_chunks(int32_t M) { return gfv_dw_slabs(M, 1, nullptr); }

extern "C" size_t gfv_linear_dw_workspace_floats(int32_t M, int32_t n_out, int32_t K) {
  return (size_t)(gfv_dw_slabs(M, 1, nullptr) + 2) * ((size_t)n_out * K + n_out + 4);
}

extern "C" int gfv_linear_dw_gs(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                                const float* in_add, int32_t a_op, const float* a_gamma, const float* a_beta, int32_t M,
                                float* dW, const float* gscale, float* db, float* workspace, int32_t accumulate,
                                void* stream) {
  if (nseg < 1 || nseg > 3 || n_out < 1 || n_out > 128 || M < 0) return GFV_ERR_ARG;
  int K = 0;
  for (int i = 0; i < nseg; ++i) K += segs[i].width;
  const long wfl = (long)n_out * K;
  const long block = ((wfl + n_out + 3) / 4) * 4;
  gfv_dw_tile_t t[3];
  int koff = 0;
  for (int i = 0; i < nseg; ++i) {
    t[i].G = G; t[i].ldg = ldg; t[i].n_out = n_out;
    t[i].A = segs[i].ptr; t[i].idx = segs[i].idx; t[i].width = segs[i].width; t[i].ld = segs[i].ld;
    t[i].in_add = (i == 0) ? in_add : nullptr;
    t[i].a_op = a_op; t[i].a_gamma = a_gamma; t[i].a_beta = a_beta;
    t[i].out_off = koff; t[i].ld_out = K;
    t[i].db_off = (i == 0 && db) ? wfl : -1;
    t[i].gscale = gscale;
    koff += segs[i].width;
  }
  const int slabs = gfv_dw_slabs(M, nseg, nullptr);
  float* tmp = workspace + (size_t)slabs * block;  // reduced block, then split into dW / db
  int rc = gfv_dw_multi(t, nseg, M, block, workspace, tmp, 0, stream);
  if (rc) return rc;
  rc = gfv_reduce_partials(tmp, 1, (int32_t)wfl, dW, accumulate, stream);
  if (rc) return rc;
  if (db) rc = gfv_reduce_partials(tmp + wfl, 1, n_out, db, accumulate, stream);
  return rc;
}

extern "C" int gfv_linear_dw_ex(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                                const float* in_add, int32_t a_op, const float* a_gamma, const float* a_beta, int32_t M,
                                float* dW, int32_t reserved, float* db, float* workspace, int32_t accumulate,
                                void* stream) {
  (void)reserved;
  return gfv_linear_dw_gs(G, ldg, n_out, segs, nseg, in_add, a_op, a_gamma, a_beta, M, dW, nullptr, db, workspace, accumulate,
                          stream);
}

extern "C" int gfv_linear_dw(const float* G, int32_t ldg, int32_t n_out, const gfv_seg_t* segs, int32_t nseg,
                             const float* in_add, int32_t a_gelu, int32_t M, float* dW, float* db, float* workspace,
                             int32_t accumulate, void* stream) {
  return gfv_linear_dw_ex(G, ldg, n_out, segs, nseg, in_add, a_gelu ? 1 : 0, nullptr, nullptr, M, dW, 0, db,
                          workspace, accumulate, stream);
}
