// Mesh plan handle of the C ABI ("an opaque per-mesh plan handle gfv_plan_create / destroy"): every
// index table the hot-path kernels walk, built on the device from the reference's own int64 index tensors - so that a
// caller of the C ABI needs no Python to prepare a mesh batch.  gfv/plan.py builds the same tables with torch ops; the GPU
// tests hold the two bit-identical (tests/test_plan_gpu.py).
//
// The one primitive is a STABLE counting sort by destination row: (rowptr, order) of a key vector.  Stability keeps the
// reference's summation order inside every segment (scatter_add walks its source rows in index order), so the segmented
// sums of the hot path add in the same order whichever builder made the plan.  rocPRIM's radix sort (through hipCUB) is
// stable; the row pointer is a binary search of the sorted keys.
#include <hipcub/hipcub.hpp>

#include "../../include/gfv.h"
#include "gfv_common.h"
#include "gfv_launch.h"

namespace {

struct Table {
  void* ptr = nullptr;
  int64_t count = 0;
};

}  // namespace

struct gfv_plan {
  Table t[GFV_PLAN_TABLE_COUNT];
  int64_t N = 0, E = 0, C = 0, Sigma = 0, S = 0;
};

namespace {

__global__ void iota_kernel(int32_t* v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (int32_t)i;
}

// rowptr[r] = number of sorted keys < r, r = 0 ... n_rows
__global__ void rowptr_kernel(const int32_t* __restrict__ keys, int64_t m, int32_t* __restrict__ rowptr, int64_t n_rows) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  int64_t lo = 0, hi = m;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < r) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = (int32_t)lo;
}

// keys of the three relations, narrowed to int32 (bad[0] |= 1 when an index falls outside [0, n_rows))
__global__ void keys_twoway_kernel(const int64_t* __restrict__ ei, int64_t E, int64_t N, int32_t* __restrict__ key, int32_t* bad) {
  // blocks.py:25-31: receiving node of the two-way adjacency = cat(senders, receivers)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * E) return;
  int64_t v = ei[i];   // [2, E] row-major: first all senders, then all receivers
  if (v < 0 || v >= N) { atomicOr(bad, 1); v = 0; }
  key[i] = (int32_t)v;
}

__global__ void narrow_kernel(const int64_t* __restrict__ src, int64_t n, int64_t limit, int32_t* __restrict__ dst, int32_t* bad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t v = src[i];
  if (v < 0 || v >= limit) { atomicOr(bad, 1); v = 0; }
  dst[i] = (int32_t)v;
}

// directed WLSQ stencil (FVgrad.py:264-271): pairs [fx, fx.flip(0), support]: in = cat(fx[1], fx[0], sup[1]), out = cat(fx[0], fx[1], sup[0])
__global__ void stencil_keys_kernel(const int64_t* __restrict__ fx, int64_t Ex, const int64_t* __restrict__ sup, int64_t Es, int64_t N,
                                    int32_t* __restrict__ in_idx, int32_t* __restrict__ out_idx, int32_t* bad) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t S = 2 * Ex + Es;
  if (i >= S) return;
  int64_t a, b;   // a = out (sending node), b = in (receiving node)
  if (i < Ex) { a = fx[i]; b = fx[Ex + i]; }
  else if (i < 2 * Ex) { a = fx[Ex + (i - Ex)]; b = fx[i - Ex]; }
  else { a = sup[i - 2 * Ex]; b = sup[Es + (i - 2 * Ex)]; }
  if (a < 0 || a >= N || b < 0 || b >= N) { atomicOr(bad, 1); a = b = 0; }
  out_idx[i] = (int32_t)a;
  in_idx[i] = (int32_t)b;
}

__global__ void adjacency_cols_kernel(const int32_t* __restrict__ key2, const int32_t* __restrict__ order, int64_t E,
                                      int32_t* __restrict__ col_node, int32_t* __restrict__ col_edge2) {
  // entry k of the receiver-ordered adjacency: o < E - node o's SENDER receives from its receiver through edge o (slot 2 o);
  // o >= E - the receiver of edge o - E receives from the sender (slot 2 (o - E) + 1)          (blocks.py:25-31)
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 2 * E) return;
  int32_t o = order[k];
  if (o < E) { col_node[k] = key2[E + o]; col_edge2[k] = 2 * o; }
  else { col_node[k] = key2[o - E]; col_edge2[k] = 2 * (int32_t)(o - E) + 1; }
}

__global__ void inv_deg_kernel(const int32_t* __restrict__ rowptr, int64_t N, float* __restrict__ inv_deg) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  float d = (float)(rowptr[r + 1] - rowptr[r]);
  inv_deg[r] = 1.0f / fmaxf(d, 1.0f);
}

__global__ void gather_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ order, int64_t n, int32_t* __restrict__ dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[order[i]];
}

constexpr int TPB = 256;
inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + TPB - 1) / TPB)); }

struct Builder {
  hipStream_t st;
  gfv_plan* p;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  int32_t* sorted_keys = nullptr;
  int32_t* iota = nullptr;
  int64_t cap = 0;
  bool ok = true;

  template <typename T>
  T* table(int which, int64_t count) {
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)(count > 0 ? count : 1) * sizeof(T)) != hipSuccess) { ok = false; return nullptr; }
    p->t[which].ptr = d;
    p->t[which].count = count;
    return (T*)d;
  }
  int32_t* scratch(int64_t count) {
    void* d = nullptr;
    if (hipMalloc(&d, (size_t)(count > 0 ? count : 1) * sizeof(int32_t)) != hipSuccess) { ok = false; return nullptr; }
    owned.push_back(d);
    return (int32_t*)d;
  }
  std::vector<void*> owned;

  bool reserve(int64_t m) {
    if (m <= cap) return true;
    sorted_keys = scratch(m);
    iota = scratch(m);
    if (!ok) return false;
    iota_kernel<<<grid_for(m), TPB, 0, st>>>(iota, m);
    size_t need = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, need, (const int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr,
                                       (int32_t*)nullptr, (int)m, 0, 32, st);
    if (need > tmp_bytes) {
      void* d = nullptr;
      if (hipMalloc(&d, need) != hipSuccess) { ok = false; return false; }
      owned.push_back(d);
      tmp = d;
      tmp_bytes = need;
    }
    cap = m;
    return true;
  }

  // stable CSR of `key` [m] by row: rowptr [n_rows + 1], order [m]
  bool csr(const int32_t* key, int64_t m, int64_t n_rows, int32_t* rowptr, int32_t* order) {
    if (!reserve(m)) return false;
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < n_rows) ++bits;
    if (m > 0) {
      size_t bytes = tmp_bytes;
      if (hipcub::DeviceRadixSort::SortPairs(tmp, bytes, key, sorted_keys, (const int32_t*)iota, order, (int)m, 0, bits, st) != hipSuccess) {
        ok = false;
        return false;
      }
    }
    rowptr_kernel<<<grid_for(n_rows + 1), TPB, 0, st>>>(sorted_keys, m, rowptr, n_rows);
    return true;
  }

  void release() {
    for (void* d : owned) hipFree(d);
    owned.clear();
  }
};

}  // namespace

extern "C" int gfv_plan_destroy(gfv_plan_t* plan) {
  if (plan == nullptr) return GFV_OK;
  for (auto& t : plan->t)
    if (t.ptr) hipFree(t.ptr);
  delete plan;
  return GFV_OK;
}

extern "C" int gfv_plan_create(const gfv_plan_desc_t* d, gfv_plan_t** out, void* stream_) {
  // (once per batch, synchronising, launches of its own that a command list does not note: never inside a recorded step)
  if (gfv_rec_active()) return GFV_ERR_ARG;
  if (d == nullptr || out == nullptr) return GFV_ERR_ARG;
  *out = nullptr;
  const int64_t N = d->n_nodes, E = d->n_faces, C = d->n_cells, Sg = d->n_incidences, Ex = d->n_stencil_pairs, Es = d->n_support_pairs;
  const int64_t LIM = (int64_t)1 << 30;   // every table is int32 and 2 E, 2 Ex + Es must fit
  if (N <= 0 || E < 0 || C < 0 || Sg < 0 || Ex < 0 || Es < 0 || N >= LIM || 2 * E >= LIM || C >= LIM || Sg >= LIM || 2 * Ex + Es >= LIM)
    return GFV_ERR_ARG;
  if ((E > 0 && d->edge_index == nullptr) || (Sg > 0 && (!d->cells_node || !d->cells_face || !d->cells_index)) ||
      (Ex > 0 && d->face_node_x == nullptr) || (Es > 0 && d->support_edge == nullptr))
    return GFV_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  gfv_plan* p = new gfv_plan();
  p->N = N; p->E = E; p->C = C; p->Sigma = Sg; p->S = 2 * Ex + Es;
  Builder b;
  b.st = st;
  b.p = p;
  int32_t* bad = b.scratch(1);
  if (!b.ok) { b.release(); gfv_plan_destroy(p); return GFV_ERR_LAUNCH; }
  hipMemsetAsync(bad, 0, sizeof(int32_t), st);

  // ---- two-way node adjacency, edges by sender / by receiver (blocks.py:24-31, 82-90) -------------------------------
  {
    int32_t* key2 = b.table<int32_t>(GFV_PLAN_ES, 2 * E);     // [senders | receivers]: ES = first half, ER = second
    int32_t* n_rowptr = b.table<int32_t>(GFV_PLAN_N_ROWPTR, N + 1);
    int32_t* n_col_node = b.table<int32_t>(GFV_PLAN_N_COL_NODE, 2 * E);
    int32_t* n_col_edge2 = b.table<int32_t>(GFV_PLAN_N_COL_EDGE2, 2 * E);
    float* inv_deg = b.table<float>(GFV_PLAN_INV_DEG, N);
    int32_t* s_rowptr = b.table<int32_t>(GFV_PLAN_S_ROWPTR, N + 1);
    int32_t* s_col = b.table<int32_t>(GFV_PLAN_S_COL, E);
    int32_t* r_rowptr = b.table<int32_t>(GFV_PLAN_R_ROWPTR, N + 1);
    int32_t* r_col = b.table<int32_t>(GFV_PLAN_R_COL, E);
    int32_t* order2 = b.scratch(2 * E);
    if (!b.ok) { b.release(); gfv_plan_destroy(p); return GFV_ERR_LAUNCH; }
    p->t[GFV_PLAN_ES].count = E;
    p->t[GFV_PLAN_ER].ptr = nullptr;   // alias into ES's allocation: resolved in gfv_plan_table
    p->t[GFV_PLAN_ER].count = E;
    if (E > 0) keys_twoway_kernel<<<grid_for(2 * E), TPB, 0, st>>>(d->edge_index, E, N, key2, bad);
    b.csr(key2, 2 * E, N, n_rowptr, order2);
    if (E > 0) adjacency_cols_kernel<<<grid_for(2 * E), TPB, 0, st>>>(key2, order2, E, n_col_node, n_col_edge2);
    inv_deg_kernel<<<grid_for(N), TPB, 0, st>>>(n_rowptr, N, inv_deg);
    b.csr(key2, E, N, s_rowptr, s_col);
    b.csr(key2 + E, E, N, r_rowptr, r_col);
  }

  // ---- directed WLSQ stencil by receiving and by sending node (FVgrad.py:264-271) ----------------------------------
  {
    const int64_t S = 2 * Ex + Es;
    int32_t* in_idx = b.scratch(S);
    int32_t* out_idx = b.scratch(S);
    int32_t* x_rowptr = b.table<int32_t>(GFV_PLAN_X_ROWPTR, N + 1);
    int32_t* x_order = b.table<int32_t>(GFV_PLAN_X_ORDER, S);
    int32_t* x_out = b.table<int32_t>(GFV_PLAN_X_OUT, S);
    int32_t* xo_rowptr = b.table<int32_t>(GFV_PLAN_XO_ROWPTR, N + 1);
    int32_t* xo_order = b.table<int32_t>(GFV_PLAN_XO_ORDER, S);
    int32_t* xo_in = b.table<int32_t>(GFV_PLAN_XO_IN, S);
    if (!b.ok) { b.release(); gfv_plan_destroy(p); return GFV_ERR_LAUNCH; }
    if (S > 0) stencil_keys_kernel<<<grid_for(S), TPB, 0, st>>>(d->face_node_x, Ex, d->support_edge, Es, N, in_idx, out_idx, bad);
    b.csr(in_idx, S, N, x_rowptr, x_order);
    if (S > 0) gather_kernel<<<grid_for(S), TPB, 0, st>>>(out_idx, x_order, S, x_out);
    b.csr(out_idx, S, N, xo_rowptr, xo_order);
    if (S > 0) gather_kernel<<<grid_for(S), TPB, 0, st>>>(in_idx, xo_order, S, xo_in);
  }

  // ---- (cell, face, node) incidences by cell, by face, by node (FVscheme.py:60-120) ---------------------------------
  {
    int32_t* ci = b.scratch(Sg);
    int32_t* cf = b.scratch(Sg);
    int32_t* cn = b.scratch(Sg);
    int32_t* crow = b.table<int32_t>(GFV_PLAN_CROW, C + 1);
    int32_t* korder = b.table<int32_t>(GFV_PLAN_K_ORDER, Sg);
    int32_t* kface = b.table<int32_t>(GFV_PLAN_KFACE, Sg);
    int32_t* knode = b.table<int32_t>(GFV_PLAN_KNODE, Sg);
    int32_t* kcell = b.table<int32_t>(GFV_PLAN_KCELL, Sg);
    int32_t* frow = b.table<int32_t>(GFV_PLAN_FROW, E + 1);
    int32_t* fk = b.table<int32_t>(GFV_PLAN_FK, Sg);
    int32_t* nrow = b.table<int32_t>(GFV_PLAN_NROW, N + 1);
    int32_t* ncell = b.table<int32_t>(GFV_PLAN_NCELL, Sg);
    int32_t* on = b.scratch(Sg);
    if (!b.ok) { b.release(); gfv_plan_destroy(p); return GFV_ERR_LAUNCH; }
    if (Sg > 0) {
      narrow_kernel<<<grid_for(Sg), TPB, 0, st>>>(d->cells_index, Sg, C, ci, bad);
      narrow_kernel<<<grid_for(Sg), TPB, 0, st>>>(d->cells_face, Sg, E, cf, bad);
      narrow_kernel<<<grid_for(Sg), TPB, 0, st>>>(d->cells_node, Sg, N, cn, bad);
    }
    b.csr(ci, Sg, C, crow, korder);
    if (Sg > 0) {
      gather_kernel<<<grid_for(Sg), TPB, 0, st>>>(cf, korder, Sg, kface);
      gather_kernel<<<grid_for(Sg), TPB, 0, st>>>(cn, korder, Sg, knode);
      gather_kernel<<<grid_for(Sg), TPB, 0, st>>>(ci, korder, Sg, kcell);
    }
    b.csr(kface, Sg, E, frow, fk);
    b.csr(knode, Sg, N, nrow, on);
    if (Sg > 0) gather_kernel<<<grid_for(Sg), TPB, 0, st>>>(kcell, on, Sg, ncell);
  }

  int32_t bad_host = 0;
  hipError_t e = hipMemcpyAsync(&bad_host, bad, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);   // plan creation is per-batch set-up, not the per-step path
  b.release();
  if (e != hipSuccess || hipGetLastError() != hipSuccess || !b.ok) { gfv_plan_destroy(p); return GFV_ERR_LAUNCH; }
  if (bad_host) { gfv_plan_destroy(p); return GFV_ERR_ARG; }   // an index outside its range
  *out = p;
  return GFV_OK;
}

extern "C" int gfv_plan_table(const gfv_plan_t* plan, int32_t which, const void** ptr, int64_t* count) {
  if (plan == nullptr || which < 0 || which >= GFV_PLAN_TABLE_COUNT || ptr == nullptr || count == nullptr) return GFV_ERR_ARG;
  if (which == GFV_PLAN_ER) {
    *ptr = (const int32_t*)plan->t[GFV_PLAN_ES].ptr + plan->E;
    *count = plan->E;
    return GFV_OK;
  }
  *ptr = plan->t[which].ptr;
  *count = plan->t[which].count;
  return GFV_OK;
}

extern "C" int gfv_plan_sizes(const gfv_plan_t* plan, int64_t* sizes5) {
  if (plan == nullptr || sizes5 == nullptr) return GFV_ERR_ARG;
  sizes5[0] = plan->N; sizes5[1] = plan->E; sizes5[2] = plan->C; sizes5[3] = plan->Sigma; sizes5[4] = plan->S;
  return GFV_OK;
}
