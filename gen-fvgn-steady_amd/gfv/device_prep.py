"""Per-mesh preprocessing on the device: the k-hop WLSQ stencil (``build_k_hop_edge_index``,
parse_to_h5.py:228-254 / Load_mesh.py:421-521) and the WLSQ moment matrices A, B (``calc_WLSQ_A_B_normal_matrix``,
Load_mesh.py:247-272; FVgrad.py:183-232) on the GPU, both as HIP kernels (`gfv_khop_count` / `gfv_khop_fill`,
csrc/prep.hip; `gfv_wlsq_moments`, csrc/fvm.hip) - the two steps that dominate the host-side mesh set-up (1.3 s + 3.4 s of
~6 s for the 50 k-cell mesh with the numpy code of gfv.meshgen, which stays the reference implementation and the checker:
tests/test_pool_gpu.py::test_device_preprocessing_matches_host).  float64 / int64 throughout, like the host code;
deterministic (integer atomics only where the order does not reach the result, fixed summation orders)."""
from __future__ import annotations

import torch


def k_hop_pairs(face_node, n_nodes, k_hop):
    """Unordered node pairs (i < j) within k_hop edges of each other, columns sorted lexicographically
    (= gfv.meshgen.k_hop_pairs / np.unique(axis=1)): HIP kernels (`gfv_khop_count` / `gfv_khop_fill`, csrc/prep.hip - CSR
    adjacency by integer atomics, a thread per node walks its neighbourhood breadth first and sorts its partners), one
    device-to-host read of the pair count in between."""
    from . import lib as L
    if not face_node.is_cuda:
        raise RuntimeError("gfv.device_prep runs on the GPU (gfv.meshgen has the host form)")
    lib = L.load()
    fn = face_node.to(torch.int64).contiguous()
    F, n = int(fn.shape[1]), int(n_nodes)
    ws = torch.empty(int(lib.gfv_khop_workspace_ints(n, F)), dtype=torch.int32, device=fn.device)
    L.check(lib.gfv_khop_count(fn[0].data_ptr(), fn[1].data_ptr(), F, n, int(k_hop), ws.data_ptr(), L.stream_ptr()), "gfv_khop_count")
    total, flag = (int(v) for v in ws[4 * (n + 1) - 1:4 * (n + 1) + 1].tolist())
    if flag:
        raise RuntimeError(f"a {k_hop}-hop neighbourhood of this mesh has more than 512 nodes")
    out = torch.empty((2, total), dtype=torch.int64, device=fn.device)
    if total:
        L.check(lib.gfv_khop_fill(n, int(k_hop), ws.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), L.stream_ptr()), "gfv_khop_fill")
    return out


def taylor_displacement(d, order="2nd"):
    """= gfv.meshgen.taylor_displacement (FVorder.py:23-72) with torch ops."""
    x, y = d[:, 0:1], d[:, 1:2]
    cols = [d]
    if order in ("2nd", "3rd", "4th"):
        cols += [0.5 * d ** 2, x * y]
    if order in ("3rd", "4th"):
        cols += [(1 / 6) * d ** 3, 0.5 * x ** 2 * y, 0.5 * y ** 2 * x]
    if order == "4th":
        cols += [(1 / 24) * x ** 4, (1 / 6) * x ** 3 * y, (1 / 4) * x ** 2 * y ** 2, (1 / 6) * x * y ** 3, (1 / 24) * y ** 4]
    if order not in ("1st", "2nd", "3rd", "4th"):
        raise NotImplementedError(f"{order} Order not implemented")
    return torch.cat(cols, dim=1)


def wlsq_moments(pos, face_node_x, support_edge, order="2nd"):
    """A [N,M,M], one-way B [Ex,M,1], extra B [2,M,1] in float64 (= gfv.meshgen.wlsq_moments): the directed stencil
    [fx, fx.flip(0), support_edge] in CSR order of the receiving node (one stable sort), then ONE HIP kernel
    (`gfv_wlsq_moments`, csrc/fvm.hip: a thread per node accumulates w t t^T over its entries in that order and writes
    w t of every entry back to its place in edge order) - no atomics, deterministic."""
    from . import lib as L
    M = {"1st": 2, "2nd": 5, "3rd": 9, "4th": 14}.get(order)
    if M is None:
        raise NotImplementedError(f"{order} Order not implemented")
    if not pos.is_cuda:
        raise RuntimeError("gfv.device_prep runs on the GPU (gfv.meshgen has the host form)")
    comp = torch.cat((face_node_x, face_node_x.flip(0), support_edge), dim=1)
    out_idx, in_idx = comp[0], comp[1]
    n, S = int(pos.shape[0]), int(in_idx.shape[0])
    perm = torch.argsort(in_idx, stable=True)
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=pos.device)
    rowptr[1:] = torch.cumsum(torch.bincount(in_idx, minlength=n), 0)
    rp32, out32, ent32 = rowptr.to(torch.int32), out_idx[perm].to(torch.int32).contiguous(), perm.to(torch.int32).contiguous()
    p64 = pos.to(torch.float64).contiguous()
    A = torch.empty((n, M, M), dtype=torch.float64, device=pos.device)
    B = torch.empty((S, M), dtype=torch.float64, device=pos.device)
    L.check(L.load().gfv_wlsq_moments(p64.data_ptr(), rp32.data_ptr(), out32.data_ptr(), ent32.data_ptr(), A.data_ptr(),
                                      B.data_ptr(), n, M, L.stream_ptr()), "gfv_wlsq_moments")
    B = B.unsqueeze(2)
    ex = face_node_x.shape[1]
    return A, B[:ex], B[2 * ex:]
