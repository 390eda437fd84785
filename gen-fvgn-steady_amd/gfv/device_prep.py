"""Per-mesh preprocessing on the device: the k-hop WLSQ stencil (``build_k_hop_edge_index``,
parse_to_h5.py:228-254 / Load_mesh.py:421-521) and the WLSQ moment matrices A, B (``calc_WLSQ_A_B_normal_matrix``,
Load_mesh.py:247-272; FVgrad.py:183-232) with torch tensor ops on the GPU - the two steps that dominate the host-side
mesh set-up (1.3 s + 3.4 s of ~6 s for the 50 k-cell mesh with the numpy code of gfv.meshgen, which stays the reference
implementation and the checker: tests/test_pool_gpu.py::test_device_preprocessing_matches_host).  float64 / int64
throughout, like the host code; deterministic (sorts, scans and segment differences, no atomics)."""
from __future__ import annotations

import torch


def _csr_neighbours(face_node, n_nodes):
    two = torch.cat((face_node, face_node.flip(0)), dim=1)
    two = torch.unique(two, dim=1)                          # sorted by row, then column; duplicates dropped
    counts = torch.bincount(two[0], minlength=n_nodes)
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=face_node.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr, two[1], two


def k_hop_pairs(face_node, n_nodes, k_hop):
    """Unordered node pairs (i < j) within k_hop edges of each other, columns sorted lexicographically
    (= gfv.meshgen.k_hop_pairs / np.unique(axis=1))."""
    rowptr, nbr, two = _csr_neighbours(face_node, n_nodes)
    deg = rowptr[1:] - rowptr[:-1]
    cur = two
    out = [two]
    for _ in range(1, k_hop):
        # extend every path (i ... j) by the neighbours of j
        dj = deg[cur[1]]
        src = torch.repeat_interleave(cur[0], dj)
        start = torch.repeat_interleave(rowptr[cur[1]], dj)
        within = torch.arange(src.shape[0], device=src.device) - torch.repeat_interleave(torch.cumsum(dj, 0) - dj, dj)
        cur = torch.unique(torch.stack((src, nbr[start + within])), dim=1)
        out.append(cur)
    e = torch.cat(out, dim=1)
    e = e[:, e[0] != e[1]]
    e = torch.stack((torch.minimum(e[0], e[1]), torch.maximum(e[0], e[1])))
    return torch.unique(e, dim=1)


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
    """A [N,M,M], one-way B [Ex,M,1], extra B [2,M,1] in float64 (= gfv.meshgen.wlsq_moments)."""
    comp = torch.cat((face_node_x, face_node_x.flip(0), support_edge), dim=1)
    out_idx, in_idx = comp[0], comp[1]
    d = pos[out_idx] - pos[in_idx]
    disp = taylor_displacement(d, order)                                          # [S,M]
    M = disp.shape[1]
    w = 1.0 / torch.linalg.norm(d, dim=1, keepdim=True)
    left = ((disp * w).unsqueeze(2) * disp.unsqueeze(1)).reshape(-1, M * M)
    n = int(pos.shape[0])
    perm = torch.argsort(in_idx, stable=True)
    counts = torch.bincount(in_idx, minlength=n)
    rp = torch.zeros(n + 1, dtype=torch.int64, device=pos.device)
    rp[1:] = torch.cumsum(counts, 0)
    cs = torch.zeros((M * M, left.shape[0] + 1), dtype=torch.float64, device=pos.device)
    cs[:, 1:] = torch.cumsum(left[perm].t().contiguous(), 1)
    A = (cs[:, rp[1:]] - cs[:, rp[:-1]]).t().reshape(n, M, M)
    B = (w * disp).unsqueeze(2)
    ex = face_node_x.shape[1]
    return A, B[:ex], B[2 * ex:]
