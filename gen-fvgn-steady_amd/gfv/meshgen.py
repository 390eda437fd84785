"""Build-owned synthetic mesh generator + per-mesh preprocessing (CPU, numpy, float64, one-off per mesh).

Produces the same mesh dictionary schema the reference hands to its graph datasets, so that the
hot path sees exactly the input contract of SURVEY.md 8(a-0):

* raw connectivity in the layout of the reference's COMSOL reader
  (``Extract_mesh/parse_comsol.py:455-500``: ``face|face_node`` = lexicographically unique node pairs with
  row0 < row1, flat ``cells_node / cells_face / cells_index`` grouped tri-block then quad-block),
* derived geometry following ``Extract_mesh/parse_to_h5.py:257-496`` (centroids, CCW ordering of the node and the
  face list of each cell - sorted independently by angle -, face types, outward unit normals, cell areas,
  cell-sharing node pairs ``face_node_x``),
* the k-hop WLSQ stencil and the 2nd-order moment matrices following ``Load_mesh/Load_mesh.py:247-272,421-521``
  and ``FVMmodel/FVdiscretization/FVgrad.py:183-232`` / ``FVorder.py:7-86``,
* PDE coefficients / non-dimensionalisation following ``Load_mesh/Load_mesh.py:134-211`` and the Dirichlet
  targets of ``Load_mesh/Load_mesh.py:80-131`` (``Set_BC.py:6-66`` for the profiles).

This is host-side, one-off preprocessing (SURVEY.md row f2/f3 "next"); the per-step hot path never calls it.
Nothing here is copied from the reference; tests/golden pins the integer outputs bit-exactly against it.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp

# NodeType (utils/utilities.py:7-13)
NORMAL, INFLOW, OUTFLOW, WALL, PRESS_POINT, IN_WALL = 0, 1, 2, 3, 4, 5


# ------------------------------------------------------------------------------------------------------------
# raw meshes
# ------------------------------------------------------------------------------------------------------------
def _elements_to_faces(elem_blocks):
    """Unique sorted node pairs + inverse map, as parse_comsol.py:427-486 builds them."""
    edges = []
    for elements in elem_blocks:
        nc, k = elements.shape
        cyc = np.stack([np.stack((elements[:, i], elements[:, (i + 1) % k]), axis=1) for i in range(k)], axis=1)
        edges.append(np.sort(cyc.reshape(-1, 2), axis=1).T)
    full = np.concatenate(edges, axis=1)
    face_node, cells_face = np.unique(full, axis=1, return_inverse=True)
    return face_node.astype(np.int64), np.asarray(cells_face).reshape(-1).astype(np.int64)


def _assemble_raw(pos, elem_blocks, node_type, extra=None):
    cells_node, cells_index, count = [], [], 0
    for el in elem_blocks:
        cells_node.append(el.reshape(-1))
        cells_index.append(np.repeat(np.arange(count, count + el.shape[0]), el.shape[1]))
        count += el.shape[0]
    face_node, cells_face = _elements_to_faces(elem_blocks)
    raw = {
        "node|pos": np.asarray(pos, dtype=np.float64),
        "node|node_type": np.asarray(node_type, dtype=np.int64),
        "face|face_node": face_node,
        "cells_node": np.concatenate(cells_node).astype(np.int64),
        "cells_index": np.concatenate(cells_index).astype(np.int64),
        "cells_face": cells_face,
    }
    if extra:
        raw.update(extra)
    return raw


def _type_nodes_from_boundary_edges(n_nodes, bnd_edges, bnd_kind, press_nodes=()):
    """Node typing with the precedence of parse_comsol.py:348-424 (inflow, then wall, then outflow, then pressure point)."""
    node_type = np.full((n_nodes,), NORMAL, dtype=np.int64)
    e = bnd_edges[bnd_kind == INFLOW]
    node_type[e[:, 0]] = INFLOW
    node_type[e[:, 1]] = INFLOW
    e = bnd_edges[bnd_kind == WALL]
    was_in_l = node_type[e[:, 0]] == INFLOW
    was_in_r = node_type[e[:, 1]] == INFLOW
    node_type[e[:, 0]] = WALL
    node_type[e[:, 1]] = WALL
    node_type[e[was_in_l, 0]] = IN_WALL
    node_type[e[was_in_r, 1]] = IN_WALL
    e = bnd_edges[bnd_kind == OUTFLOW]
    if e.size:
        wl, wr = node_type[e[:, 0]] == WALL, node_type[e[:, 1]] == WALL
        il, ir = node_type[e[:, 0]] == INFLOW, node_type[e[:, 1]] == INFLOW
        node_type[e[:, 0]] = OUTFLOW
        node_type[e[:, 1]] = OUTFLOW
        node_type[e[wl, 0]] = WALL
        node_type[e[wr, 1]] = WALL
        node_type[e[il, 0]] = INFLOW
        node_type[e[ir, 1]] = INFLOW
    for p in press_nodes:
        node_type[p] = PRESS_POINT
    return node_type


def _boundary_edges(elem_blocks):
    face_node, cells_face = _elements_to_faces(elem_blocks)
    cnt = np.bincount(cells_face, minlength=face_node.shape[1])
    return face_node[:, cnt == 1].T  # [Eb,2]


def raw_quad_cavity(n=8, jitter=0.0, seed=1234, tri_fraction=0.0):
    """Lid-driven cavity on [0,1]^2 with n x n quad cells (optionally a band of cells split into triangles).

    Lid (y=1) = INFLOW, the other three sides = WALL, pressure point at the middle of the lid
    (cf. mesh_example/lid_driven_cavity/*/BC.json)."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    pos = np.stack((X.reshape(-1), Y.reshape(-1)), axis=1)
    if jitter > 0:
        h = 1.0 / n
        inner = (pos[:, 0] > 1e-9) & (pos[:, 0] < 1 - 1e-9) & (pos[:, 1] > 1e-9) & (pos[:, 1] < 1 - 1e-9)
        pos[inner] += rng.uniform(-jitter * h, jitter * h, size=(int(inner.sum()), 2))
    nid = lambda i, j: j * (n + 1) + i
    quads, tris = [], []
    n_tri_rows = int(round(tri_fraction * n))
    for j in range(n):
        for i in range(n):
            a, b, c, d = nid(i, j), nid(i + 1, j), nid(i + 1, j + 1), nid(i, j + 1)
            if j < n_tri_rows:
                if (i + j) % 2 == 0:
                    tris += [(a, b, c), (a, c, d)]
                else:
                    tris += [(a, b, d), (b, c, d)]
            else:
                quads.append((a, b, c, d))
    blocks = []
    if tris:
        blocks.append(np.asarray(tris, dtype=np.int64))
    if quads:
        blocks.append(np.asarray(quads, dtype=np.int64))
    be = _boundary_edges(blocks)
    mid = 0.5 * (pos[be[:, 0]] + pos[be[:, 1]])
    kind = np.where(mid[:, 1] > 1 - 1e-9, INFLOW, WALL)
    press = [nid(n // 2, n)]
    node_type = _type_nodes_from_boundary_edges(pos.shape[0], be, kind, press)
    bc = {
        "stencil|khops": 2,
        "theta_PDE": {"unsteady": 1, "continuity": 1, "convection": 1, "grad_p": 1},
        "U": 1.0, "rho": 1.0, "mu": 0.01, "source": 0.0, "aoa": 0.0, "dt": 0.1, "L": 1.0,
        "sigma": [1, 1, 1], "inlet_type": "uniform",
    }
    return _assemble_raw(pos, blocks, node_type, {"bc": bc, "case_name": f"cavity_{n}x{n}"})


def raw_tri_channel_cylinder(nx=44, ny=8, jitter=0.2, seed=1234, quad_fraction=0.0):
    """Channel [0,2.2]x[0,0.41] with a cylinder hole at (0.2,0.2), r=0.05, triangulated structured grid.

    Left = INFLOW (parabolic), right = OUTFLOW, top/bottom/cylinder = WALL
    (cf. mesh_example/cylinder_flow_full_tri/BC.json).  ``quad_fraction`` keeps the rightmost part as quads
    to exercise ragged (tri + quad) cell lists."""
    rng = np.random.default_rng(seed)
    Lx, Ly, cx, cy, rad = 2.2, 0.41, 0.2, 0.2, 0.05
    xs, ys = np.linspace(0, Lx, nx + 1), np.linspace(0, Ly, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    pos = np.stack((X.reshape(-1), Y.reshape(-1)), axis=1)
    hx, hy = Lx / nx, Ly / ny
    inner = (pos[:, 0] > 1e-9) & (pos[:, 0] < Lx - 1e-9) & (pos[:, 1] > 1e-9) & (pos[:, 1] < Ly - 1e-9)
    jit = rng.uniform(-jitter, jitter, size=(int(inner.sum()), 2)) * np.array([hx, hy])
    pos[inner] += jit
    nid = lambda i, j: j * (nx + 1) + i
    tris, quads = [], []
    first_quad_col = nx - int(round(quad_fraction * nx))
    for j in range(ny):
        for i in range(nx):
            a, b, c, d = nid(i, j), nid(i + 1, j), nid(i + 1, j + 1), nid(i, j + 1)
            if i >= first_quad_col:
                quads.append((a, b, c, d))
            elif (i + j) % 2 == 0:
                tris += [(a, b, c), (a, c, d)]
            else:
                tris += [(a, b, d), (b, c, d)]
    tris = np.asarray(tris, dtype=np.int64)
    cen = pos[tris].mean(axis=1)
    keep = (cen[:, 0] - cx) ** 2 + (cen[:, 1] - cy) ** 2 > rad ** 2
    tris = tris[keep]
    blocks = [tris] + ([np.asarray(quads, dtype=np.int64)] if quads else [])
    # drop orphan nodes, renumber
    used = np.zeros(pos.shape[0], dtype=bool)
    for b in blocks:
        used[b.reshape(-1)] = True
    remap = -np.ones(pos.shape[0], dtype=np.int64)
    remap[used] = np.arange(int(used.sum()))
    pos = pos[used]
    blocks = [remap[b] for b in blocks]
    be = _boundary_edges(blocks)
    mid = 0.5 * (pos[be[:, 0]] + pos[be[:, 1]])
    kind = np.full(be.shape[0], WALL, dtype=np.int64)
    kind[mid[:, 0] < 1e-9] = INFLOW
    kind[mid[:, 0] > Lx - 1e-9] = OUTFLOW
    node_type = _type_nodes_from_boundary_edges(pos.shape[0], be, kind)
    bc = {
        "stencil|khops": 2,
        "theta_PDE": {"unsteady": 1, "continuity": 1, "convection": 1, "grad_p": 1},
        "U": 0.2, "rho": 1.0, "mu": 0.001, "source": 0.0, "aoa": 0.0, "dt": 0.5, "L": 0.1,
        "sigma": [1, 1, 1], "inlet_type": "parabolic",
    }
    return _assemble_raw(pos, blocks, node_type, {"bc": bc, "case_name": f"cylinder_tri_{nx}x{ny}"})


def raw_poisson_cavity(n=6, seed=1234):
    """Poisson problem on the unit square: sigma=[1,0,0], only the x 'momentum' residual is active."""
    raw = raw_quad_cavity(n=n, jitter=0.1, seed=seed, tri_fraction=0.5)
    raw["bc"] = {
        "stencil|khops": 2,
        "theta_PDE": {"unsteady": 0, "continuity": 0, "convection": 0, "grad_p": 0},
        "U": 5.0, "rho": 1.0, "mu": 0.1, "source": 10.0, "aoa": 0.0, "dt": 1.0, "L": 1.0,
        "sigma": [1, 0, 0], "inlet_type": "uniform",
    }
    raw["case_name"] = f"poisson_{n}x{n}"
    return raw


def cylinder_grid_for_cells(target_cells=50000):
    """Pick (nx, ny) of raw_tri_channel_cylinder so the mesh has ~target_cells triangles."""
    ny = max(4, int(round(math.sqrt(target_cells / 2.0 * 0.41 / 2.2))))
    nx = max(8, int(round((target_cells / 2.0 + 0.00785 / (2.2 * 0.41) * target_cells / 2.0) / ny)))
    return nx, ny


# ------------------------------------------------------------------------------------------------------------
# derived geometry  (parse_to_h5.py:257-496)
# ------------------------------------------------------------------------------------------------------------
def _seg_sum(values, index, n):
    out = np.zeros((n,) + values.shape[1:], dtype=values.dtype)
    np.add.at(out, index, values)
    return out


def _unique_cols(a):
    return np.unique(a, axis=1)


def _separate_domains(cells_node, cells_face, cells_index):
    """Masks per cell type in ascending type order (parse_to_h5.py:196-226)."""
    n_cells = int(cells_index.max()) + 1
    ctype = np.bincount(cells_index, minlength=n_cells)
    out = []
    for ct in np.unique(ctype):
        m = (ctype == ct)[cells_index]
        out.append((int(ct), cells_node[m], cells_face[m], cells_index[m]))
    return out


def _within_cell_pairs(ct, cells_node):
    """All node pairs sharing a cell, one-way, unique (parse_to_h5.py:132-150)."""
    orig = cells_node.copy()
    cur = cells_node.copy()
    pairs = []
    for _ in range(ct - 1):
        cur = np.roll(cur.reshape(-1, ct), 1, axis=1).reshape(-1)
        pairs.append(np.stack((orig, cur), axis=0))
    p = np.concatenate(pairs, axis=1)
    p = p[:, p[0] != p[1]]
    return _unique_cols(np.sort(p, axis=0))


def derive_geometry(raw):
    mesh = dict(raw)
    pos = raw["node|pos"]
    node_type = raw["node|node_type"]
    face_node = raw["face|face_node"]
    cells_node, cells_face, cells_index = raw["cells_node"], raw["cells_face"], raw["cells_index"]
    n_cells = int(cells_index.max()) + 1

    cnt = np.bincount(cells_index, minlength=n_cells).astype(np.float64)
    centroid = _seg_sum(pos[cells_node], cells_index, n_cells) / np.maximum(cnt, 1)[:, None]
    face_center = (pos[face_node[0]] + pos[face_node[1]]) / 2.0

    # CCW ordering, node list and face list sorted independently (parse_to_h5.py:55-110)
    new_node, new_face, new_index = [], [], []
    for ct, sn, sf, si in _separate_domains(cells_node, cells_face, cells_index):
        nc = sn.shape[0] // ct
        n2, f2 = sn.reshape(nc, ct), sf.reshape(nc, ct)
        cc = centroid[si.reshape(nc, ct)[:, 0]]
        rv = pos[n2] - cc[:, None, :]
        n2 = np.take_along_axis(n2, np.argsort(np.arctan2(rv[:, :, 1], rv[:, :, 0]), axis=1, kind="stable"), axis=1)
        rf = face_center[f2] - cc[:, None, :]
        f2 = np.take_along_axis(f2, np.argsort(np.arctan2(rf[:, :, 1], rf[:, :, 0]), axis=1, kind="stable"), axis=1)
        new_node.append(n2.reshape(-1))
        new_face.append(f2.reshape(-1))
        new_index.append(si)
    cells_node = np.concatenate(new_node)
    cells_face = np.concatenate(new_face)
    cells_index = np.concatenate(new_index)

    # face types (parse_to_h5.py:306-371), later assignment wins
    lt, rt = node_type[face_node[0]], node_type[face_node[1]]
    isb = lambda t: (t == INFLOW) | (t == WALL) | (t == OUTFLOW) | (t == PRESS_POINT) | (t == IN_WALL)
    isb_noin = lambda t: (t == WALL) | (t == IN_WALL) | (t == OUTFLOW) | (t == PRESS_POINT)
    face_type = np.full((face_node.shape[1],), NORMAL, dtype=np.int64)
    face_type[(isb(lt) & (rt == INFLOW)) | (isb(rt) & (lt == INFLOW))] = INFLOW
    face_type[(isb(lt) & (rt == WALL)) | (isb_noin(rt) & (lt == WALL))] = WALL
    face_type[(isb(lt) & (rt == OUTFLOW)) | (isb_noin(rt) & (lt == OUTFLOW))] = OUTFLOW

    d = pos[face_node[0]] - pos[face_node[1]]
    face_area = np.linalg.norm(d, axis=1, keepdims=True)

    # neighbour cells (min, max) over the incidences of each face (parse_to_h5.py:385-402)
    n_faces = face_node.shape[1]
    cmax = np.full(n_faces, -1, dtype=np.int64)
    cmin = np.full(n_faces, np.iinfo(np.int64).max, dtype=np.int64)
    np.maximum.at(cmax, cells_face, cells_index)
    np.minimum.at(cmin, cells_face, cells_index)
    neighbour_cell = np.stack((cmin, cmax), axis=0)

    # outward unit normals per (cell, face) incidence (parse_to_h5.py:405-439)
    unv = np.concatenate((-d[:, 1:2], d[:, 0:1]), axis=1)
    unv = unv / np.linalg.norm(unv, axis=1, keepdims=True)
    f2c = face_center[cells_face] - centroid[cells_index]
    cf_unv = unv[cells_face]
    outward = (np.sum(f2c * cf_unv, axis=1, keepdims=True) > 0.0)
    cf_unv = np.where(outward, cf_unv, -cf_unv)
    surf = cf_unv * face_area[cells_face]
    closure = _seg_sum(surf, cells_index, n_cells)
    if not np.allclose(closure, 0.0, rtol=1e-5, atol=1e-8):
        raise ValueError("surface vectors of a cell do not close")

    # cell areas: divergence theorem, shoelace as check/fallback (parse_to_h5.py:449-472)
    area = _seg_sum(np.sum(0.5 * face_center[cells_face] * surf, axis=1), cells_index, n_cells)
    shoelace = np.zeros(n_cells)
    for ct, sn, _, si in _separate_domains(cells_node, cells_face, cells_index):
        p = pos[sn.reshape(-1, ct)]
        x, y = p[:, :, 0], p[:, :, 1]
        shoelace[si.reshape(-1, ct)[:, 0]] = 0.5 * np.abs(
            np.sum(x * np.roll(y, 1, axis=1), axis=1) - np.sum(y * np.roll(x, 1, axis=1), axis=1))
    if not np.allclose(area, shoelace, rtol=1e-5, atol=1e-8):
        area = shoelace

    # cell-sharing node pairs (parse_to_h5.py:474-491)
    fx = [_within_cell_pairs(ct, sn) for ct, sn, _, _ in _separate_domains(cells_node, cells_face, cells_index)]
    fx = np.concatenate(fx, axis=1)
    fx = _unique_cols(fx[:, fx[0] != fx[1]])

    mesh.update({
        "cells_node": cells_node, "cells_face": cells_face, "cells_index": cells_index,
        "cell|centroid": centroid, "face|face_center_pos": face_center, "face|face_type": face_type,
        "face|face_area": face_area, "face|neighbour_cell": neighbour_cell, "unit_norm_v": cf_unv,
        "cell|cells_area": area, "face_node_x_base": fx,
    })
    return mesh


# ------------------------------------------------------------------------------------------------------------
# stencil + WLSQ moments (Load_mesh.py:421-521, 247-272; FVgrad.py:183-232; FVorder.py:7-86)
# ------------------------------------------------------------------------------------------------------------
def k_hop_pairs(face_node, n_nodes, k_hop):
    two = np.concatenate((face_node, face_node[::-1]), axis=1)
    adj = sp.csr_matrix((np.ones(two.shape[1]), (two[0], two[1])), shape=(n_nodes, n_nodes))
    adj.sum_duplicates()
    adj.data[:] = 1.0
    cur, out = adj, []
    for k in range(1, k_hop + 1):
        if k > 1:
            cur = (cur @ adj).tocsr()
        coo = cur.tocoo()
        out.append(np.stack((coo.row.astype(np.int64), coo.col.astype(np.int64)), axis=0))
    e = np.concatenate(out, axis=1)
    e = e[:, e[0] != e[1]]
    return _unique_cols(np.sort(e, axis=0))


ORDER_TERMS = {"1st": 2, "2nd": 5, "3rd": 9, "4th": 14}   # Taylor terms of the reconstruction (FVorder.py:23-72)


def taylor_displacement(d, order="2nd"):
    """[S, ORDER_TERMS[order]]: the Taylor monomials of the position difference, in the reference's column order
    (FVorder.py:23-72): (dx, dy | dx^2/2, dy^2/2, dx dy | dx^3/6, dy^3/6, dx^2 dy/2, dy^2 dx/2 | dx^4/24, dx^3 dy/6,
    dx^2 dy^2/4, dx dy^3/6, dy^4/24)."""
    if order not in ORDER_TERMS:
        raise NotImplementedError(f"{order} Order not implemented")
    x, y = d[:, 0:1], d[:, 1:2]
    cols = [d]
    if order in ("2nd", "3rd", "4th"):
        cols += [0.5 * d ** 2, x * y]
    if order in ("3rd", "4th"):
        cols += [(1 / 6) * d ** 3, 0.5 * x ** 2 * y, 0.5 * y ** 2 * x]
    if order == "4th":
        cols += [(1 / 24) * x ** 4, (1 / 6) * x ** 3 * y, (1 / 4) * x ** 2 * y ** 2, (1 / 6) * x * y ** 3, (1 / 24) * y ** 4]
    return np.concatenate(cols, axis=1)


def second_order_displacement(d):
    return taylor_displacement(d, "2nd")  # [S,5]


def wlsq_moments(pos, face_node_x, support_edge, order="2nd"):
    """A [N,M,M], one-way B [Ex,M,1], extra B [2,M,1] in float64, M = ORDER_TERMS[order] (cast to f32 by the caller,
    Load_mesh.py:264-269)."""
    comp = np.concatenate((face_node_x, face_node_x[::-1], support_edge), axis=1)
    out_idx, in_idx = comp[0], comp[1]
    d = pos[out_idx] - pos[in_idx]
    disp = taylor_displacement(d, order)
    w = 1.0 / np.linalg.norm(d, axis=1, keepdims=True)
    left = (disp * w)[:, :, None] * disp[:, None, :]
    A = _seg_sum(left, in_idx, int(in_idx.max()) + 1)
    B = (w * disp)[:, :, None]
    ex = face_node_x.shape[1]
    return A, B[:ex], B[2 * ex:]


def velocity_profile(pos, mean_u, kind):
    uv = np.zeros_like(pos)
    if pos.shape[0] == 0:
        return uv
    if kind == "parabolic":
        y = pos[:, 1] - pos[:, 1].min()
        ymax, ymin = y.max(), y.min()
        uv[:, 0] = 6 * mean_u * y * (((ymax - ymin) - y) / (ymax - ymin) ** 2)
    elif kind == "uniform":
        uv[:, 0] = mean_u
    else:
        raise ValueError(kind)
    return uv


def pde_coefficients(bc):
    """bc (sampled U, rho, mu, source, aoa, dt, L + theta_PDE switches) -> theta_PDE [1,9], dt_graph [1,1], uvp_dim [1,3]
    (Load_mesh.py:134-211: set_theta_PDE / makedimless)."""
    th = bc["theta_PDE"]
    Uin, rho, mu = float(bc["U"]), float(bc["rho"]), float(bc["mu"])
    aoa = float(bc["aoa"])
    Re = rho * Uin * float(bc["L"]) / mu if mu != 0 else 0.0
    diffusion = (mu / Uin) if th["convection"] == 0 else (mu / (rho * Uin))
    theta = np.array([[th["unsteady"], th["continuity"], th["convection"], th["grad_p"] / rho, diffusion,
                       float(bc["source"]) / Uin, Uin * math.cos(math.radians(aoa)),
                       Uin * math.sin(math.radians(aoa)), Re]], dtype=np.float32)
    return (theta, np.array([[float(bc["dt"]) * Uin]], dtype=np.float32),
            np.array([[Uin, Uin, Uin * Uin]], dtype=np.float32))


def finish_mesh(raw, U=None, device=None, order="2nd"):
    """raw mesh -> full mesh dict with stencil, moments, PDE coefficients and Dirichlet targets (float64/int64).
    order: WLSQ reconstruction order the moment matrices are built for (Load_mesh.py:543-546, params.order).

    device: run the two heavy steps (k-hop stencil, WLSQ moments) with torch ops on that device (gfv.device_prep,
    SURVEY.md row f2) instead of the host numpy code; same stencil, moments equal to ~1e-12 in float64."""
    mesh = derive_geometry(raw)
    bc = dict(raw["bc"])
    if U is not None:
        bc["U"] = float(U)
    pos = mesh["node|pos"]
    n_nodes = pos.shape[0]
    support_edge = np.array([[0, 1], [1, 0]], dtype=np.int64)
    if device is None:
        extra = k_hop_pairs(mesh["face|face_node"], n_nodes, int(bc["stencil|khops"]))
        face_node_x = np.concatenate((mesh["face_node_x_base"], extra), axis=1)  # duplicates kept (Load_mesh.py:485)
        A, B1, Bx = wlsq_moments(pos, face_node_x, support_edge, order)
    else:
        import torch
        from . import device_prep
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        extra = device_prep.k_hop_pairs(t(mesh["face|face_node"]), n_nodes, int(bc["stencil|khops"])).cpu().numpy()
        face_node_x = np.concatenate((mesh["face_node_x_base"], extra), axis=1)
        A, B1, Bx = (x.cpu().numpy() for x in device_prep.wlsq_moments(t(pos), t(face_node_x), t(support_edge), order))
    theta, dt_graph, uvp_dim = pde_coefficients(bc)
    Uin = float(bc["U"])
    nt = mesh["node|node_type"]
    inlet = (nt == INFLOW) | (nt == IN_WALL) | (nt == PRESS_POINT)
    uv = velocity_profile(pos, Uin, bc["inlet_type"]).astype(np.float32)
    uv[inlet] = velocity_profile(pos[inlet], Uin, bc["inlet_type"]).astype(np.float32)
    uv[nt == WALL] = 0
    uv[nt == IN_WALL] = uv[nt == IN_WALL] / 2.0
    mesh.update({
        "face_node_x": face_node_x, "support_edge": support_edge,
        "A_node_to_node": A.astype(np.float32), "single_B_node_to_node": B1.astype(np.float32),
        "extra_B_node_to_node": Bx.astype(np.float32),
        "theta_PDE": theta, "dt_graph": dt_graph, "uvp_dim": uvp_dim,
        "sigma": np.array([bc["sigma"]], dtype=np.float32), "bc": bc,
        "target|uvp": (uv / np.float32(Uin)).astype(np.float32),
        "init_uvp": np.concatenate((uv, np.zeros((n_nodes, 1), np.float32)), axis=1),
    })
    mesh.pop("face_node_x_base")
    return mesh


def random_fields(mesh, seed=1):
    """'random-feature' node state: x[:,0:3] ~ U(-1,1) * uvp_dim."""
    rng = np.random.default_rng(seed)
    n = mesh["node|pos"].shape[0]
    return (rng.uniform(-1.0, 1.0, size=(n, 3)) * mesh["uvp_dim"].astype(np.float64)).astype(np.float32)
