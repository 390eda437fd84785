"""Mesh ingest: COMSOL ``.mphtxt`` + ``BC.json`` -> the raw mesh dict that ``gfv.meshgen.finish_mesh``
turns into a full mesh (geometry, WLSQ stencil and moments) - pure CPU, one-off per mesh.

Restates the reader half of the reference's ``Cosmol_manager`` (Extract_mesh/parse_comsol.py): the text format
(:107-348: vertices, element types ``vtx / edg / tri / quad`` with their geometric entity indices, quads re-ordered
counter-clockwise by angle about their centroid), the boundary-condition node typing with its order-dependent corner
rules (:350-424) and the element -> unique faces step (:426-485).  ``write_mphtxt`` emits the same format (used to make
the small fixture tests/golden/ingest_small.mphtxt; the reference's own parser is run on that file by
tests/golden/make_ingest_golden.py)."""
from __future__ import annotations

import json

import numpy as np

NORMAL, INFLOW, OUTFLOW, WALL, PRESS_POINT, IN_WALL = 0, 1, 2, 3, 4, 5   # utils/utilities.py NodeType


# ------------------------------------------------------------------------------------------------------------
# text format
# ------------------------------------------------------------------------------------------------------------
def read_mphtxt(path):
    """-> {"vertices": [N, sdim] float64, "<type>": {"Elements": [n, k] int64 (0-based), "Geometric entity indices":
    [n] int64 (1-based, as the COMSOL GUI shows them)}} (parse_comsol.py:107-348)."""
    with open(path, "r") as f:
        lines = [ln.strip() for ln in f.readlines()]
    i = 0

    def seek(pred, what):
        nonlocal i
        while i < len(lines):
            ln = lines[i]
            i += 1
            if pred(ln):
                return ln
        raise ValueError(f"{what} not found in {path}")

    def tokens(count):
        nonlocal i
        out = []
        while len(out) < count:
            if i >= len(lines):
                raise ValueError(f"unexpected end of {path}")
            out.extend(lines[i].split())
            i += 1
        return out[:count]

    seek(lambda s: s.startswith("# --------- Object 0 ----------"), "start of the mesh object")
    sdim = int(seek(lambda s: s.endswith("# sdim"), "sdim").split()[0])
    nv = int(seek(lambda s: s.endswith("# number of mesh vertices"), "number of mesh vertices").split()[0])
    lowest = int(seek(lambda s: s.endswith("# lowest mesh vertex index"), "lowest mesh vertex index").split()[0])
    seek(lambda s: s.startswith("# Mesh vertex coordinates"), "vertex coordinates")
    verts = np.empty((nv, sdim), dtype=np.float64)
    for v in range(nv):
        while i < len(lines) and not lines[i]:
            i += 1
        verts[v] = [float(x) for x in tokens(sdim)]
    out = {"vertices": verts}
    ntypes = int(seek(lambda s: s.endswith("# number of element types"), "number of element types").split()[0])
    for _ in range(ntypes):
        seek(lambda s: s.startswith("# Type #"), "element type")
        while i < len(lines) and not lines[i]:
            i += 1
        name = lines[i].split()[1]
        i += 1
        k = int(seek(lambda s: s.endswith("# number of vertices per element"), "vertices per element").split()[0])
        n = int(seek(lambda s: s.endswith("# number of elements"), "number of elements").split()[0])
        while i < len(lines) and (not lines[i] or lines[i].startswith("#")):
            i += 1
        elems = np.empty((n, k), dtype=np.int64)
        for e in range(n):
            while i < len(lines) and not lines[i]:
                i += 1
            ev = np.array([int(x) - lowest for x in tokens(k)], dtype=np.int64)
            if k > 3:   # counter-clockwise by angle about the centroid (parse_comsol.py:296-304)
                xy = verts[ev, :]
                d = xy - xy.mean(axis=0)
                ev = ev[np.argsort(np.arctan2(d[:, 1], d[:, 0]))]
            elems[e] = ev
        ng = int(seek(lambda s: s.endswith("# number of geometric entity indices"), "geometric entity indices").split()[0])
        while i < len(lines) and (not lines[i] or lines[i].startswith("#")):
            i += 1
        geo = np.empty((ng,), dtype=np.int64)
        for gidx in range(ng):
            while i < len(lines) and not lines[i]:
                i += 1
            geo[gidx] = int(lines[i])
            i += 1
        out[name] = {"Elements": elems, "Geometric entity indices": geo + 1}
    return out


def write_mphtxt(path, vertices, types):
    """types: {"vtx"|"edg"|"tri"|"quad": (elements [n,k] 0-based, geometric entity indices [n] 0-based)}."""
    with open(path, "w") as f:
        f.write("# Created by gfv.ingest.write_mphtxt\n\n# Major & minor version\n0 1\n1 # number of tags\n# Tags\n5 mesh1\n"
                "1 # number of types\n# Types\n3 obj\n\n# --------- Object 0 ----------\n\n0 0 1\n4 Mesh # class\n"
                "4 # version\n")
        f.write(f"{vertices.shape[1]} # sdim\n{vertices.shape[0]} # number of mesh vertices\n0 # lowest mesh vertex index\n\n"
                "# Mesh vertex coordinates\n")
        for v in vertices:
            f.write(" ".join(repr(float(x)) for x in v) + " \n")
        f.write(f"\n{len(types)} # number of element types\n")
        for t, (name, (elems, geo)) in enumerate(types.items()):
            elems = np.asarray(elems).reshape(len(geo), -1)
            f.write(f"\n# Type #{t}\n\n{len(name)} {name} # type name\n\n\n{elems.shape[1]} # number of vertices per element\n"
                    f"{elems.shape[0]} # number of elements\n# Elements\n")
            for e in elems:
                f.write(" ".join(str(int(x)) for x in e) + " \n")
            f.write(f"\n{len(geo)} # number of geometric entity indices\n# Geometric entity indices\n")
            for gidx in geo:
                f.write(f"{int(gidx)}\n")


# ------------------------------------------------------------------------------------------------------------
# boundary conditions -> node types (parse_comsol.py:69-105,350-424)
# ------------------------------------------------------------------------------------------------------------
def expand_bc(bc):
    """BC.json lists may hold ints, "a-b" ranges and nested lists; non-list entries pass through."""
    def item(x):
        if isinstance(x, str) and "-" in x:
            a, b = map(int, x.split("-"))
            return list(range(a, b + 1))
        if isinstance(x, list):
            return [item(y) for y in x]
        return int(x)

    def flat(xs):
        out = []
        for x in xs:
            out.extend(flat(x) if isinstance(x, list) else [x])
        return out

    return {k: (flat([item(x) for x in v]) if isinstance(v, list) else v) for k, v in bc.items()}


def node_types(mesh_file, bc):
    """Order-dependent typing of the reference: the BC.json entries are applied in file order; a WALL edge keeps
    INFLOW corners as IN_WALL, an OUTFLOW edge leaves WALL / INFLOW corners as they are."""
    n = mesh_file["vertices"].shape[0]
    nt = np.full((n,), NORMAL, dtype=np.int64)
    surf = np.zeros((n,), dtype=bool)
    edges = mesh_file["edg"]["Elements"]
    egeo = mesh_file["edg"]["Geometric entity indices"]
    for kind, idx_list in bc.items():
        if idx_list is None or not isinstance(idx_list, list):
            continue
        for b in idx_list:
            e = edges[egeo == b]
            if kind == "inflow":
                nt[e[:, 0]] = INFLOW
                nt[e[:, 1]] = INFLOW
            elif kind == "wall":
                was_l, was_r = nt[e[:, 0]] == INFLOW, nt[e[:, 1]] == INFLOW
                nt[e[:, 0]] = WALL
                nt[e[:, 1]] = WALL
                nt[e[was_l, 0]] = IN_WALL
                nt[e[was_r, 1]] = IN_WALL
            elif kind == "outflow":
                wl, wr = nt[e[:, 0]] == WALL, nt[e[:, 1]] == WALL
                il, ir = nt[e[:, 0]] == INFLOW, nt[e[:, 1]] == INFLOW
                nt[e[:, 0]] = OUTFLOW
                nt[e[:, 1]] = OUTFLOW
                nt[e[wl, 0]] = WALL
                nt[e[wr, 1]] = WALL
                nt[e[il, 0]] = INFLOW
                nt[e[ir, 1]] = INFLOW
            elif kind == "pressure_point":
                vt = mesh_file["vtx"]["Elements"].reshape(-1)
                nt[vt[mesh_file["vtx"]["Geometric entity indices"] == b]] = PRESS_POINT
            elif kind == "surf":
                surf[e[:, 0]] = True
                surf[e[:, 1]] = True
    return nt, surf


# ------------------------------------------------------------------------------------------------------------
# elements -> faces (parse_comsol.py:426-485)
# ------------------------------------------------------------------------------------------------------------
def comsol_to_raw(mesh_file, bc_json, **physics):
    """-> raw mesh dict for ``gfv.meshgen.finish_mesh``: node|pos, node|node_type, face|face_node (unique sorted edges),
    cells_node / cells_index / cells_face (tri block, then quad block, reader order), bc.

    `physics` overrides the sampled entries of ``bc`` that ``finish_mesh`` reads (U, rho, mu, source, aoa, dt, L; the
    reference draws them per mesh from the ranges in BC.json, Load_mesh.py:134-211); defaults: the first value of each
    range."""
    bc = expand_bc(bc_json)
    nt, _surf = node_types(mesh_file, bc)
    cells_node, cells_index, edge_blocks = [], [], []
    count = 0
    for name in ("tri", "quad"):
        if name not in mesh_file:
            continue
        el = mesh_file[name]["Elements"]
        cells_node.append(el.reshape(-1))
        cells_index.append(np.repeat(np.arange(count, count + el.shape[0]), el.shape[1]))
        count += el.shape[0]
        k = el.shape[1]
        e = np.stack([np.stack((el[:, j], el[:, (j + 1) % k]), axis=1) for j in range(k)], axis=1).reshape(-1, 2)
        edge_blocks.append(np.sort(e, axis=1).T)
    face_node, cells_face = np.unique(np.concatenate(edge_blocks, axis=1), axis=1, return_inverse=True)
    th = bc_json["theta_PDE"]
    first = lambda v: float(v[0] if isinstance(v, list) else v)
    pb = {"stencil|khops": int(bc_json.get("stencil|khops", 2)),
          "theta_PDE": {k: th[k] for k in ("unsteady", "continuity", "convection", "grad_p")},
          "U": first(th["inlet"]), "rho": first(th["rho"]), "mu": first(th["mu"]), "source": first(th["source"]),
          "aoa": first(th["aoa"]), "dt": float(th["dt"]), "L": float(th["L"]), "sigma": list(bc_json["sigma"]),
          "inlet_type": bc_json["inlet_type"]}
    pb.update(physics)
    return {"node|pos": mesh_file["vertices"][:, 0:2].astype(np.float64), "node|node_type": nt,
            "face|face_node": face_node.astype(np.int64), "cells_node": np.concatenate(cells_node).astype(np.int64),
            "cells_index": np.concatenate(cells_index).astype(np.int64),
            "cells_face": np.asarray(cells_face).reshape(-1).astype(np.int64), "bc": pb}


def load_comsol_mesh(mphtxt_path, bc_json_path, **physics):
    with open(bc_json_path, "r") as f:
        bc_json = json.load(f)
    return comsol_to_raw(read_mphtxt(mphtxt_path), bc_json, **physics)
