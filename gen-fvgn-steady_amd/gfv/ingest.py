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


# ------------------------------------------------------------------------------------------------------------
# Tecplot FEPolygon meshes (Extract_mesh/parse_tecplot.py:50-699): polygon cells of any size (config 5 of BASELINE.json)
# ------------------------------------------------------------------------------------------------------------
def read_tecplot(path):
    """Parse a Tecplot ASCII ``.dat`` mesh as the reference's `TecplotMesh` does (parse_tecplot.py:74-351): ONE interior
    zone (ZONETYPE FEPolygon / FETriangle / FEQuadrilateral, BLOCK packing: all X, then all Y; sections ``# face nodes``,
    ``# left elements``, ``# right elements``, 1-based, 0 = outside) followed by boundary line zones whose node
    coordinates mark the obstacle surface.  Returns a dict: ``pos`` [N,2] float64, ``face_node`` [E,2] int64 (0-based, file
    order and orientation), ``left`` / ``right`` [E] int64 (0-based, -1 = outside), ``boundary_pos`` [Nb,2] float64
    (coordinates of every boundary-zone node, zones concatenated in file order), ``variables``, ``zones``."""
    with open(path, "r") as f:
        lines = f.read().split("\n")
    i, n = 0, len(lines)
    variables, zones = [], []
    out = {"boundary_pos": []}
    while i < n:
        line = lines[i].strip()
        if line.startswith("VARIABLES"):
            variables = [line.split("=", 1)[1].strip().strip('"')]
            i += 1
            while i < n and not lines[i].strip().startswith("ZONE"):
                s = lines[i].strip()
                if s and not s.startswith("DATASETAUXDATA"):
                    variables.append(s.strip('"'))
                i += 1
            continue
        if line.startswith("ZONE"):
            info = {"ZONE": line.split("=", 1)[1].strip().strip('"')}
            i += 1
            while i < n and not lines[i].strip().startswith("DT"):
                for kv in lines[i].strip().split(","):
                    if "=" in kv:
                        k, v = kv.split("=", 1)
                        info[k.strip()] = v.strip()
                i += 1
            i += 1                                                        # the DT=(...) line
            n_nodes = int(info.get("Nodes", 0))
            need = n_nodes * len(variables)
            vals = []
            while len(vals) < need:
                vals.extend(lines[i].split())
                i += 1
            data = np.asarray(vals[:need], dtype=np.float64).reshape(len(variables), n_nodes)
            xy = np.stack((data[variables.index("X")], data[variables.index("Y")]), axis=1)
            sections, key = {}, None
            while i < n and not lines[i].startswith("ZONE"):
                s = lines[i].strip()
                i += 1
                if not s:
                    continue
                if s.startswith("#"):
                    key = "_".join(s.split("#", 1)[1].split())
                    sections[key] = []
                    continue
                sections.setdefault(key, []).append(s)
            ints = {k: np.asarray(" ".join(v).split(), dtype=np.int64) for k, v in sections.items()}
            zones.append(info)
            if info.get("ZONETYPE", "").lower() in ("fepolygon", "fetriangle", "fequadrilateral"):
                out["pos"] = xy
                out["face_node"] = ints["face_nodes"].reshape(-1, 2) - 1
                out["left"] = ints["left_elements"] - 1
                out["right"] = ints["right_elements"] - 1
            else:
                out["boundary_pos"].append(xy)
            continue
        i += 1
    out["boundary_pos"] = np.concatenate(out["boundary_pos"], axis=0) if out["boundary_pos"] else np.zeros((0, 2))
    out["variables"], out["zones"] = variables, zones
    return out


def _ensure_ccw(items, cells_index, coords):
    """parse_base.py:142-191 (`is_convex`, `reorder_polygon`, `ensure_counterclockwise`) for every cell at once: a cell whose
    list, as given, turns the same way at every corner ((a - b) x (c - b) >= 0) is kept; any other is sorted by angle
    about the mean of its points (stable, like Python's `sorted`)."""
    out = items.copy()
    counts = np.bincount(cells_index)
    start = np.concatenate(([0], np.cumsum(counts)))[:-1]
    for ct in np.unique(counts[counts > 0]):
        cells = np.nonzero(counts == ct)[0]
        idx = start[cells][:, None] + np.arange(ct)[None, :]
        ids = items[idx]                                                   # [nc, ct]
        p = coords[ids]
        a, b, c = p, np.roll(p, -1, axis=1), np.roll(p, -2, axis=1)
        ba, bc = a - b, c - b
        cross = ba[:, :, 0] * bc[:, :, 1] - ba[:, :, 1] * bc[:, :, 0]
        bad = (cross < 0).any(axis=1)
        if bad.any():
            pb = p[bad]
            cen = pb.mean(axis=1, keepdims=True)
            order = np.argsort(np.arctan2(pb[:, :, 1] - cen[:, :, 1], pb[:, :, 0] - cen[:, :, 0]), axis=1, kind="stable")
            ids[bad] = np.take_along_axis(ids[bad], order, axis=1)
            out[idx] = ids
    return out


def tecplot_cells(face_node, left, right, pos=None):
    """Flat (cell, face) / (cell, node) incidence lists in the reader order of parse_tecplot.py:176-207: the faces of a cell
    are its occurrences in [left elements | right elements] in that order (ascending face id inside each half), its nodes
    the ascending unique node ids of those faces; with `pos`, each list then goes through the reference's
    `ensure_counterclockwise` (faces by their centres, nodes by their coordinates).  The final CCW ordering is applied
    afterwards by extract_mesh_state = meshgen.derive_geometry, whatever this order is.
    Cells are numbered as in the file; cells of different size interleave (derive_geometry regroups them by size)."""
    E = face_node.shape[0]
    two_way = np.concatenate((left, right))
    face_index = np.tile(np.arange(E, dtype=np.int64), 2)
    keep = two_way >= 0
    order = np.argsort(two_way[keep], kind="stable")
    cells_index = two_way[keep][order]
    cells_face = face_index[keep][order]
    pairs = np.stack((np.repeat(cells_index, 2), face_node[cells_face].reshape(-1)), axis=1)
    pairs = np.unique(pairs, axis=0)                                     # sorted by (cell, node)
    n_cells = int(cells_index.max()) + 1
    if not np.array_equal(np.bincount(pairs[:, 0], minlength=n_cells), np.bincount(cells_index, minlength=n_cells)):
        raise ValueError("a cell does not have as many nodes as faces: not a polygon mesh")
    cells_node = pairs[:, 1].copy()
    if pos is not None:
        fc = (pos[face_node[:, 0]] + pos[face_node[:, 1]]) / 2.0
        cells_face = _ensure_ccw(cells_face, cells_index, fc)
        cells_node = _ensure_ccw(cells_node, cells_index, pos)
    return cells_node, cells_face, cells_index


def pipe_flow_node_types(pos, boundary_pos):
    """Node typing of the reference's Tecplot path (parse_tecplot.py:563-643, `extract_pipe_flow_boundary`): float32
    coordinates shifted by |min|; left edge INFLOW, top / bottom WALL, right edge OUTFLOW (corners belong to the walls),
    interior nodes whose coordinates coincide exactly with a boundary-zone node are the obstacle (WALL + surf mask)."""
    shift = np.abs(pos.astype(np.float32).min(axis=0))
    p = pos.astype(np.float32) + shift
    bp = boundary_pos.astype(np.float32) + shift
    top, bottom = p[:, 1].max(), p[:, 1].min()
    outlet, inlet = p[:, 0].max(), p[:, 0].min()
    x, y = p[:, 0].astype(np.float64), p[:, 1].astype(np.float64)

    def is_equal(v, pivot):                                               # parse_base.py:193-211
        return (np.abs(v) >= abs(float(pivot)) - 1e-8) & (np.abs(v) <= abs(float(pivot)) + 1e-8)

    inside_y = (y > float(bottom) + 1e-12) & (y < float(top) - 1e-12)
    is_in = is_equal(x, inlet) & inside_y
    is_wall = ~is_in & ((p[:, 1] >= top) | (p[:, 1] <= bottom))
    is_out = ~is_in & ~is_wall & is_equal(x, outlet) & inside_y
    key = lambda a: np.ascontiguousarray(a).view(np.dtype((np.void, 8))).reshape(-1)
    on_zone = np.isin(key(p), key(bp))
    is_obst = ~is_in & ~is_wall & ~is_out & on_zone & (x > 0) & (x < float(outlet) - 1e-12) & (y > 0) & (y < float(top) - 1e-12)
    nt = np.full((pos.shape[0],), NORMAL, dtype=np.int64)
    nt[is_in] = INFLOW
    nt[is_wall] = WALL
    nt[is_out] = OUTFLOW
    nt[is_obst] = WALL
    return nt, is_obst


def tecplot_to_raw(tec, bc_json, **physics):
    """-> raw mesh dict for ``gfv.meshgen.finish_mesh`` (same contract as `comsol_to_raw`); face|face_node keeps the file's
    order and orientation (parse_tecplot.py:656)."""
    nt, surf = pipe_flow_node_types(tec["pos"], tec["boundary_pos"])
    cells_node, cells_face, cells_index = tecplot_cells(tec["face_node"], tec["left"], tec["right"], tec["pos"])
    th = bc_json["theta_PDE"]
    first = lambda v: float(v[0] if isinstance(v, list) else v)
    pb = {"stencil|khops": int(bc_json.get("stencil|khops", 2)),
          "theta_PDE": {k: th[k] for k in ("unsteady", "continuity", "convection", "grad_p")},
          "U": first(th["inlet"]), "rho": first(th["rho"]), "mu": first(th["mu"]), "source": first(th["source"]),
          "aoa": first(th["aoa"]), "dt": float(th["dt"]), "L": float(th["L"]), "sigma": list(bc_json["sigma"]),
          "inlet_type": bc_json["inlet_type"]}
    pb.update(physics)
    return {"node|pos": tec["pos"].astype(np.float64), "node|node_type": nt, "node|surf_mask": surf,
            "face|face_node": np.ascontiguousarray(tec["face_node"].T).astype(np.int64), "cells_node": cells_node,
            "cells_index": cells_index, "cells_face": cells_face, "bc": pb}


def load_tecplot_mesh_from(tec, bc_json, **physics):
    """`tecplot_to_raw` under the name the loaders use (tec: result of `read_tecplot`)."""
    return tecplot_to_raw(tec, bc_json, **physics)


def load_tecplot_mesh(dat_path, bc_json_path, **physics):
    with open(bc_json_path, "r") as f:
        bc_json = json.load(f)
    return tecplot_to_raw(read_tecplot(dat_path), bc_json, **physics)


def write_tecplot(path, pos, face_node, left, right, n_cells, boundary_zones=()):
    """Write a polygon mesh in the Tecplot ASCII layout `read_tecplot` (and the reference's reader) takes: 1-based
    `face_node` [E,2], `left` / `right` [E] (0 = outside); boundary_zones: [(name, coordinates [n,2])]."""
    pos = np.asarray(pos, dtype=np.float64)

    def block(f, values, per_line, fmt):
        values = list(values)
        for i in range(0, len(values), per_line):
            f.write(" " + " ".join(fmt % v for v in values[i:i + per_line]) + "\n")

    with open(path, "w") as f:
        f.write('TITLE     = "gfv polygon mesh"\nVARIABLES = "X"\n"Y"\n')
        f.write('ZONE T="Surf: tank"\n STRANDID=1, SOLUTIONTIME=0\n')
        f.write(f" Nodes={pos.shape[0]}, Faces={len(left)}, Elements={n_cells}, ZONETYPE=FEPolygon\n DATAPACKING=BLOCK\n")
        f.write(" NumConnectedBoundaryFaces=0, TotalNumBoundaryConnections=0\n DT=(SINGLE SINGLE )\n")
        block(f, pos[:, 0], 5, "%.9E")
        block(f, pos[:, 1], 5, "%.9E")
        f.write("# face nodes\n")
        block(f, np.asarray(face_node).reshape(-1), 10, "%d")
        f.write("# left elements\n")
        block(f, left, 10, "%d")
        f.write("# right elements\n")
        block(f, right, 10, "%d")
        for name, xy in boundary_zones:
            xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
            f.write(f'ZONE T="{name}"\n STRANDID=2, SOLUTIONTIME=0\n')
            f.write(f" Nodes={xy.shape[0]}, Elements={max(xy.shape[0] - 1, 1)}, ZONETYPE=FELineSeg\n DATAPACKING=BLOCK\n")
            f.write(" DT=(SINGLE SINGLE )\n")
            block(f, xy[:, 0], 5, "%.9E")
            block(f, xy[:, 1], 5, "%.9E")
            block(f, [v for i in range(max(xy.shape[0] - 1, 1)) for v in (i + 1, min(i + 2, xy.shape[0]))], 10, "%d")
