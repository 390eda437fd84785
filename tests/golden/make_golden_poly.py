"""Golden vectors on the reference's POLYGON example mesh (mesh_example/cylinder_flow_poly: Tecplot FEPolygon, 27 778 nodes,
45 214 faces, 17 436 cells of 3 ... 9 nodes - BASELINE.json config 5's cell kind).  The reference's own Tecplot reader
(Extract_mesh/parse_tecplot.py `TecplotMesh`), its extract_mesh_state / transform_mesh pipeline and its NNmodel forward /
backward run here, in the build container; committed as tests/golden/poly_cylinder.npz are the RAW reader arrays (data, not
code: coordinates, face nodes, left / right elements, boundary-zone coordinates), the sampled PDE parameters, the node
field and the reference's outputs.  Also written: tests/golden/poly_small.dat, a small Tecplot-format polygon mesh made by
this script (a Voronoi-like dual of a jittered grid), with the arrays the reference's reader returns for it.
Run: python tests/golden/make_golden_poly.py"""
import json
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refstubs"))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402

sys.path.insert(0, cases.ROOT)
from oracle import fvgn_oracle as O  # noqa: E402
from gfv import ingest, meshgen  # noqa: E402
from gfv.graph import build_batch  # noqa: E402
import make_golden as MG  # noqa: E402

MESH_DIR = "/root/reference/mesh_example/cylinder_flow_poly"


def reference_reader(dat_path, case_name, out_dir):
    """The reference's TecplotMesh on `dat_path` -> (its mesh dict after extract_mesh_state, its raw reader arrays)."""
    import importlib
    pt = importlib.import_module("Extract_mesh.parse_tecplot")
    pt.write_vtp_file = lambda *a, **k: None
    pt.TecplotMesh.save_to_vtu = lambda *a, **k: None
    ph = importlib.import_module("Extract_mesh.parse_to_h5")
    ph.write_point_cloud_to_vtk = lambda *a, **k: None
    os.makedirs(out_dir, exist_ok=True)
    path = {"file_dir": out_dir, "case_name": case_name, "mesh_only": True}
    captured = {}
    orig = pt.extract_mesh_state

    def capture(dataset, path=None):
        for k in ("cells_node", "cells_face", "cells_index"):
            captured[k] = np.asarray(dataset[k]).astype(np.int64).reshape(-1).copy()
        captured["node_type"] = np.asarray(dataset["node|node_type"]).astype(np.int64).reshape(-1).copy()
        captured["surf"] = np.asarray(dataset["node|surf_mask"]).astype(bool).reshape(-1).copy()
        return orig(dataset, path=path)

    pt.extract_mesh_state = capture
    mgr = pt.TecplotMesh(mesh_file=dat_path, data_file=None, file_dir=out_dir, case_name=case_name, path=path)
    mesh = dict(mgr.extract_mesh())
    return mesh, captured


def check_reader(dat_path, bc, mesh_ref, captured, label):
    """gfv.ingest's reader against the reference's, array by array (integers bit-exact)."""
    tec = ingest.read_tecplot(dat_path)
    raw = ingest.tecplot_to_raw(tec, bc)
    assert np.array_equal(raw["node|node_type"], captured["node_type"]), "node types"
    assert np.array_equal(raw["node|surf_mask"], captured["surf"]), "surf mask"
    for k in ("cells_node", "cells_face", "cells_index"):
        assert np.array_equal(raw[k], captured[k]), k
    assert np.array_equal(raw["face|face_node"], np.asarray(mesh_ref["face|face_node"]).astype(np.int64)), "face_node"
    assert np.array_equal(raw["node|pos"], np.asarray(mesh_ref["node|pos"]).astype(np.float64)), "pos"
    geo = meshgen.derive_geometry(raw)
    n = lambda k: np.asarray(mesh_ref[k])
    for k in ("cells_node", "cells_face", "cells_index", "face|face_type", "face|neighbour_cell"):
        assert np.array_equal(np.asarray(geo[k]).reshape(-1), n(k).astype(np.int64).reshape(-1)), k
    assert np.array_equal(geo["face_node_x_base"], n("face_node_x").astype(np.int64)), "face_node_x (cell-sharing pairs)"
    for k in ("cell|centroid", "face|face_center_pos", "face|face_area", "unit_norm_v", "cell|cells_area"):
        a_, b_ = np.asarray(geo[k], dtype=np.float64).reshape(-1), n(k).astype(np.float64).reshape(-1)
        err = np.abs(a_ - b_).max() / (np.abs(b_).max() + 1e-30)
        print(f"  [{label}] ingest + meshgen vs reference reader {k:24s} rel {err:.2e}")
        assert err < 1e-6, k
    ct = np.bincount(np.bincount(raw["cells_index"]))
    print(f"  [{label}] cells by node count:", {i: int(c) for i, c in enumerate(ct) if c})
    return tec, raw


def write_small_poly(path, nx=13, ny=7, seed=5):
    """A small polygon mesh in the Tecplot layout the reference reads: the dual of a jittered triangulated grid clipped
    to a channel (cells of 3 ... 8 nodes), a square obstacle of removed cells, boundary zones listing the obstacle / wall /
    inflow / outflow nodes.  Pure numpy; written with gfv.ingest.write_tecplot."""
    rng = np.random.default_rng(seed)
    # primal points: regular grid, interior points jittered; dual cells = polygons around primal points (clipped at the box)
    import scipy.spatial as sps
    W, Hh = 0.6, 0.3
    xs, ys = (np.arange(nx) + 0.5) * W / nx, (np.arange(ny) + 0.5) * Hh / ny
    P = np.stack(np.meshgrid(xs, ys, indexing="ij"), -1).reshape(-1, 2)
    P += rng.uniform(-0.3, 0.3, size=P.shape) * np.array([W / nx, Hh / ny])
    # mirror the points across the four sides so that the Voronoi cells of the box points are clipped exactly at the box
    mir = [P, P * [-1, 1], P * [-1, 1] + [2 * W, 0], P * [1, -1], P * [1, -1] + [0, 2 * Hh]]
    vor = sps.Voronoi(np.concatenate(mir))
    cells = []
    hole = lambda c: (0.18 < c[0] < 0.30) and (0.10 < c[1] < 0.20)
    for i in range(P.shape[0]):
        reg = vor.regions[vor.point_region[i]]
        if -1 in reg or len(reg) < 3 or hole(P[i]):
            continue
        cells.append(reg)
    used = np.unique(np.concatenate(cells))
    V = vor.vertices[used]
    V = np.round(V, 9)
    remap = {int(u): k for k, u in enumerate(used)}
    # merge duplicate vertices (rounded coordinates)
    _, first, inv = np.unique(V, axis=0, return_index=True, return_inverse=True)
    V = V[np.sort(first)]
    rank = np.argsort(np.argsort(first))
    vid = lambda u: int(rank[inv[remap[int(u)]]])
    faces, left, right = {}, [], []
    for ci, reg in enumerate(cells):
        ids = [vid(u) for u in reg]
        ids = [a for a, b in zip(ids, ids[1:] + ids[:1]) if a != b]
        for a, b in zip(ids, ids[1:] + ids[:1]):
            key = (min(a, b), max(a, b))
            if key not in faces:
                faces[key] = [len(faces), (a, b), ci + 1, 0]
            else:
                faces[key][3] = ci + 1
    fl = sorted(faces.values())
    face_node = np.array([f[1] for f in fl]) + 1
    left = np.array([f[2] for f in fl])
    right = np.array([f[3] for f in fl])
    bnd = face_node[right == 0] - 1
    bn = np.unique(bnd)
    bp = V[bn]
    on_outer = (np.isclose(bp[:, 0], 0) | np.isclose(bp[:, 0], W) | np.isclose(bp[:, 1], 0) | np.isclose(bp[:, 1], Hh))
    zones = [("Line: Cylinder.Cylinder Surface", bp[~on_outer]), ("Line: Block.wall", bp[on_outer])]
    ingest.write_tecplot(path, V, face_node, left, right, len(cells), zones)


def main():
    R = ref_import.reference_modules()
    M = ref_import.reference_mesh_modules()
    bc = json.load(open(f"{MESH_DIR}/BC.json"))

    # ---- small polygon mesh: reader only ----------------------------------------------------------------------------
    small = os.path.join(HERE, "poly_small.dat")
    write_small_poly(small)
    mesh_s, cap_s = reference_reader(small, "cylinder_small_poly", "/tmp/poly_small_out")
    check_reader(small, bc, mesh_s, cap_s, "poly_small")
    np.savez_compressed(os.path.join(HERE, "poly_small_reader.npz"),
                        node_type=cap_s["node_type"].astype(np.int8), surf=cap_s["surf"],
                        cells_node=cap_s["cells_node"].astype(np.int32), cells_face=cap_s["cells_face"].astype(np.int32),
                        cells_index=cap_s["cells_index"].astype(np.int32),
                        ccw_cells_node=np.asarray(mesh_s["cells_node"]).astype(np.int32).reshape(-1),
                        ccw_cells_face=np.asarray(mesh_s["cells_face"]).astype(np.int32).reshape(-1),
                        ccw_cells_index=np.asarray(mesh_s["cells_index"]).astype(np.int32).reshape(-1),
                        face_type=np.asarray(mesh_s["face|face_type"]).astype(np.int8).reshape(-1),
                        face_node_x=np.asarray(mesh_s["face_node_x"]).astype(np.int32),
                        cells_area=np.asarray(mesh_s["cell|cells_area"]).astype(np.float64).reshape(-1),
                        unit_norm_v=np.asarray(mesh_s["unit_norm_v"]).astype(np.float64))

    # ---- the reference's example mesh: reader + pipeline + model ------------------------------------------------------
    mesh, cap = reference_reader(f"{MESH_DIR}/mesh.dat", "cylinder_flow_poly", "/tmp/poly_out")
    tec, raw = check_reader(f"{MESH_DIR}/mesh.dat", bc, mesh, cap, "cylinder_flow_poly")
    mesh.update(bc)
    mesh["case_name"] = "cylinder_flow_poly"
    mesh["theta_PDE_bak"] = mesh["theta_PDE"]
    th = mesh["theta_PDE_bak"]
    mesh["theta_PDE_list"] = R.get_param.generate_combinations(
        U_range=th["inlet"], rho_range=th["rho"], mu_range=th["mu"], source_range=th["source"], aoa_range=th["aoa"],
        dt=th["dt"], L=th["L"], Re_max=th["Re_max"], Re_min=th["Re_min"])
    random.seed(11)
    params = R.get_param.params()
    mesh, init_uvp = M.Load_mesh.CFDdatasetBase.transform_mesh(mesh, params)
    n = lambda k: np.asarray(mesh[k])
    md = {
        "node|pos": n("node|pos").astype(np.float64), "node|node_type": n("node|node_type").astype(np.int64).reshape(-1),
        "face|face_node": n("face|face_node").astype(np.int64), "cells_node": n("cells_node").astype(np.int64).reshape(-1),
        "cells_face": n("cells_face").astype(np.int64).reshape(-1), "cells_index": n("cells_index").astype(np.int64).reshape(-1),
        "cell|centroid": n("cell|centroid"), "face|face_center_pos": n("face|face_center_pos"),
        "face|face_type": n("face|face_type").astype(np.int64).reshape(-1), "face|face_area": n("face|face_area"),
        "face|neighbour_cell": n("face|neighbour_cell").astype(np.int64), "unit_norm_v": n("unit_norm_v"),
        "cell|cells_area": n("cell|cells_area").reshape(-1), "face_node_x": n("face_node_x").astype(np.int64),
        "support_edge": n("support_edge").astype(np.int64), "A_node_to_node": n("A_node_to_node"),
        "single_B_node_to_node": n("single_B_node_to_node"), "extra_B_node_to_node": n("extra_B_node_to_node"),
        "theta_PDE": n("theta_PDE").astype(np.float32), "dt_graph": n("dt_graph").astype(np.float32),
        "uvp_dim": n("uvp_dim").astype(np.float32), "sigma": n("sigma").astype(np.float32),
        "target|uvp": n("target|uvp").astype(np.float32), "init_uvp": init_uvp.numpy().astype(np.float32),
    }
    rng = np.random.default_rng(123)
    field = (rng.uniform(-1, 1, size=(md["node|pos"].shape[0], 3)) * md["uvp_dim"].astype(np.float64)).astype(np.float32)
    graphs = build_batch([md], [field])

    P0 = O.init_parameters(cases.WEIGHT_SEED)
    model = R.importer.NNmodel(params)
    sd = model.state_dict()
    for k, v in P0.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    gn, gx, ge, gc, gi = MG.to_ref_graphs(graphs)
    gn.norm_uvp, gn.norm_global = True, True
    lc, lmx, lmy, lp, uvp_node, uvp_cell = model(graph_node=gn, graph_node_x=gx, graph_edge=ge, graph_cell=gc, graph_Index=gi,
                                                 is_training=True)
    loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lmx + params.loss_mom * lmy))
    loss.backward()
    names = list(P0)
    grads = {k: p.grad for k, p in model.named_parameters()}
    og = tuple(g.clone() for g in graphs)
    oout = O.model_forward({k: v.clone() for k, v in P0.items()}, O.new_normalizer_buffers(), og)
    print("poly mesh: N", md["node|pos"].shape[0], "E", md["face|face_node"].shape[1], "C", md["cell|centroid"].shape[0],
          "Ex", md["face_node_x"].shape[1], "loss ref", float(loss))
    for nm, a, b in (("loss_cont", oout[0], lc), ("loss_mom_x", oout[1], lmx), ("loss_mom_y", oout[2], lmy),
                     ("loss_press", oout[3], lp), ("uvp_node", oout[4], uvp_node), ("uvp_cell", oout[5], uvp_cell)):
        print("  oracle vs reference", nm, MG.rel(a, b))

    U = float(mesh["mean_u"])
    bcd = {"stencil|khops": int(bc["stencil|khops"]), "theta_PDE": {k: th[k] for k in ("unsteady", "continuity", "convection", "grad_p")},
           "U": U, "rho": float(mesh["rho"]), "mu": float(mesh["mu"]), "source": float(mesh["source"]),
           "aoa": float(mesh["aoa"]), "dt": float(mesh["dt"]), "L": float(mesh["L"]), "sigma": bc["sigma"],
           "inlet_type": bc["inlet_type"]}
    raw["bc"] = bcd
    mine = meshgen.finish_mesh(raw)
    for k in ("cells_node", "cells_face", "cells_index", "face|face_type", "face|neighbour_cell", "face_node_x", "support_edge"):
        assert np.array_equal(np.asarray(mine[k]), md[k]), k
    for k in ("cell|centroid", "face|face_center_pos", "face|face_area", "unit_norm_v", "cell|cells_area", "A_node_to_node",
              "single_B_node_to_node", "extra_B_node_to_node", "theta_PDE", "dt_graph", "uvp_dim", "sigma", "target|uvp"):
        a_, b_ = np.asarray(mine[k], dtype=np.float64).reshape(-1), np.asarray(md[k], dtype=np.float64).reshape(-1)
        err = np.abs(a_ - b_).max() / (np.abs(b_).max() + 1e-30)
        print(f"  meshgen vs reference pipeline {k:24s} rel {err:.2e}")
        assert err < 1e-6, k
    save = {"tec.pos": tec["pos"], "tec.face_node": tec["face_node"].astype(np.int32), "tec.left": tec["left"].astype(np.int32),
            "tec.right": tec["right"].astype(np.int32), "tec.boundary_pos": tec["boundary_pos"],
            "raw.bc": np.array(json.dumps(bcd)), "field": field,
            "node_type": cap["node_type"].astype(np.int8),
            "fp.cells_node": cases.fingerprint(md["cells_node"]), "fp.cells_face": cases.fingerprint(md["cells_face"]),
            "fp.cells_index": cases.fingerprint(md["cells_index"]), "fp.face_node_x": cases.fingerprint(md["face_node_x"]),
            "fp.face_type": cases.fingerprint(md["face|face_type"])}
    save.update({"loss": np.float64(loss.item()), "loss_cont": lc.detach().numpy(), "loss_mom_x": lmx.detach().numpy(),
                 "loss_mom_y": lmy.detach().numpy(), "loss_press": lp.detach().numpy(),
                 "uvp_node": uvp_node.detach().numpy(), "uvp_cell": uvp_cell.detach().numpy(),
                 "grad_fp": np.stack([cases.fingerprint(grads[k].numpy()) if grads[k] is not None else np.full(3, np.nan)
                                      for k in names])})
    out = os.path.join(HERE, "poly_cylinder.npz")
    np.savez_compressed(out, **save)
    print("saved", os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
