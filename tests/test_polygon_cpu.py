"""Polygon cells (BASELINE.json config 5's mesh kind) on the CPU side: the Tecplot reader and the geometry pipeline
against what the REFERENCE's reader / extract_mesh_state returned (fixtures made by tests/golden/make_golden_poly.py, which
imports the reference in the build container), and the oracle on the polygon mesh against the reference's outputs."""
import os

import numpy as np
import torch

import cases
from oracle import fvgn_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def test_tecplot_reader_small_polygon_mesh_matches_reference_reader():
    from gfv import ingest, meshgen
    import json
    tec = ingest.read_tecplot(os.path.join(GOLD, "poly_small.dat"))
    ref = np.load(os.path.join(GOLD, "poly_small_reader.npz"))
    bc = {"stencil|khops": 2, "sigma": [1, 1, 1], "inlet_type": "parabolic",
          "theta_PDE": {"unsteady": 1, "continuity": 1, "convection": 1, "grad_p": 1, "inlet": [0.1], "rho": [1], "mu": [0.001],
                        "source": [0], "aoa": [0], "dt": 0.5, "L": 0.1}}
    raw = ingest.tecplot_to_raw(tec, bc)
    assert np.array_equal(raw["node|node_type"], ref["node_type"].astype(np.int64))
    assert np.array_equal(raw["node|surf_mask"], ref["surf"])
    for k in ("cells_node", "cells_face", "cells_index"):       # the reader's own (pre-CCW) lists, bit for bit
        assert np.array_equal(raw[k], ref[k].astype(np.int64)), k
    sizes = np.bincount(np.bincount(raw["cells_index"]))
    assert sizes[5:].sum() > 0 and len(sizes) > 7, "the fixture must hold cells of more than 4 nodes"
    geo = meshgen.derive_geometry(raw)
    for k in ("cells_node", "cells_face", "cells_index"):       # after extract_mesh_state's CCW sort + regrouping by size
        assert np.array_equal(geo[k], ref["ccw_" + k].astype(np.int64)), k
    assert np.array_equal(geo["face|face_type"], ref["face_type"].astype(np.int64))
    assert np.array_equal(geo["face_node_x_base"], ref["face_node_x"].astype(np.int64))
    assert np.allclose(geo["cell|cells_area"], ref["cells_area"], rtol=1e-12, atol=0)
    assert np.allclose(geo["unit_norm_v"], ref["unit_norm_v"], rtol=0, atol=1e-14)
    # geometric invariants of the reference's own checks (parse_to_h5.py:430-472) on ragged cells
    S = geo["unit_norm_v"] * geo["face|face_area"][geo["cells_face"]]
    closure = np.zeros((geo["cell|centroid"].shape[0], 2))
    np.add.at(closure, geo["cells_index"], S)
    assert np.abs(closure).max() < 1e-12


def test_tecplot_round_trip(tmp_path):
    from gfv import ingest
    tec = ingest.read_tecplot(os.path.join(GOLD, "poly_small.dat"))
    n_cells = int(max(tec["left"].max(), tec["right"].max())) + 1
    out = str(tmp_path / "rt.dat")
    ingest.write_tecplot(out, tec["pos"], tec["face_node"] + 1, tec["left"] + 1, tec["right"] + 1, n_cells,
                         [("Line: all", tec["boundary_pos"])])
    again = ingest.read_tecplot(out)
    assert np.array_equal(again["face_node"], tec["face_node"]) and np.array_equal(again["left"], tec["left"])
    assert np.array_equal(again["right"], tec["right"]) and np.allclose(again["pos"], tec["pos"], rtol=0, atol=1e-12)


def test_reference_polygon_mesh_pipeline_and_oracle():
    """mesh_example/cylinder_flow_poly: 17 436 cells of 3 ... 9 nodes.  Reader arrays -> node types, ragged CCW lists, face
    types, stencil: fingerprints of the reference's arrays; oracle forward = the reference's outputs."""
    graphs, fx, mesh = cases.poly_cylinder(GOLD)
    assert graphs[0].x.shape[0] == 27778 and graphs[3].pos.shape[0] == 17436 and graphs[0].edge_index.shape[1] == 45214
    sizes = np.bincount(np.bincount(mesh["cells_index"]))
    assert {i: int(c) for i, c in enumerate(sizes) if c} == {3: 3, 4: 7145, 5: 2542, 6: 6206, 7: 1404, 8: 128, 9: 8}
    assert np.array_equal(mesh["node|node_type"], fx["node_type"].astype(np.int64))
    for k, name in (("cells_node", "cells_node"), ("cells_face", "cells_face"), ("cells_index", "cells_index"),
                    ("face_node_x", "face_node_x"), ("face_type", "face|face_type")):
        assert np.array_equal(cases.fingerprint(mesh[name]), fx["fp." + k]), k     # integer arrays: exact sums
    P = O.init_parameters(cases.WEIGHT_SEED)
    og = tuple(g.clone() for g in graphs)
    with torch.no_grad():
        out = O.model_forward(P, O.new_normalizer_buffers(), og)
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        a, b = out[i].numpy().astype(np.float64), fx[key].astype(np.float64)
        assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), key   # (moments rebuilt by gfv.meshgen: last-ulp input noise)
