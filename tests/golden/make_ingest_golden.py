"""Golden data for SURVEY.md row f3 (mesh ingest).

1. Writes a small COMSOL-format mesh (tests/golden/ingest_small.mphtxt + ingest_small_BC.json; synthetic tri + quad
   channel with a cylinder, boundary edges grouped into entities like mesh_example/cylinder_flow_full_tri) with
   gfv.ingest.write_mphtxt, runs the REFERENCE's reader on it (Extract_mesh/parse_comsol.py: Cosmol_manager.read_mesh_file,
   set_node_type, and the arrays handed to extract_mesh_state) and commits what it returns
   (tests/golden/ingest_small.npz).
2. Checks gfv.ingest against the reference's reader on the reference's own example files
   (mesh_example/cylinder_flow_full_tri, cylinder_flow_tri_quad - they cannot travel, so this check runs here only and
   its result is logged in make_ingest_golden.log).
Build container only (needs /root/reference):  python tests/golden/make_ingest_golden.py"""
from __future__ import annotations

import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refstubs"))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import ref_import  # noqa: E402

sys.path.insert(0, cases.ROOT)
from gfv import ingest, meshgen  # noqa: E402


def reference_reader(mphtxt, bc_dir):
    pc = importlib.import_module("Extract_mesh.parse_comsol")
    pc.write_vtp_file = lambda *a, **k: None
    pc.Cosmol_manager.save_to_vtu = lambda *a, **k: None
    out_dir = "/tmp/ingest_out"
    os.makedirs(out_dir, exist_ok=True)
    mgr = pc.Cosmol_manager(mesh_file=mphtxt, data_file=None, file_dir=bc_dir, case_name="x", path={"file_dir": out_dir, "case_name": "x"})
    mgr.file_dir = out_dir
    captured = {}

    def capture(dataset, path=None):   # the arrays the reader hands to extract_mesh_state (before its CCW pass)
        for k, v in dataset.items():
            captured[k] = np.asarray(v).copy()
        raise StopIteration

    pc.extract_mesh_state = capture
    try:
        mgr.extract_mesh()
    except StopIteration:
        pass
    return mgr.mesh_file, captured


def compare(tag, mphtxt, bc_dir, bc_name="BC.json"):
    ref_file, ref = reference_reader(mphtxt, bc_dir)
    mine_file = ingest.read_mphtxt(mphtxt)
    assert set(mine_file) == set(ref_file), (set(mine_file), set(ref_file))
    assert np.array_equal(mine_file["vertices"], ref_file["vertices"])
    for t in mine_file:
        if t == "vertices":
            continue
        assert np.array_equal(mine_file[t]["Elements"].reshape(-1), np.asarray(ref_file[t]["Elements"]).reshape(-1)), t
        assert np.array_equal(mine_file[t]["Geometric entity indices"], ref_file[t]["Geometric entity indices"]), t
    raw = ingest.comsol_to_raw(mine_file, json.load(open(os.path.join(bc_dir, bc_name))))
    for k_ref, k in (("node|pos", "node|pos"), ("node|node_type", "node|node_type"), ("face|face_node", "face|face_node"),
                     ("cells_node", "cells_node"), ("cells_index", "cells_index"), ("cells_face", "cells_face")):
        a, b = np.asarray(ref[k_ref]).reshape(-1), np.asarray(raw[k]).reshape(-1)
        assert a.shape == b.shape and np.array_equal(a.astype(np.float64), b.astype(np.float64)), (tag, k)
    print(f"   {tag}: gfv.ingest == reference reader  (N {raw['node|pos'].shape[0]}, E {raw['face|face_node'].shape[1]}, "
          f"C {int(raw['cells_index'].max()) + 1}; types {sorted(t for t in mine_file if t != 'vertices')})")
    return ref


def main():
    ref_import.install()
    ref_import.reference_modules()
    ref_import.reference_mesh_modules()
    # ---- 1. small synthetic file ---------------------------------------------------------------------------
    raw = meshgen.raw_tri_channel_cylinder(nx=22, ny=6, quad_fraction=0.25, seed=77)
    pos = raw["node|pos"]
    ci, cn = raw["cells_index"], raw["cells_node"]
    counts = np.bincount(ci)
    starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
    tris = np.stack([cn[s:s + 3] for s, c in zip(starts, counts) if c == 3])
    quads = np.stack([cn[s:s + 4] for s, c in zip(starts, counts) if c == 4])
    rng = np.random.default_rng(5)
    quads = np.stack([q[rng.permutation(4)] if i % 3 == 0 else q for i, q in enumerate(quads)])   # reader must re-order
    be = meshgen._boundary_edges([tris, quads])
    mid = 0.5 * (pos[be[:, 0]] + pos[be[:, 1]])
    ent = np.full(be.shape[0], 4, dtype=np.int64)                 # 0-based entity ids in the file; +1 in the GUI / BC.json
    ent[mid[:, 1] < 1e-9] = 1                                     # bottom wall -> entity 2
    ent[mid[:, 1] > 0.41 - 1e-9] = 2                              # top wall    -> entity 3
    ent[mid[:, 0] < 1e-9] = 0                                     # inflow      -> entity 1
    ent[mid[:, 0] > 2.2 - 1e-9] = 3                               # outflow     -> entity 4
    cyl = (ent == 4)
    ent[cyl] = 4 + (np.arange(int(cyl.sum())) % 4)                # cylinder    -> entities 5-8
    corner = np.array([[0], [int(np.argmax(pos[:, 0] + pos[:, 1]))]])
    mph = os.path.join(HERE, "ingest_small.mphtxt")
    ingest.write_mphtxt(mph, pos, {"vtx": (corner, np.array([0, 1])), "edg": (be, ent), "tri": (tris, np.zeros(len(tris), int)),
                                   "quad": (quads, np.zeros(len(quads), int))})
    bc = {"inflow": [1], "wall": [2, 3, "5-8"], "outflow": [4], "pressure_point": None, "periodic": None, "surf": ["5-8"],
          "stencil|BC_extra_points": 4, "stencil|khops": 2,
          "theta_PDE": {"unsteady": 1, "continuity": 1, "convection": 1, "grad_p": 1, "inlet": [0.2, 0.1, 0.3], "rho": [1, 1, 1],
                        "mu": [0.001, 0.001, 0.001], "source": [0, 0, 0], "aoa": [0, 0, 0], "dt": 0.5, "L": 0.1, "Re_max": 30,
                        "Re_min": 2},
          "sigma": [1, 1, 1], "inlet_type": "parabolic", "init_field_type": "parabolic"}
    os.makedirs("/tmp/ingest_small", exist_ok=True)
    json.dump(bc, open(os.path.join(HERE, "ingest_small_BC.json"), "w"), indent=1)
    json.dump(bc, open("/tmp/ingest_small/BC.json", "w"))
    ref = compare("synthetic ingest_small.mphtxt", mph, "/tmp/ingest_small")
    np.savez_compressed(os.path.join(HERE, "ingest_small.npz"),
                        **{k: np.asarray(ref[k]) for k in ("node|pos", "node|node_type", "face|face_node", "cells_node",
                                                           "cells_index", "cells_face")})
    print("wrote ingest_small.mphtxt / ingest_small_BC.json / ingest_small.npz")
    # ---- 2. the reference's own example files (build container only) --------------------------------------
    for d, f in (("cylinder_flow_full_tri", "mesh_full_tri.mphtxt"), ("cylinder_flow_tri_quad", "mesh.mphtxt")):
        compare(f"mesh_example/{d}", f"/root/reference/mesh_example/{d}/{f}", f"/root/reference/mesh_example/{d}")


if __name__ == "__main__":
    main()
