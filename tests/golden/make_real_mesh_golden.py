"""Golden vectors on one of the reference's OWN example meshes (mesh_example/cylinder_flow_full_tri: COMSOL mesh,
7 798 nodes / 15 074 tri cells): the reference's mesh pipeline (Cosmol_manager.extract_mesh -> transform_mesh) and the
reference's NNmodel forward / backward are run here, in the build container; the derived mesh arrays (data, not code)
and the reference's outputs are committed as tests/golden/real_cylinder.npz.  Run: python tests/golden/make_real_mesh_golden.py
"""
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
from gfv.graph import build_batch  # noqa: E402
import make_golden as MG  # noqa: E402

MESH_ROOT = "/root/reference/mesh_example"
# (directory under mesh_example, mesh file, fixture name, random seed of the PDE-parameter draw, field seed)
MESHES = {
    "real_cylinder": ("cylinder_flow_full_tri", "mesh_full_tri.mphtxt", 7, 99),
    "real_cavity101": ("lid_driven_cavity/lid_driven_cavity_101x101-Re=100", "mesh.mphtxt", 8, 100),
    "real_poisson_quad_tri": ("poisson/cavity_poisson_quad_tri", "mesh_tri.mphtxt", 9, 101),
    "real_naca0012": ("airfoil_L=1/farfield_NACA0012_with_quad_bc", "mesh_2.mphtxt", 10, 102),
}


def reference_arrays(fixture):
    """The reference's own mesh pipeline on one of its example meshes: (R, md, captured, mesh, bc, th, params, init_uvp) -
    md holds every array `transform_mesh` produced (what make_prep_golden.py fingerprints)."""
    import importlib
    import json
    sub, mesh_file, pde_seed, field_seed = MESHES[fixture]
    MESH_DIR = os.path.join(MESH_ROOT, sub)
    case_name = os.path.basename(sub)
    R = ref_import.reference_modules()
    M = ref_import.reference_mesh_modules()
    pc = importlib.import_module("Extract_mesh.parse_comsol")
    pc.write_vtp_file = lambda *a, **k: None
    pc.Cosmol_manager.save_to_vtu = lambda *a, **k: None
    path = {"file_dir": "/tmp/real_mesh_out/" + fixture, "case_name": case_name}
    os.makedirs(path["file_dir"], exist_ok=True)
    mgr = pc.Cosmol_manager(mesh_file=f"{MESH_DIR}/{mesh_file}", data_file=None, file_dir=MESH_DIR,
                            case_name=case_name, path=path)
    mgr.file_dir = path["file_dir"]
    captured = {}
    orig_ems = pc.extract_mesh_state

    def capture(dataset, path=None):  # keep the reader's raw (pre-CCW) arrays
        for k in ("cells_node", "cells_face", "cells_index"):
            captured[k] = np.asarray(dataset[k]).astype(np.int64).reshape(-1).copy()
        return orig_ems(dataset, path=path)

    pc.extract_mesh_state = capture
    mesh = dict(mgr.extract_mesh())
    bc = json.load(open(f"{MESH_DIR}/BC.json"))
    mesh.update(bc)
    mesh["case_name"] = case_name
    mesh["theta_PDE_bak"] = mesh["theta_PDE"]
    th = mesh["theta_PDE_bak"]
    mesh["theta_PDE_list"] = R.get_param.generate_combinations(
        U_range=th["inlet"], rho_range=th["rho"], mu_range=th["mu"], source_range=th["source"], aoa_range=th["aoa"],
        dt=th["dt"], L=th["L"], Re_max=th["Re_max"], Re_min=th["Re_min"])
    random.seed(pde_seed)
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
    return R, md, captured, mesh, bc, th, params, init_uvp


def generate(fixture):
    import json
    _sub, _mesh_file, _pde_seed, field_seed = MESHES[fixture]
    R, md, captured, mesh, bc, th, params, init_uvp = reference_arrays(fixture)
    rng = np.random.default_rng(field_seed)
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
    out = model(graph_node=gn, graph_node_x=gx, graph_edge=ge, graph_cell=gc, graph_Index=gi, is_training=True)
    lc, lmx, lmy, lp, uvp_node, uvp_cell = out
    loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lmx + params.loss_mom * lmy))
    loss.backward()
    names = list(P0)
    grads = {k: p.grad for k, p in model.named_parameters()}

    # oracle on the same inputs
    og = tuple(g.clone() for g in graphs)
    oout = O.model_forward({k: v.clone() for k, v in P0.items()}, O.new_normalizer_buffers(), og)
    print(fixture, "N", md["node|pos"].shape[0], "E", md["face|face_node"].shape[1], "C", md["cell|centroid"].shape[0],
          "Ex", md["face_node_x"].shape[1])
    print("loss ref", float(loss))
    for nm, a, b in (("loss_cont", oout[0], lc), ("loss_mom_x", oout[1], lmx), ("loss_mom_y", oout[2], lmy),
                     ("loss_press", oout[3], lp), ("uvp_node", oout[4], uvp_node), ("uvp_cell", oout[5], uvp_cell)):
        print("  oracle vs reference", nm, MG.rel(a, b))

    # Store only the RAW mesh (as the COMSOL reader emits it) + the sampled PDE parameters; every derived array is
    # re-derived at test time by gfv/meshgen.py - checked here, array by array, against the reference's own pipeline.
    from gfv import meshgen
    U = float(mesh["mean_u"])
    raw = {k: md[k] for k in ("node|pos", "node|node_type", "face|face_node")}
    raw.update(captured)  # the reader's own (pre-CCW) cell lists
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
    save = {"raw.node|pos": md["node|pos"], "raw.node|node_type": md["node|node_type"].astype(np.int8),
            "raw.face|face_node": md["face|face_node"].astype(np.int32), "raw.cells_node": captured["cells_node"].astype(np.int32),
            "raw.cells_face": captured["cells_face"].astype(np.int32), "raw.cells_index": captured["cells_index"].astype(np.int32),
            "raw.bc": np.array(json.dumps(bcd))}
    save["field"] = field
    save.update({"loss": np.float64(loss.item()), "loss_cont": lc.detach().numpy(), "loss_mom_x": lmx.detach().numpy(),
                 "loss_mom_y": lmy.detach().numpy(), "loss_press": lp.detach().numpy(),
                 "uvp_node": uvp_node.detach().numpy(), "uvp_cell": uvp_cell.detach().numpy(),
                 "grad_fp": np.stack([cases.fingerprint(grads[k].numpy()) if grads[k] is not None else np.full(3, np.nan)
                                      for k in names])})
    np.savez_compressed(os.path.join(HERE, fixture + ".npz"), **save)
    print("saved", os.path.getsize(os.path.join(HERE, fixture + ".npz")) / 1e6, "MB")


if __name__ == "__main__":
    for name in (sys.argv[1:] or list(MESHES)):
        generate(name)
