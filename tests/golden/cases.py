"""Deterministic parity cases shared by the golden generator and the tests (inputs are re-generated, not stored)."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "gen-fvgn-steady_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

from gfv import meshgen  # noqa: E402
from gfv.graph import build_batch  # noqa: E402

CASES = {
    # name: list of (raw-mesh factory, kwargs, U override, field seed)
    "cavity_mixed_b1": [("raw_quad_cavity", dict(n=8, jitter=0.15, tri_fraction=0.25, seed=11), None, 1)],
    "cyl_cavity_b2": [
        ("raw_tri_channel_cylinder", dict(nx=44, ny=8, quad_fraction=0.2, seed=12), 0.2714, 2),
        ("raw_quad_cavity", dict(n=6, jitter=0.1, tri_fraction=0.0, seed=13), 1.0, 3),
    ],
    "poisson_b1": [("raw_poisson_cavity", dict(n=6, seed=14), None, 4)],
    "cyl_b3": [
        ("raw_tri_channel_cylinder", dict(nx=30, ny=6, quad_fraction=0.0, seed=21), 0.15, 5),
        ("raw_tri_channel_cylinder", dict(nx=36, ny=6, quad_fraction=0.3, seed=22), 0.25, 6),
        ("raw_tri_channel_cylinder", dict(nx=28, ny=7, quad_fraction=0.0, seed=23), 0.30, 7),
    ],
}

WEIGHT_SEED = 0


def make_meshes(name, order="2nd"):
    meshes, fields = [], []
    for fac, kw, U, fseed in CASES[name]:
        m = meshgen.finish_mesh(getattr(meshgen, fac)(**kw), U=U, order=order)
        meshes.append(m)
        fields.append(meshgen.random_fields(m, seed=fseed))
    return meshes, fields


def make_graphs(name, device="cpu", order="2nd"):
    meshes, fields = make_meshes(name, order)
    return build_batch(meshes, fields, device=device)


def projection(n):
    """Fixed pseudo-random direction used to fingerprint large tensors."""
    i = np.arange(n, dtype=np.float64)
    return np.cos(0.37 * i + 0.11 * np.sqrt(i + 1.0))


def fingerprint(a):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    return np.array([a.sum(), np.sqrt((a * a).sum()), (a * projection(a.size)).sum()])


# The reference's own example meshes (mesh_example/..., COMSOL .mphtxt): raw reader arrays + the reference's outputs on
# them, committed as data by make_real_mesh_golden.py.  name: (nodes, faces, cells, what it is)
REAL_MESHES = {
    "real_cylinder": (7798, 22872, 15074, "cylinder_flow_full_tri (all triangles, parabolic inlet)"),
    "real_cavity101": (10404, 20604, 10201, "lid_driven_cavity_101x101-Re=100 (all quads, pressure point)"),
    "real_poisson_quad_tri": (2497, 6192, 3696, "poisson/cavity_poisson_quad_tri (Poisson: continuity/convection/grad_p off)"),
    "real_naca0012": (16861, 47545, 30684, "airfoil_L=1/farfield_NACA0012_with_quad_bc (tri + quad boundary layer, unsteady=1)"),
}


def int_fingerprint(a):
    """[columns, sum of row 0, sum of row 1, position-dependent checksum] of a [2, n] int64 index array."""
    a = np.asarray(a, dtype=np.int64)
    pos = np.arange(a.shape[1], dtype=np.int64) % 8191
    return np.array([a.shape[1], int(a[0].sum()), int(a[1].sum()), int(((a[0] * 31 + a[1] * 17 + pos) % 1000003).sum())], dtype=np.int64)


def real_mesh(name, golden_dir=None, device=None):
    """One of REAL_MESHES -> (graphs, fixture, mesh dict).  device: prepare the stencil / moments with gfv.device_prep there."""
    import json
    golden_dir = golden_dir or os.path.dirname(os.path.abspath(__file__))
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    raw = {k[4:]: (fx[k].astype(np.int64) if fx[k].dtype.kind == "i" else fx[k]) for k in fx.files
           if k.startswith("raw.") and k != "raw.bc"}
    raw["bc"] = json.loads(str(fx["raw.bc"]))
    mesh = meshgen.finish_mesh(raw, device=device)
    return build_batch([mesh], [fx["field"]]), fx, mesh


def real_cylinder(golden_dir=None):
    """The reference's own example mesh mesh_example/cylinder_flow_full_tri (raw reader arrays committed as data in
    real_cylinder.npz by make_real_mesh_golden.py) -> (graphs, fixture)."""
    import json
    golden_dir = golden_dir or os.path.dirname(os.path.abspath(__file__))
    fx = np.load(os.path.join(golden_dir, "real_cylinder.npz"))
    raw = {k[4:]: (fx[k].astype(np.int64) if fx[k].dtype.kind == "i" else fx[k]) for k in fx.files
           if k.startswith("raw.") and k != "raw.bc"}
    raw["bc"] = json.loads(str(fx["raw.bc"]))
    mesh = meshgen.finish_mesh(raw)
    return build_batch([mesh], [fx["field"]]), fx


def poly_cylinder(golden_dir=None):
    """The reference's polygon example mesh mesh_example/cylinder_flow_poly (Tecplot FEPolygon, cells of 3 ... 9 nodes; raw
    reader arrays committed as data in poly_cylinder.npz by make_golden_poly.py) -> (graphs, fixture)."""
    import json
    from gfv import ingest
    golden_dir = golden_dir or os.path.dirname(os.path.abspath(__file__))
    fx = np.load(os.path.join(golden_dir, "poly_cylinder.npz"))
    tec = {"pos": fx["tec.pos"], "face_node": fx["tec.face_node"].astype(np.int64), "left": fx["tec.left"].astype(np.int64),
           "right": fx["tec.right"].astype(np.int64), "boundary_pos": fx["tec.boundary_pos"]}
    bcd = json.loads(str(fx["raw.bc"]))
    bc_json = {"stencil|khops": bcd["stencil|khops"], "sigma": bcd["sigma"], "inlet_type": bcd["inlet_type"],
               "theta_PDE": dict(bcd["theta_PDE"], inlet=[bcd["U"]], rho=[bcd["rho"]], mu=[bcd["mu"]], source=[bcd["source"]],
                                 aoa=[bcd["aoa"]], dt=bcd["dt"], L=bcd["L"])}
    raw = ingest.tecplot_to_raw(tec, bc_json)
    raw["bc"] = bcd
    mesh = meshgen.finish_mesh(raw)
    return build_batch([mesh], [fx["field"]]), fx, mesh


# Stated tolerances of the reduced-precision product form (gfv_set_f16split(2): BASELINE configs 3 / 5) against the fp32
# oracle, relative: fields (of the field's maximum), each residual loss, the scalar log-loss, all gradients norm-wise
# (tests/test_model_gpu.py::test_reduced_precision_form_against_the_fp32_oracle, tests/test_fullsize_gpu.py)
LOWP_TOL = dict(field=2e-4, losses=2e-3, logloss=1e-5, grad_norm=2e-3)
# ... and of the bf16 single-product form (gfv_set_f16split(3): config 3's "bf16 MLP GEMMs on MFMA" to the letter; 8
# significand bits where the fp16 form has 11)
BF16_TOL = dict(field=2e-3, losses=2e-2, logloss=1e-4, grad_norm=2e-2)
