"""CPU: the oracle (oracle/fvgn_oracle.py) reproduces the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  Tolerance: 1e-5 relative fp32 (BASELINE.json north_star); the generator log
(tests/golden/make_golden.log) shows the forward agreeing bit-for-bit in the build container."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import fvgn_oracle as O

TOL = 1e-5


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference_golden(name, golden_dir):
    fx = np.load(os.path.join(golden_dir, f"{name}.npz"))
    graphs = cases.make_graphs(name)
    P = O.init_parameters(cases.WEIGHT_SEED)
    names = list(P)
    assert names == list(fx["param_names"])
    # inputs regenerated identically (fingerprints committed with the golden vectors)
    assert _rel(cases.fingerprint(graphs[0].x.numpy()), fx["in.x"]) < 1e-12
    assert _rel(cases.fingerprint(graphs[0].edge_index.numpy()), fx["in.edge_index"]) == 0
    assert _rel(cases.fingerprint(graphs[1].face_node_x.numpy()), fx["in.face_node_x"]) == 0
    assert _rel(cases.fingerprint(graphs[1].A_node_to_node.numpy()), fx["in.A"]) < 1e-12
    w = np.stack([cases.fingerprint(P[k].numpy()) for k in names])
    assert _rel(w, fx["in.weights"]) < 1e-12

    buffers = O.new_normalizer_buffers()
    Pg = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    out, inter = O.model_forward(Pg, buffers, graphs, return_intermediates=True)
    loss = O.training_loss(out)
    for key, val in (("loss_cont", out[0]), ("loss_mom_x", out[1]), ("loss_mom_y", out[2]), ("loss_press", out[3]),
                     ("uvp_node", out[4]), ("uvp_cell", out[5]), ("x_norm", graphs[0].x),
                     ("edge_attr", graphs[0].edge_attr), ("dec", inter["dec"])):
        assert _rel(val.detach().numpy(), fx[key]) < TOL, key
    assert abs(float(loss) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))
    for k in inter:
        if "tap_fp." + k in fx:
            assert _rel(cases.fingerprint(inter[k].detach().numpy())[:2], fx["tap_fp." + k][:2]) < TOL, k

    grads = dict(zip(names, torch.autograd.grad(loss, [Pg[k] for k in names], allow_unused=True)))
    gfp = fx["grad_fp"]
    gscale = np.nanmax(gfp[:, 1])  # largest gradient L2 norm in the model: noise floor for ~zero gradients
    for i, k in enumerate(names):
        if np.isnan(gfp[i, 0]):
            assert grads[k] is None, k  # ln_1 / Attn.temperature are unused
            continue
        mine = cases.fingerprint(grads[k].numpy())
        assert abs(mine[1] - gfp[i, 1]) < 1e-4 * gfp[i, 1] + 1e-7 * gscale, k
    for k in fx.files:
        if k.startswith("grad."):
            g = grads[k[5:]].numpy()
            assert np.abs(g - fx[k]).max() < 1e-4 * np.abs(fx[k]).max() + 1e-7 * gscale, k
    # Normalizer buffers after one accumulation (utils/normalization.py:52-66)
    assert _rel(buffers["acc_count"].numpy(), fx["norm.acc_count"]) < 1e-6
    assert _rel(buffers["acc_sum"].numpy(), fx["norm.acc_sum"]) < 1e-5
    assert _rel(buffers["acc_sum_squared"].numpy(), fx["norm.acc_sum_squared"]) < 1e-5


def test_wlsq_known_answer(golden_dir):
    """The reference's only known-answer material (grad_rec_acc_test.py:87-181): WLSQ gradient of the analytic field."""
    import math
    fx = np.load(os.path.join(golden_dir, "wlsq_analytic.npz"))
    meshes, _ = cases.make_meshes("cyl_cavity_b2")
    m = meshes[0]
    pos = torch.from_numpy(m["node|pos"]).float()
    x, y = pos[:, 0:1], pos[:, 1:2]
    phi = (1.0 + 0.01 * torch.sin(5 * math.pi * x) + 0.01 * torch.sin(5 * math.pi * y)
           + 0.01 * torch.cos(5 * math.pi * x * y))
    t = lambda k: torch.from_numpy(m[k])
    g = O.node_based_WLSQ(phi, t("face_node_x"), t("support_edge"), t("A_node_to_node"),
                          t("single_B_node_to_node"), t("extra_B_node_to_node"))
    assert _rel(g.numpy(), fx["grad"]) < TOL


def test_oracle_on_reference_example_mesh(golden_dir):
    """mesh_example/cylinder_flow_full_tri (7 798 nodes / 15 074 cells, COMSOL): raw arrays -> gfv.meshgen -> oracle,
    against the outputs the reference produced on its own mesh pipeline (tests/golden/make_real_mesh_golden.py)."""
    graphs, fx = cases.real_cylinder(golden_dir)
    assert graphs[0].x.shape[0] == 7798 and graphs[3].pos.shape[0] == 15074 and graphs[0].edge_index.shape[1] == 22872
    out = O.model_forward(O.init_parameters(cases.WEIGHT_SEED), O.new_normalizer_buffers(), graphs)
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert _rel(out[i].detach().numpy(), fx[key]) < TOL, key
    assert abs(float(O.training_loss(out)) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))


@pytest.mark.parametrize("name", ["real_cavity101", "real_poisson_quad_tri", "real_naca0012"])
def test_oracle_on_more_reference_example_meshes(golden_dir, name):
    """Three more of the reference's own meshes: the 101 x 101 lid-driven cavity (all quads, pressure
    point), the Poisson cavity on a quad + tri mesh (theta_PDE switches the continuity / convection / pressure terms off)
    and the NACA0012 far-field mesh (30 684 cells, quad boundary layer): raw reader arrays -> gfv.meshgen -> oracle against
    the outputs the reference produced with its own mesh pipeline (make_real_mesh_golden.py; the generator also checks
    every derived mesh array against the reference pipeline's and found the oracle's forward bit-identical)."""
    graphs, fx, _ = cases.real_mesh(name, golden_dir)
    n, e, c, _what = cases.REAL_MESHES[name]
    assert (graphs[0].x.shape[0], graphs[0].edge_index.shape[1], graphs[3].pos.shape[0]) == (n, e, c)
    out = O.model_forward(O.init_parameters(cases.WEIGHT_SEED), O.new_normalizer_buffers(), graphs)
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert _rel(out[i].detach().numpy(), fx[key]) < TOL, key
    assert abs(float(O.training_loss(out)) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))


def _check_prep_against_reference(mesh, fp, name):
    """k-hop stencil (integers: exact) and WLSQ moment arrays (floats: fingerprints to 1e-6) of a finished mesh against what the
    REFERENCE's pipeline produced on the same example mesh (tests/golden/make_prep_golden.py)."""
    fx = np.asarray(mesh["face_node_x"], dtype=np.int64)
    assert np.array_equal(cases.int_fingerprint(fx), fp[name + ".stencil"]), name
    assert np.array_equal(fx[:, :64], fp[name + ".stencil_head"].astype(np.int64)), name
    for k in ("A_node_to_node", "single_B_node_to_node", "extra_B_node_to_node"):
        mine, want = cases.fingerprint(np.asarray(mesh[k], dtype=np.float64)), fp[name + "." + k]
        assert np.all(np.abs(mine - want) <= 1e-6 * np.abs(want[1]) + 1e-12), (name, k, mine, want)


@pytest.mark.parametrize("name", sorted(cases.REAL_MESHES))
def test_host_preprocessing_matches_reference_generated_fingerprints(golden_dir, name):
    """Row f2 on the host: gfv.meshgen's k-hop stencil and moment matrices against the reference's own (these were
    pinned only transitively)."""
    fp = np.load(os.path.join(golden_dir, "real_prep_fp.npz"))
    _g, _fx, mesh = cases.real_mesh(name, golden_dir)
    _check_prep_against_reference(mesh, fp, name)


def test_oracle_transfvgn_v1_matches_reference(golden_dir):
    """net='TransFVGN_v1' (one processor).  Fixture = the reference itself run with that net
    (tests/golden/make_golden_v1.py); the oracle's forward agrees bit for bit there (make_golden_v1.log)."""
    fx = np.load(os.path.join(golden_dir, "v1_cyl_cavity_b2.npz"))
    hyper = {"net": "TransFVGN_v1"}
    shapes = O.parameter_shapes(hyper)
    assert list(shapes) == [str(k) for k in fx["param_names"]] and len(shapes) == 91
    assert sum(int(np.prod(v)) for v in shapes.values()) == 642611
    graphs = cases.make_graphs("cyl_cavity_b2")
    out = O.model_forward(O.init_parameters(cases.WEIGHT_SEED, hyper=hyper), O.new_normalizer_buffers(), graphs, hyper=hyper)
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert _rel(out[i].detach().numpy(), fx[key]) < TOL, key
    assert abs(float(O.training_loss(out)) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))


def test_oracle_hidden_size_64_matches_reference(golden_dir):
    """`--hidden_size 64` (utils/get_param.py:69).  Fixture = the reference itself run at that width
    (tests/golden/make_golden_hidden.py); the oracle's forward agrees bit for bit there (make_golden_hidden.log) - the pin
    behind tests/test_model_gpu.py::test_hidden_size_below_128_matches_oracle."""
    fx = np.load(os.path.join(golden_dir, "hidden64_cyl_cavity_b2.npz"))
    hyper = {"hidden_size": 64}
    shapes = O.parameter_shapes(hyper)
    assert list(shapes) == [str(k) for k in fx["param_names"]]
    assert sum(int(np.prod(v)) for v in shapes.values()) == 299619
    graphs = cases.make_graphs("cyl_cavity_b2")
    out = O.model_forward(O.init_parameters(cases.WEIGHT_SEED, hyper=hyper), O.new_normalizer_buffers(), graphs, hyper=hyper)
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert _rel(out[i].detach().numpy(), fx[key]) < TOL, key
    assert abs(float(O.training_loss(out)) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))


def test_oracle_non_conserved_form_matches_reference(golden_dir):
    """conserved_form=False (FVscheme.py:276-511).  Fixture = the reference itself run with that
    switch (tests/golden/make_golden_nc.py); the oracle's forward agrees bit for bit there."""
    fx = np.load(os.path.join(golden_dir, "nc_cyl_cavity_b2.npz"))
    hyper = {"conserved_form": False}
    graphs = cases.make_graphs("cyl_cavity_b2")
    out = O.model_forward(O.init_parameters(cases.WEIGHT_SEED), O.new_normalizer_buffers(), graphs, hyper=hyper)
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert _rel(out[i].detach().numpy(), fx[key]) < TOL, key
    assert abs(float(O.training_loss(out)) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))


@pytest.mark.parametrize("order", ["1st", "3rd", "4th"])
def test_oracle_wlsq_orders_match_reference(order, golden_dir):
    """WLSQ reconstruction orders 1st / 3rd / 4th (2 / 9 / 14 Taylor terms, FVorder.py:23-72,
    FVgrad.py:299-312).  Fixture = the reference itself run with params.order on meshes whose moment matrices are built
    for that order (tests/golden/make_golden_orders.py); the oracle's forward agrees bit for bit there."""
    fx = np.load(os.path.join(golden_dir, f"order_{order}_cyl_cavity_b2.npz"))
    graphs = cases.make_graphs("cyl_cavity_b2", order=order)
    assert graphs[1].A_node_to_node.shape[-1] == {"1st": 2, "3rd": 9, "4th": 14}[order]
    out = O.model_forward(O.init_parameters(cases.WEIGHT_SEED), O.new_normalizer_buffers(), graphs, hyper={"order": order})
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert _rel(out[i].detach().numpy(), fx[key]) < TOL, key
    assert abs(float(O.training_loss(out)) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))
