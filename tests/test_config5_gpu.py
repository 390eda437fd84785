"""BASELINE.json config 5 (polygon mesh, unsteady inner iterations) on the HIP path.

* the reference's polygon example mesh (mesh_example/cylinder_flow_poly, cells of 3 ... 9 nodes, read by gfv.ingest's
  Tecplot reader from the committed raw arrays) through NNmodel: against the REFERENCE's outputs (fixture made by
  tests/golden/make_golden_poly.py) and against the oracle run in float64;
* the solve loop of solve_with_grad_GPU.py:133-197 - per time step `max_inner_steps` iterations of (restore x, re-arm the norm
  flags, forward, log-loss, backward, Adam) and then the time advance x[:, 0:3] <- prediction - on a polygon mesh: the drop-in
  module under torch.optim.Adam (the reference driver's own call sequence) and the fused TrainStep (eager and command-list
  replay), each against the oracle's loop."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import fvgn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _model(P, **kw):
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    m = NNmodel(default_params(**kw))
    sd = m.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    m.load_state_dict(sd)
    return m.cuda()


def test_reference_polygon_mesh_matches_reference_and_fp64_oracle():
    from test_fullsize_gpu import compare_to_fp64, check_gradients
    graphs, fx, mesh = cases.poly_cylinder(GOLD)
    sizes = np.bincount(np.bincount(mesh["cells_index"]))
    assert sizes[5:].sum() > 10000, "polygon cells (5 ... 9 nodes) dominate this mesh"
    P = O.init_parameters(cases.WEIGHT_SEED)
    report, (o64, o32, hip) = compare_to_fp64(graphs, P, "reference polygon mesh, 17 436 cells of 3 ... 9 nodes")
    keys = ("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")
    for key in keys + ("loss",):
        assert report[key][0] < TOL, (key, report[key])
    check_gradients(report, "reference polygon mesh")
    # and against what the reference itself returned on this mesh (fixture: the reference's default dataset_size = 100, i.e.
    # an accumulating Normalizer on the first forward; its fp32 pooling noise bounds loss_cont)
    model = _model(P)
    hg = tuple(g.clone().to("cuda") for g in graphs)
    hg[0].norm_uvp, hg[0].norm_global = True, True
    out = model(*hg)
    for i, key in enumerate(keys):
        r = rel(out[i], fx[key])
        assert r < (1e-4 if key == "loss_cont" else TOL), (key, r)
    loss = torch.mean(torch.log(1.0 * out[3] + 6e4 * out[0] + 5e4 * out[1] + 5e4 * out[2]))
    assert abs(float(loss) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))
    loss.backward()
    gfp = fx["grad_fp"]
    gscale = np.nanmax(gfp[:, 1])
    for i, (k, p_) in enumerate(model.named_parameters()):
        if np.isnan(gfp[i, 0]):
            assert p_.grad is None, k
            continue
        mine = cases.fingerprint(p_.grad.cpu().numpy())
        # fingerprints (sum, norm, fixed projection) of every gradient tensor against the reference's
        assert abs(mine[1] - gfp[i, 1]) < 2e-3 * gfp[i, 1] + 1e-5 * gscale, (k, mine, gfp[i])


def _small_polygon_graphs():
    import json
    from gfv import ingest, meshgen
    from gfv.graph import build_batch
    bc = {"stencil|khops": 2, "sigma": [1, 1, 1], "inlet_type": "parabolic",
          "theta_PDE": {"unsteady": 1, "continuity": 1, "convection": 1, "grad_p": 1, "inlet": [0.2], "rho": [1], "mu": [0.001],
                        "source": [0], "aoa": [0], "dt": 0.5, "L": 0.1}}
    raw = ingest.load_tecplot_mesh_from(ingest.read_tecplot(os.path.join(GOLD, "poly_small.dat")), bc)
    mesh = meshgen.finish_mesh(raw)
    assert np.bincount(np.bincount(mesh["cells_index"]))[5:].sum() > 40
    # start from the reference's initial field (Load_mesh.py:80-131: boundary-conditioned parabolic profile, p = 0)
    return build_batch([mesh], [mesh["init_uvp"].astype(np.float32)])


INNER, STEPS = 20, 2   # params.max_inner_steps of the reference (get_param.py), two time steps


def _oracle_loop(graphs, P0):
    P = {k: v.clone() for k, v in P0.items()}
    buffers, state = O.new_normalizer_buffers(), {}
    g = tuple(x.clone() for x in graphs)
    losses = []
    for _ in range(STEPS):
        backup = g[0].x.clone()
        for _ in range(INNER):
            g[0].x = backup.clone()
            loss, out, _ = O.train_step(P, buffers, g, state, hyper={"dataset_size": 1})
            losses.append(float(loss))
        g[0].x = torch.cat((out[4].detach(), backup[:, 3:]), 1)        # solve_with_grad_GPU.py:197
    return P, losses, g[0].x.clone(), out[4].detach()


def test_unsteady_inner_iterations_and_time_advance_drop_in_driver():
    """The reference driver's own call sequence (solve_with_grad_GPU.py:133-197) on the drop-in module with torch's Adam."""
    graphs = _small_polygon_graphs()
    P0 = O.init_parameters(cases.WEIGHT_SEED)
    Po, lo, xo, uvpo = _oracle_loop(graphs, P0)
    model = _model(P0, dataset_size=1)
    opt = torch.optim.Adam(model.parameters(), lr=5e-5)
    gn, gx, ge, gc, gi = tuple(x.clone().to("cuda") for x in graphs)
    lh = []
    for _ in range(STEPS):
        backup = gn.x.clone()
        for _ in range(INNER):
            gn.x = backup.clone()
            gn.norm_uvp, gn.norm_global = True, True
            opt.zero_grad()
            lc, lmx, lmy, lp, uvp_node, uvp_cell = model(graph_node=gn, graph_node_x=gx, graph_edge=ge, graph_cell=gc,
                                                         graph_Index=gi, is_training=True)
            loss = torch.mean(torch.log(1.0 * lp + 6e4 * lc + 5e4 * lmx + 5e4 * lmy))
            loss.backward()
            opt.step()
            lh.append(float(loss))
        gn.x = torch.cat((uvp_node.detach(), backup[:, 3:]), 1)
    _compare_loops(model, Po, lh, lo, gn.x, xo, "drop-in NNmodel + torch.optim.Adam")


@pytest.mark.parametrize("mode", [False, "list"])
def test_unsteady_inner_iterations_and_time_advance_trainstep(mode):
    from gfv.trainer import TrainStep
    graphs = _small_polygon_graphs()
    P0 = O.init_parameters(cases.WEIGHT_SEED)
    Po, lo, xo, uvpo = _oracle_loop(graphs, P0)
    model = _model(P0, dataset_size=1)
    ts = TrainStep(model, tuple(x.clone().to("cuda") for x in graphs), use_graph=mode)
    lh = []
    for _ in range(STEPS):
        for _ in range(INNER):
            ts.step()
            lh.append(float(ts.loss))
        ts.advance_time()
    _compare_loops(model, Po, lh, lo, ts.x_backup, xo, f"TrainStep(use_graph={mode!r})")


def _compare_loops(model, Po, lh, lo, x_final, x_oracle, label):
    # 40 Adam steps: Adam turns every gradient, however small, into a step of ~lr, so rounding-level differences of tiny
    # gradients move single weights by up to 2 lr per step; the trajectories are compared through what they produce
    lerr = max(abs(a - b) / abs(b) for a, b in zip(lh, lo))
    xerr = rel(x_final[:, 0:3], x_oracle[:, 0:3])
    perr = max(float((p.detach().cpu() - Po[k]).abs().max()) for k, p in model.named_parameters())
    print(f"[{label}] worst loss deviation over {len(lo)} inner iterations {lerr:.2e}; advanced state {xerr:.2e}; "
          f"worst parameter |delta| {perr:.2e} (lr 5e-5); loss {lo[0]:.4f} -> {lo[-1]:.4f}")
    assert lo[-1] < lo[0], "the inner iterations must lower the PDE loss"
    assert lerr < 1e-4, lerr
    assert xerr < 1e-3, xerr
    assert perr < 40 * 2 * 5e-5
