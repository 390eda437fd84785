"""The BASELINE.json configurations that earlier rounds timed or listed but did not test.

* config 4's per-GPU share - a batch of EIGHT 50 000-cell cylinder meshes in one step: block-diagonality (every member's
  losses and fields = that mesh alone; the batch gradient = the mean of the members' gradients), float64-oracle parity
  on one member, and 3 training steps bit-identical in the three launch modes;
* config 2 - a 5 041-cell lid-driven cavity (71 x 71 quads), fp32 product form and the default form, forward + backward
  against the float64 oracle and 3 training steps against the oracle's;
* three more of the reference's own meshes (lid_driven_cavity_101x101, poisson/cavity_poisson_quad_tri,
  airfoil_L=1/farfield_NACA0012: raw reader arrays committed by tests/golden/make_real_mesh_golden.py) against the
  REFERENCE's outputs and against the float64 oracle;
* config 5's loop - one time step of 20 inner iterations + the time advance on the NACA0012 mesh - against the oracle's."""
import numpy as np
import pytest
import torch

import cases
from oracle import fvgn_oracle as O
from test_fullsize_gpu import _loss, _mesh, _model, check_gradients, compare_to_fp64, global_grad_error, hip_run, rel

pytestmark = pytest.mark.gpu
TOL = 1e-5
KEYS = ("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")


# ------------------------------------------------------------------------------------------------------------------
# batch of eight 50k-cell meshes
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def batch8():
    """Eight DIFFERENT 50k-cell cylinder meshes (different jitter seeds, inlet velocities from the reference's cylinder
    range 0.1 ... 0.3) - the stencil and the WLSQ moments are prepared on the GPU (gfv.device_prep), as bench.py does."""
    from gfv import meshgen
    nx, ny = meshgen.cylinder_grid_for_cells(50000)
    meshes, fields = [], []
    for i in range(8):
        raw = meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, jitter=0.2, seed=500 + i)
        m = meshgen.finish_mesh(raw, U=0.1 + 0.025 * i, device="cuda")
        meshes.append(m)
        fields.append(meshgen.random_fields(m, seed=600 + i))
    return meshes, fields


def _forward_backward(meshes, fields, P):
    from gfv.graph import build_batch
    model = _model(P)
    hg = tuple(g.clone().to("cuda") for g in build_batch(meshes, fields))
    hg[0].norm_uvp, hg[0].norm_global = True, True
    out = model(*hg)
    _loss(out).backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
    return [o.detach().cpu() for o in out], grads


def test_batch_of_eight_50k_meshes_is_block_diagonal_and_matches_fp64_oracle(batch8):
    from gfv.graph import build_batch
    meshes, fields = batch8
    assert all(49000 < m["cell|centroid"].shape[0] < 51000 for m in meshes)
    P = O.init_parameters(cases.WEIGHT_SEED)
    both, gboth = _forward_backward(meshes, fields, P)
    assert both[0].shape[0] == 8
    gsum, n0 = None, 0
    for i, (m, f) in enumerate(zip(meshes, fields)):
        alone, galone = _forward_backward([m], [f], P)
        for j in range(4):
            assert rel(both[j][i], alone[j][0]) < TOL, (i, KEYS[j])
        n = alone[4].shape[0]
        assert rel(both[4][n0:n0 + n], alone[4]) < TOL, i
        n0 += n
        gsum = galone if gsum is None else {k: gsum[k] + galone[k] for k in gsum}
    # loss = mean over the graphs of log(...) (pre_train_Adam.py:177-184): batch gradient = mean of the members'
    gmean = {k: v / 8.0 for k, v in gsum.items()}
    assert set(gmean) == set(gboth)
    assert global_grad_error(gboth, gmean) < TOL
    # float64-oracle parity of one member (number 5) THROUGH the batch: its losses and its rows of the fields
    k = 5
    one = build_batch([meshes[k]], [fields[k]])
    Pg = {kk: v.detach().double() for kk, v in P.items()}
    from test_fullsize_gpu import graphs_to
    buf = {kk: v.double() for kk, v in O.new_normalizer_buffers().items()}
    with torch.no_grad():
        o64 = O.model_forward(Pg, buf, graphs_to(one, torch.float64), hyper={"dataset_size": 1})
    for j in range(4):
        assert rel(both[j][k], o64[j][0]) < TOL, KEYS[j]
    lo = sum(m["node|pos"].shape[0] for m in meshes[:k])
    assert rel(both[4][lo:lo + one[0].x.shape[0]], o64[4]) < TOL


def test_batch_of_eight_50k_meshes_three_steps_in_three_launch_modes(batch8):
    """TrainStep on the B = 8 batch: eager launches, hipGraph replay and command-list replay give bit-identical parameters,
    moments, losses and fields after 3 steps, and the per-graph residuals of step 1 are those of the plain module."""
    from gfv.graph import build_batch
    from gfv.trainer import TrainStep
    meshes, fields = batch8
    graphs = build_batch(meshes, fields)
    P = O.init_parameters(cases.WEIGHT_SEED)
    finals, first = {}, {}
    for mode in (False, True, "list"):
        model = _model(P)
        ts = TrainStep(model, tuple(g.clone().to("cuda") for g in graphs), use_graph=mode)
        for i in range(3):
            ts.step()
            if i == 0:
                first[mode] = ts.losses.detach().clone()
        torch.cuda.synchronize()
        finals[mode] = torch.cat([t.reshape(-1) for pmv in ts.named_state().values() for t in pmv]
                                 + [ts.loss.reshape(-1), ts.losses.reshape(-1), ts.uvp_node.reshape(-1)]).clone()
        assert torch.isfinite(finals[mode]).all()
    assert torch.equal(finals[False], finals[True]), "hipGraph replay differs from eager"
    assert torch.equal(finals[False], finals["list"]), "command-list replay differs from eager"
    out = hip_run(graphs, P)[0]
    ref = torch.stack([o.reshape(-1) for o in out[:4]], 1)         # TrainStep.losses: [B, 4]
    assert first[False].numel() == ref.numel() and rel(first[False].reshape(ref.shape), ref) < TOL


# ------------------------------------------------------------------------------------------------------------------
# config 2: ~5k-cell lid-driven cavity, fp32
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("form", [0, 1])
def test_cavity_5041_cells_matches_fp64_oracle_and_trains_like_the_oracle(form):
    """71 x 71 quad cavity = 5 041 cells (BASELINE.json configs[1]; jittered interior nodes) in the fp32 product form
    (gfv_set_f16split(0): every chain product an fp32 FMA) and in the default split-fp16 form."""
    from gfv import lib as L
    from gfv import meshgen
    from gfv.graph import build_batch
    from gfv.trainer import TrainStep
    mesh = meshgen.finish_mesh(meshgen.raw_quad_cavity(n=71, jitter=0.1, seed=71))
    assert mesh["cell|centroid"].shape[0] == 5041
    graphs = build_batch([mesh], [meshgen.random_fields(mesh, seed=72)])
    P = O.init_parameters(cases.WEIGHT_SEED)
    lib = L.load()
    try:
        lib.gfv_set_f16split(form)
        report, _ = compare_to_fp64(graphs, P, f"cavity, 5 041 cells, product form {form}")
        for key in KEYS + ("loss",):
            assert report[key][0] < TOL, (key, report[key])
        check_gradients(report, "cavity 5041")
        # three training steps (forward + log-loss + backward + Adam) against the oracle's
        Po = {k: v.clone() for k, v in P.items()}
        buffers, state, og = O.new_normalizer_buffers(), {}, tuple(g.clone() for g in graphs)
        x0 = og[0].x.clone()
        model = _model(P)
        ts = TrainStep(model, tuple(g.clone().to("cuda") for g in graphs), use_graph="list")
        for i in range(3):
            og[0].x = x0.clone()
            oloss, oout, _ = O.train_step(Po, buffers, og, state, hyper={"dataset_size": 1})
            ts.step()
            assert abs(float(ts.loss) - float(oloss)) < TOL * abs(float(oloss)), (i, float(ts.loss), float(oloss))
            assert rel(ts.uvp_node, oout[4]) < (TOL if i == 0 else 1e-4), i
    finally:
        lib.gfv_set_f16split(1)


# ------------------------------------------------------------------------------------------------------------------
# the reference's own meshes
# ------------------------------------------------------------------------------------------------------------------
MORE = ["real_cavity101", "real_poisson_quad_tri", "real_naca0012"]


@pytest.mark.parametrize("name", MORE)
def test_more_reference_meshes_match_the_reference_outputs(golden_dir, name):
    """HIP path vs the numbers the REFERENCE produced on its own mesh (fixture).  As on cylinder_flow_full_tri
    (tests/test_model_gpu.py) the pooled residual norms are held to 1e-5 against the exactly accumulated pool of the
    oracle's fp32 residuals and to 1e-4 against the reference's sequentially rounded fp32 index_add."""
    graphs, fx, _ = cases.real_mesh(name, golden_dir)
    P = O.init_parameters(cases.WEIGHT_SEED)
    model = _model_default(P)
    hg = tuple(g.clone().to("cuda") for g in graphs)
    hg[0].norm_uvp, hg[0].norm_global = True, True
    out = model(*hg)
    oout, inter = O.model_forward(P, O.new_normalizer_buffers(), tuple(g.clone() for g in graphs), return_intermediates=True)
    theta, sigma = graphs[4].theta_PDE.double(), graphs[4].sigma.double()
    exact = {
        "loss_cont": torch.sqrt((inter["div"].detach().double() ** 2).sum()) * theta[0, 1],
        "loss_mom_x": torch.sqrt((inter["mom"][:, 0].detach().double() ** 2).sum()) * sigma[0, 0],
        "loss_mom_y": torch.sqrt((inter["mom"][:, 1].detach().double() ** 2).sum()) * sigma[0, 1],
    }
    for i, key in enumerate(KEYS):
        ref = torch.from_numpy(fx[key])
        if float(ref.abs().max()) == 0.0:       # a term theta_PDE / sigma switches off (Poisson)
            assert float(out[i].abs().max()) == 0.0, key
            continue
        r = rel(out[i], ref)
        assert r < (TOL if key in ("uvp_node", "uvp_cell", "loss_press") else 1e-4), (key, r)
        if key in exact:
            e = abs(float(out[i]) - float(exact[key])) / float(exact[key])
            assert e < TOL, (key, "vs exactly pooled fp32 residuals", e)
    loss = torch.mean(torch.log(1.0 * out[3] + 6e4 * out[0] + 5e4 * out[1] + 5e4 * out[2]))
    assert abs(float(loss) - float(fx["loss"])) < TOL * abs(float(fx["loss"]))
    loss.backward()
    gfp = fx["grad_fp"]
    gscale = np.nanmax(gfp[:, 1])
    for i, (k, p) in enumerate(model.named_parameters()):
        if np.isnan(gfp[i, 0]):
            assert p.grad is None, k
            continue
        mine = cases.fingerprint(p.grad.cpu().numpy())
        assert abs(mine[1] - gfp[i, 1]) < 2e-3 * gfp[i, 1] + 1e-5 * gscale, (k, mine, gfp[i])


def _model_default(P):
    """NNmodel with the reference's default dataset_size (the fixtures were made with it: an accumulating Normalizer)."""
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    m = NNmodel(default_params())
    sd = m.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("name", MORE)
def test_more_reference_meshes_match_fp64_oracle(golden_dir, name):
    graphs, _fx, _ = cases.real_mesh(name, golden_dir)
    P = O.init_parameters(cases.WEIGHT_SEED)
    report, _ = compare_to_fp64(graphs, P, cases.REAL_MESHES[name][3])
    for key in KEYS + ("loss",):
        if report[key][1] == 0.0 and report[key][0] == 0.0:
            continue
        assert report[key][0] < TOL, (key, report[key])
    check_gradients(report, name)


# ------------------------------------------------------------------------------------------------------------------
# config 5's loop on the airfoil mesh
# ------------------------------------------------------------------------------------------------------------------
def test_unsteady_twenty_inner_iterations_on_the_naca0012_mesh(golden_dir):
    """One time step of the solve loop (solve_with_grad_GPU.py:133-197): max_inner_steps = 20 iterations of (restore x,
    re-arm the norm flags, forward, log-loss, backward, Adam) from the reference's initial field, then the time advance -
    fused TrainStep in its default launch mode on the 30 684-cell NACA0012 mesh against the oracle's loop."""
    from gfv.graph import build_batch
    from gfv.trainer import TrainStep
    _g, _fx, mesh = cases.real_mesh("real_naca0012", golden_dir)
    graphs = build_batch([mesh], [mesh["init_uvp"].astype(np.float32)])
    P0 = O.init_parameters(cases.WEIGHT_SEED)
    Po = {k: v.clone() for k, v in P0.items()}
    buffers, state = O.new_normalizer_buffers(), {}
    og = tuple(x.clone() for x in graphs)
    backup = og[0].x.clone()
    model = _model(P0)
    ts = TrainStep(model, tuple(x.clone().to("cuda") for x in graphs))
    worst = 0.0
    first = last = None
    for it in range(20):
        og[0].x = backup.clone()
        oloss, oout, _ = O.train_step(Po, buffers, og, state, hyper={"dataset_size": 1})
        ts.step()
        worst = max(worst, abs(float(ts.loss) - float(oloss)) / abs(float(oloss)))
        first = float(oloss) if first is None else first
        last = float(oloss)
    ts.advance_time()
    x_oracle = torch.cat((oout[4].detach(), backup[:, 3:]), 1)
    xerr = rel(ts.x_backup[:, 0:3], x_oracle[:, 0:3])
    perr = max(float((p.detach().cpu() - Po[k]).abs().max()) for k, p in model.named_parameters())
    print(f"[naca0012, 20 inner iterations] worst loss deviation {worst:.2e}; advanced state {xerr:.2e}; worst parameter "
          f"|delta| {perr:.2e}; loss {first:.4f} -> {last:.4f}")
    assert last < first
    assert worst < 1e-4 and xerr < 1e-3 and perr < 20 * 2 * 5e-5


@pytest.mark.parametrize("name", sorted(cases.REAL_MESHES))
def test_device_preprocessing_matches_reference_generated_fingerprints(golden_dir, name):
    """Row f2 on the device: the HIP kernels behind gfv.device_prep (gfv_khop_count / gfv_khop_fill, gfv_wlsq_moments) on the
    reference's own example meshes against the stencil and the moment arrays the REFERENCE's pipeline produced
    (tests/golden/make_prep_golden.py) - not against the build's host code."""
    import os
    from test_oracle_golden import _check_prep_against_reference
    fp = np.load(os.path.join(golden_dir, "real_prep_fp.npz"))
    _g, _fx, mesh = cases.real_mesh(name, golden_dir, device="cuda")
    _check_prep_against_reference(mesh, fp, name)
