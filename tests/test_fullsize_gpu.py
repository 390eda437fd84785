"""Parity at BASELINE.json's full size (the 50 020-cell cylinder mesh bench.py times): direct comparison with the oracle
(its forward + backward takes a few seconds on the host at this size) and the size-independent properties the domain
offers - block-diagonal batching (per-graph losses of a batch = the losses of each mesh alone), exactness of the
2nd-order WLSQ gradient on a quadratic field, run-to-run bit identity."""
import numpy as np
import pytest
import torch

import cases
from oracle import fvgn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _mesh(cells, seed):
    from gfv import meshgen
    nx, ny = meshgen.cylinder_grid_for_cells(cells)
    m = meshgen.finish_mesh(meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, jitter=0.2, seed=seed))
    return m, meshgen.random_fields(m, seed=seed + 7)


def _model(P):
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    m = NNmodel(default_params(dataset_size=1))
    sd = m.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    m.load_state_dict(sd)
    return m.cuda()


def _loss(out):
    hp = O.DEFAULT_HYPER
    return torch.mean(torch.log(hp["loss_press"] * out[3] + hp["loss_cont"] * out[0] + hp["loss_mom"] * out[1]
                                + hp["loss_mom"] * out[2]))


@pytest.fixture(scope="module")
def bench_mesh():
    from gfv.graph import build_batch
    mesh, field = _mesh(50000, 1234)   # bench.py's mesh (rank 0, mesh 0)
    return build_batch([mesh], [field])


def test_full_size_forward_backward_matches_oracle(bench_mesh):
    graphs = bench_mesh
    assert graphs[3].pos.shape[0] == 50020
    P = O.init_parameters(cases.WEIGHT_SEED)
    Pg = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    og = tuple(g.clone() for g in graphs)
    oout, inter = O.model_forward(Pg, O.new_normalizer_buffers(), og, hyper={"dataset_size": 1}, return_intermediates=True)
    oloss = O.training_loss(oout)
    names = list(Pg)
    ograds = dict(zip(names, torch.autograd.grad(oloss, [Pg[k] for k in names], allow_unused=True)))
    model = _model(P)
    hg = tuple(g.clone().to("cuda") for g in graphs)
    hg[0].norm_uvp, hg[0].norm_global = True, True
    out = model(*hg)
    # pooled residual norms over 50 020 cells: the oracle (= the reference) pools sequentially in fp32, which at this size
    # is itself ~1e-4 off the exact sum of its own fp32 terms (tests/test_model_gpu.py::test_reference_example_mesh...):
    # the HIP pool is held to 1e-5 against the exactly (fp64) pooled oracle residuals and to 5e-4 against the
    # sequentially rounded values
    theta, sigma = graphs[4].theta_PDE.double(), graphs[4].sigma.double()
    exact = [torch.sqrt((inter["div"].detach().double() ** 2).sum()) * theta[0, 1],
             torch.sqrt((inter["mom"][:, 0].detach().double() ** 2).sum()) * sigma[0, 0],
             torch.sqrt((inter["mom"][:, 1].detach().double() ** 2).sum()) * sigma[0, 1]]
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert rel(out[i], oout[i]) < (5e-4 if i < 4 else TOL), (key, rel(out[i], oout[i]))
        if i < 3:
            e = abs(float(out[i]) - float(exact[i])) / float(exact[i])
            assert e < TOL, (key, "vs exactly pooled fp32 residuals", e)
    loss = _loss(out)
    assert abs(float(loss) - float(oloss)) < 3 * TOL * abs(float(oloss))   # carries the oracle's pooling error
    loss.backward()
    import os
    from gfv import lib as L
    # the bench configuration in the form bench.py times by default: chain products as split-fp16 on the f16 MFMA pipe
    assert (L.load().gfv_rowtile_last_path() >= 5) == (os.environ.get("GFV_F16SPLIT", "1") != "0")
    # gradients are sums over 25 k nodes / 75 k edges of mixed-sign terms: the fp32 oracle itself carries ~1e-4 of
    # summation noise in the small tensors at this size, so the comparison is norm-wise per tensor (3e-3, with a floor
    # of 1e-4 of the global gradient norm) - the element-wise 1e-4 bar is applied at the sizes of test_model_gpu.py
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in ograds.values() if g is not None)))
    worst = 0.0
    for k, p in model.named_parameters():
        if ograds[k] is None:
            assert p.grad is None, k
            continue
        err = float((p.grad.cpu().double() - ograds[k].double()).norm())
        ref = float(ograds[k].double().norm())
        worst = max(worst, err / (ref + 1e-4 * gnorm))
        assert err < 3e-3 * ref + 1e-4 * gnorm, (k, err, ref, gnorm)
    print("full-size worst norm-wise gradient error:", worst)


def test_full_size_run_to_run_bit_identity(bench_mesh):
    P = O.init_parameters(cases.WEIGHT_SEED)
    results = []
    for _ in range(2):
        model = _model(P)
        hg = tuple(g.clone().to("cuda") for g in bench_mesh)
        hg[0].norm_uvp, hg[0].norm_global = True, True
        out = model(*hg)
        _loss(out).backward()
        torch.cuda.synchronize()
        results.append(([o.clone() for o in out], {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    for a, b in zip(results[0][0], results[1][0]):
        assert torch.equal(a, b)
    for k in results[0][1]:
        assert torch.equal(results[0][1][k], results[1][1][k]), k


def test_batching_is_block_diagonal():
    """Per-graph losses and fields of a 2-mesh batch = those of each mesh run alone (no cross-graph coupling anywhere:
    graph-wise normalisation, slice tokens, residual pooling)."""
    from gfv.graph import build_batch
    (m0, f0), (m1, f1) = _mesh(12000, 31), _mesh(9000, 32)
    P = O.init_parameters(cases.WEIGHT_SEED)

    def run(ms, fs):
        model = _model(P)
        hg = tuple(g.clone().to("cuda") for g in build_batch(ms, fs))
        hg[0].norm_uvp, hg[0].norm_global = True, True
        return model(*hg)

    both, a, b = run([m0, m1], [f0, f1]), run([m0], [f0]), run([m1], [f1])
    for i in range(4):
        assert rel(both[i][0], a[i][0]) < TOL and rel(both[i][1], b[i][0]) < TOL, i
    n0 = a[4].shape[0]
    assert rel(both[4][:n0], a[4]) < TOL and rel(both[4][n0:], b[4]) < TOL


def test_wlsq_exact_on_quadratic_field_full_size(bench_mesh):
    """2nd-order WLSQ reproduces the gradient of a quadratic field (FVgrad.py:235-367; the reference's own accuracy
    test, grad_rec_acc_test.py:87-181, at the bench mesh)."""
    from FVMmodel.FVdiscretization.FVgrad import node_based_WLSQ
    gn, gx = bench_mesh[0], bench_mesh[1]
    pos = gn.pos.double()
    x, y = pos[:, 0], pos[:, 1]
    phi = torch.stack((1.0 + 2.0 * x - 3.0 * y + 0.5 * x * x - 0.25 * x * y + 1.5 * y * y,
                       -2.0 + 0.5 * x + y - x * x + 2.0 * x * y + 0.75 * y * y), 1)
    gref = torch.stack((torch.stack((2.0 + x - 0.25 * y, -3.0 - 0.25 * x + 3.0 * y), 1),
                        torch.stack((0.5 - 2.0 * x + 2.0 * y, 1.0 + 2.0 * x + 1.5 * y), 1)), 1)   # [N, 2, 2]
    d = lambda t: t.to("cuda")
    g = node_based_WLSQ(phi_node=d(phi.float()), edge_index=d(gx.face_node_x), extra_edge_index=d(gx.support_edge),
                        mesh_pos=d(gn.pos), order="2nd",
                        precompute_Moments=[d(gx.A_node_to_node), d(gx.single_B_node_to_node), d(gx.extra_B_node_to_node)])
    err = (g[:, :, 0:2].double().cpu() - gref).abs().max() / gref.abs().max()
    assert float(err) < 2e-3, float(err)   # fp32 solve of the row-normalised 5x5 system at h ~ 1e-2
