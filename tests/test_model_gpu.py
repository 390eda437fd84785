"""GPU parity proper: the HIP path behind `FVMmodel.importer.NNmodel` against the oracle on identical meshes,
fields and weights (the cases whose oracle outputs are pinned to the reference by tests/golden/*.npz).
Tolerance 1e-5 relative fp32 on fields / losses (BASELINE.json north_star); gradients: 1e-4 of each tensor's
scale with a floor of 1e-6 of the global gradient scale (tiny-gradient tensors are rounding noise in the
reference itself, see tests/golden/make_golden.log)."""
import os

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


def _hip_model(P, dataset_size=100, **kw):
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    m = NNmodel(default_params(dataset_size=dataset_size, **kw))
    sd = m.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    m.load_state_dict(sd)
    return m.cuda()


@pytest.mark.parametrize("name", list(cases.CASES))
def test_forward_backward_matches_oracle(name, golden_dir):
    assert torch.cuda.is_available()
    graphs = cases.make_graphs(name)
    P = O.init_parameters(cases.WEIGHT_SEED)
    # oracle (CPU)
    buffers = O.new_normalizer_buffers()
    Pg = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    og = tuple(g.clone() for g in graphs)
    oout, ointer = O.model_forward(Pg, buffers, og, return_intermediates=True)
    oloss = O.training_loss(oout)
    names = list(Pg)
    ograds = dict(zip(names, torch.autograd.grad(oloss, [Pg[k] for k in names], allow_unused=True)))
    # HIP
    model = _hip_model(P)
    hg = tuple(g.clone().to("cuda") for g in graphs)
    hg[0].norm_uvp, hg[0].norm_global = True, True
    out = model(*hg)
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert rel(out[i], oout[i]) < TOL, (key, rel(out[i], oout[i]))
    assert rel(hg[0].x, og[0].x) < TOL and rel(hg[0].edge_attr, og[0].edge_attr) < TOL
    assert hg[0].norm_uvp is False and hg[0].norm_global is False
    # golden (reference-generated) values too
    fx = np.load(os.path.join(golden_dir, f"{name}.npz"))
    for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        assert rel(out[i], torch.from_numpy(fx[key])) < TOL, key
    hp = O.DEFAULT_HYPER
    loss = torch.mean(torch.log(hp["loss_press"] * out[3] + hp["loss_cont"] * out[0] + hp["loss_mom"] * out[1]
                                + hp["loss_mom"] * out[2]))
    assert abs(float(loss) - float(oloss)) < TOL * abs(float(oloss))
    loss.backward()
    gscale = max(float(g.abs().max()) for g in ograds.values() if g is not None)
    worst = 0.0
    for k, p in model.named_parameters():
        if ograds[k] is None:
            assert p.grad is None, k
            continue
        assert p.grad is not None, k
        err = float((p.grad.cpu().double() - ograds[k].double()).abs().max())
        bound = 1e-4 * float(ograds[k].abs().max()) + 1e-6 * gscale
        worst = max(worst, err / bound)
        assert err < bound, (k, err, bound)
    # Normalizer buffers (utils/normalization.py:52-66)
    for k in ("acc_count", "num_accumulations", "acc_sum", "acc_sum_squared"):
        assert rel(getattr(model.node_norm, k), buffers[k]) < TOL, k
    # second call without re-arming the flags must raise like the reference (importer.py:123-124)
    with pytest.raises(ValueError):
        model(*hg)


def test_adam_training_steps_track_oracle():
    """Three optimiser steps (torch.optim.Adam on the HIP model vs the oracle's restated Adam)."""
    name = "cyl_cavity_b2"
    graphs = cases.make_graphs(name)
    P = O.init_parameters(cases.WEIGHT_SEED)
    Po = {k: v.clone() for k, v in P.items()}
    buffers = O.new_normalizer_buffers()
    model = _hip_model(P)
    opt = torch.optim.Adam(model.parameters(), lr=5e-5)
    state = {}
    hp = O.DEFAULT_HYPER
    for step in range(3):
        og = tuple(g.clone() for g in graphs)
        oloss, _, _ = O.train_step(Po, buffers, og, state)
        hg = tuple(g.clone().to("cuda") for g in graphs)
        hg[0].norm_uvp, hg[0].norm_global = True, True
        opt.zero_grad()
        out = model(*hg)
        loss = torch.mean(torch.log(hp["loss_press"] * out[3] + hp["loss_cont"] * out[0] + hp["loss_mom"] * out[1]
                                    + hp["loss_mom"] * out[2]))
        loss.backward()
        opt.step()
        assert abs(float(loss) - float(oloss)) < 2e-5 * abs(float(oloss)), step
