"""GPU: the stand-alone operator API (same class names / signatures as the reference's FVMmodel sub-modules) and the
fused TrainStep (flat buffers, fused Adam, hipGraph replay) against the oracle."""
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


def _load(module, P, prefix):
    sd = module.state_dict()
    for k in sd:
        sd[k].copy_(P[f"{prefix}.{k}"])
    module.load_state_dict(sd)
    return module.cuda()


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available()
    graphs = cases.make_graphs("cyl_cavity_b2")
    P = O.init_parameters(cases.WEIGHT_SEED)
    g = torch.Generator().manual_seed(5)
    N, E = graphs[0].x.shape[0], graphs[0].edge_index.shape[1]
    x = torch.randn(N, 128, generator=g)
    e = torch.randn(E, 128, generator=g)
    return graphs, P, x, e


def test_gnblock_edgeblock_nodeblock(setup):
    from FVMmodel.Models.FVGN.EPD import GnBlock
    from gfv.graph import Data
    graphs, P, x, e = setup
    pre = "simulator.processpr_list.0.GN_block_list.1"
    blk = _load(GnBlock(128), P, pre)
    ei = graphs[0].edge_index
    xg, eg = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items() if k.startswith(pre)}
    ox, oe = O.gn_block(Pg, pre, xg, eg, ei)
    wx, we = torch.randn_like(ox), torch.randn_like(oe)
    ((ox * wx).sum() + (oe * we).sum()).backward()
    xd, ed = x.cuda().requires_grad_(True), e.cuda().requires_grad_(True)
    gd = Data(x=xd, edge_attr=ed, edge_index=ei.cuda(), face=None, num_graphs=2, batch=graphs[0].batch.cuda())
    out = blk(gd)
    assert rel(out.x, ox) < TOL and rel(out.edge_attr, oe) < TOL
    ((out.x * wx.cuda()).sum() + (out.edge_attr * we.cuda()).sum()).backward()
    assert rel(xd.grad, xg.grad) < 1e-4 and rel(ed.grad, eg.grad) < 1e-4
    for k, p in blk.named_parameters():
        assert rel(p.grad, Pg[f"{pre}.{k}"].grad) < 1e-4, k
    # the two halves as stand-alone operators
    oe2 = O.edge_block(P, pre + ".eb_module", x, e, ei)
    out_e = blk.eb_module(Data(x=x.cuda(), edge_attr=e.cuda(), edge_index=ei.cuda(), face=None, num_graphs=2, batch=None))
    assert rel(out_e.edge_attr, oe2) < TOL
    ox2 = O.node_block(P, pre + ".nb_module", x, e, ei)
    out_n = blk.nb_module(Data(x=x.cuda(), edge_attr=e.cuda(), edge_index=ei.cuda(), face=None, num_graphs=2, batch=None))
    assert rel(out_n.x, ox2) < TOL


def test_transolver_block(setup):
    from FVMmodel.Models.GraphTransolver.GraphTransolver import Transolver_block
    graphs, P, x, _ = setup
    pre = "simulator.processpr_list.1.TransBlock"
    blk = _load(Transolver_block(8, 128, 0, "gelu", 2, 32), P, pre)
    batch = graphs[0].batch
    xg = x.clone().requires_grad_(True)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items() if k.startswith(pre)}
    o = O.transolver_block(Pg, pre, xg, batch, 2)
    w = torch.randn_like(o)
    (o * w).sum().backward()
    xd = x.cuda().requires_grad_(True)
    out = blk(xd, batch.cuda())
    assert rel(out, o) < TOL
    (out * w.cuda()).sum().backward()
    assert rel(xd.grad, xg.grad) < 1e-4
    gscale = max(float(v.grad.abs().max()) for v in Pg.values() if v.grad is not None)
    for k, p in blk.named_parameters():
        ref = Pg[f"{pre}.{k}"].grad
        if ref is None:
            assert p.grad is None, k
            continue
        err = float((p.grad.cpu().double() - ref.double()).abs().max())
        assert err < 1e-4 * float(ref.abs().max()) + 1e-6 * gscale, (k, err)


def test_simulator_encoder_decoder(setup):
    from FVMmodel.Models.TransFVGN.TransFVGN_v2 import Simulator
    from gfv.graph import Data
    graphs, P, _, _ = setup
    sim = _load(Simulator(3, 15, 12, 3), P, "simulator")
    g0 = graphs[0]
    xin = torch.randn(g0.x.shape[0], 12, generator=torch.Generator().manual_seed(1))
    ea = O.relative_edge_attr(xin, g0.pos, g0.edge_index)
    ref = O.simulator_v2(P, xin, ea, g0.edge_index, g0.batch, 2)
    gd = Data(x=xin.cuda(), edge_attr=ea.cuda(), edge_index=g0.edge_index.cuda(), face=None, num_graphs=2,
              batch=g0.batch.cuda())
    out = sim(gd, None, None)
    assert rel(out, ref) < TOL
    latent, node_ = sim.encoder(gd)
    assert rel(node_, O.mlp3(P, "simulator.encoder.nb_encoder", xin)) < TOL
    assert rel(latent.edge_attr, O.mlp3(P, "simulator.encoder.eb_encoder", ea)) < TOL
    assert rel(sim.decoder(latent), O.decoder(P, "simulator.decoder", O.mlp3(P, "simulator.encoder.nb_encoder", xin))) < TOL


def test_integrator_and_wlsq_operators(setup):
    from FVMmodel.FVdiscretization.FVscheme import Intergrator
    from FVMmodel.FVdiscretization.FVgrad import node_based_WLSQ
    from gfv.params import default_params
    graphs, _, _, _ = setup
    G = O.graph_tensors(*graphs)
    gen = torch.Generator().manual_seed(3)
    N = graphs[0].x.shape[0]
    uvp = torch.randn(N, 3, generator=gen) * 0.3
    uv_old = torch.randn(N, 2, generator=gen) * 0.3
    uvp_g = uvp.clone().requires_grad_(True)
    uv_hat = (uv_old + uvp_g[:, 0:2]) / 2
    ref = O.integrator_conserved(uvp_g, uv_hat, uv_old, G, O.DEFAULT_HYPER)
    w = torch.tensor([[6e4, 5e4, 5e4, 1.0]])
    (torch.cat(ref[0:4], 1) * w).sum().backward()
    cg = tuple(g.clone().to("cuda") for g in graphs)
    ud = uvp.cuda().requires_grad_(True)
    out = Intergrator()(uvp_new_node=ud, uv_hat_node=(uv_old.cuda() + ud[:, 0:2]) / 2, uv_old_node=uv_old.cuda(),
                        graph_node=cg[0], graph_node_x=cg[1], graph_edge=cg[2], graph_cell=cg[3], graph_Index=cg[4],
                        params=default_params())
    for i in range(4):
        assert rel(out[i], ref[i]) < TOL, i
    # outputs 4 / 5 as the reference's Intergrator returns them (FVscheme.py:253-262,718-724): the smoothed node field BEFORE the
    # Dirichlet overwrite and neither field re-dimensionalised (importer.py:223-231 does both afterwards)
    assert rel(out[4], ref[4]) < TOL
    assert rel(out[5], ref[5]) < TOL
    (torch.cat(out[0:4], 1) * w.cuda()).sum().backward()
    assert rel(ud.grad, uvp_g.grad) < 1e-4
    # ... and on the Poisson fixture (sigma = [1, 0, 0]: the re-dimensionalisation of importer.py:228 would zero v and p)
    pgraphs = cases.make_graphs("poisson_b1")
    PG = O.graph_tensors(*pgraphs)
    Np = pgraphs[0].x.shape[0]
    pu, po = torch.randn(Np, 3, generator=gen) * 0.3, torch.randn(Np, 2, generator=gen) * 0.3
    pref = O.integrator_conserved(pu, (po + pu[:, 0:2]) / 2, po, PG, O.DEFAULT_HYPER)
    pc = tuple(g.clone().to("cuda") for g in pgraphs)
    pout = Intergrator()(uvp_new_node=pu.cuda(), uv_hat_node=(po.cuda() + pu.cuda()[:, 0:2]) / 2, uv_old_node=po.cuda(),
                         graph_node=pc[0], graph_node_x=pc[1], graph_edge=pc[2], graph_cell=pc[3], graph_Index=pc[4],
                         params=default_params())
    for i in (0, 1, 2, 3, 4, 5):
        assert pout[i].shape == pref[i].shape and rel(pout[i], pref[i]) < TOL, i
    assert float(pref[4][:, 1:3].abs().max()) > 0   # (the un-scaled field keeps v and p)
    # node_based_WLSQ stand-alone: all five 2nd-order entries and its adjoint
    phi = torch.randn(N, 7, generator=gen)
    pg = phi.clone().requires_grad_(True)
    gref = O.node_based_WLSQ(pg, G["face_node_x"], G["support_edge"], G["A"], G["B1"], G["Bx"])
    wg = torch.randn_like(gref)
    (gref * wg).sum().backward()
    pd = phi.cuda().requires_grad_(True)
    gout = node_based_WLSQ(phi_node=pd, edge_index=cg[1].face_node_x, extra_edge_index=cg[1].support_edge,
                           mesh_pos=cg[0].pos, order="2nd",
                           precompute_Moments=[cg[1].A_node_to_node, cg[1].single_B_node_to_node, cg[1].extra_B_node_to_node])
    assert rel(gout, gref) < 2e-5
    (gout * wg.cuda()).sum().backward()
    assert rel(pd.grad, pg.grad) < 1e-4
    with pytest.raises(ValueError):
        node_based_WLSQ(phi_node=pd, order="5th")


@pytest.mark.parametrize("use_graph", [False, True])
def test_trainstep_tracks_oracle(use_graph):
    """Flat-buffer TrainStep (fused loss / Adam, optional hipGraph replay) vs the oracle's train_step, 4 steps."""
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    graphs = cases.make_graphs("cyl_b3")
    P = O.init_parameters(cases.WEIGHT_SEED)
    Po = {k: v.clone() for k, v in P.items()}
    model = NNmodel(default_params(dataset_size=1))
    sd = model.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    model = model.cuda()
    hg = tuple(g.clone().to("cuda") for g in graphs)
    ts = TrainStep(model, hg, use_graph=use_graph)
    buffers, state = O.new_normalizer_buffers(), {}
    for step in range(4):
        og = tuple(g.clone() for g in graphs)
        oloss, oout, _ = O.train_step(Po, buffers, og, state, hyper={"dataset_size": 1})
        loss = ts.step()
        assert abs(float(loss) - float(oloss)) < 2e-5 * abs(float(oloss)), (step, float(loss), float(oloss))
    assert rel(ts.uvp_node, oout[4]) < 1e-4
    worst = max(rel(ts.P[k], Po[k]) for k in Po)
    assert worst < 1e-4, worst


def test_step_modes_are_bit_identical():
    """Eager launches, hipGraph replay and command-list replay (gfv/cmdlist.py) are the SAME launches on the same two
    streams: parameters, Adam moments and the loss after 8 steps on a 3 000-cell mesh must be bit-identical - with an
    accumulating Normalizer on the first steps (a second recording once it freezes), with a learning-rate change half way
    (the hyper-parameters are device resident: replayed steps follow it), and after an in-place edit of a boundary value."""
    from FVMmodel.importer import NNmodel
    from gfv import meshgen
    from gfv.graph import build_batch
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    nx, ny = meshgen.cylinder_grid_for_cells(3000)
    mesh = meshgen.finish_mesh(meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, seed=5), U=0.3)
    graphs = build_batch([mesh], [meshgen.random_fields(mesh, seed=9)])
    P = O.init_parameters(cases.WEIGHT_SEED)
    finals, losses_seen = {}, {}
    for mode in (False, True, "list", "no-edit"):
        edit, mode = mode != "no-edit", (False if mode == "no-edit" else mode)
        model = NNmodel(default_params(dataset_size=3))
        sd = model.state_dict()
        for k, v in P.items():
            sd[k].copy_(v)
        model.load_state_dict(sd)
        model = model.cuda()
        ts = TrainStep(model, tuple(g.clone().to("cuda") for g in graphs), use_graph=mode)
        for i in range(8):
            if i == 5:
                ts.set_lr(2e-5)
            if i == 6 and edit:
                # a Dirichlet target and a PDE coefficient edited IN PLACE between two steps: the plan's copies follow
                # (TrainStep.step compares data pointers / version counters; the reference re-reads them every forward)
                ts.graphs[0].y.mul_(1.25)
                ts.graphs[4].theta_PDE[:, 4].mul_(0.5)
            ts.step()
        torch.cuda.synchronize()
        if not edit:
            finals["no-edit"] = ts.loss.reshape(-1).clone()
            continue
        losses_seen[mode] = ts.loss.reshape(-1).clone()
        if mode == "list":
            assert any(isinstance(k, tuple) and k[0] == "list" for k in ts._graphs), "the command list was never recorded"
            assert len(next(v for k, v in ts._graphs.items() if k[0] == "list")[0]) > 100
        finals[mode] = torch.cat([t.reshape(-1) for pmv in ts.named_state().values() for t in pmv]
                                 + [ts.adam_state[0:1], ts.loss.reshape(-1), ts.losses.reshape(-1), ts.uvp_node.reshape(-1)]).clone()
    assert torch.equal(finals[False], finals[True]), "hipGraph replay differs from eager"
    assert torch.equal(finals[False], finals["list"]), "command-list replay differs from eager"
    assert float(losses_seen[False]) != float(finals["no-edit"]), "the in-place edit never reached the step"


def test_reduced_precision_form_tracks_the_fp32_forms():
    """BASELINE configs 3 / 5 ("bf16 MLP GEMMs", "mixed precision"): `gfv_set_f16split(2)` runs every GEMM product of the
    chain and weight-gradient kernels as ONE fp16 x fp16 product with fp32 accumulation (the high parts of the same
    operands, fp32 everywhere else).  It is not the form any parity claim is made on; this test pins what it is: after the
    same 4 training steps the loss agrees with the split-fp16 (fp32-accurate) form to 1e-2 relative and the parameters
    to 2e-2 of the step they took - and it is NOT identical (the switch reaches the kernels).  `gfv_set_f16split(3)` is the
    same single product on bf16 operands (v_mfma_f32_16x16x32_bf16)."""
    from FVMmodel.importer import NNmodel
    from gfv import lib as L
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    graphs = cases.make_graphs("cavity_mixed_b1")
    P = O.init_parameters(cases.WEIGHT_SEED)
    lib = L.load()
    out = {}
    try:
        for mode in (1, 2, 3):
            lib.gfv_set_f16split(mode)
            model = NNmodel(default_params(dataset_size=1))
            sd = model.state_dict()
            for k, v in P.items():
                sd[k].copy_(v)
            model.load_state_dict(sd)
            model = model.cuda()
            ts = TrainStep(model, tuple(g.clone().to("cuda") for g in graphs), use_graph=False)
            for _ in range(4):
                ts.step()
            torch.cuda.synchronize()
            out[mode] = (float(ts.loss), torch.cat([pmv[0].reshape(-1) for pmv in ts.named_state().values()]).clone())
    finally:
        lib.gfv_set_f16split(1)
    p0 = torch.cat([P[k].reshape(-1) for k in ts.named_state().keys()]).cuda() if set(ts.named_state().keys()) <= set(P.keys()) else None
    (l1, w1), (l2, w2), (l3, w3) = out[1], out[2], out[3]
    assert l1 != l2 or not torch.equal(w1, w2), "the reduced-precision switch did not reach the kernels"
    assert abs(l1 - l2) <= 1e-2 * abs(l1), (l1, l2)
    moved = (w1 - p0).norm() if p0 is not None else w1.norm() * 1e-3
    assert float((w1 - w2).norm()) <= 2e-2 * float(moved) + 1e-6 * float(w1.norm()), (float((w1 - w2).norm()), float(moved))
    # the bf16 form (gfv_set_f16split(3), 8 significand bits): neither of the other two, and within 5e-2 / 1e-1 of them
    assert not torch.equal(w3, w2) and not torch.equal(w3, w1)
    assert abs(l1 - l3) <= 5e-2 * abs(l1), (l1, l3)
    assert float((w1 - w3).norm()) <= 1e-1 * float(moved) + 1e-6 * float(w1.norm()), (float((w1 - w3).norm()), float(moved))


def test_trainstep_state_dict_resume_equals_uninterrupted(tmp_path):
    """the fused Adam's state (moments, step count, lr) is saved under the reference's `optimizer0`
    key by NNmodel.save_checkpoint(optimizer=ts) and restored by load_checkpoint: 3 steps + save + load into a fresh
    model / TrainStep + 3 steps == 6 uninterrupted steps, bit for bit."""
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    graphs = cases.make_graphs("cavity_mixed_b1")
    P = O.init_parameters(cases.WEIGHT_SEED)

    def fresh():
        model = NNmodel(default_params(dataset_size=1))
        sd = model.state_dict()
        for k, v in P.items():
            sd[k].copy_(v)
        model.load_state_dict(sd)
        model = model.cuda()
        return model, TrainStep(model, tuple(g.clone().to("cuda") for g in graphs))

    model, ts = fresh()
    for _ in range(6):
        ts.step()
    torch.cuda.synchronize()
    want = torch.cat([t.reshape(-1) for pmv in ts.named_state().values() for t in pmv] + [ts.adam_state[0:1]]).clone()

    model, ts = fresh()
    ts.set_lr(5e-5)
    for _ in range(3):
        ts.step()
    path = str(tmp_path / "ckpt.pth")
    model.save_checkpoint(path, optimizer=ts)
    saved = torch.load(path, map_location="cpu", weights_only=False)
    assert "optimizer0" in saved and len(saved["optimizer0"]["state"]) > 100
    model2, ts2 = fresh()
    model2.load_checkpoint(optimizer=ts2, ckpdir=path, device="cuda")
    ts2.sync_from_model()
    for _ in range(3):
        ts2.step()
    torch.cuda.synchronize()
    got = torch.cat([t.reshape(-1) for pmv in ts2.named_state().values() for t in pmv] + [ts2.adam_state[0:1]])
    assert torch.equal(got, want)


def test_side_stream_is_bit_identical_and_repeatable():
    """The side stream only reorders independent launches: parameters after 6 hipGraph-replayed steps on a 3 000-cell
    mesh must be bit-identical with and without it, and from run to run (a buffer-lifetime race between the two
    streams would show up here as a mismatch)."""
    from FVMmodel.importer import NNmodel
    from gfv import meshgen
    from gfv.graph import build_batch
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    nx, ny = meshgen.cylinder_grid_for_cells(3000)
    mesh = meshgen.finish_mesh(meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, seed=5), U=0.3)
    graphs = build_batch([mesh], [meshgen.random_fields(mesh, seed=9)])
    P = O.init_parameters(cases.WEIGHT_SEED)
    finals = []
    for overlap in (True, False, True):
        model = NNmodel(default_params(dataset_size=1))
        sd = model.state_dict()
        for k, v in P.items():
            sd[k].copy_(v)
        model.load_state_dict(sd)
        model = model.cuda()
        model.engine().overlap = overlap
        ts = TrainStep(model, tuple(g.clone().to("cuda") for g in graphs), use_graph=True)
        for _ in range(6):
            ts.step()
        torch.cuda.synchronize()
        finals.append(({k: v.clone() for k, v in ts.P.items()}, float(ts.loss)))  # (alignment padding of the flat buffer excluded)
    for other in finals[1:]:
        assert other[1] == finals[0][1]
        bad = [k for k in finals[0][0] if not torch.equal(other[0][k], finals[0][0][k])]
        assert not bad, bad


def test_lbfgs_closure_driver_runs_on_the_drop_in_model():
    """SURVEY.md row f4 (LBFGS driver): the reference's closure loop (solve_with_grad_GPU_LBFGS.py:67-160: restore x,
    re-arm the norm flags, forward, clamp + log loss, backward inside torch.optim.LBFGS with strong-Wolfe line search)
    runs unchanged on the drop-in nn.Module and lowers the PDE loss."""
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    graphs = tuple(g.to("cuda") for g in cases.make_graphs("cavity_mixed_b1"))
    torch.manual_seed(0)
    params = default_params(dataset_size=1)
    model = NNmodel(params).cuda()
    gn = graphs[0]
    backup_x = gn.x[:, 3:].clone()
    uvp_node = gn.x[:, 0:3].clone()
    opt = torch.optim.LBFGS(model.parameters(), max_iter=6, history_size=10, tolerance_grad=1e-6, tolerance_change=1e-8,
                            line_search_fn="strong_wolfe")
    history = []

    def closure():
        opt.zero_grad()
        gn.x = torch.cat((uvp_node.detach(), backup_x), dim=-1)
        gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
        lc, lmx, lmy, lp, _, _ = model(*graphs)
        lb = params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lmx + params.loss_mom * lmy
        loss = torch.mean(torch.log(torch.clamp(lb, min=1e-10, max=1e10)))
        loss.backward()
        history.append(float(loss))
        return loss

    opt.step(closure)
    assert len(history) >= 3 and all(np.isfinite(history))
    assert min(history) < history[0] - 1e-3, history


def test_lbfgs_loop_tracks_the_oracle_under_the_same_optimizer():
    """Row f4, LBFGS: the closure loop of solve_with_grad_GPU_LBFGS.py:67-160 on the drop-in module and the SAME
    torch.optim.LBFGS (strong Wolfe) on the oracle, from the same weights: the closure evaluations - the line search's trial
    points included - see the same losses (first one to 1e-5, the following ones to 1e-3: each depends on every gradient
    before it)."""
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    cpu_graphs = cases.make_graphs("cavity_mixed_b1")
    P = O.init_parameters(cases.WEIGHT_SEED)
    params = default_params(dataset_size=1)
    hyper = {"dataset_size": 1}
    kw = dict(max_iter=4, history_size=10, tolerance_grad=1e-9, tolerance_change=1e-12, line_search_fn="strong_wolfe")

    # oracle
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    buffers = O.new_normalizer_buffers()
    x0 = cpu_graphs[0].x.clone()
    oh = []
    used = [p for k, p in Pg.items()]
    oopt = torch.optim.LBFGS(used, **kw)

    def oclosure():
        oopt.zero_grad()
        g = tuple(t.clone() for t in cpu_graphs)
        g[0].x = x0.clone()
        out = O.model_forward(Pg, buffers, g, hyper)
        lb = params.loss_press * out[3] + params.loss_cont * out[0] + params.loss_mom * out[1] + params.loss_mom * out[2]
        loss = torch.mean(torch.log(torch.clamp(lb, min=1e-10, max=1e10)))
        loss.backward()
        for p_ in used:          # parameters the forward never touches (ln_1, temperature): LBFGS needs a gradient tensor
            if p_.grad is None:
                p_.grad = torch.zeros_like(p_)
        oh.append(float(loss))
        return loss

    oopt.step(oclosure)

    # HIP
    model = NNmodel(params)
    sd = model.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    model = model.cuda()
    graphs = tuple(g.clone().to("cuda") for g in cpu_graphs)
    gn = graphs[0]
    xg = gn.x.clone()
    hh = []
    hopt = torch.optim.LBFGS(model.parameters(), **kw)

    def hclosure():
        hopt.zero_grad()
        gn.x = xg.clone()
        gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
        lc, lmx, lmy, lp, _, _ = model(*graphs)
        lb = params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lmx + params.loss_mom * lmy
        loss = torch.mean(torch.log(torch.clamp(lb, min=1e-10, max=1e10)))
        loss.backward()
        for p_ in model.parameters():
            if p_.grad is None:
                p_.grad = torch.zeros_like(p_)
        hh.append(float(loss))
        return loss

    hopt.step(hclosure)
    n = min(len(oh), len(hh), 5)
    assert n >= 3, (oh, hh)
    assert abs(hh[0] - oh[0]) < 1e-5 * abs(oh[0]), (hh[0], oh[0])
    for a, b in zip(hh[:n], oh[:n]):
        assert abs(a - b) < 1e-3 * abs(b), (hh[:n], oh[:n])
    assert min(hh) < hh[0] - 1e-3


@pytest.mark.parametrize("mode", ["eager", "list"])
def test_data_parallel_trainstep_two_ranks(mode):
    """SURVEY.md 8e on the device path: two ranks (one mesh each, sharing this GPU, gloo) run the sharded TrainStep; both ranks
    must end bit-identical, and equal (to rounding) to a single process stepping on the two-mesh batch.  `eager`: an
    accumulating Normalizer, every step exchanges its statistics; `list`: command-list replay, the early gradient bucket (fork to
    the communication stream + all-reduce) is part of the recorded list.  Child processes via
    torch.distributed.run (tests/dp_gpu_worker.py)."""
    import os
    import re
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_gpu_worker.py")
    # (a port of its own per parametrisation; ONE more attempt on another port when the launcher produced no result line at all - a
    # rendezvous that did not come up is the test bed's problem; a result line with wrong values is never retried)
    for attempt in range(2):
        port = 29600 + (os.getpid() % 300) + (0 if mode == "eager" else 311) + 523 * attempt
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), worker],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR="127.0.0.1", GFV_TEST_MODE=mode))
        m = re.search(r"DPRESULT same=(\d) param_err=(\S+) norm_err=(\S+)", r.stdout)
        if m:
            break
    assert r.returncode == 0 and m, r.stdout[-1500:] + r.stderr[-1500:]
    assert m.group(1) == "1", "ranks diverged"
    # (the workers inherit the product form: under the single-product forms - GFV_F16SPLIT=2 / 3 - the single process on the
    # two-mesh batch and the two ranks agree to that form's accuracy, measured 6e-5)
    tol = 2e-5 if os.environ.get("GFV_F16SPLIT", "1") in ("0", "1") else 5e-4
    assert float(m.group(2)) < tol and float(m.group(3)) < 1e-6, m.group(0)


def test_command_list_recording_refuses_stray_torch_ops():
    """A torch op that launches device work during recording without going through cmdlist.call would run once and be
    missing from every replay: the recording guard (gfv/cmdlist.py) raises instead; views and pool allocations pass."""
    from gfv import cmdlist
    x = torch.ones(8, device="cuda")
    with pytest.raises(RuntimeError, match="dropped on replay"):
        with cmdlist.record():
            _ = x + 1
    assert cmdlist.active() is None
    with cmdlist.record() as cl:
        cmdlist.call(x.add_, 1.0)            # recorded with its stream
        _v = x[2:4]                          # a view: nothing to record
        _e = torch.empty(16, device="cuda")  # allocation from the recording pool: no kernel
    assert len(cl) == 1
    cl.replay()
    torch.cuda.synchronize()
    assert float(x[0]) == 3.0


def test_cell_to_node_with_cell_grad_uses_the_cells_own_gradient():
    """Interplot.cell_to_node_2nd_order with `cell_grad`: the correction is the gradient of the incidence's OWN cell
    (cell_grad[cells_index]); the reference's line FVInterpolation.py:248 indexes the cell array with node ids
    (cell_grad[cells_node]) - documented as a deliberate deviation in the module docstring; no caller passes cell_grad."""
    from FVMmodel.FVdiscretization.FVInterpolation import Interplot
    graphs = cases.make_graphs("cyl_cavity_b2")
    gn, gc = graphs[0], graphs[3]
    gen = torch.Generator().manual_seed(11)
    C = gc.pos.shape[0]
    cphi = torch.randn(C, 3, generator=gen)
    cgrad = torch.randn(C, 3, 2, generator=gen)
    cells_node, cells_index = gn.face, gc.face
    d = (gn.pos[cells_node] - gc.pos[cells_index]).double()
    w = 1.0 / d.norm(dim=-1, keepdim=True)
    corr = (cgrad.double()[cells_index] * d.unsqueeze(1)).sum(-1)
    num = torch.zeros(gn.pos.shape[0], 3, dtype=torch.float64).index_add_(0, cells_node, (cphi.double()[cells_index] + corr) * w)
    den = torch.zeros(gn.pos.shape[0], 1, dtype=torch.float64).index_add_(0, cells_node, w)
    ref = num / den
    cg = tuple(g.clone().to("cuda") for g in graphs)
    pd, gd = cphi.cuda().requires_grad_(True), cgrad.cuda().requires_grad_(True)
    out = Interplot().cell_to_node_2nd_order(cell_phi=pd, cell_grad=gd, cells_node=cg[0].face, cells_index=cg[3].face,
                                             centroid=cg[3].pos, mesh_pos=cg[0].pos)
    assert rel(out, ref) < TOL
    wgt = torch.randn(out.shape, generator=gen)
    (out * wgt.cuda()).sum().backward()
    pr, gr = cphi.double().requires_grad_(True), cgrad.double().requires_grad_(True)
    corr = (gr[cells_index] * d.unsqueeze(1)).sum(-1)
    ref2 = torch.zeros(gn.pos.shape[0], 3, dtype=torch.float64).index_add(0, cells_node, (pr[cells_index] + corr) * w) / den
    (ref2 * wgt.double()).sum().backward()
    assert rel(pd.grad, pr.grad) < 1e-4 and rel(gd.grad, gr.grad) < 1e-4


def test_per_graph_norm_statistics_many_workgroup_form():
    """gfv_graph_norm_stats_ws (64 workgroups per graph, double partial sums) against the one-workgroup form and float64:
    mean and population std of x[:, 0:3] per graph (importer.py:80-93), graphs of very different sizes, a large offset."""
    from gfv import lib as L
    lib = L.load()
    st = L.stream_ptr()
    g = torch.Generator().manual_seed(2)
    sizes = [25479, 7, 3000, 1]
    N = sum(sizes)
    x = torch.randn(N, 12, generator=g)
    x[:, 2] = x[:, 2] * 1e-3 + 50.0            # a column whose mean dwarfs its spread
    ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32).cuda()
    xd = x.cuda()
    B = len(sizes)
    s1, s2 = torch.empty(B, 6, device="cuda"), torch.empty(B, 6, device="cuda")
    ws = torch.empty(lib.gfv_graph_norm_workspace_bytes(B) // 4, device="cuda")
    L.check(lib.gfv_graph_norm_stats(xd.data_ptr(), 12, ptr.data_ptr(), B, s1.data_ptr(), st), "one workgroup")
    L.check(lib.gfv_graph_norm_stats_ws(xd.data_ptr(), 12, ptr.data_ptr(), B, s2.data_ptr(), ws.data_ptr(), st), "many")
    torch.cuda.synchronize()
    o = 0
    for b, n in enumerate(sizes):
        seg = x[o:o + n, 0:3].double()
        mean, std = seg.mean(0), seg.var(0, unbiased=False).sqrt()
        for c in range(3):
            assert abs(float(s2[b, c]) - float(mean[c])) <= 1e-6 * max(1.0, abs(float(mean[c])))
            assert abs(float(s2[b, 3 + c]) - float(std[c])) <= 1e-5 * float(std[c]) + 1e-9
            assert abs(float(s1[b, 3 + c]) - float(std[c])) <= 2e-3 * float(std[c]) + 1e-9   # (two fp32 sweeps: looser)
        o += n
