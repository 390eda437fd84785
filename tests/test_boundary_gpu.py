"""The Python operator boundary of SURVEY.md 8(b), operator by operator, on the HIP path against the oracle:
`Interplot` (FVInterpolation.py:36-265), the scatter primitives the reference imports from torch_scatter / torch_geometric
(through the import shims), `utils.utilities.calc_*`, `node_based_WLSQ` without precomputed moments and with `rt_cond`
(FVgrad.py:273-294,363-364; the reference's known-answer script grad_rec_acc_test.py:87-181 uses exactly that call), and the
call sequence of pre_train_Adam.py:158-191 through the reference's import paths."""
import os
import sys

import numpy as np
import pytest
import torch

import cases
from oracle import fvgn_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def graphs():
    return cases.make_graphs("cyl_cavity_b2")


def test_interplot_operators_match_oracle_forward_and_backward(graphs):
    from FVMmodel.FVdiscretization.FVInterpolation import Interplot
    gn, gx, ge, gc, gi = graphs
    cg = tuple(g.clone().to("cuda") for g in graphs)
    ip = Interplot()
    gen = torch.Generator().manual_seed(4)
    N, C_ = gn.x.shape[0], gc.pos.shape[0]
    phi, grad = torch.randn(N, 7, generator=gen), torch.randn(N, 7, 2, generator=gen)

    def both(fn_ref, fn_hip, inputs):
        ri = [t.clone().requires_grad_(True) for t in inputs]
        hi = [t.clone().cuda().requires_grad_(True) for t in inputs]
        r, h = fn_ref(*ri), fn_hip(*hi)
        assert rel(h, r) < TOL
        w = torch.randn(r.shape, generator=gen)
        (r * w).sum().backward()
        (h * w.cuda()).sum().backward()
        for a, b in zip(hi, ri):
            assert rel(a.grad, b.grad) < 1e-5

    # node -> cell (FVInterpolation.py:36-109), both calling conventions of the reference
    both(lambda p, g: O.node_to_cell_2nd_order(p, g, gn.face, gc.face, gn.pos, gc.pos),
         lambda p, g: ip.node_to_cell_2nd_order(node_phi=p, node_grad=g, graph_node=cg[0], graph_cell=cg[3]), [phi, grad])
    both(lambda p, g: O.node_to_cell_2nd_order(p, g, gn.face, gc.face, gn.pos, gc.pos),
         lambda p, g: ip.node_to_cell_2nd_order(node_phi=p, node_grad=g, cells_node=cg[0].face, cells_index=cg[3].face,
                                                mesh_pos=cg[0].pos, centroid=cg[3].pos), [phi[:, :3], grad[:, :3]])
    # node -> face with the Taylor correction, and a gradient field without it (FVscheme.py:109-122)
    both(lambda p, g: O.node_to_face_2nd_order(p, g, gn.edge_index, gn.pos, ge.pos),
         lambda p, g: ip.node_to_face_2nd_order(node_phi=p, node_grad=g, graph_node=cg[0], graph_edge=cg[2]),
         [phi[:, :5], grad[:, :5]])
    both(lambda g: O.node_to_face_2nd_order(g, None, gn.edge_index, gn.pos, ge.pos),
         lambda g: ip.node_to_face_2nd_order(node_phi=g, node_grad=None, graph_node=cg[0], graph_edge=cg[2]), [grad[:, :5]])
    # cell -> node, inverse-distance weights (FVInterpolation.py:218-265)
    cphi = torch.randn(C_, 3, generator=gen)
    both(lambda p: O.cell_to_node_2nd_order(p, gn.face, gc.face, gc.pos, gn.pos),
         lambda p: ip.cell_to_node_2nd_order(cell_phi=p, cell_grad=None, cells_node=cg[0].face, cells_index=cg[3].face,
                                             centroid=cg[3].pos, mesh_pos=cg[0].pos), [cphi])
    with pytest.raises(ValueError):
        ip.cell_to_node_2nd_order(cell_phi=cphi.cuda(), cell_grad=cphi.cuda(), cells_node=cg[0].face, cells_index=cg[3].face,
                                  centroid=cg[3].pos, mesh_pos=cg[0].pos)


def test_scatter_shims_and_utilities(graphs):
    """`import torch_scatter`, `from torch_geometric.nn import global_add_pool`, `from torch_geometric.data import Data` resolve
    to the shims (neither wheel is installed here) and give torch_scatter's results, forward and backward."""
    shim_dir = os.path.join(ROOT, "gen-fvgn-steady_amd", "shims")
    sys.path.append(shim_dir)
    try:
        import torch_scatter
        from torch_geometric.data import Data
        from torch_geometric.nn import global_add_pool
        assert os.path.abspath(torch_scatter.__file__).startswith(shim_dir)
    finally:
        sys.path.remove(shim_dir)
    from utils.utilities import calc_cell_centered_with_node_attr, calc_node_centered_with_cell_attr
    gn, gx, ge, gc, gi = graphs
    gen = torch.Generator().manual_seed(6)
    N, E = gn.x.shape[0], gn.edge_index.shape[1]
    for F in (1, 3, 64, 128, 35):
        src = torch.randn(2 * E, F, generator=gen)
        idx = torch.cat((gn.edge_index[0], gn.edge_index[1]))
        for mean in (False, True):
            s_ref = src.clone().requires_grad_(True)
            ref = (O.scatter_mean if mean else O.scatter_add)(s_ref, idx, N)
            s_hip = src.clone().cuda().requires_grad_(True)
            fn = torch_scatter.scatter_mean if mean else torch_scatter.scatter_add
            out = fn(s_hip, idx.cuda(), dim=0, dim_size=N)
            assert rel(out, ref) < TOL, (F, mean)
            w = torch.randn(ref.shape, generator=gen)
            (ref * w).sum().backward()
            (out * w.cuda()).sum().backward()
            assert rel(s_hip.grad, s_ref.grad) < TOL
    v = torch.randn(2 * E, generator=gen)                                   # 1-D src, default dim
    idx = torch.cat((gn.edge_index[0], gn.edge_index[1]))
    assert rel(torch_scatter.scatter(v.cuda(), idx.cuda(), reduce="sum"), O.scatter_add(v, idx, N)) < TOL
    into = torch.ones(N, 4).cuda()                                          # scatter_add into `out` (blocks.py:35 style)
    s4 = torch.randn(2 * E, 4, generator=gen)
    torch_scatter.scatter_add(s4.cuda(), idx.cuda(), dim=0, out=into)
    assert rel(into, 1.0 + O.scatter_add(s4, idx, N)) < TOL
    x = torch.randn(gc.pos.shape[0], 2, generator=gen)
    assert rel(global_add_pool(x.cuda(), gc.batch.cuda(), size=2), O.global_add_pool(x, gc.batch, 2)) < TOL
    assert rel(global_add_pool(x.cuda(), gc.batch.cuda()), O.global_add_pool(x, gc.batch, 2)) < TOL
    with pytest.raises(NotImplementedError):
        torch_scatter.scatter(x.cuda(), gc.batch.cuda(), dim=0, reduce="max")
    d = Data(x=x, edge_index=gn.edge_index, num_graphs=2)
    assert d.x is x and d.num_graphs == 2
    # utils.utilities.calc_* (utilities.py:16-57)
    na = torch.randn(N, 3, generator=gen)
    ref = O.scatter_mean(na[gn.face], gc.face, gc.pos.shape[0])
    assert rel(calc_cell_centered_with_node_attr(na.cuda(), gn.face.cuda(), gc.face.cuda()), ref) < TOL
    ca = torch.randn(gc.pos.shape[0], 3, generator=gen)
    ref = O.scatter_add(ca[gc.face], gn.face, N)
    assert rel(calc_node_centered_with_cell_attr(ca.cuda(), gn.face.cuda(), gc.face.cuda(), reduce="sum"), ref) < TOL
    with pytest.raises(ValueError):
        calc_cell_centered_with_node_attr(na.cuda(), gn.face.cuda()[:-1], gc.face.cuda())


def _analytic_field(pos):
    """utils/utilities.py:212-217 (`Scalar_Eular_solution`) with grad_rec_acc_test.py:87-97's parameters."""
    x, y = pos[:, 0].double(), pos[:, 1].double()
    a = 5 * np.pi
    phi = 1.0 + 0.01 * torch.sin(a * x) + 0.01 * torch.sin(a * y) + 0.01 * torch.cos(a * x * y)
    gx = 0.01 * a * torch.cos(a * x) - 0.01 * a * y * torch.sin(a * x * y)
    gy = 0.01 * a * torch.cos(a * y) - 0.01 * a * x * torch.sin(a * x * y)
    return phi.float()[:, None], torch.stack((gx, gy), 1)


def test_wlsq_without_precomputed_moments_and_known_answer(golden_dir):
    """grad_rec_acc_test.py:87-181 on the reference's example mesh (mesh_example/cylinder_flow_full_tri): the analytic field's
    gradient, reconstructed with moments built on the fly (FVgrad.py:273-294) and with precomputed ones - the same numbers -
    with the relative L2 errors the reference's arithmetic gives on this mesh (recorded from the reference:
    2nd order (0.01332, 0.03231), 1st order (0.02407, 0.06644)), and `rt_cond`."""
    from FVMmodel.FVdiscretization.FVgrad import node_based_WLSQ
    graphs, _ = cases.real_cylinder(golden_dir)
    gn, gx = graphs[0], graphs[1]
    phi, gref = _analytic_field(gn.pos)
    d = lambda t: t.to("cuda")
    pre = node_based_WLSQ(phi_node=d(phi), edge_index=d(gx.face_node_x), extra_edge_index=d(gx.support_edge), mesh_pos=d(gn.pos),
                          order="2nd", precompute_Moments=[d(gx.A_node_to_node), d(gx.single_B_node_to_node),
                                                           d(gx.extra_B_node_to_node)])
    direct, cond = node_based_WLSQ(phi_node=d(phi), edge_index=d(gx.face_node_x), extra_edge_index=d(gx.support_edge),
                                   mesh_pos=d(gn.pos), order="2nd", precompute_Moments=None, rt_cond=True)
    assert direct.shape == (gn.pos.shape[0], 1, 5) and cond.shape == (gn.pos.shape[0],)
    # the reference: "precomputed-moments path equals direct path" (SURVEY.md 8c).  Here the on-the-fly moments are summed
    # in float64 (gfv/device_prep.py), the handed-in ones are the reference's fp32 sums: the gradient entries agree to
    # 2e-5, the second-derivative entries (differences of larger terms) to 1e-4
    assert rel(direct[:, :, 0:2], pre[:, :, 0:2]) < 2e-5 and rel(direct, pre) < 1e-4
    assert bool(torch.isfinite(cond).all()) and float(cond.min()) >= 1.0
    oracle = O.node_based_WLSQ(phi, gx.face_node_x, gx.support_edge, gx.A_node_to_node, gx.single_B_node_to_node,
                               gx.extra_B_node_to_node)
    assert rel(pre, oracle) < 2e-5
    for order, want in (("2nd", (0.01332, 0.03231)), ("1st", (0.02407, 0.06644))):
        g = node_based_WLSQ(phi_node=d(phi), edge_index=d(gx.face_node_x), extra_edge_index=d(gx.support_edge),
                            mesh_pos=d(gn.pos), order=order, precompute_Moments=None)
        err = torch.norm(g[:, 0, 0:2].double().cpu() - gref, dim=0) / torch.norm(gref, dim=0)
        assert abs(float(err[0]) - want[0]) < 2e-4 and abs(float(err[1]) - want[1]) < 2e-4, (order, err, want)
    # differentiable through the on-the-fly branch too
    p = d(phi).clone().requires_grad_(True)
    node_based_WLSQ(phi_node=p, edge_index=d(gx.face_node_x), extra_edge_index=d(gx.support_edge), mesh_pos=d(gn.pos),
                    order="2nd").sum().backward()
    assert bool(torch.isfinite(p.grad).all())


def test_pre_train_adam_call_sequence_through_reference_import_paths():
    """pre_train_Adam.py:158-191 as the reference writes it - the model imported as `FVMmodel.importer.NNmodel`, re-armed norm
    flags, `loss.backward()`, `optimizer.step()`, detached predictions written back - for several batches of different
    meshes, against the oracle's train_step on the same batches (dataset training: the Normalizer accumulates)."""
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    params = default_params(dataset_size=3)
    P0 = O.init_parameters(cases.WEIGHT_SEED)
    model = NNmodel(params)
    sd = model.state_dict()
    for k, v in P0.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    model = model.to("cuda")
    model.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=params.lr)
    Po = {k: v.clone() for k, v in P0.items()}
    buffers, state = O.new_normalizer_buffers(), {}
    for name in ("cavity_mixed_b1", "cyl_cavity_b2", "cyl_b3", "cavity_mixed_b1"):
        graphs = cases.make_graphs(name)
        og = tuple(g.clone() for g in graphs)
        oloss, oout, _ = O.train_step(Po, buffers, og, state, hyper={"dataset_size": 3})
        graph_node, graph_node_x, graph_edge, graph_cell, graph_Index = tuple(g.clone().to("cuda") for g in graphs)
        graph_node.norm_uvp = params.norm_uvp
        graph_node.norm_global = params.norm_global
        optimizer.zero_grad()
        (loss_cont, loss_mom_x, loss_mom_y, loss_press, uvp_node_new, uvp_cell_new) = model(
            graph_node=graph_node, graph_node_x=graph_node_x, graph_edge=graph_edge, graph_cell=graph_cell,
            graph_Index=graph_Index, is_training=True)
        loss_batch = (params.loss_press * loss_press + params.loss_cont * loss_cont + params.loss_mom * loss_mom_x
                      + params.loss_mom * loss_mom_y)
        loss = torch.mean(torch.log(loss_batch))
        loss.backward()
        optimizer.step()
        graph_node.x[:, 0:3] = uvp_node_new.detach()
        assert abs(float(loss) - float(oloss)) < 2e-5 * abs(float(oloss)), name
        assert rel(uvp_node_new, oout[4]) < 2e-5 and rel(uvp_cell_new, oout[5]) < 2e-5, name
        assert graph_node.norm_uvp is False and graph_node.edge_attr.shape[1] == 15
    assert float(model.node_norm.num_accumulations) == float(buffers["num_accumulations"]) == 3.0
    assert rel(model.node_norm.acc_sum, buffers["acc_sum"]) < 1e-6
    worst = max(float((p.detach().cpu() - Po[k]).abs().max()) for k, p in model.named_parameters())
    assert worst < 4 * 2 * params.lr   # (4 Adam steps; see tests/test_config5_gpu.py on why parameters are judged loosely)


def test_plain_encoder_processer_decoder_matches_oracle(graphs):
    """`EncoderProcesserDecoder` (FVMmodel/Models/FVGN/EPD.py:222-270: Encoder, mp GnBlocks, Decoder - the FVGN simulator
    without Transolver blocks) as the reference composes it from the block operators, forward and backward."""
    from FVMmodel.Models.FVGN.EPD import EncoderProcesserDecoder
    from gfv.graph import Data
    P1 = O.init_parameters(cases.WEIGHT_SEED, hyper={"net": "TransFVGN_v1"})   # holds encoder / GN_block_list / decoder under `simulator.`
    net = EncoderProcesserDecoder(message_passing_num=3, edge_input_size=15, node_input_size=12, node_output_size=3)
    sd = net.state_dict()
    for k in sd:
        sd[k].copy_(P1["simulator." + k])
    net.load_state_dict(sd)
    net = net.cuda()
    g0 = graphs[0]
    gen = torch.Generator().manual_seed(2)
    xin = torch.randn(g0.x.shape[0], 12, generator=gen)
    ea = O.relative_edge_attr(xin, g0.pos, g0.edge_index)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P1.items() if "TransBlock" not in k}
    xn = O.mlp3(Pg, "simulator.encoder.nb_encoder", xin)
    en = O.mlp3(Pg, "simulator.encoder.eb_encoder", ea)
    for i in range(3):
        xn, en = O.gn_block(Pg, f"simulator.GN_block_list.{i}", xn, en, g0.edge_index)
    ref = O.decoder(Pg, "simulator.decoder", xn)
    w = torch.randn(ref.shape, generator=gen)
    (ref * w).sum().backward()
    gd = Data(x=xin.cuda(), edge_attr=ea.cuda(), edge_index=g0.edge_index.cuda(), face=None, num_graphs=2, batch=g0.batch.cuda())
    out = net(graph_node=gd, graph_cell=None)
    assert rel(out, ref) < TOL
    (out * w.cuda()).sum().backward()
    gscale = max(float(v.grad.abs().max()) for v in Pg.values() if v.grad is not None)
    for k, p in net.named_parameters():
        r = Pg["simulator." + k].grad
        if r is None:
            continue
        err = float((p.grad.cpu().double() - r.double()).abs().max())
        assert err < 1e-4 * float(r.abs().max()) + 1e-6 * gscale, (k, err)


def test_plan_cache_follows_in_place_edits_of_boundary_data(graphs):
    """the cached plan holds copies of graph_node.y / node types / PDE coefficients; the reference re-reads them on
    every forward, so an in-place edit of a reused batch (a re-selected boundary value, another viscosity) must reach the next
    forward.  Same model, same graph objects: forward, edit in place, forward -> equals a forward on freshly built graphs."""
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    P = O.init_parameters(cases.WEIGHT_SEED)
    model = NNmodel(default_params(dataset_size=1))
    sd = model.state_dict()
    for k, v in P.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    model = model.cuda()

    def run(gs, x0):
        gs[0].x = x0.clone()
        gs[0].norm_uvp, gs[0].norm_global = True, True
        with torch.no_grad():
            return [o.clone() for o in model(*gs)]

    hg = tuple(g.clone().to("cuda") for g in graphs)
    x0 = hg[0].x.clone()
    first = run(hg, x0)
    inflow = hg[0].node_type == 1
    hg[0].y[inflow] *= 1.5                     # in place: another inlet velocity
    hg[4].theta_PDE[:, 4] *= 2.0               # in place: another diffusion coefficient
    edited = run(hg, x0)
    fresh_graphs = tuple(g.clone().to("cuda") for g in graphs)
    fresh_graphs[0].y[fresh_graphs[0].node_type == 1] *= 1.5
    fresh_graphs[4].theta_PDE[:, 4] *= 2.0
    fresh = run(fresh_graphs, x0)
    for a, b in zip(edited, fresh):
        assert torch.equal(a, b)
    assert not torch.equal(edited[1], first[1]), "the edit must change the momentum residual"
