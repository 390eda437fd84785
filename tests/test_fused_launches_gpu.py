"""Round 6: the launches that were merged (input preparation 6 -> 2, finite-volume forward tail 3 -> 1 and backward 6 -> 3, a
Transolver block's six reductions -> 1, Adam's tick into the update) against the launches they replace: the SAME sums in the same
order, so everything is compared bit for bit - through the drop-in model (forward outputs, in-place side effects, every gradient)
and through the fused TrainStep (losses, fields, parameters, Adam moments after several steps), on meshes with several graphs,
mixed cell types and an OUTFLOW boundary.  The stand-alone kernels being compared against are the ones the operator tests hold to
the oracle (tests/test_operators_gpu.py, tests/test_model_gpu.py)."""
import os

import pytest
import torch

import cases
from oracle import fvgn_oracle as O

pytestmark = pytest.mark.gpu


def _model():
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    params = default_params(dataset_size=4)
    P0 = O.init_parameters(cases.WEIGHT_SEED)
    model = NNmodel(params)
    sd = model.state_dict()
    for k, v in P0.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    return model.to("cuda"), params


def _unfuse(model, monkeypatch):
    """The launches of rounds 1 - 5 for the input preparation and the finite-volume tail / adjoint."""
    e = model.engine()
    e._fvm_fuse = False
    monkeypatch.setenv("GFV_PREP_FUSE", "0")


@pytest.mark.parametrize("case", ["cyl_cavity_b2", "cavity_mixed_b1", "cyl_b3"])
def test_drop_in_model_with_merged_launches_is_bit_identical(case, monkeypatch):
    """Two accumulating calls (the Normalizer's running statistics move) and one that does not."""
    res = {}
    for fused in (True, False):
        model, params = _model()
        model._replay.enabled = False
        if not fused:
            _unfuse(model, monkeypatch)
        outs = []
        for it in range(3):
            graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs(case))
            gn = graphs[0]
            gn.norm_uvp, gn.norm_global = True, it < 2
            model.zero_grad()
            o = model(*graphs)
            loss = torch.mean(torch.log(o[3] + 6e4 * o[0] + 5e4 * o[1] + 5e4 * o[2]))
            loss.backward()
            torch.cuda.synchronize()
            outs.append([t.detach().clone() for t in o] + [gn.x.clone(), gn.edge_attr.clone()]
                        + [p.grad.clone() for p in model.parameters() if p.grad is not None]
                        + [b.clone() for b in model.node_norm.buffers()])
        res[fused] = outs
        monkeypatch.undo()
    for a, b in zip(res[True], res[False]):
        assert len(a) == len(b)
        for i, (ta, tb) in enumerate(zip(a, b)):
            assert torch.equal(ta, tb), (case, i, float((ta - tb).abs().max()))


@pytest.mark.parametrize("mode", [False, "list"])
def test_trainstep_with_merged_launches_is_bit_identical(mode, monkeypatch):
    from gfv.trainer import TrainStep
    res = {}
    for fused in (True, False):
        model, params = _model()
        if not fused:
            _unfuse(model, monkeypatch)
        graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))
        ts = TrainStep(model, graphs, use_graph=mode)
        for _ in range(6):
            ts.step()
        torch.cuda.synchronize()
        res[fused] = [ts.loss.clone(), ts.losses.clone(), ts.uvp_node.clone(), ts.uvp_cell.clone(), ts.x.clone(), ts.flat_p.clone(),
                      ts.flat_m.clone(), ts.flat_v.clone(), ts.flat_g.clone(), ts.adam_state[0:4].clone()]
        monkeypatch.undo()
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    assert float(res[True][-1][0]) == 6.0


def test_adam_launch_equals_torch_adam_bias_corrections():
    """gfv_adam_step_dev alone (one launch: update + the next step's corrections by its last workgroup) against torch.optim.Adam on
    one flat tensor over 40 steps with a learning rate that changes on the way: the corrections are those of torch's host code
    (double), the element arithmetic is fp32 in both."""
    from gfv import lib as L
    lib = L.load()
    n = 300_001
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    p = p0.clone().cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state = torch.zeros(16, device="cuda")
    hyper = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 1.0, 0, 0, 0], device="cuda")
    L.check(lib.gfv_adam_state_init(state.data_ptr(), 0.9, 0.999, 0.0, L.stream_ptr()), "init")
    q = torch.nn.Parameter(p0.clone().cuda())
    opt = torch.optim.Adam([q], lr=1e-3)
    for step in range(40):
        grad = torch.randn(n, generator=g).cuda() * (1.0 + step)
        if step == 17:
            hyper[0] = 3e-4
            opt.param_groups[0]["lr"] = 3e-4
        L.check(lib.gfv_adam_step_dev(p.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, state.data_ptr(),
                                      hyper.data_ptr(), L.stream_ptr()), "adam")
        q.grad = grad.clone()
        opt.step()
    torch.cuda.synchronize()
    assert float(state[0]) == 40.0 and int(state.view(torch.int32)[4]) == 0
    st = opt.state[q]
    # (torch: exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2) with 1 - beta in double)
    assert float((m - st["exp_avg"]).abs().max()) <= 2e-6 * float(st["exp_avg"].abs().max())
    assert float((v - st["exp_avg_sq"]).abs().max()) <= 2e-6 * float(st["exp_avg_sq"].abs().max())
    assert float((p - q.detach()).abs().max()) < 40 * 1e-3 * 2e-5


def test_merged_reductions_agree_with_the_separate_launches():
    """The merge that changes a summation ORDER (a Transolver block's reductions in one gfv_reduce_multi launch instead of the
    weight-gradient launches' own reductions): every output and gradient within 2e-6 of scale of the separate launches."""
    res = {}
    for merged in (True, False):
        model, params = _model()
        model._replay.enabled = False
        model.engine()._trans_reduce_merge = merged
        graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))
        graphs[0].norm_uvp, graphs[0].norm_global = True, True
        o = model(*graphs)
        torch.mean(torch.log(o[3] + 6e4 * o[0] + 5e4 * o[1] + 5e4 * o[2])).backward()
        torch.cuda.synchronize()
        res[merged] = ([t.detach().clone() for t in o[:6]], {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    gscale = max(float(g.abs().max()) for g in res[False][1].values())
    for n, g in res[False][1].items():
        err = float((res[True][1][n] - g).abs().max())
        assert err <= 2e-6 * float(g.abs().max()) + 2e-7 * gscale, (n, err)


def test_recorded_step_with_delayed_side_bursts_is_bit_identical(monkeypatch):
    """gfv_record_delay_side: the recorded step with every side-stream burst issued behind up to six of the main stream's
    following launches - a different ISSUE order of the same launches with the same dependencies: parameters, moments, losses and
    fields after eight steps equal those of the list as recorded, bit for bit."""
    from gfv.trainer import TrainStep
    res = {}
    for delay in (0, 6):
        monkeypatch.setenv("GFV_SIDE_DELAY", str(delay))
        model, params = _model()
        graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_b3"))
        ts = TrainStep(model, graphs, use_graph="list")
        for _ in range(8):
            ts.step()
        torch.cuda.synchronize()
        cl = ts._graphs[("list", False, False)][0] if ("list", False, False) in ts._graphs else next(iter(ts._graphs.values()))[0]
        res[delay] = ([ts.loss.clone(), ts.losses.clone(), ts.uvp_node.clone(), ts.flat_p.clone(), ts.flat_m.clone(), ts.flat_v.clone()],
                      getattr(cl, "delayed", 0))
    from gfv import cmdlist
    # (the Python-level list of GFV_CMDLIST_NATIVE=0 has no such pass, GFV_OVERLAP=0 no side stream, and with the launch merges
    # of round 6 switched off the bursts sit directly in front of joins: the two runs are then the same list)
    eng = model.engine()
    if cmdlist.NATIVE and eng.overlap and eng._fvm_fuse and eng._trans_reduce_merge and os.environ.get("GFV_PREP_FUSE", "1") != "0":
        assert res[6][1] > 0 and res[0][1] == 0      # the pass found runs to move
    for a, b in zip(res[0][0], res[6][0]):
        assert torch.equal(a, b)


def test_layernorm_on_load_aggregation_kernel_is_bit_identical_to_gathering_the_layernorm_output():
    """gfv_seg_gather_sum_ln against gfv_seg_gather_sum over the LayerNorm output formed by the same fp32 expression
    ((y - mean) * rstd * gamma + beta, one rounding per operation): rows with 0 .. 11 entries, both halves, the same CSR order."""
    from gfv import ops
    g = torch.Generator().manual_seed(21)
    E, N = 5003, 1777
    y = (torch.randn(E, 128, generator=g) * 3.0 + 0.5).cuda()
    gamma, beta = (1.0 + 0.3 * torch.randn(128, generator=g)).cuda(), (0.2 * torch.randn(128, generator=g)).cuda()
    mean = y.mean(1)
    rstd = torch.rsqrt(y.var(1, unbiased=False) + 1e-5)
    stats = torch.stack((mean, rstd), 1).contiguous()
    ln = (y - mean[:, None]) * rstd[:, None] * gamma[None, :] + beta[None, :]
    counts = torch.randint(0, 12, (N,), generator=g)
    counts[5] = 0
    rowptr = torch.zeros(N + 1, dtype=torch.int32)
    rowptr[1:] = counts.cumsum(0).to(torch.int32)
    col = torch.randint(0, 2 * E, (int(rowptr[-1]),), generator=g).to(torch.int32)
    want = ops.seg_gather_sum(ln.view(2 * E, 64), rowptr.cuda(), col.cuda(), N)
    got = ops.seg_gather_sum_ln(y, stats, gamma, beta, rowptr.cuda(), col.cuda(), N)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = torch.zeros(N, 64, dtype=torch.float64)
    lnd = ln.double().cpu().view(2 * E, 64)
    for r in range(N):
        for k in range(int(rowptr[r]), int(rowptr[r + 1])):
            ref[r] += lnd[int(col[k])]
    assert float((got.double().cpu() - ref).abs().max()) < 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("family", ["small-tile", "row-owner"])
@pytest.mark.parametrize("case", ["cyl_cavity_b2", "cavity_mixed_b1"])
def test_aggregation_from_the_saved_rows_is_bit_identical(case, family, gfv_limits, monkeypatch):
    """Engine._agg_ln (GFV_AGG_LN): the EdgeBlock forward without its second, residual-free output - the node aggregation forms
    LayerNorm(y3) on the way in - against the forward that writes it: outputs, side effects and every gradient bit for bit, with the
    EdgeBlock MLP on the column-owner small-tile forward and on the row-owner chain (what 8 meshes per GPU run)."""
    from gfv import ops
    # (the families this test is about, whatever the environment says: the row statistics the aggregation reads are saved for the
    # small-tile backward of launches this short)
    gfv_limits(GFV_CBWD=1, GFV_CFWD=0 if family == "row-owner" else 1)
    calls = {True: 0, False: 0}
    plain = ops.seg_gather_sum_ln
    res = {}
    for on in (True, False):
        def counted(*a, _on=on, **kw):
            calls[_on] += 1
            return plain(*a, **kw)
        monkeypatch.setattr(ops, "seg_gather_sum_ln", counted)
        model, params = _model()
        model._replay.enabled = False
        model.engine()._agg_ln = on
        graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs(case))
        graphs[0].norm_uvp, graphs[0].norm_global = True, True
        o = model(*graphs)
        torch.mean(torch.log(o[3] + 6e4 * o[0] + 5e4 * o[1] + 5e4 * o[2])).backward()
        torch.cuda.synchronize()
        res[on] = [t.detach().clone() for t in o] + [graphs[0].x.clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
    assert calls == {True: 6, False: 0}, calls     # one aggregation per GnBlock
    assert len(res[True]) == len(res[False])
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
