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


def graphs_to(graphs, dtype):
    """Copy of the five graph objects with every floating tensor cast to `dtype` (the oracle computes in its inputs' dtype)."""
    out = []
    for g in graphs:
        c = g.clone()
        for k, v in list(vars(c).items()):
            if torch.is_tensor(v) and v.is_floating_point():
                setattr(c, k, v.to(dtype))
        out.append(c)
    return tuple(out)


def oracle_run(graphs, P, dtype):
    """Oracle forward + backward in `dtype`: (outputs, scalar loss, gradients, intermediates)."""
    Pg = {k: v.detach().to(dtype).requires_grad_(True) for k, v in P.items()}
    og = graphs_to(graphs, dtype)
    buf = {k: v.to(dtype) for k, v in O.new_normalizer_buffers().items()}
    oout, inter = O.model_forward(Pg, buf, og, hyper={"dataset_size": 1}, return_intermediates=True)
    oloss = O.training_loss(oout)
    names = list(Pg)
    ograds = dict(zip(names, torch.autograd.grad(oloss, [Pg[k] for k in names], allow_unused=True)))
    return [o.detach() for o in oout], oloss.detach(), ograds, inter


def hip_run(graphs, P, recompute=None):
    model = _model(P)
    if recompute is not None:
        model.engine().recompute = recompute     # (gfv.engine.Engine: GFV_RECOMPUTE)
    hg = tuple(g.clone().to("cuda") for g in graphs)
    hg[0].norm_uvp, hg[0].norm_global = True, True
    out = model(*hg)
    loss = _loss(out)
    loss.backward()
    grads = {k: (None if p.grad is None else p.grad.detach().cpu()) for k, p in model.named_parameters()}
    return [o.detach().cpu() for o in out], loss.detach().cpu(), grads


def grad_errors(grads, ref):
    """Element-wise error of every gradient tensor in units of that tensor's scale (max |ref|, floored at 1e-6 of the
    global gradient scale - tensors whose gradient is rounding noise of the whole computation): {name: error}."""
    gmax = max(float(g.abs().max()) for g in ref.values() if g is not None)
    errs = {}
    for k, g in ref.items():
        if g is None:
            assert grads[k] is None, k
            continue
        errs[k] = float((grads[k].double() - g.double()).abs().max()) / max(float(g.abs().max()), 1e-6 * gmax)
    return errs


def global_grad_error(grads, ref):
    num = sum(float(((grads[k].double() - g.double()) ** 2).sum()) for k, g in ref.items() if g is not None)
    den = sum(float((g.double() ** 2).sum()) for g in ref.values() if g is not None)
    return (num / den) ** 0.5


ILL_CONDITIONED = ".TransBlock.Attn."   # slice-attention parameters: every entry is a sum over ALL nodes of cancelling terms


def check_gradients(report, label):
    """Against the float64 oracle.  (1) Norm-wise over all gradients the HIP path is within 1e-5, or at least as close as the
    reference's own fp32 arithmetic (the fp32 oracle on the same inputs).  (2) Element-wise, every gradient tensor is
    within 1e-4 of its scale - except tensors of the slice-attention parameter group (temperatures, slice projection,
    in_project_x, the 16 x 16 q / k weights: sums over all nodes of cancelling terms that no fp32 evaluation order resolves;
    the fp32 oracle itself is 1e-4 ... 3e-3 away there, and by how much varies from run to run with the thread schedule of
    its index_add), which are held to 5e-3.  (3) The median tensor is within 2e-5.
    Measured (profiles/r02_parity_fp64.txt): bench mesh - global 2.1e-6 (fp32 oracle 4.6e-4), median 2.7e-6 (4.2e-4), 9 of
    154 tensors beyond 1e-4, worst 2.6e-3 (graph_temperature; fp32 oracle 2.7e-3)."""
    eh, e32 = report["grad_hip"], report["grad_o32"]
    over = {k: (eh[k], e32[k]) for k in eh if eh[k] >= 1e-4}
    print(f"[{label}] gradient tensors beyond 1e-4 of scale (HIP | fp32 oracle), {len(over)} of {len(eh)}:")
    for k, (a, b) in sorted(over.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:70s} {a:.2e} | {b:.2e}")
    for k, (a, b) in over.items():
        assert ILL_CONDITIONED in k and a < 5e-3, (k, a, b)
    assert report["grad_global"][0] < max(1e-5, report["grad_global"][1]), report["grad_global"]
    assert report["grad_median_elementwise"][0] < 2e-5, report["grad_median_elementwise"]


def compare_to_fp64(graphs, P, label):
    """The HIP path and the fp32 oracle, each against the SAME oracle run in float64: the float64 run
    is the exact value of the reference's algorithm on these inputs; the fp32 oracle's distance to it is the reference's
    own rounding noise at this size."""
    o64 = oracle_run(graphs, P, torch.float64)
    o32 = oracle_run(graphs, P, torch.float32)
    hip = hip_run(graphs, P)
    keys = ("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")
    report = {}
    for i, key in enumerate(keys):
        report[key] = (rel(hip[0][i], o64[0][i]), rel(o32[0][i], o64[0][i]))
    report["loss"] = (abs(float(hip[1]) - float(o64[1])) / abs(float(o64[1])), abs(float(o32[1]) - float(o64[1])) / abs(float(o64[1])))
    eh, e32 = grad_errors(hip[2], o64[2]), grad_errors(o32[2], o64[2])
    report["grad_worst_elementwise"] = (max(eh.values()), max(e32.values()))
    report["grad_median_elementwise"] = (float(np.median(list(eh.values()))), float(np.median(list(e32.values()))))
    report["grad_global"] = (global_grad_error(hip[2], o64[2]), global_grad_error(o32[2], o64[2]))
    print(f"[{label}] distance to the float64 oracle: HIP path | fp32 oracle")
    for k, (a, b) in report.items():
        print(f"  {k:24s} {a:.3e} | {b:.3e}")
    report["grad_hip"], report["grad_o32"] = eh, e32
    path = __import__("os").environ.get("GFV_PARITY_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"[{label}] distance to the float64 oracle: HIP path | fp32 oracle\n")
            for k, v in report.items():
                if isinstance(v, tuple):
                    f.write(f"  {k:24s} {v[0]:.3e} | {v[1]:.3e}\n")
            over = {k: (eh[k], e32[k]) for k in eh if eh[k] >= 1e-4}
            f.write(f"  gradient tensors beyond 1e-4 of scale: {len(over)} of {len(eh)}\n")
            for k, (a, b) in sorted(over.items(), key=lambda kv: -kv[1][0]):
                f.write(f"    {k:70s} {a:.2e} | {b:.2e}\n")
    return report, (o64, o32, hip)


def test_full_size_forward_backward_matches_fp64_oracle(bench_mesh):
    """50 020 cells (the bench mesh): fields, residual losses and the scalar loss within 1e-5, every gradient element
    within 1e-4 of its tensor's scale - against the oracle run in FLOAT64, with the fp32 oracle's own distance beside it."""
    graphs = bench_mesh
    assert graphs[3].pos.shape[0] == 50020
    P = O.init_parameters(cases.WEIGHT_SEED)
    report, (o64, o32, hip) = compare_to_fp64(graphs, P, "bench mesh, 50 020 cells")
    import os
    from gfv import lib as L
    # the bench configuration in the form bench.py times by default: chain products as split-fp16 on the f16 MFMA pipe
    assert (L.load().gfv_rowtile_last_path() >= 5) == (os.environ.get("GFV_F16SPLIT", "1") != "0")
    for key in ("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell", "loss"):
        assert report[key][0] < TOL, (key, report[key])
    check_gradients(report, "bench mesh")
    # BASELINE configs 3 / 5: the reduced-precision product forms (gfv_set_f16split(2): single fp16 products; (3): single bf16
    # products, config 3's wording) on the same mesh against the fp32 ORACLE, to the tolerances stated in tests/golden/cases.py
    lib = L.load()
    for form, tol, label in ((2, cases.LOWP_TOL, "fp16"), (3, cases.BF16_TOL, "bf16")):
        try:
            lib.gfv_set_f16split(form)
            low = hip_run(graphs, P)
        finally:
            lib.gfv_set_f16split(1)
        meas = {key: rel(low[0][i], o32[0][i]) for i, key in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell"))}
        meas["logloss"] = abs(float(low[1]) - float(o32[1])) / abs(float(o32[1]))
        meas["grad_norm"] = global_grad_error(low[2], o32[2])
        print(f"[bench mesh] reduced-precision form ({label}) vs fp32 oracle:", {k: f"{v:.2e}" for k, v in meas.items()})
        path = os.environ.get("GFV_PARITY_REPORT")
        if path:
            with open(path, "a") as f:
                f.write(f"[bench mesh, 50 020 cells] reduced-precision product form (gfv_set_f16split({form}): single {label} products) "
                        "against the fp32 oracle:\n")
                for k, v in meas.items():
                    f.write(f"  {k:24s} {v:.3e}\n")
        for key in ("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press"):
            assert meas[key] < tol["losses"], (label, key, meas)
        assert meas["uvp_node"] < tol["field"] and meas["uvp_cell"] < tol["field"], (label, meas)
        assert meas["logloss"] < tol["logloss"] and meas["grad_norm"] < tol["grad_norm"], (label, meas)


def test_reference_example_mesh_matches_fp64_oracle(golden_dir):
    """The same comparison on the reference's own example mesh (mesh_example/cylinder_flow_full_tri, 15 074 cells; raw
    reader arrays committed as tests/golden/real_cylinder.npz by make_real_mesh_golden.py)."""
    graphs, _fx = cases.real_cylinder(golden_dir)
    P = O.init_parameters(cases.WEIGHT_SEED)
    report, _ = compare_to_fp64(graphs, P, "reference example mesh, 15 074 cells")
    for key in ("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell", "loss"):
        assert report[key][0] < TOL, (key, report[key])
    check_gradients(report, "reference example mesh")


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


def test_full_size_recompute_form_equals_read_form(bench_mesh, monkeypatch):
    """VERDICT r3 item 1(a) on the bench mesh: with `Engine.recompute` (GFV_RECOMPUTE=1) the forward of the fourteen big MLPs
    keeps z1 + row statistics only and the persistent backward rebuilds z2 and the LayerNorm input from z1 on the matrix cores
    (include/gfv.h, rc_Wh).  The forward is the same arithmetic (bit-identical outputs); every gradient tensor agrees with the
    read form's to 1e-5 of its scale outside the slice-attention group (5e-3 there, as against float64), the whole gradient
    norm-wise to 1e-6.  (Since every one of those forwards runs on the column-owner small-tile kernel - round 6: the encoders' too -
    the rebuilt rows can be bit-identical to the saved ones: the MFMA order and the split scales of csrc/cfwd.hip and of the
    persistent backward are the same.  That the switch reached the kernels is therefore checked on the launches themselves:
    fourteen fused backward launches of MLPs whose forward saved no z2.)"""
    from gfv import engine as E
    graphs = bench_mesh
    P = O.init_parameters(cases.WEIGHT_SEED)
    read = hip_run(graphs, P, recompute=False)
    lean_launches = []

    def counted(fused):
        def run(self, P_, sv, *a, **kw):
            done = fused(self, P_, sv, *a, **kw)
            if done and sv["z2"] is None:
                lean_launches.append(sv["prefix"])
            return done
        return run
    monkeypatch.setattr(E.Engine, "_mlp3_bwd_fused", counted(E.Engine._mlp3_bwd_fused))
    monkeypatch.setattr(E.Engine, "_edge_bwd_fused", counted(E.Engine._edge_bwd_fused))   # (the factored EdgeBlock's)
    rc = hip_run(graphs, P, recompute=True)
    monkeypatch.undo()
    for a, b in zip(read[0], rc[0]):
        assert torch.equal(a, b)
    assert float(read[1]) == float(rc[1])
    errs = grad_errors(rc[2], read[2])
    assert global_grad_error(rc[2], read[2]) < 1e-6, global_grad_error(rc[2], read[2])
    for k, e in errs.items():
        tol = 5e-3 if ILL_CONDITIONED in k else 1e-5
        assert e < tol, (k, e)
    from gfv.engine import Engine
    probe = Engine()
    if probe.fuse_dw and probe.factor:                # (GFV_FUSE_DW=0: no fused backward at all; GFV_EDGE_FACTOR=0: the plain
        assert len(lean_launches) == 14, lean_launches   # EdgeBlock's gathered adjoint runs unfused) 2 encoders + 6 EdgeBlock + 6 NodeBlock MLPs
