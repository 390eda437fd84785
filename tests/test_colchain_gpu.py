"""GPU: the column-owner persistent backward family (csrc/colchain_kernel.h) against float64 autograd of the fused MLP
(EPD.py:10-33 build_mlp inside blocks.py EdgeBlock / NodeBlock, the encoders); tolerance 1e-5 relative (fp32), as for every
other kernel (tests/test_kernels_gpu.py).  Every test runs in the read form (z2 and the LayerNorm input saved by the forward)
and in the RECOMPUTE form (rebuilt from z1 inside the launch: include/gfv.h, rc_Wh)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from gfv import lib
    lib.load()
    return torch.device("cuda:0")


def _params(g, kin, scale=0.3):
    return dict(W1=torch.randn(128, kin, generator=g) * scale / kin ** 0.5 * 4, b1=torch.randn(128, generator=g) * 0.1,
                W2=torch.randn(128, 128, generator=g) * scale / 3, b2=torch.randn(128, generator=g) * 0.1,
                W3=torch.randn(128, 128, generator=g) * scale / 3, b3=torch.randn(128, generator=g) * 0.1,
                gamma=1 + 0.1 * torch.randn(128, generator=g), beta=0.1 * torch.randn(128, generator=g))


def _ref(P, X, add=None):
    P = {k: v.double() for k, v in P.items()}
    z1 = F.linear(X, P["W1"], P["b1"])
    if add is not None:
        z1 = z1 + add
    z2 = F.linear(F.gelu(z1), P["W2"], P["b2"])
    y3 = F.linear(F.gelu(z2), P["W3"], P["b3"])
    return z1, z2, y3, F.layer_norm(y3, (128,), P["gamma"], P["beta"], 1e-5)


def _images(dev, Ws):
    from gfv import ops
    wmax = torch.stack([w.abs().max() for w in Ws]).max().reshape(1).to(dev)
    wi = ops.WeightImages(dev, wmax)
    wi.static = [(0, 1 << 62)]
    return wi


@pytest.mark.parametrize("M", [4000, 97, 1024])
@pytest.mark.parametrize("extras", [True, False])
@pytest.mark.parametrize("form", ["read", "rc"])
def test_column_owner_backward_with_fused_weight_gradients(dev, M, extras, form):
    """EdgeBlock backward in its factored form: LayerNorm backward, the three transposed layers, residual; with the weight
    gradients of the third and second Linear, their bias gradients and (dgamma, dbeta) accumulated by the same launch
    (include/gfv.h, gfv_rowtile_args_t.dw_partial) - every piece against float64 autograd."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + 7 * extras)
    n_nodes = 300
    e = torch.randn(M, 128, generator=g)
    P = _params(g, 128)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    X = e.double().requires_grad_(True)
    z1, z2, y3, ln = _ref(Pg, X)
    go = torch.randn(M, 128, generator=g) * torch.logspace(-5, 0, M)[:, None]     # gradient rows over five decades
    gagg = torch.randn(n_nodes, 64, generator=g) * 1e-2
    s = torch.randint(0, n_nodes, (M,), generator=g)
    r = torch.randint(0, n_nodes, (M,), generator=g)
    gadd2 = torch.randn(M, 128, generator=g) * 1e-3
    go_total = go.double()
    if extras:
        go_total = go_total + torch.cat((gagg[s], gagg[r]), 1).double() + gadd2.double()
    (ln * go_total).sum().backward()
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.detach().float()), d(z2.detach().float()), d(y3.detach().float())
    mean = y3.detach().mean(1)
    rstd = (y3.detach().var(1, unbiased=False) + 1e-5).rsqrt()
    stats = d(torch.stack((mean, rstd), 1).float())
    W3t, W2t, W1t = ops.transpose(Pd["W3"]), ops.transpose(Pd["W2"]), ops.transpose(Pd["W1"])
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    god = d(go)
    gz1 = torch.full((M, 128), float("nan"), device=dev)
    ge = torch.full((M, 128), float("nan"), device=dev)
    nwg = L.load().gfv_rowtile_dw_partials_m(M)
    part = torch.full((nwg, L.DW_FUSED_FLOATS), float("nan"), device=dev)
    kw = dict(gadd=d(gagg), gadd_s=d(s.int()), gadd_r=d(r.int()), in_add=d(gadd2)) if extras else {}
    rc = form == "rc"     # recompute form: neither z2 nor the LayerNorm input is handed over
    if rc:
        kw["rc"] = (Pd["W2"], Pd["b2"], Pd["W3"], Pd["b3"])
        z2d = y3d = None
    args = dict(in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, res=[god], dw_partial=part, wimg=wi,
                family=L.CHAIN_COLUMN_OWNER, **kw)
    layers = [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, aux=z2d), ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=z1d),
              ops.LayerSpec(W1t)]
    assert ops.rowtile_chain(M, [ops.Seg(god)], layers, [ge], query_fused=True, **args)
    ops.rowtile_chain(M, [ops.Seg(god)], layers, [ge], **args)
    assert L.load().gfv_rowtile_last_path() == 5 + 16, L.load().gfv_rowtile_last_path()
    # the chain's own results
    gz1_ref = torch.autograd.grad((ln * go_total).sum(), z1, retain_graph=True)[0] if False else None
    assert rel(ge, X.grad + go.double()) < TOL
    # fused weight gradients: sum of the workgroups' blocks
    tot = part.double().sum(0).cpu()
    dW3, db3 = tot[:16384].view(128, 128), tot[16384:16512]
    dW2, db2 = tot[16512:16512 + 16384].view(128, 128), tot[16512 + 16384:16512 + 16384 + 128]
    dgam, dbet = tot[2 * 16384 + 256:2 * 16384 + 384], tot[2 * 16384 + 384:2 * 16384 + 512]
    pieces = [(dW3, "W3"), (db3, "b3"), (dW2, "W2"), (db2, "b2"), (dgam, "gamma"), (dbet, "beta")]
    for mine, name in pieces:
        assert rel(mine, Pg[name].grad) < TOL, (name, rel(mine, Pg[name].grad))
    # gz1 feeds the separate first-layer weight gradient and the node-level scatter
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(d(e))], M)
    assert rel(dW1, Pg["W1"].grad) < TOL and rel(db1, Pg["b1"].grad) < TOL
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value == 0


@pytest.mark.parametrize("M", [3000, 333])
@pytest.mark.parametrize("rc", [False, True])
def test_column_owner_backward_node_mlp_192_wide(dev, M, rc):
    """NodeBlock dX chain: last layer 192 wide, written as [x part 128 (+ residual) | neighbour-mean part 64] (blocks.py:54
    adjoint); weight gradients of the third and second Linear fused, the first one's by the weight-gradient kernel with the
    row scales the chain launch leaves behind."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M)
    nbm, x = torch.randn(M, 64, generator=g), torch.randn(M, 128, generator=g)
    P = _params(g, 192)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    X = torch.cat((nbm, x), 1).double().requires_grad_(True)
    z1, z2, y3, ln = _ref(Pg, X)
    go = torch.randn(M, 128, generator=g) * torch.logspace(-3, 0, M)[:, None]
    (ln * go.double()).sum().backward()
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.detach().float()), d(z2.detach().float()), d(y3.detach().float())
    stats = d(torch.stack((y3.detach().mean(1), (y3.detach().var(1, unbiased=False) + 1e-5).rsqrt()), 1).float())
    W1t = torch.empty(192, 128, device=dev)          # rows for x first, then nbm (engine._T(perm=True))
    ops.transpose(Pd["W1"], out=W1t[0:128], col0=64, ncols=128)
    ops.transpose(Pd["W1"], out=W1t[128:192], col0=0, ncols=64)
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    gx, gnbm = torch.full((M, 128), float("nan"), device=dev), torch.full((M, 64), float("nan"), device=dev)
    gz1 = torch.full((M, 128), float("nan"), device=dev)
    part = torch.full((L.load().gfv_rowtile_dw_partials_m(M), L.DW_FUSED_FLOATS), float("nan"), device=dev)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    god = d(go)
    rckw = {}
    if rc:
        rckw["rc"] = (Pd["W2"], Pd["b2"], Pd["W3"], Pd["b3"])
        z2d = y3d = None
    layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, aux=z2d),
              ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, save=gz1, aux=z1d), ops.LayerSpec(W1t)]
    kw = dict(res=[god, None], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, dw_partial=part, gscale=gs,
              wimg=wi, family=L.CHAIN_COLUMN_OWNER, **rckw)
    assert ops.rowtile_chain(M, [ops.Seg(god)], layers, [gx, (gnbm, 64)], query_fused=True, **kw)
    ops.rowtile_chain(M, [ops.Seg(god)], layers, [gx, (gnbm, 64)], **kw)
    assert L.load().gfv_rowtile_last_path() == 5 + 16
    assert rel(gx, X.grad[:, 64:] + go.double()) < TOL and rel(gnbm, X.grad[:, :64]) < TOL
    tot = part.double().sum(0).cpu()
    for mine, name in ((tot[:16384].view(128, 128), "W3"), (tot[16384:16512], "b3"), (tot[16512:16512 + 16384].view(128, 128), "W2"),
                       (tot[16512 + 16384:16512 + 16384 + 128], "b2"), (tot[2 * 16384 + 256:2 * 16384 + 384], "gamma"),
                       (tot[2 * 16384 + 384:2 * 16384 + 512], "beta")):
        assert rel(mine, Pg[name].grad) < TOL, (name, rel(mine, Pg[name].grad))
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(d(nbm)), ops.Seg(d(x))], M, gscale=gs[2])
    assert rel(dW1, Pg["W1"].grad) < TOL and rel(db1, Pg["b1"].grad) < TOL


@pytest.mark.parametrize("M", [3000, 333, 16])
@pytest.mark.parametrize("rc", [False, True])
def test_column_owner_backward_without_input_gradient(dev, M, rc):
    """Encoder backward (EPD.py:92-119: the raw inputs need no gradient): a TWO-layer launch whose output is gz1; weight
    gradients of the third and second Linear, bias and LayerNorm gradients fused; the narrow first Linear's (16 input
    columns) by the weight-gradient kernel with the row scales the chain launch leaves behind."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + 1)
    x = torch.randn(M, 16, generator=g) * torch.tensor([1.0] * 12 + [1e-3] * 4)     # geometric columns at mesh-spacing scale
    P = _params(g, 16)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    z1, z2, y3, ln = _ref(Pg, x.double())
    go = torch.randn(M, 128, generator=g) * torch.logspace(-4, 0, M)[:, None]
    (ln * go.double()).sum().backward()
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.detach().float()), d(z2.detach().float()), d(y3.detach().float())
    stats = d(torch.stack((y3.detach().mean(1), (y3.detach().var(1, unbiased=False) + 1e-5).rsqrt()), 1).float())
    wi = _images(dev, [P["W2"], P["W3"]])
    gz1 = torch.full((M, 128), float("nan"), device=dev)
    part = torch.full((L.load().gfv_rowtile_dw_partials_m(M), L.DW_FUSED_FLOATS), float("nan"), device=dev)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    god = d(go)
    rckw = {}
    if rc:
        rckw["rc"] = (Pd["W2"], Pd["b2"], Pd["W3"], Pd["b3"])
        z2d = y3d = None
    layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, aux=z2d),
              ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, aux=z1d)]
    kw = dict(in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, dw_partial=part, gscale=gs, wimg=wi,
              family=L.CHAIN_COLUMN_OWNER, **rckw)
    assert ops.rowtile_chain(M, [ops.Seg(god)], layers, [gz1], query_fused=True, **kw)
    ops.rowtile_chain(M, [ops.Seg(god)], layers, [gz1], **kw)
    assert L.load().gfv_rowtile_last_path() == 5 + 16
    tot = part.double().sum(0).cpu()
    for mine, name in ((tot[:16384].view(128, 128), "W3"), (tot[16384:16512], "b3"), (tot[16512:16512 + 16384].view(128, 128), "W2"),
                       (tot[16512 + 16384:16512 + 16384 + 128], "b2"), (tot[2 * 16384 + 256:2 * 16384 + 384], "gamma"),
                       (tot[2 * 16384 + 384:2 * 16384 + 512], "beta")):
        assert rel(mine, Pg[name].grad) < TOL, (name, rel(mine, Pg[name].grad))
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(d(x), width=16, ld=16)], M, gscale=gs[2], col_scale=True)
    assert rel(dW1, Pg["W1"].grad) < TOL and rel(db1, Pg["b1"].grad) < TOL
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value == 0


@pytest.mark.parametrize("mode", ["recompute", "rematerialize"])
def test_engine_recompute_form_and_its_fallback(dev, mode):
    """The engine's use of the recompute form (GFV_RECOMPUTE=1 / Engine.recompute): the forward of an MLP whose backward runs
    fused keeps z1 only, the backward launch rebuilds z2 and the LayerNorm input.  `rematerialize`: the backward turns out not to
    run fused after all (here: fusing switched off between forward and backward) and gets them from one extra two-layer launch.
    Input gradient and every parameter gradient against the read form of the same engine (1e-5) and against float64 autograd."""
    from gfv import lib as L, ops
    from gfv.engine import Engine, GradStore
    from gfv.ops import Seg
    g = torch.Generator().manual_seed(11)
    M = 30000                                            # (> GFV_CBWD_MAX_M, >= GFV_COLCHAIN_BWD_MIN_M: the fused backward takes it)
    x = torch.randn(M, 128, generator=g)
    P = _params(g, 128)
    names = ["mlp.0.0.weight", "mlp.0.0.bias", "mlp.0.2.weight", "mlp.0.2.bias", "mlp.0.4.weight", "mlp.0.4.bias", "mlp.1.weight", "mlp.1.bias"]
    vals = [P["W1"], P["b1"], P["W2"], P["b2"], P["W3"], P["b3"], P["gamma"], P["beta"]]
    Pd = {n: v.to(dev).contiguous() for n, v in zip(names, vals)}
    # persistent parameter storage: the weight images are built for static address ranges only
    go = torch.randn(M, 128, generator=g) * torch.logspace(-3, 0, M)[:, None]
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    X = x.double().requires_grad_(True)
    (_ref(Pg, X)[3] * go.double()).sum().backward()
    want = dict(zip(names, (Pg[k].grad for k in ("W1", "b1", "W2", "b2", "W3", "b3", "gamma", "beta"))))
    res = {}
    for form in ("read", mode):
        eng = Engine()
        eng.fuse_dw = True                               # (the recompute form IS the fused backward: GFV_FUSE_DW=0 must not reach this test)
        eng.recompute = form != "read"
        xd, god = x.to(dev), go.to(dev)
        grads = GradStore(names, [Pd[n].shape for n in names], dev)
        wi_prev = eng._wi_enter("fwd", Pd)
        try:
            out, _, sv = eng.mlp3_fwd(Pd, "mlp", M, [Seg(xd)])
        finally:
            eng._wi_exit("fwd", wi_prev)
        assert (sv["z2"] is None) == (form != "read") and sv["stats"] is not None
        if form == "rematerialize":
            eng.fuse_dw = False
        gx = torch.empty(M, 128, device=dev)
        eng.prepare_transposes(Pd)
        wi_prev = eng._wi_enter("bwd", Pd)
        try:
            eng.mlp3_bwd(Pd, sv, god, grads, outs=[gx])
            eng.join()
        finally:
            eng._wt_live = False
            eng._wi_exit("bwd", wi_prev)
        torch.cuda.synchronize()
        res[form] = (out.clone(), gx.clone(), {n: grads.view(n).clone() for n in names})
        assert rel(gx, X.grad) < TOL, (form, rel(gx, X.grad))
        for n in names:
            assert rel(grads.view(n), want[n]) < TOL, (form, n, rel(grads.view(n), want[n]))
    assert torch.equal(res["read"][0], res[mode][0])           # the forward's outputs do not depend on what it saves
    assert rel(res[mode][1], res["read"][1]) < TOL


@pytest.mark.parametrize("M", [75499, 2300, 257, 5])
def test_narrow_input_weight_gradient_kernel(dev, M):
    """The encoders' first Linear (EPD.py:92-119: edge_attr [E,15] / node inputs [N,12]): the weight gradient against an input of
    <= 16 columns runs as plain fp32 FMAs (csrc/dw.hip dw_narrow_kernel, round 4) whatever the product form - widths 15 / 12 / 16
    / 1, row strides 16 / 12 / 16 / 4, row counts that leave slabs ragged or empty; every element against its own sum |g x| at
    fp32 accumulation accuracy, the bias gradient too."""
    from gfv import ops
    g = torch.Generator().manual_seed(M)
    d = lambda t: t.to(dev).contiguous()
    G = torch.randn(M, 128, generator=g) * torch.logspace(-4, 0, M)[:, None]
    for width, ld in ((15, 16), (12, 12), (16, 16), (1, 4)):
        x = torch.randn(M, ld, generator=g) * torch.tensor([1.0] * max(width - 3, 1) + [1e-4] * (ld - max(width - 3, 1)))[:ld]
        dW, db = ops.linear_dw(d(G), 128, [ops.Seg(d(x), width=width, ld=ld)], M, col_scale=True)
        assert dW.shape == (128, width)
        ref = G.double().T @ x[:, :width].double()
        mag = G.double().abs().T @ x[:, :width].double().abs()
        e = float(((dW.double().cpu() - ref).abs() / (mag + 1e-300)).max())
        assert e < 2e-6, (width, ld, e)
        assert rel(db, G.double().sum(0)) < 2e-6, (width, ld)
