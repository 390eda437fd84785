"""GPU: the column-owner persistent chain family (csrc/colchain_kernel.h) against a float64 torch restatement of the
fused MLP (EPD.py:10-33 build_mlp inside blocks.py EdgeBlock / NodeBlock) and against the row-owner family on the same
launch; tolerance 1e-5 relative (fp32), as for every other kernel (tests/test_kernels_gpu.py)."""
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


@pytest.mark.parametrize("M", [5000, 2049, 16, 129])
@pytest.mark.parametrize("padd", [True, False])
def test_column_owner_edge_mlp_forward(dev, M, padd):
    """EdgeBlock forward in its factored form: one 128-wide segment + the gathered first-layer addend, every saved tensor."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + padd)
    n_nodes = 700
    e = torch.randn(M, 128, generator=g) * (1 + 3 * torch.rand(M, 1, generator=g))
    e[3] *= 1e-4          # a tiny row and a big row: the input rows carry their own power-of-two scale
    e[min(5, M - 1)] *= 30.0
    pab = torch.randn(n_nodes, 256, generator=g)
    s = torch.randint(0, n_nodes, (M,), generator=g)
    r = torch.randint(0, n_nodes, (M,), generator=g)
    P = _params(g, 128)
    add = (pab[s, :128] + pab[r, 128:]).double() if padd else None
    z1, z2, y3, ln = _ref(P, e.double(), add)
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    ed, pabd, sd, rd = d(e), d(pab), d(s.int()), d(r.int())
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    res = {}
    for fam in (L.CHAIN_COLUMN_OWNER, L.CHAIN_ROW_OWNER):
        z1d, z2d, y3d, outd, enew = (torch.full((M, 128), float("nan"), device=dev) for _ in range(5))
        ops.rowtile_chain(
            M, [ops.Seg(ed)],
            [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU, save=z1d),
             ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=z2d), ops.LayerSpec(Pd["W3"], Pd["b3"])],
            [outd], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"], fin_beta=Pd["beta"], fin_presave=y3d, res=[ed], out_nores=enew,
            wimg=wi, family=fam, **(dict(padd=pabd, padd_s=sd, padd_r=rd) if padd else {}))
        path = L.load().gfv_rowtile_last_path()
        assert path == (13 if fam == L.CHAIN_COLUMN_OWNER else 5), path
        for mine, want, name in ((z1d, z1, "z1"), (z2d, z2, "z2"), (y3d, y3, "y3"), (enew, ln, "ln"), (outd, ln + e.double(), "out")):
            assert rel(mine, want) < TOL, (fam, name, rel(mine, want))
        res[fam] = outd
    assert rel(res[L.CHAIN_COLUMN_OWNER], res[L.CHAIN_ROW_OWNER]) < 2e-6
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value == 0


@pytest.mark.parametrize("M", [3000, 97])
def test_column_owner_node_mlp_forward(dev, M):
    """NodeBlock forward: input [nbm (64) | x (128)], K = 192, row-gathered first segment."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M)
    nbm, x = torch.randn(M + 50, 64, generator=g), torch.randn(M, 128, generator=g)
    idx = torch.randperm(M + 50, generator=g)[:M]
    P = _params(g, 192)
    z1, z2, y3, ln = _ref(P, torch.cat((nbm[idx], x), 1).double())
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    xd, nbmd = d(x), d(nbm)
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    z1d, z2d, y3d, outd = (torch.full((M, 128), float("nan"), device=dev) for _ in range(4))
    ops.rowtile_chain(M, [ops.Seg(nbmd, d(idx.int())), ops.Seg(xd)],
                      [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU, save=z1d),
                       ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=z2d), ops.LayerSpec(Pd["W3"], Pd["b3"])],
                      [outd], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"], fin_beta=Pd["beta"], fin_presave=y3d, res=[xd],
                      wimg=wi, family=L.CHAIN_COLUMN_OWNER)
    assert L.load().gfv_rowtile_last_path() == 13
    for mine, want, name in ((z1d, z1, "z1"), (z2d, z2, "z2"), (y3d, y3, "y3"), (outd, ln + x.double(), "out")):
        assert rel(mine, want) < TOL, (name, rel(mine, want))


def test_column_owner_hidden_range_flag(dev):
    """Hidden activations are split after a fixed scale: beyond 2^11 the status word says so (include/gfv.h)."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(0)
    M = 64
    P = _params(g, 128)
    P["b1"] = P["b1"] + 5000.0
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    ed = d(torch.randn(M, 128, generator=g))
    out = torch.empty(M, 128, device=dev)
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    ops.rowtile_chain(M, [ops.Seg(ed)],
                      [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU), ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU),
                       ops.LayerSpec(Pd["W3"], Pd["b3"])], [out], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"],
                      fin_beta=Pd["beta"], wimg=wi, family=L.CHAIN_COLUMN_OWNER)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value & 2


@pytest.mark.parametrize("M", [4000, 97, 1024])
@pytest.mark.parametrize("extras", [True, False])
@pytest.mark.parametrize("dw1", [True, False])
def test_column_owner_backward_with_fused_weight_gradients(dev, M, extras, dw1):
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
    nwg = L.load().gfv_rowtile_dw_partials()
    part = torch.full((nwg, L.DW_FUSED_FLOATS_IN if dw1 else L.DW_FUSED_FLOATS), float("nan"), device=dev)
    kw = dict(gadd=d(gagg), gadd_s=d(s.int()), gadd_r=d(r.int()), in_add=d(gadd2)) if extras else {}
    if dw1:
        kw["dw_in"] = d(e)
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
    if dw1:
        pieces += [(tot[2 * 16384 + 512:3 * 16384 + 512].view(128, 128), "W1"), (tot[3 * 16384 + 512:], "b1")]
    for mine, name in pieces:
        assert rel(mine, Pg[name].grad) < TOL, (name, rel(mine, Pg[name].grad))
    # gz1 feeds the separate first-layer weight gradient and the node-level scatter
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(d(e))], M)
    assert rel(dW1, Pg["W1"].grad) < TOL and rel(db1, Pg["b1"].grad) < TOL
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value == 0


@pytest.mark.parametrize("M", [3000, 333])
def test_column_owner_backward_node_mlp_192_wide(dev, M):
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
    part = torch.full((L.load().gfv_rowtile_dw_partials(), L.DW_FUSED_FLOATS), float("nan"), device=dev)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    god = d(go)
    layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, aux=z2d),
              ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, save=gz1, aux=z1d), ops.LayerSpec(W1t)]
    kw = dict(res=[god, None], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, dw_partial=part, gscale=gs,
              wimg=wi, family=L.CHAIN_COLUMN_OWNER)
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
def test_column_owner_backward_without_input_gradient(dev, M):
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
    part = torch.full((L.load().gfv_rowtile_dw_partials(), L.DW_FUSED_FLOATS), float("nan"), device=dev)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    god = d(go)
    layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, aux=z2d),
              ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, aux=z1d)]
    kw = dict(in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, dw_partial=part, gscale=gs, wimg=wi,
              family=L.CHAIN_COLUMN_OWNER)
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
