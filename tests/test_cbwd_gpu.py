"""GPU: the column-owner small-tile backward (csrc/cbwd.hip) - the dX chain of a 3-layer LayerNorm MLP on short launches -
against float64 autograd of the fused MLP (EPD.py:10-33 build_mlp inside blocks.py EdgeBlock / NodeBlock, the encoders);
tolerance 1e-5 relative (fp32), as for every other kernel.  Every piece the launch leaves behind is checked: the input
gradient(s), g3 / gz2 / gz1 through the weight gradients the side-queue launch forms from them (with the row scales the chain
launch wrote), and the per-tile (dgamma, dbeta) partial sums - one row per 32 rows here (include/gfv.h)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _families_on(gfv_limits):
    """These are the tests OF the small-tile families: they take their launches whatever GFV_CBWD / GFV_CFWD / GFV_CTRANS /
    GFV_LIN1S say in the environment (the suite is also run with them switched off: the model-level tests then cover the
    large-launch kernels at every size)."""
    gfv_limits(GFV_CBWD=1, GFV_CFWD=1, GFV_CTRANS=1, GFV_LIN1S=1)

TOL = 1e-5
PATH = 5 + 128


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


def _ref(P, X):
    z1 = F.linear(X, P["W1"], P["b1"])
    z2 = F.linear(F.gelu(z1), P["W2"], P["b2"])
    y3 = F.linear(F.gelu(z2), P["W3"], P["b3"])
    return z1, z2, y3, F.layer_norm(y3, (128,), P["gamma"], P["beta"], 1e-5)


def _images(dev, Ws):
    from gfv import ops
    wmax = torch.stack([w.abs().max() for w in Ws]).max().reshape(1).to(dev)
    wi = ops.WeightImages(dev, wmax)
    wi.static = [(0, 1 << 62)]
    return wi


def _stats(y3, d):
    return d(torch.stack((y3.detach().mean(1), (y3.detach().var(1, unbiased=False) + 1e-5).rsqrt()), 1).float())


def _no_flags():
    from gfv import lib as L
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value == 0


def _ln_sums(part, M):
    from gfv import ops
    n = ops.last_ln_rows()
    assert n == (M + 31) // 32 == ops.ln_rows(M)
    tot = part[:n].double().sum(0).cpu()
    return tot[0], tot[1]


@pytest.mark.parametrize("M", [4000, 97, 1024, 33, 16384, 41003])
@pytest.mark.parametrize("extras", [True, False])
def test_small_tile_backward_edge_mlp(dev, M, extras, gfv_limits):
    """EdgeBlock backward in its factored form: LayerNorm backward (+ the gathered and the plain addend of the incoming
    gradient), the three transposed layers, residual.  (41 003 rows: beyond the default row limit of the family: moved for
    the test, gfv_set_limit.)"""
    from gfv import lib as L, ops
    if M > 16384:
        gfv_limits(GFV_CBWD_MAX_M=100000)
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
    W3t, W2t, W1t = ops.transpose(Pd["W3"]), ops.transpose(Pd["W2"]), ops.transpose(Pd["W1"])
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    god = d(go)
    nan = lambda *shape: torch.full(shape, float("nan"), device=dev)
    g3, gz2, gz1, ge = nan(M, 128), nan(M, 128), nan(M, 128), nan(M, 128)
    part = nan(ops.ln_rows(M), 2, 128)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    kw = dict(gadd=d(gagg), gadd_s=d(s.int()), gadd_r=d(r.int()), in_add=d(gadd2)) if extras else {}
    layers = [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=z2d), ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=z1d),
              ops.LayerSpec(W1t)]
    have = ops.rowtile_chain(M, [ops.Seg(god)], layers, [ge], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=_stats(y3, d),
                             in_save=g3, ln_partial=part, res=[god], gscale=gs, wimg=wi, **kw)
    assert L.load().gfv_rowtile_last_path() == PATH, L.load().gfv_rowtile_last_path()
    assert have
    assert rel(ge, X.grad + go.double()) < TOL
    dgam, dbet = _ln_sums(part, M)
    assert rel(dgam, Pg["gamma"].grad) < TOL and rel(dbet, Pg["beta"].grad) < TOL
    # what the launch leaves for the weight-gradient launch of the side queue, through that launch
    a2, a1 = d(F.gelu(z2.detach()).float()), d(F.gelu(z1.detach()).float())
    for G, A, sc, wn, bn in ((g3, a2, gs[0], "W3", "b3"), (gz2, a1, gs[1], "W2", "b2"), (gz1, d(e), gs[2], "W1", "b1")):
        dW, db = ops.linear_dw(G, 128, [ops.Seg(A)], M, gscale=sc)
        assert rel(dW, Pg[wn].grad) < TOL and rel(db, Pg[bn].grad) < TOL, (wn, rel(dW, Pg[wn].grad), rel(db, Pg[bn].grad))
    _no_flags()


@pytest.mark.parametrize("M", [3000, 333, 5184])
def test_small_tile_backward_node_mlp_192_wide(dev, M):
    """NodeBlock dX chain: last layer 192 wide, written as [x part 128 (+ residual) | neighbour-mean part 64] (blocks.py:54
    adjoint)."""
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
    W1t = torch.empty(192, 128, device=dev)          # rows for x first, then nbm (engine._T(perm=True))
    ops.transpose(Pd["W1"], out=W1t[0:128], col0=64, ncols=128)
    ops.transpose(Pd["W1"], out=W1t[128:192], col0=0, ncols=64)
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    nan = lambda *shape: torch.full(shape, float("nan"), device=dev)
    gx, gnbm, g3, gz2, gz1 = nan(M, 128), nan(M, 64), nan(M, 128), nan(M, 128), nan(M, 128)
    part = nan(ops.ln_rows(M), 2, 128)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    god = d(go)
    layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, save=gz2, aux=z2d),
              ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, save=gz1, aux=z1d), ops.LayerSpec(W1t)]
    ops.rowtile_chain(M, [ops.Seg(god)], layers, [gx, (gnbm, 64)], res=[god, None], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d,
                      in_stats=_stats(y3, d), in_save=g3, ln_partial=part, gscale=gs, wimg=wi)
    assert L.load().gfv_rowtile_last_path() == PATH, L.load().gfv_rowtile_last_path()
    assert rel(gx, X.grad[:, 64:] + go.double()) < TOL and rel(gnbm, X.grad[:, :64]) < TOL
    dgam, dbet = _ln_sums(part, M)
    assert rel(dgam, Pg["gamma"].grad) < TOL and rel(dbet, Pg["beta"].grad) < TOL
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(d(nbm)), ops.Seg(d(x))], M, gscale=gs[2])
    assert rel(dW1, Pg["W1"].grad) < TOL and rel(db1, Pg["b1"].grad) < TOL
    dW3, db3 = ops.linear_dw(g3, 128, [ops.Seg(d(F.gelu(z2.detach()).float()))], M, gscale=gs[0])
    assert rel(dW3, Pg["W3"].grad) < TOL and rel(db3, Pg["b3"].grad) < TOL
    _no_flags()


@pytest.mark.parametrize("M", [3000, 333, 16])
def test_small_tile_backward_without_input_gradient(dev, M):
    """Encoder backward (EPD.py:92-119: the raw inputs need no gradient): a TWO-layer launch whose output is gz1."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + 1)
    x = torch.randn(M, 16, generator=g) * torch.tensor([1.0] * 12 + [1e-3] * 4)     # geometric columns at mesh-spacing scale
    P = _params(g, 16)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    z1, z2, y3, ln = _ref(Pg, x.double())
    go = torch.randn(M, 128, generator=g) * torch.logspace(-4, 0, M)[:, None]
    gadd2 = torch.randn(M, 128, generator=g) * 1e-2
    (ln * (go.double() + gadd2.double())).sum().backward()
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.detach().float()), d(z2.detach().float()), d(y3.detach().float())
    wi = _images(dev, [P["W2"], P["W3"]])
    nan = lambda *shape: torch.full(shape, float("nan"), device=dev)
    g3, gz2, gz1 = nan(M, 128), nan(M, 128), nan(M, 128)
    part = nan(ops.ln_rows(M), 2, 128)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, save=gz2, aux=z2d),
              ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, aux=z1d)]
    ops.rowtile_chain(M, [ops.Seg(d(go))], layers, [gz1], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=_stats(y3, d),
                      in_save=g3, in_add=d(gadd2), ln_partial=part, gscale=gs, wimg=wi)
    assert L.load().gfv_rowtile_last_path() == PATH, L.load().gfv_rowtile_last_path()
    dgam, dbet = _ln_sums(part, M)
    assert rel(dgam, Pg["gamma"].grad) < TOL and rel(dbet, Pg["beta"].grad) < TOL
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(d(x), width=16, ld=16)], M, gscale=gs[2], col_scale=True)
    assert rel(dW1, Pg["W1"].grad) < TOL and rel(db1, Pg["b1"].grad) < TOL
    dW2, db2 = ops.linear_dw(gz2, 128, [ops.Seg(d(F.gelu(z1.detach()).float()))], M, gscale=gs[1])
    assert rel(dW2, Pg["W2"].grad) < TOL and rel(db2, Pg["b2"].grad) < TOL
    dW3, db3 = ops.linear_dw(g3, 128, [ops.Seg(d(F.gelu(z2.detach()).float()))], M, gscale=gs[0])
    assert rel(dW3, Pg["W3"].grad) < TOL and rel(db3, Pg["b3"].grad) < TOL
    _no_flags()


def test_small_tile_backward_matches_the_row_owner_chain(dev):
    """The same launch through the row-owner chain (family = CHAIN_ROW_OWNER) and through the small-tile kernel: input gradient
    and saved gradients agree to 1e-5 of their maxima; the row-owner launch fills one ln_partial row per 64 rows."""
    from gfv import lib as L, ops
    M = 2500
    g = torch.Generator().manual_seed(5)
    e = torch.randn(M, 128, generator=g)
    P = _params(g, 128)
    z1, z2, y3, _ = _ref({k: v.double() for k, v in P.items()}, e.double())
    go = torch.randn(M, 128, generator=g) * torch.logspace(-4, 0, M)[:, None]
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.float()), d(z2.float()), d(y3.float())
    W3t, W2t, W1t = ops.transpose(Pd["W3"]), ops.transpose(Pd["W2"]), ops.transpose(Pd["W1"])
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    god = d(go)
    res = {}
    for fam in (0, L.CHAIN_ROW_OWNER):
        nan = lambda *shape: torch.full(shape, float("nan"), device=dev)
        g3, gz2, gz1, ge = nan(M, 128), nan(M, 128), nan(M, 128), nan(M, 128)
        part = nan(ops.ln_rows(M), 2, 128)
        layers = [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=z2d), ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=z1d),
                  ops.LayerSpec(W1t)]
        ops.rowtile_chain(M, [ops.Seg(god)], layers, [ge], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=_stats(y3, d),
                          in_save=g3, ln_partial=part, res=[god], wimg=wi, family=fam)
        assert L.load().gfv_rowtile_last_path() == (PATH if fam == 0 else 5)
        n = ops.last_ln_rows()
        assert n == ((M + 31) // 32 if fam == 0 else (M + 63) // 64)
        res[fam] = (ge, g3, gz2, gz1, part[:n].sum(0))
    for a, b in zip(res[0], res[L.CHAIN_ROW_OWNER]):
        assert rel(a, b) < TOL


@pytest.mark.parametrize("form", [2, 3])
def test_small_tile_backward_single_product_forms(dev, form):
    """The reduced-precision forms (one fp16 x fp16 / bf16 x bf16 product per term): the small-tile kernel against the row-owner
    chain in the same form (both round the same operands; their gradient-row scales differ) and, loosely, float64."""
    from gfv import lib as L, ops
    lib = L.load()
    M = 1500
    g = torch.Generator().manual_seed(form)
    e = torch.randn(M, 128, generator=g)
    P = _params(g, 128)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    X = e.double().requires_grad_(True)
    z1, z2, y3, ln = _ref(Pg, X)
    go = torch.randn(M, 128, generator=g) * torch.logspace(-3, 0, M)[:, None]
    (ln * go.double()).sum().backward()
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.detach().float()), d(z2.detach().float()), d(y3.detach().float())
    W3t, W2t, W1t = ops.transpose(Pd["W3"]), ops.transpose(Pd["W2"]), ops.transpose(Pd["W1"])
    god = d(go)
    res = {}
    lib.gfv_set_f16split(form)
    try:
        wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
        for fam in (0, L.CHAIN_ROW_OWNER):
            ge, gz1 = torch.full((M, 128), float("nan"), device=dev), torch.full((M, 128), float("nan"), device=dev)
            part = torch.full((ops.ln_rows(M), 2, 128), float("nan"), device=dev)
            layers = [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, aux=z2d), ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=z1d),
                      ops.LayerSpec(W1t)]
            ops.rowtile_chain(M, [ops.Seg(god)], layers, [ge], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=_stats(y3, d),
                              ln_partial=part, res=[god], wimg=wi, family=fam)
            assert bool(lib.gfv_rowtile_last_path() & 128) == (fam == 0)
            res[fam] = (ge, gz1, part[:ops.last_ln_rows()].sum(0))
    finally:
        lib.gfv_set_f16split(1)
    tol = 4e-3 if form == 2 else 3e-2
    assert rel(res[0][0], X.grad + go.double()) < tol
    for a, b in zip(res[0], res[L.CHAIN_ROW_OWNER]):
        assert rel(a, b) < tol, rel(a, b)
    # (dgamma, dbeta) do not pass through a product: fp32 sums in both kernels
    assert rel(res[0][2][0], Pg["gamma"].grad) < TOL and rel(res[0][2][1], Pg["beta"].grad) < TOL


@pytest.mark.parametrize("h", [64, 32])
def test_small_tile_backward_narrow_model_layernorm_width(dev, h):
    """A model of hidden_size h < 128 runs zero padded to 128 columns (FVMmodel/padding.py): LayerNorm statistics over the h real
    columns, gamma = beta = 0 and zero weight rows / columns in the padding - input gradient and (dgamma, dbeta) of the real
    columns against float64 autograd of the h-wide MLP, the padded columns of the input gradient exactly zero."""
    from gfv import lib as L, ops
    lib = L.load()
    M = 2100
    g = torch.Generator().manual_seed(h)
    pad = lambda w, r, c: F.pad(w, (0, c - w.shape[-1], 0, r - w.shape[0])) if w.dim() == 2 else F.pad(w, (0, r - w.shape[0]))
    Ph = dict(W1=torch.randn(h, h, generator=g) * 0.3, b1=torch.randn(h, generator=g) * 0.1,
              W2=torch.randn(h, h, generator=g) * 0.3, b2=torch.randn(h, generator=g) * 0.1,
              W3=torch.randn(h, h, generator=g) * 0.3, b3=torch.randn(h, generator=g) * 0.1,
              gamma=1 + 0.1 * torch.randn(h, generator=g), beta=0.1 * torch.randn(h, generator=g))
    x = torch.randn(M, h, generator=g)
    Pg = {k: v.double().requires_grad_(True) for k, v in Ph.items()}
    X = x.double().requires_grad_(True)
    z1 = F.linear(X, Pg["W1"], Pg["b1"])
    z2 = F.linear(F.gelu(z1), Pg["W2"], Pg["b2"])
    y3 = F.linear(F.gelu(z2), Pg["W3"], Pg["b3"])
    ln = F.layer_norm(y3, (h,), Pg["gamma"], Pg["beta"], 1e-5)
    go = torch.randn(M, h, generator=g) * torch.logspace(-3, 0, M)[:, None]
    (ln * go.double()).sum().backward()
    d = lambda t: t.to(dev).contiguous()
    P = {k: pad(v, 128, 128) for k, v in Ph.items()}
    Pd = {k: d(v) for k, v in P.items()}
    wide = lambda t: d(F.pad(t.detach().float(), (0, 128 - h)))
    stats = d(torch.stack((y3.detach().mean(1), (y3.detach().var(1, unbiased=False) + 1e-5).rsqrt()), 1).float())
    W3t, W2t, W1t = ops.transpose(Pd["W3"]), ops.transpose(Pd["W2"]), ops.transpose(Pd["W1"])
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    god = wide(go)
    ge, gz1 = torch.full((M, 128), float("nan"), device=dev), torch.full((M, 128), float("nan"), device=dev)
    part = torch.full((ops.ln_rows(M), 2, 128), float("nan"), device=dev)
    layers = [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, aux=wide(z2)), ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=wide(z1)),
              ops.LayerSpec(W1t)]
    assert lib.gfv_set_hidden_size(h) == 0
    try:
        ops.rowtile_chain(M, [ops.Seg(god)], layers, [ge], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=wide(y3), in_stats=stats,
                          ln_partial=part, res=[god], wimg=wi)
        assert lib.gfv_rowtile_last_path() == PATH
    finally:
        lib.gfv_set_hidden_size(128)
    assert rel(ge[:, :h], X.grad + go.double()) < TOL
    assert bool((ge[:, h:] == 0).all())
    dgam, dbet = _ln_sums(part, M)
    assert rel(dgam[:h], Pg["gamma"].grad) < TOL and rel(dbet[:h], Pg["beta"].grad) < TOL
    _no_flags()
