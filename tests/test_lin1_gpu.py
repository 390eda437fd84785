"""The lean single-layer kernel (csrc/lin1.hip: one Linear over the node rows, image staged in LDS once per 128-row workgroup)
against the chain kernel it stands in for (pinned with family=CHAIN_ROW_OWNER) and against float64: plain Linear + bias +
residual, two row-stacked Linears of one input with in_add / in_save (in_project_fx | in_project_x, the EdgeBlock's node
projection), two column-stacked Linears of two inputs (their adjoint), a 256-wide input behind a GELU (linear_post)."""
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


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _wi(ws):
    from gfv import ops
    wmax = torch.stack([w.abs().max() for w in ws]).max().reshape(1).cuda()
    wi = ops.WeightImages(torch.device("cuda"), wmax)
    wi.static = [(0, 1 << 62)]
    return wi


def _both(M, segs, layer, outs, wi, per_run=None, **kw):
    """Run the launch on the lean kernel (outs[0]) and on the chain kernel (outs[1]); returns the last-path codes.
    per_run: {keyword: [value for the lean run, value for the chain run]}."""
    from gfv import lib as L, ops
    paths = []
    for i, fam in enumerate((0, L.CHAIN_ROW_OWNER)):
        extra = {k: v[i] for k, v in (per_run or {}).items()}
        ops.rowtile_chain(M, segs, [layer], outs[i], wimg=wi, family=fam, **kw, **extra)
        paths.append(L.load().gfv_rowtile_last_path())
    torch.cuda.synchronize()
    return paths


@pytest.mark.parametrize("M", [5000, 1025, 2048, 17000, 33])   # (<= 16 384 rows: the small-tile form, csrc/lin1s.hip; above: csrc/lin1.hip)
def test_lean_single_layer_kernel_against_the_chain_kernel_and_float64(M):
    from gfv import ops
    g = torch.Generator().manual_seed(M)
    d = lambda t: t.cuda().contiguous()
    x = torch.randn(M, 128, generator=g) * torch.logspace(-3, 1, M)[:, None]      # rows over four decades
    e = torch.randn(M, 128, generator=g)
    r = torch.randn(M, 128, generator=g)
    W = torch.randn(128, 384, generator=g) * 0.1
    Wp = torch.randn(128, 256, generator=g) * 0.1
    b1, b2 = torch.randn(128, generator=g), torch.randn(128, generator=g)
    Wd, Wpd, b1d, b2d, xd, ed, rd = d(W), d(Wp), d(b1), d(b2), d(x), d(e), d(r)
    wi = _wi([W, Wp])
    new = lambda *s: [torch.full(s, float("nan"), device="cuda") for _ in range(2)]

    # (1) plain Linear + bias + residual (to_out)
    o = new(M, 128)
    paths = _both(M, [ops.Seg(xd)], ops.LayerSpec(Wd[:, 0:128], b1d), [[o[0]], [o[1]]], wi, res=[rd])
    assert paths == [5 + 32, 5], paths
    ref = x.double() @ W[:, 0:128].double().T + b1.double() + r.double()
    assert rel(o[0], ref) < TOL and rel(o[0], o[1]) < 2e-6

    # (2) two Linears of one input, in_add + in_save, outputs side by side in one [M, 256] buffer (node projection / in_project)
    o, xs = new(M, 256), new(M, 128)
    paths = _both(M, [ops.Seg(xd)], ops.LayerSpec(Wd[:, 0:128], b1d, stack=Wd[:, 128:256], bias2=b2d),
                  [[(o[0], 256), (o[0].data_ptr() + 512, 256)], [(o[1], 256), (o[1].data_ptr() + 512, 256)]], wi,
                  per_run={"in_save": xs}, in_add=ed)
    assert paths == [5 + 32, 5], paths
    xin = (x + e).double()
    ref = torch.cat((xin @ W[:, 0:128].double().T + b1.double(), xin @ W[:, 128:256].double().T + b2.double()), 1)
    assert rel(o[0], ref) < TOL and rel(o[0], o[1]) < 2e-6 and rel(xs[0], xin) < 1e-7 and torch.equal(xs[0], xs[1])

    # (3) the adjoint: two inputs, two column-stacked blocks, residual
    o = new(M, 128)
    paths = _both(M, [ops.Seg(xd), ops.Seg(ed)], ops.LayerSpec(Wd[:, 0:128], stack_cols=Wd[:, 256:384]), [[o[0]], [o[1]]], wi, res=[rd])
    assert paths == [5 + 32, 5], paths
    ref = x.double() @ W[:, 0:128].double().T + e.double() @ W[:, 256:384].double().T + r.double()
    assert rel(o[0], ref) < TOL and rel(o[0], o[1]) < 2e-6

    # (4) a 256-wide input (two halves of one [M, 256] buffer) behind a GELU, bias, residual (linear_post)
    from gfv import lib as L
    z = torch.cat((x, e), 1)
    zd = d(z)
    o = new(M, 128)
    segs = [ops.Seg(zd, width=128, ld=256), ops.Seg(zd, width=128, ld=256, offset=128)]
    paths = _both(M, segs, ops.LayerSpec(Wpd, b1d), [[o[0]], [o[1]]], wi, in_op=L.IN_GELU, res=[rd])
    assert paths == [5 + 32, 5], paths
    ref = F.gelu(z.double()) @ Wp.double().T + b1.double() + r.double()
    assert rel(o[0], ref) < TOL and rel(o[0], o[1]) < 2e-6


def test_lean_kernel_leaves_what_it_does_not_cover_to_the_chain():
    """A one-pass GELU' launch that asks for the scales of its output rows or a gathered segment stays on the chain kernel."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(3)
    M = 3000
    x = (torch.randn(M, 128, generator=g)).cuda()
    W = (torch.randn(128, 128, generator=g) * 0.1).cuda()
    wi = _wi([W])
    o = torch.empty(M, 128, device="cuda")
    gs = torch.zeros(3, ops.gscale_ld(M), device="cuda")
    z = torch.randn(M, 128, generator=g).cuda()
    # a one-pass GELU' launch without residual also leaves the scales of its OUTPUT rows (slot 1): the chain kernel's
    ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W, None, L.OP_MUL_DGELU, aux=z)], [o], wimg=wi, gscale=gs)
    assert L.load().gfv_rowtile_last_path() == 5
    idx = torch.randint(0, M, (M,), generator=g).int().cuda()
    ops.rowtile_chain(M, [ops.Seg(x, idx)], [ops.LayerSpec(W)], [o], wimg=wi)
    assert L.load().gfv_rowtile_last_path() == 5
    ops.rowtile_chain(512, [ops.Seg(x)], [ops.LayerSpec(W)], [o], wimg=wi)   # (round 5: short launches take the small-tile form)
    assert L.load().gfv_rowtile_last_path() == 5 + 32
    ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W)], [o], wimg=wi)
    assert L.load().gfv_rowtile_last_path() == 5 + 32


@pytest.mark.parametrize("M", [3000, 1100, 16500])
def test_lean_kernel_layernorm_prologue_and_gelu_prime_epilogue(M):
    """The two remaining Transolver launches the lean kernel takes: linear_pre behind LayerNorm ln_2 (128 -> 256) and the
    adjoint of linear_post (gradient (+ addend, kept) x W_post -> 256 wide, x gelu'(z)), with the per-16-row scales of the
    kept gradient rows for the weight-gradient launch - against the chain kernel and float64."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + 9)
    d = lambda t: t.cuda().contiguous()
    x = torch.randn(M, 128, generator=g) * torch.logspace(-2, 1, M)[:, None] + 0.3
    Wpre = torch.randn(256, 128, generator=g) * 0.1
    bpre = torch.randn(256, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
    wi = _wi([Wpre])
    Wd, bd, gd, btd, xd = d(Wpre), d(bpre), d(gam), d(bet), d(x)
    o = [torch.full((M, 256), float("nan"), device="cuda") for _ in range(2)]
    paths = _both(M, [ops.Seg(xd)], ops.LayerSpec(Wd, bd),
                  [[(o[0], 256), (o[0].data_ptr() + 512, 256)], [(o[1], 256), (o[1].data_ptr() + 512, 256)]], wi,
                  in_op=L.IN_LN, in_gamma=gd, in_beta=btd)
    assert paths == [5 + 32, 5], paths
    ref = F.layer_norm(x.double(), (128,), gam.double(), bet.double(), 1e-5) @ Wpre.double().T + bpre.double()
    assert rel(o[0], ref) < TOL and rel(o[0], o[1]) < 2e-6
    # adjoint of linear_post: [M,128] gradient + addend -> [M,256], times gelu'(z)
    go = torch.randn(M, 128, generator=g) * torch.logspace(-4, 0, M)[:, None]
    ga = torch.randn(M, 128, generator=g) * 1e-2
    z = torch.randn(M, 256, generator=g)
    Wpost_t = torch.randn(256, 128, generator=g) * 0.1          # = W_post^T: [256 outputs of this launch, 128 inputs]
    wi2 = _wi([Wpost_t])
    Wt, god, gad, zd = d(Wpost_t), d(go), d(ga), d(z)
    o = [torch.full((M, 256), float("nan"), device="cuda") for _ in range(2)]
    gsum = [torch.full((M, 128), float("nan"), device="cuda") for _ in range(2)]
    gs = [torch.zeros(3, ops.gscale_ld(M), device="cuda") for _ in range(2)]
    paths = _both(M, [ops.Seg(god)], ops.LayerSpec(Wt, None, L.OP_MUL_DGELU, aux=zd),
                  [[(o[0], 256), (o[0].data_ptr() + 512, 256)], [(o[1], 256), (o[1].data_ptr() + 512, 256)]], wi2,
                  per_run={"in_save": gsum, "gscale": gs}, in_add=gad)
    assert paths == [5 + 32, 5], paths
    zz = z.double().requires_grad_(True)
    dg = torch.autograd.grad(F.gelu(zz).sum(), zz)[0]
    ref = ((go + ga).double() @ Wpost_t.double().T) * dg
    assert rel(o[0], ref) < TOL and rel(o[0], o[1]) < 2e-6
    assert torch.equal(gsum[0], gsum[1]) and torch.equal(gs[0][0, :(M + 15) // 16], gs[1][0, :(M + 15) // 16])


@pytest.mark.parametrize("M", [3000, 1100, 1089, 16500])
def test_lean_kernel_layernorm_backward_epilogue(M):
    """The adjoint of linear_pre behind LayerNorm ln_2: g_fx1 = LNbwd(g_z [M,256] W; fx1, gamma) + residual, and the per-tile
    (dgamma, dbeta) partials for the reduction launch - against the chain kernel and float64 autograd."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + 21)
    d = lambda t: t.cuda().contiguous()
    fx1 = torch.randn(M, 128, generator=g) * 2 + 0.5
    gz = torch.randn(M, 256, generator=g) * torch.logspace(-4, 0, M)[:, None]
    gout = torch.randn(M, 128, generator=g) * 1e-2
    Wpre = torch.randn(256, 128, generator=g) * 0.1                 # linear_pre.weight [256, 128]; the launch multiplies by W^T's transpose
    gam = 1 + 0.1 * torch.randn(128, generator=g)
    Wt = Wpre.t().contiguous()                                      # [128, 256]: out[n] = sum_k g_z[k] Wpre[k, n]
    wi = _wi([Wt])
    Wd, gzd, fxd, gd, god = d(Wt), d(gz), d(fx1), d(gam), d(gout)
    tiles = ops.rowtile_tiles(M)
    o = [torch.full((M, 128), float("nan"), device="cuda") for _ in range(2)]
    part = [torch.full((tiles, 2, 128), float("nan"), device="cuda") for _ in range(2)]
    segs = [ops.Seg(gzd, width=128, ld=256), ops.Seg(gzd, width=128, ld=256, offset=128)]
    paths = _both(M, segs, ops.LayerSpec(Wd), [[o[0]], [o[1]]], wi, per_run={"ln_partial": part},
                  fin_op=L.FIN_LNBWD, fin_gamma=gd, fin_aux=fxd, res=[god])
    assert paths == [5 + 32, 5], paths
    x = fx1.double().requires_grad_(True)
    gm = gam.double().requires_grad_(True)
    bt = torch.zeros(128, dtype=torch.float64, requires_grad=True)
    yln = F.layer_norm(x, (128,), gm, bt, 1e-5)
    gy = gz.double() @ Wpre.double()                                # gradient wrt the LayerNorm output
    gx, ggam, gbet = torch.autograd.grad(yln, (x, gm, bt), gy)
    assert rel(o[0], gx + gout.double()) < TOL and rel(o[0], o[1]) < 2e-6
    assert rel(part[0][:, 0].sum(0), ggam) < TOL and rel(part[0][:, 1].sum(0), gbet) < TOL
    assert rel(part[0].sum(0), part[1].sum(0)) < 2e-6


@pytest.mark.parametrize("M,E", [(3000, 9000), (1100, 2500), (16500, 50000)])
def test_lean_kernel_segmented_sum_prologue(M, E):
    """The per-side scatter of the factored EdgeBlock's adjoint in front of its node-level Linear: rows = sums of gz1 rows by
    sender / by receiver (CSR, some rows empty), the assembled rows written out - against the chain kernel and float64."""
    from gfv import ops
    g = torch.Generator().manual_seed(M)
    d = lambda t: t.cuda().contiguous()
    gz1 = torch.randn(E, 128, generator=g) * torch.logspace(-3, 0, E)[:, None]
    W = torch.randn(128, 256, generator=g) * 0.1
    wi = _wi([W])
    csrs, dense = [], []
    for side in range(2):
        idx = torch.randint(0, M - 7, (E,), generator=g)           # the last rows stay empty
        order = torch.argsort(idx, stable=True)
        rowptr = torch.zeros(M + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(torch.bincount(idx, minlength=M), 0)
        csrs.append((d(rowptr.int()), d(order.int())))
        dense.append(torch.zeros(M, 128, dtype=torch.float64).index_add_(0, idx, gz1.double()))
    gzd, Wd = d(gz1), d(W)
    o = [torch.full((M, 128), float("nan"), device="cuda") for _ in range(2)]
    sv = [[torch.full((M, 128), float("nan"), device="cuda") for _ in range(2)] for _ in range(2)]
    from gfv import lib as L
    paths = []
    for i, fam in enumerate((0, L.CHAIN_ROW_OWNER)):
        ops.rowtile_chain(M, [ops.Seg(gzd, csr=csrs[0], save=sv[i][0]), ops.Seg(gzd, csr=csrs[1], save=sv[i][1])],
                          [ops.LayerSpec(Wd)], [o[i]], wimg=wi, family=fam)
        paths.append(L.load().gfv_rowtile_last_path())
    torch.cuda.synchronize()
    assert paths == [5 + 32, 5], paths
    ref = dense[0] @ W[:, 0:128].double().T + dense[1] @ W[:, 128:256].double().T
    assert rel(o[0], ref) < TOL and rel(o[0], o[1]) < 2e-6
    for side in range(2):
        assert rel(sv[0][side], dense[side]) < 1e-6 and torch.equal(sv[0][side], sv[1][side])
