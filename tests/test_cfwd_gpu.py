"""The column-owner small-tile forward (csrc/cfwd.hip: the 3-layer LayerNorm MLP forward of short launches - a wave owns 32
output columns of a 32- / 64-row tile) against float64 and against the row-owner chain it stands in for (pinned with
family=CHAIN_ROW_OWNER): the NodeBlock shape [nbm 64 | x 128] with residual, the factored EdgeBlock shape (gathered first-layer
addend, out_nores), a plain 128-wide input, the encoders' narrow raw inputs; every saved tensor (z1, z2, the LayerNorm input, the
row statistics), both tile heights, partial tiles, the three split-fp16 product forms, a narrower model (LayerNorm over h < 128
real columns).  Reference: build_mlp, /root/reference/src/FVMmodel/Models/FVGN/EPD.py:10-33."""
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


def _real_cols(h, layout="node"):
    """The real columns of a latent of a model of hidden size h inside the kernels' 128 columns (FVMmodel/padding.py): node
    latents at the front, the two halves of an edge latent at columns 0 and 64."""
    m = torch.zeros(128, dtype=torch.bool)
    if layout == "node":
        m[:h] = True
    else:
        m[:h // 2] = True
        m[64:64 + h // 2] = True
    return m


def _params(g, k_in, h=128, layout="node"):
    s = lambda *sh: torch.randn(*sh, generator=g)
    P = {"W1": s(128, k_in) * (1.0 / k_in ** 0.5), "b1": s(128) * 0.1, "W2": s(128, 128) * 0.09, "b2": s(128) * 0.1,
         "W3": s(128, 128) * 0.09, "b3": s(128) * 0.1, "gamma": 1.0 + 0.1 * s(128), "beta": 0.1 * s(128)}
    if h < 128:   # a narrower model, zero padded to the kernels' 128 columns (FVMmodel/padding.py)
        dead = ~_real_cols(h, layout)
        for k in ("W1", "W2", "W3"):
            P[k][dead, :] = 0
        for k in ("W2", "W3"):
            P[k][:, dead] = 0
        for k in ("b1", "b2", "b3", "gamma", "beta"):
            P[k][dead] = 0
    return P


def _ref(P, X, add=None, h=128, layout="node"):
    P = {k: v.double() for k, v in P.items()}
    z1 = X @ P["W1"].T + P["b1"] + (0 if add is None else add)
    z2 = F.gelu(z1) @ P["W2"].T + P["b2"]
    y3 = F.gelu(z2) @ P["W3"].T + P["b3"]
    real = _real_cols(h, layout)
    mean = y3[:, real].mean(1, keepdim=True)
    var = ((y3[:, real] - mean) ** 2).mean(1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    ln = (y3 - mean) * rstd * P["gamma"] + P["beta"]
    return z1, z2, y3, ln, torch.cat((mean, rstd), 1)


def _run(M, segs, Pd, wi, fam, *, res=None, padd=None, nores=False, stats=True, w1=None):
    from gfv import lib as L, ops
    dev = torch.device("cuda")
    new = lambda *s: torch.full(s, float("nan"), device=dev)
    z1, z2, y3, out = new(M, 128), new(M, 128), new(M, 128), new(M, 128)
    st = new(M, 2) if stats else None
    onr = new(M, 128) if nores else None
    kw = dict(padd=padd[0], padd_s=padd[1], padd_r=padd[2]) if padd is not None else {}
    ops.rowtile_chain(M, segs, [ops.LayerSpec(Pd["W1"] if w1 is None else w1, Pd["b1"], L.OP_BIAS_GELU, save=z1),
                                ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=z2), ops.LayerSpec(Pd["W3"], Pd["b3"])],
                      [out], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"], fin_beta=Pd["beta"], fin_presave=y3, fin_stats=st,
                      res=[res] if res is not None else None, out_nores=onr, wimg=wi, family=fam, **kw)
    path = L.load().gfv_rowtile_last_path()
    torch.cuda.synchronize()
    return dict(z1=z1, z2=z2, y3=y3, out=out, stats=st, nores=onr), path


# 32-row tiles on 8 waves: a partial last tile, one tile, many
@pytest.mark.parametrize("M", [777, 32, 97, 20001, 41003])
def test_small_tile_forward_node_and_edge_shapes(M):
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M)
    d = lambda t: t.cuda().contiguous()
    # (1) NodeBlock: [nbm 64 | x 128], residual x, rows over three decades
    nbm = torch.randn(M, 64, generator=g)
    x = torch.randn(M, 128, generator=g) * torch.logspace(-2, 1, M)[:, None]
    P = _params(g, 192)
    Pd = {k: d(v) for k, v in P.items()}
    wi = _wi([P["W1"], P["W2"], P["W3"]])
    z1, z2, y3, ln, st = _ref(P, torch.cat((nbm, x), 1).double())
    xd = d(x)
    a, pa = _run(M, [ops.Seg(d(nbm)), ops.Seg(xd)], Pd, wi, 0, res=xd)
    b, pb = _run(M, [ops.Seg(d(nbm)), ops.Seg(xd)], Pd, wi, L.CHAIN_ROW_OWNER, res=xd)
    assert pa == 5 + 64 and pb == 5, (pa, pb)
    for k, ref in (("z1", z1), ("z2", z2), ("y3", y3), ("out", ln + x.double()), ("stats", st)):
        assert rel(a[k], ref) < TOL, (k, rel(a[k], ref))
        assert rel(a[k], b[k]) < 4e-6, (k, rel(a[k], b[k]))
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value & 2 == 0   # no hidden activation beyond the fixed split scale's range
    # (2) factored EdgeBlock: e [M,128] + (pab[s][:128] + pab[r][128:]) added to the first pre-activation; out + out_nores
    N = max(8, M // 3)
    e = torch.randn(M, 128, generator=g)
    pab = torch.randn(N, 256, generator=g)
    si, ri = torch.randint(0, N, (M,), generator=g), torch.randint(0, N, (M,), generator=g)
    Wfull = torch.randn(128, 384, generator=g) * 0.06   # [W1a | W1b | W1c]: the launch multiplies the c block (a column block)
    P = _params(g, 128)
    P["W1"] = Wfull[:, 256:384].clone()
    Pd = {k: d(v) for k, v in P.items()}
    Wd = d(Wfull)
    wi = _wi([Wfull, P["W2"], P["W3"]])
    add = pab[si, :128].double() + pab[ri, 128:].double()
    z1, z2, y3, ln, st = _ref(P, e.double(), add=add)
    ed = d(e)
    padd = (d(pab), d(si.int()), d(ri.int()))
    a, pa = _run(M, [ops.Seg(ed)], Pd, wi, 0, res=ed, padd=padd, nores=True, w1=Wd[:, 256:384])
    b, pb = _run(M, [ops.Seg(ed)], Pd, wi, L.CHAIN_ROW_OWNER, res=ed, padd=padd, nores=True, w1=Wd[:, 256:384])
    assert pa == 5 + 64 and pb == 5, (pa, pb)
    for k, ref in (("z1", z1), ("z2", z2), ("y3", y3), ("nores", ln), ("out", ln + e.double()), ("stats", st)):
        assert rel(a[k], ref) < TOL, (k, rel(a[k], ref))
        assert rel(a[k], b[k]) < 4e-6, (k, rel(a[k], b[k]))


@pytest.mark.parametrize("width", [12, 15, 128])
def test_small_tile_forward_encoder_and_plain_inputs(width):
    """One narrow raw-input segment (the encoders: x [N,12], edge_attr [E,15] stored with row stride 16), geometric columns at
    mesh-spacing scale; and a plain 128-wide input without residual, without saved statistics."""
    from gfv import lib as L, ops
    M = 1500
    g = torch.Generator().manual_seed(width)
    d = lambda t: t.cuda().contiguous()
    ld = 16 if width == 15 else width
    buf = torch.zeros(M, ld)
    buf[:, :width] = torch.randn(M, width, generator=g)
    if width < 128:
        buf[:, :4] *= 1e-3
    P = _params(g, width)
    Pd = {k: d(v) for k, v in P.items()}
    wi = _wi([P["W1"], P["W2"], P["W3"]])
    z1, z2, y3, ln, st = _ref(P, buf[:, :width].double())
    seg = ops.Seg(d(buf), width=width, ld=ld)
    a, pa = _run(M, [seg], Pd, wi, 0, stats=(width != 128))
    b, pb = _run(M, [seg], Pd, wi, L.CHAIN_ROW_OWNER, stats=(width != 128))
    assert pa & 64 and not (pb & 64), (pa, pb)
    for k, ref in (("z1", z1), ("z2", z2), ("y3", y3), ("out", ln)):
        assert rel(a[k], ref) < TOL, (k, rel(a[k], ref))
        assert rel(a[k], b[k]) < 4e-6, (k, rel(a[k], b[k]))


@pytest.mark.parametrize("form", [2, 3])
def test_small_tile_forward_single_product_forms(form):
    """The reduced-precision forms (one fp16 x fp16 / bf16 x bf16 product per term): the small-tile kernel against the row-owner
    chain in the same form (both round the same operands; the hidden activations' split scales differ) and, loosely, float64."""
    from gfv import lib as L, ops
    lib = L.load()
    M = 900
    g = torch.Generator().manual_seed(form)
    d = lambda t: t.cuda().contiguous()
    nbm, x = torch.randn(M, 64, generator=g), torch.randn(M, 128, generator=g)
    P = _params(g, 192)
    Pd = {k: d(v) for k, v in P.items()}
    _, _, _, ln, _ = _ref(P, torch.cat((nbm, x), 1).double())
    lib.gfv_set_f16split(form)
    try:
        wi = _wi([P["W1"], P["W2"], P["W3"]])
        a, pa = _run(M, [ops.Seg(d(nbm)), ops.Seg(d(x))], Pd, wi, 0)
        b, pb = _run(M, [ops.Seg(d(nbm)), ops.Seg(d(x))], Pd, wi, L.CHAIN_ROW_OWNER)
    finally:
        lib.gfv_set_f16split(1)
    assert pa & 64 and not (pb & 64), (pa, pb)
    tol = 4e-3 if form == 2 else 3e-2
    assert rel(a["out"], ln) < tol and rel(a["out"], b["out"]) < tol, (rel(a["out"], ln), rel(a["out"], b["out"]))


@pytest.mark.parametrize("h,layout", [(64, "node"), (32, "edge"), (112, "edge")])
def test_small_tile_forward_narrow_model_layernorm_width(h, layout):
    """hidden_size h < 128 zero padded to 128 columns: LayerNorm statistics over the h real columns WHEREVER they sit (node
    latents at the front, the halves of an edge latent at columns 0 and 64), padded columns stay exactly 0."""
    from gfv import lib as L, ops
    lib = L.load()
    M = 640
    g = torch.Generator().manual_seed(h)
    d = lambda t: t.cuda().contiguous()
    real = _real_cols(h, layout)
    x = torch.randn(M, 128, generator=g)
    x[:, ~real] = 0
    P = _params(g, 128, h=h, layout=layout)
    P["W1"][:, ~real] = 0
    Pd = {k: d(v) for k, v in P.items()}
    wi = _wi([P["W1"], P["W2"], P["W3"]])
    z1, z2, y3, ln, st = _ref(P, x.double(), h=h, layout=layout)
    assert lib.gfv_set_hidden_size(h) == 0
    try:
        xd = d(x)
        a, pa = _run(M, [ops.Seg(xd)], Pd, wi, 0, res=xd)
        b, pb = _run(M, [ops.Seg(xd)], Pd, wi, L.CHAIN_ROW_OWNER, res=xd)
    finally:
        lib.gfv_set_hidden_size(128)
    assert pa & 64 and not (pb & 64), (pa, pb)
    for k, ref in (("y3", y3), ("out", ln + x.double()), ("stats", st)):
        assert rel(a[k], ref) < TOL, (k, rel(a[k], ref))
        assert rel(a[k], b[k]) < 4e-6, (k, rel(a[k], b[k]))
    assert bool((a["out"][:, ~real.cuda()] == 0).all())


@pytest.mark.parametrize("M", [900, 33])
def test_small_tile_forward_decoder_shape(M):
    """The decoder (EPD.py:199-219): 128 -> 128 -> 128 -> 3, GELUs, no LayerNorm; the output has row stride 3."""
    from gfv import lib as L, ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(M)
    d = lambda t: t.cuda().contiguous()
    x = torch.randn(M, 128, generator=g)
    P = _params(g, 128)
    P["W3"], P["b3"] = torch.randn(3, 128, generator=g) * 0.09, torch.randn(3, generator=g) * 0.1
    Pd = {k: d(v) for k, v in P.items()}
    wi = _wi([P["W1"], P["W2"], P["W3"]])
    Pq = {k: v.double() for k, v in P.items()}
    z1 = x.double() @ Pq["W1"].T + Pq["b1"]
    z2 = F.gelu(z1) @ Pq["W2"].T + Pq["b2"]
    y = F.gelu(z2) @ Pq["W3"].T + Pq["b3"]
    res = {}
    for fam in (0, L.CHAIN_ROW_OWNER):
        s1, s2 = torch.full((M, 128), float("nan"), device=dev), torch.full((M, 128), float("nan"), device=dev)
        out = torch.full((M, 3), float("nan"), device=dev)
        ops.rowtile_chain(M, [ops.Seg(d(x))], [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU, save=s1),
                                               ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=s2), ops.LayerSpec(Pd["W3"], Pd["b3"])],
                          [out], wimg=wi, family=fam)
        res[fam] = (s1, s2, out, L.load().gfv_rowtile_last_path())
    torch.cuda.synchronize()
    a, b = res[0], res[L.CHAIN_ROW_OWNER]
    assert a[3] & 64 and not (b[3] & 64), (a[3], b[3])
    for mine, other, ref, name in ((a[0], b[0], z1, "z1"), (a[1], b[1], z2, "z2"), (a[2], b[2], y, "out")):
        assert rel(mine, ref) < TOL, (name, rel(mine, ref))
        assert rel(mine, other) < 4e-6, (name, rel(mine, other))

