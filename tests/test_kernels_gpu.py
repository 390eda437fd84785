"""GPU: every C-ABI kernel against a float64 torch restatement of the same op (tolerance 1e-5 relative, fp32)."""
import os

import numpy as np
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


@pytest.fixture(params=["f32", "f16split"], autouse=True)
def chain_mode(request, dev, monkeypatch):
    """Every test of this file runs twice: with the chain launches on the fp32 MFMA, and with the products as split-fp16
    on the f16 MFMA pipe (weight images made on the spot, then once more after the batched refresh - the second result is
    what the test sees; the library must report that the split form really ran)."""
    if request.param == "f32":
        yield "f32"
        return
    from gfv import lib as L, ops
    orig = ops.rowtile_chain
    wmax = torch.zeros(1, device=dev)

    def split(M, segs, layers, *a, **k):
        # max |W| of this launch's weights, then the launch with images made on the spot
        wmax.copy_(torch.stack([ly.W.abs().max() for ly in layers]).max().reshape(1))
        wi = ops.WeightImages(dev, wmax)
        wi.static = [(0, 1 << 62)]
        orig(M, segs, layers, *a, wimg=wi, **k)
        assert L.load().gfv_rowtile_last_path() >= 5, "split-fp16 chain did not run"
        wi.build()    # the batched refresh writes the same images
        orig(M, segs, layers, *a, wimg=wi, **k)
        assert L.load().gfv_rowtile_last_path() >= 5

    monkeypatch.setattr(ops, "rowtile_chain", split)
    yield "f16split"


def _csr(index, n_rows):
    order = torch.argsort(index, stable=True)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(index, minlength=n_rows), 0)
    return rowptr.int(), order.int()


@pytest.mark.parametrize("Fdim", [128, 64, 16, 4, 7, 35])
def test_seg_gather_sum(dev, Fdim):
    from gfv import ops
    g = torch.Generator().manual_seed(Fdim)
    n_src, n_rows, nnz = 3000, 1111, 9000
    src = torch.randn(n_src, Fdim, generator=g)
    dst = torch.randint(0, n_rows, (nnz,), generator=g)
    dst[dst == 5] = 6  # an empty row
    col = torch.randint(0, n_src, (nnz,), generator=g)
    rowptr, order = _csr(dst, n_rows)
    colp = col[order.long()].int()
    scale = torch.rand(n_rows, generator=g) + 0.5
    sscale = torch.rand(n_src, generator=g) + 0.5
    ref = torch.zeros(n_rows, Fdim, dtype=torch.float64).index_add_(0, dst, (src.double() * sscale.double()[:, None])[col])
    ref = ref * scale.double()[:, None]
    out = ops.seg_gather_sum(src.to(dev), rowptr.to(dev), colp.to(dev), n_rows, scale=scale.to(dev),
                             src_scale=sscale.to(dev))
    assert rel(out, ref) < TOL
    base = torch.randn(n_rows, Fdim, generator=g)
    out2 = base.to(dev).clone()
    ops.seg_gather_sum(src.to(dev), rowptr.to(dev), colp.to(dev), n_rows, out=out2, accumulate=True)
    ref2 = base.double() + torch.zeros(n_rows, Fdim, dtype=torch.float64).index_add_(0, dst, src.double()[col])
    assert rel(out2, ref2) < TOL
    assert float(out2[5].sub(base[5].to(dev)).abs().max()) == 0.0


def test_gather_pair_transpose(dev):
    from gfv import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(500, 64, generator=g)
    s = torch.randint(0, 500, (1777,), generator=g).int()
    r = torch.randint(0, 500, (1777,), generator=g).int()
    base = torch.randn(1777, 128, generator=g)
    out = ops.gather_pair(a.to(dev), s.to(dev), r.to(dev), base=base.to(dev))
    ref = torch.cat((a[s.long()], a[r.long()]), 1) + base
    assert rel(out, ref) < 1e-7
    w = torch.randn(128, 384, generator=g)
    assert torch.equal(ops.transpose(w.to(dev)).cpu(), w.t().contiguous())
    w = torch.randn(3, 128, generator=g)
    assert torch.equal(ops.transpose(w.to(dev)).cpu(), w.t().contiguous())


def _mlp_params(g, kin, nout=128, scale=0.3):
    P = dict(W1=torch.randn(128, kin, generator=g) * scale / kin ** 0.5 * 4, b1=torch.randn(128, generator=g) * 0.1,
             W2=torch.randn(128, 128, generator=g) * scale / 3, b2=torch.randn(128, generator=g) * 0.1,
             W3=torch.randn(nout, 128, generator=g) * scale / 3, b3=torch.randn(nout, generator=g) * 0.1,
             gamma=1 + 0.1 * torch.randn(128, generator=g), beta=0.1 * torch.randn(128, generator=g))
    return P


def _mlp_ref(P, X, ln=True):
    P = {k: v.double() for k, v in P.items()}
    z1 = F.linear(X, P["W1"], P["b1"])
    z2 = F.linear(F.gelu(z1), P["W2"], P["b2"])
    y3 = F.linear(F.gelu(z2), P["W3"], P["b3"])
    out = F.layer_norm(y3, (128,), P["gamma"], P["beta"], 1e-5) if ln else y3
    return z1, z2, y3, out


@pytest.mark.parametrize("M", [1000, 64, 37])
def test_rowtile_edge_mlp_forward_backward(dev, M):
    """EdgeBlock-shaped chain: X = [nb[s] | nb[r] | e], 3 layers, LayerNorm, residual; then the dX chain + dW."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M)
    n_nodes = 300
    nb = torch.randn(n_nodes, 128, generator=g)
    e = torch.randn(M, 128, generator=g)
    s = torch.randint(0, n_nodes, (M,), generator=g)
    r = torch.randint(0, n_nodes, (M,), generator=g)
    P = _mlp_params(g, 384)
    X = torch.cat((nb[s], nb[r], e), 1).double().requires_grad_(True)
    z1, z2, y3, ln = _mlp_ref(P, X)
    ref_out = ln + e.double()

    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    nbd, ed, sd, rd = d(nb), d(e), d(s.int()), d(r.int())
    z1d, z2d, y3d, outd, enew = (torch.empty(M, 128, device=dev) for _ in range(5))
    ops.rowtile_chain(
        M, [ops.Seg(nbd, sd), ops.Seg(nbd, rd), ops.Seg(ed)],
        [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU, save=z1d),
         ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=z2d), ops.LayerSpec(Pd["W3"], Pd["b3"])],
        [outd], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"], fin_beta=Pd["beta"], fin_presave=y3d, res=[ed],
        out_nores=enew)
    assert rel(z1d, z1) < TOL and rel(z2d, z2) < TOL and rel(y3d, y3) < TOL
    assert rel(outd, ref_out) < TOL and rel(enew, ln) < TOL

    # backward: grad wrt the LN output = go (+ a gathered pair, as the NodeBlock adjoint feeds it)
    go = torch.randn(M, 128, generator=g)
    gagg = torch.randn(n_nodes, 64, generator=g)
    go_total = go + torch.cat((gagg[s], gagg[r]), 1)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    _, _, _, ln2 = _mlp_ref(Pg, X)
    (ln2 * go_total.double()).sum().backward()
    W1t, W2t, W3t = ops.transpose(Pd["W1"]), ops.transpose(Pd["W2"]), ops.transpose(Pd["W3"])
    g3, gz2, gz1 = (torch.empty(M, 128, device=dev) for _ in range(3))
    gnb = torch.empty(M, 256, device=dev)
    ge = torch.empty(M, 128, device=dev)
    tiles = ops.rowtile_tiles(M)
    part = torch.empty(tiles, 2, 128, device=dev)
    god = d(go)
    ops.rowtile_chain(
        M, [ops.Seg(god)],
        [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=z2d),
         ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=z1d), ops.LayerSpec(W1t)],
        [(gnb, 256), (gnb.data_ptr() + 4 * 128, 256), ge], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d,
        gadd=d(gagg), gadd_s=sd, gadd_r=rd, in_save=g3, ln_partial=part, res=[None, None, god])
    gX = X.grad
    assert rel(gnb[:, :128], gX[:, :128]) < TOL and rel(gnb[:, 128:], gX[:, 128:256]) < TOL
    assert rel(ge, gX[:, 256:] + go.double()) < TOL
    dgb = ops.reduce_partials(part, tiles, 256)
    assert rel(dgb[:128], Pg["gamma"].grad) < TOL and rel(dgb[128:], Pg["beta"].grad) < TOL
    dW3, db3 = ops.linear_dw(g3, 128, [ops.Seg(z2d)], M, a_op=1)
    dW2, db2 = ops.linear_dw(gz2, 128, [ops.Seg(z1d)], M, a_op=1)
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(nbd, sd), ops.Seg(nbd, rd), ops.Seg(ed)], M)
    for mine, name in ((dW3, "W3"), (db3, "b3"), (dW2, "W2"), (db2, "b2"), (dW1, "W1"), (db1, "b1")):
        assert rel(mine, Pg[name].grad) < TOL, name


def test_rowtile_node_mlp_and_small_widths(dev):
    """NodeBlock-shaped input [nbm(64) | x(128)], encoder widths 12 and 15 (scalar path), decoder N=3."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(7)
    M = 777
    d = lambda t: t.to(dev).contiguous()
    for widths in ([64, 128], [12], [15], [16]):
        segs = [torch.randn(M, w, generator=g) for w in widths]
        P = _mlp_params(g, sum(widths))
        X = torch.cat(segs, 1).double()
        z1, z2, y3, ln = _mlp_ref(P, X)
        Pd = {k: d(v) for k, v in P.items()}
        out = torch.empty(M, 128, device=dev)
        ops.rowtile_chain(M, [ops.Seg(d(t)) for t in segs],
                          [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU), ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU),
                           ops.LayerSpec(Pd["W3"], Pd["b3"])], [out], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"],
                          fin_beta=Pd["beta"])
        assert rel(out, ln) < TOL, widths
    # decoder: 128 -> 128 -> 128 -> 3, no LayerNorm; and its dX / dW with a 3-wide gradient
    x = torch.randn(M, 128, generator=g)
    P = _mlp_params(g, 128, nout=3)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    Xg = x.double().requires_grad_(True)
    z1, z2, y3, _ = _mlp_ref(Pg, Xg, ln=False)
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d = torch.empty(M, 128, device=dev), torch.empty(M, 128, device=dev)
    out = torch.empty(M, 3, device=dev)
    xd = d(x)
    ops.rowtile_chain(M, [ops.Seg(xd)],
                      [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU, save=z1d),
                       ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=z2d), ops.LayerSpec(Pd["W3"], Pd["b3"])],
                      [out])
    assert rel(out, y3) < TOL
    go = torch.randn(M, 3, generator=g)
    (y3 * go.double()).sum().backward()
    god = d(go)
    gz2, gz1, gx = (torch.empty(M, 128, device=dev) for _ in range(3))
    ops.rowtile_chain(M, [ops.Seg(god)],
                      [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, save=gz2, aux=z2d),
                       ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, save=gz1, aux=z1d),
                       ops.LayerSpec(ops.transpose(Pd["W1"]))], [gx])
    assert rel(gx, Xg.grad) < TOL
    dW3, db3 = ops.linear_dw(god, 3, [ops.Seg(z2d)], M, a_op=1)
    assert rel(dW3, Pg["W3"].grad) < TOL and rel(db3, Pg["b3"].grad) < TOL
    dW1, db1 = ops.linear_dw(gz1, 128, [ops.Seg(xd)], M)
    assert rel(dW1, Pg["W1"].grad) < TOL and rel(db1, Pg["b1"].grad) < TOL


def test_rowtile_node_mlp_backward_192_wide(dev):
    """NodeBlock dX chain: LN backward, two 128x128 transposed layers, last layer 192 wide written as
    [x part 128 (+ residual) | nbm part 64] (blocks.py:101-111 adjoint); rows not a multiple of the tile."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(11)
    M = 333
    d = lambda t: t.to(dev).contiguous()
    nbm, x = torch.randn(M, 64, generator=g), torch.randn(M, 128, generator=g)
    P = _mlp_params(g, 192)
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    X = torch.cat((nbm, x), 1).double().requires_grad_(True)
    z1, z2, y3, ln = _mlp_ref(Pg, X)
    go = torch.randn(M, 128, generator=g)
    (ln * go.double()).sum().backward()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.detach().float()), d(z2.detach().float()), d(y3.detach().float())
    W1t = torch.empty(192, 128, device=dev)          # rows for x first, then nbm (engine._T(perm=True))
    ops.transpose(Pd["W1"], out=W1t[0:128], col0=64, ncols=128)
    ops.transpose(Pd["W1"], out=W1t[128:192], col0=0, ncols=64)
    gx, gnbm = torch.empty(M, 128, device=dev), torch.empty(M, 64, device=dev)
    g3, gz2, gz1 = (torch.empty(M, 128, device=dev) for _ in range(3))
    tiles = ops.rowtile_tiles(M)
    part = torch.empty(tiles, 2, 128, device=dev)
    god = d(go)
    ops.rowtile_chain(M, [ops.Seg(god)],
                      [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, save=gz2, aux=z2d),
                       ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, save=gz1, aux=z1d),
                       ops.LayerSpec(W1t)], [gx, (gnbm, 64)], res=[god, None], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"],
                      in_aux=y3d, in_save=g3, ln_partial=part)
    assert rel(gx, X.grad[:, 64:] + go.double()) < TOL and rel(gnbm, X.grad[:, :64]) < TOL
    dgb = ops.reduce_partials(part, tiles, 256)
    assert rel(dgb[:128], Pg["gamma"].grad) < TOL and rel(dgb[128:], Pg["beta"].grad) < TOL


@pytest.mark.parametrize("M", [500, 64])
def test_rowtile_factored_first_layer(dev, M):
    """EdgeBlock first layer factored through the nodes: z1 = e W1c^T + (nb W1a^T)[s] + (nb W1b^T)[r] + b1 with W1c a
    column block of W1 (row stride 384) and the node products gathered in the first epilogue (gfv.h: padd, ldw);
    must equal the concat form (blocks.py:54 + EPD.py:21)."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(100 + M)
    n_nodes = 200
    nb = torch.randn(n_nodes, 128, generator=g)
    e = torch.randn(M, 128, generator=g)
    s = torch.randint(0, n_nodes, (M,), generator=g)
    r = torch.randint(0, n_nodes, (M,), generator=g)
    P = _mlp_params(g, 384)
    X = torch.cat((nb[s], nb[r], e), 1).double()
    z1, z2, y3, ln = _mlp_ref(P, X)
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    nbd, ed, sd, rd = d(nb), d(e), d(s.int()), d(r.int())
    W1 = Pd["W1"]
    pab = torch.empty(n_nodes, 256, device=dev)
    ops.rowtile_chain(n_nodes, [ops.Seg(nbd)], [ops.LayerSpec(W1[:, 0:128])], [(pab, 256)])
    ops.rowtile_chain(n_nodes, [ops.Seg(nbd)], [ops.LayerSpec(W1[:, 128:256])], [(pab.data_ptr() + 512, 256)])
    assert rel(pab[:, :128], nb.double() @ P["W1"][:, :128].double().t()) < TOL
    z1d, z2d, y3d, outd = (torch.empty(M, 128, device=dev) for _ in range(4))
    ops.rowtile_chain(M, [ops.Seg(ed)],
                      [ops.LayerSpec(W1[:, 256:384], Pd["b1"], L.OP_BIAS_GELU, save=z1d),
                       ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=z2d), ops.LayerSpec(Pd["W3"], Pd["b3"])],
                      [outd], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"], fin_beta=Pd["beta"], fin_presave=y3d, res=[ed],
                      padd=pab, padd_s=sd, padd_r=rd)
    assert rel(z1d, z1) < TOL and rel(z2d, z2) < TOL and rel(y3d, y3) < TOL and rel(outd, ln + e.double()) < TOL


def test_rowtile_transolver_linears(dev):
    """Single-layer uses: X=a+b, LayerNorm prologue with N=256, GELU prologue with K=256, LN-backward epilogue."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(11)
    M = 900
    d = lambda t: t.to(dev).contiguous()
    a, b = torch.randn(M, 128, generator=g), torch.randn(M, 128, generator=g)
    Wp = torch.randn(256, 128, generator=g) * 0.1
    bp = torch.randn(256, generator=g) * 0.1
    Wq = torch.randn(128, 256, generator=g) * 0.1
    bq = torch.randn(128, generator=g) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
    t64 = lambda t: t.double().requires_grad_(True)
    a6, b6, Wp6, bp6, Wq6, bq6, gam6, bet6 = map(t64, (a, b, Wp, bp, Wq, bq, gam, bet))
    fx = a6 + b6
    z = F.linear(F.layer_norm(fx, (128,), gam6, bet6, 1e-5), Wp6, bp6)
    out = F.linear(F.gelu(z), Wq6, bq6) + fx
    fxd = torch.empty(M, 128, device=dev)
    # fx = a + b through an identity-free path: use the chain with in_add and in_save
    zd = torch.empty(M, 256, device=dev)
    ad, bd = d(a), d(b)
    gd, btd, Wpd, bpd, Wqd, bqd = d(gam), d(bet), d(Wp), d(bp), d(Wq), d(bq)
    fxd = ad + bd
    ops.rowtile_chain(M, [ops.Seg(fxd)], [ops.LayerSpec(Wpd, bpd)], [(zd, 256), (zd.data_ptr() + 512, 256)],
                      in_op=L.IN_LN, in_gamma=gd, in_beta=btd)
    assert rel(zd, z) < TOL
    outd = torch.empty(M, 128, device=dev)
    ops.rowtile_chain(M, [ops.Seg(zd, width=128, ld=256), ops.Seg(zd, width=128, ld=256, offset=128)],
                      [ops.LayerSpec(Wqd, bqd)], [outd], in_op=L.IN_GELU, res=[fxd])
    assert rel(outd, out) < TOL
    # in_add: X = a + b feeding a plain linear
    o2 = torch.empty(M, 256, device=dev)
    ops.rowtile_chain(M, [ops.Seg(ad)], [ops.LayerSpec(Wpd, bpd)], [(o2, 256), (o2.data_ptr() + 512, 256)], in_add=bd)
    assert rel(o2, F.linear(fx, Wp6, bp6)) < TOL
    # backward of the two linears
    go = torch.randn(M, 128, generator=g)
    (out * go.double()).sum().backward()
    god = d(go)
    gz = torch.empty(M, 256, device=dev)
    ops.rowtile_chain(M, [ops.Seg(god)], [ops.LayerSpec(ops.transpose(Wqd), None, L.OP_MUL_DGELU, aux=zd)],
                      [(gz, 256), (gz.data_ptr() + 512, 256)])
    tiles = ops.rowtile_tiles(M)
    part = torch.empty(tiles, 2, 128, device=dev)
    gfx = torch.empty(M, 128, device=dev)
    ops.rowtile_chain(M, [ops.Seg(gz, width=128, ld=256), ops.Seg(gz, width=128, ld=256, offset=128)],
                      [ops.LayerSpec(ops.transpose(Wpd))], [gfx], fin_op=L.FIN_LNBWD, fin_gamma=gd, fin_aux=fxd,
                      ln_partial=part, res=[god])
    assert rel(gfx, a6.grad) < TOL
    dgb = ops.reduce_partials(part, tiles, 256)
    assert rel(dgb[:128], gam6.grad) < TOL and rel(dgb[128:], bet6.grad) < TOL
    dWq, dbq = ops.linear_dw(god, 128, [ops.Seg(zd, width=128, ld=256), ops.Seg(zd, width=128, ld=256, offset=128)], M,
                             a_op=1)
    assert rel(dWq, Wq6.grad) < TOL and rel(dbq, bq6.grad) < TOL
    dWp = torch.empty(256, 128, device=dev)
    dbp = torch.empty(256, device=dev)
    for h in range(2):
        ops.linear_dw(gz, 128, [ops.Seg(fxd)], M, a_op=2, a_gamma=gd, a_beta=btd, dW=dWp[128 * h:128 * h + 128],
                      db=dbp[128 * h:128 * h + 128], ldg=256, g_offset=128 * h)
    assert rel(dWp, Wp6.grad) < TOL and rel(dbp, bp6.grad) < TOL


def test_rowtile_stacked_layers(dev, chain_mode, gfv_limits):
    """Virtual layers stacked from two weight blocks, which exist as split-fp16 images only (the blocks' images back to
    back).  Rows: two Linear layers applied to the same input in one launch (`LayerSpec(stack=, bias2=)`: the node-level
    products W1a x, W1b x of the factored EdgeBlock; in_project_fx / in_project_x of the Transolver block) - one launch per
    block when the image is missing.  Columns: the sum of two Linear layers applied to two input segments
    (`stack_cols`: the adjoint of the in_project pair)."""
    from gfv import lib as L, ops
    gfv_limits(GFV_LIN1S=1)   # (the launch-path assertion below names the small-tile single-layer family)
    g = torch.Generator().manual_seed(5)
    M = 900
    d = lambda t: t.to(dev).contiguous()
    x, emb = torch.randn(M, 128, generator=g), torch.randn(M, 128, generator=g)
    W1 = d(torch.randn(128, 384, generator=g) * 0.1)
    b1, b2 = d(torch.randn(128, generator=g)), d(torch.randn(128, generator=g))
    out = torch.empty(M, 256, device=dev)
    xin = torch.empty(M, 128, device=dev)
    ops.rowtile_chain(M, [ops.Seg(d(x))], [ops.LayerSpec(W1[:, 0:128], b1, stack=W1[:, 128:256], bias2=b2)],
                      [(out, 256), (out.data_ptr() + 512, 256)], in_add=d(emb), in_save=xin)
    xs = (x + emb).double()
    ref = torch.cat((xs @ W1[:, 0:128].double().cpu().T + b1.double().cpu(),
                     xs @ W1[:, 128:256].double().cpu().T + b2.double().cpu()), 1)
    assert rel(out, ref) < TOL and rel(xin, xs) < TOL
    if chain_mode == "f16split":
        assert L.load().gfv_rowtile_last_path() == 5 + 32    # ONE launch in the split form (round 5: short launches run on the small-tile single-layer kernel)
        Wa, Wb = d(torch.randn(128, 128, generator=g) * 0.1), d(torch.randn(128, 128, generator=g) * 0.1)
        r = torch.randn(M, 128, generator=g)
        o2 = torch.empty(M, 128, device=dev)
        ops.rowtile_chain(M, [ops.Seg(d(x)), ops.Seg(d(emb))], [ops.LayerSpec(Wa, stack_cols=Wb)], [o2], res=[d(r)])
        ref2 = x.double() @ Wa.double().cpu().T + emb.double() @ Wb.double().cpu().T + r.double()
        assert rel(o2, ref2) < TOL
    else:
        # without an image a column-stacked layer has no weights to run on: the C entry refuses it
        with pytest.raises(RuntimeError):
            ops.rowtile_chain(M, [ops.Seg(d(x)), ops.Seg(d(emb))], [ops.LayerSpec(W1[:, 0:128], stack_cols=W1[:, 128:256])],
                              [torch.empty(M, 128, device=dev)])


def test_runtime_switch_between_product_forms(dev, chain_mode):
    """`gfv_set_f16split(0)` moves the whole process to the fp32 MFMA at run time (bench.py times both forms in one run):
    a launch that carries weight images then ignores them, the weight-gradient launch takes the fp32 kernel; both forms
    agree with each other far inside the parity tolerance."""
    if chain_mode != "f32":
        pytest.skip("one pass is enough")
    from gfv import lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(17)
    M = 1500
    d = lambda t: t.to(dev).contiguous()
    x, W, b = d(torch.randn(M, 128, generator=g)), d(torch.randn(128, 128, generator=g) * 0.1), d(torch.randn(128, generator=g))
    G = d(torch.randn(M, 128, generator=g) * 1e-3)
    wi = ops.WeightImages(dev, W.abs().max().reshape(1).clone())
    wi.static = [(0, 1 << 62)]
    outs, dws = {}, {}
    assert lib.gfv_f16split_enabled() == 1
    try:
        for form in (1, 0, 1):
            lib.gfv_set_f16split(form)
            o = torch.empty(M, 128, device=dev)
            ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W, b)], [o], wimg=wi)
            assert (lib.gfv_rowtile_last_path() >= 4) == bool(form)
            outs[form] = o
            dws[form] = ops.linear_dw(G, 128, [ops.Seg(x)], M)[0]
    finally:
        lib.gfv_set_f16split(1)
    ref = x.double().cpu() @ W.double().cpu().T + b.double().cpu()
    assert rel(outs[1], ref) < TOL and rel(outs[0], ref) < TOL and rel(outs[1], outs[0]) < 2e-6
    refW = G.double().cpu().T @ x.double().cpu()
    assert rel(dws[1], refW) < TOL and rel(dws[0], refW) < TOL and not torch.equal(dws[1], dws[0])


def test_rowtile_and_dw_extreme_dynamic_range(dev, chain_mode):
    """Rows spanning 40 orders of magnitude, all-zero rows, and segments of very different scale in one concat: the
    per-row (chain) and per-slab (dW) power-of-two scalings of the split-fp16 form must neither overflow nor lose the
    small rows; each row is judged against its own scale, with the error budget of an fp32 dot product."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(99)
    M = 1100
    d = lambda t: t.to(dev).contiguous()
    rs = 10.0 ** torch.randint(-20, 21, (M, 1), generator=g).float()
    rs[::7] = 0.0
    xa = torch.randn(M, 128, generator=g) * rs
    xb = torch.randn(M, 128, generator=g) * rs * 10.0 ** torch.randint(-6, 7, (M, 1), generator=g).float()
    W = torch.randn(128, 256, generator=g) * 0.1
    b = torch.zeros(128)
    out = torch.empty(M, 128, device=dev)
    ops.rowtile_chain(M, [ops.Seg(d(xa)), ops.Seg(d(xb))], [ops.LayerSpec(d(W), d(b))], [out])
    X = torch.cat((xa, xb), 1).double()
    ref = X @ W.double().T
    mag = X.abs() @ W.double().abs().T          # sum |x w| per output
    err = (out.double().cpu() - ref).abs()
    assert torch.isfinite(out).all()
    assert float((err / (mag + 1e-300)).max()) < 2e-6, float((err / (mag + 1e-300)).max())
    # weight gradient: slabs whose gradient rows are tiny next to slabs with large ones
    G = torch.randn(M, 128, generator=g) * 10.0 ** (torch.arange(M).float()[:, None] // 200 * 4 - 12)
    A = torch.randn(M, 128, generator=g)
    dW, db = ops.linear_dw(d(G), 128, [ops.Seg(d(A))], M)
    refW = G.double().T @ A.double()
    magW = G.double().abs().T @ A.double().abs()
    assert torch.isfinite(dW).all()
    assert float(((dW.double().cpu() - refW).abs() / magW).max()) < 2e-6
    assert rel(db, G.double().sum(0)) < TOL


def test_dw_activation_dynamic_range_per_column(dev, chain_mode):
    """the ACTIVATION side of the split-fp16 weight gradient.  With GFV_DW_COLSCALE (what the
    engine sets for the encoders' narrow raw inputs) the activations carry one power of two per column and slab: (a)
    columns whose scales span 1e-6 ... 1e+6, (b) an encoder-shaped input (K = 15:
    O(1) feature differences next to geometric columns at 1e-4 mesh-spacing scale, importer.py:54-78), (c) single columns
    that themselves span twelve decades over the rows.  Every output element is judged against its own sum |g a|."""
    if chain_mode != "f32":
        pytest.skip("one pass is enough (the weight-gradient form follows the process-wide switch, not the fixture)")
    assert __import__("gfv.lib", fromlist=["load"]).load().gfv_f16split_enabled() == 1
    from gfv import ops
    g = torch.Generator().manual_seed(7)
    d = lambda t: t.to(dev).contiguous()

    def check(G, A, tag, ld=None):
        K = A.shape[1]
        At = A if ld is None else torch.cat((A, torch.zeros(A.shape[0], ld - K)), 1)
        dW, _ = ops.linear_dw(d(G), G.shape[1], [ops.Seg(d(At), width=K, ld=At.shape[1])], G.shape[0], col_scale=True)
        ref = G.double().T @ A.double()
        mag = G.double().abs().T @ A.double().abs()
        assert torch.isfinite(dW).all(), tag
        e = float(((dW.double().cpu() - ref).abs() / (mag + 1e-300)).max())
        assert e < 2e-6, (tag, e)

    M = 2300
    G = torch.randn(M, 128, generator=g) * 1e-3
    col = 10.0 ** torch.linspace(-6, 6, 128)
    check(G, torch.randn(M, 128, generator=g) * col[None, :], "column scales 1e-6 .. 1e+6")
    feat = torch.randn(M, 12, generator=g)
    geo = torch.randn(M, 2, generator=g) * 1e-4
    ea = torch.cat((feat, geo, geo.norm(dim=1, keepdim=True)), 1)           # [M, 15] edge_attr-shaped
    check(G, ea, "encoder-shaped, K = 15", ld=16)
    rows = 10.0 ** (torch.rand(M, 1, generator=g) * 12 - 6)
    check(G, torch.randn(M, 128, generator=g) * rows, "rows spanning 1e-6 .. 1e+6 inside every column")
    check(G * rows.flip(0), torch.randn(M, 128, generator=g) * col[None, :], "both operands wide")


def test_dw_unscaled_activation_overflow_raises_flag(dev, chain_mode):
    """GELU / LayerNorm outputs (a_op 1 / 2) are split unscaled; a value beyond the fp16 range used to be clamped silently
    (csrc/dw.hip r1) - now the device status word reports it."""
    if chain_mode != "f32":
        pytest.skip("one pass is enough (the weight-gradient form follows the process-wide switch, not the fixture)")
    assert __import__("gfv.lib", fromlist=["load"]).load().gfv_f16split_enabled() == 1
    import ctypes as C
    from gfv import lib as L, ops
    lib = L.load()
    flags = C.c_int32(0)
    lib.gfv_status_flags(C.byref(flags))                                   # clear
    g = torch.Generator().manual_seed(8)
    M = 700
    d = lambda t: t.to(dev).contiguous()
    G = torch.randn(M, 128, generator=g)
    Z = torch.randn(M, 128, generator=g)
    ops.linear_dw(d(G), 128, [ops.Seg(d(Z))], M, a_op=1)
    torch.cuda.synchronize()
    lib.gfv_status_flags(C.byref(flags))
    assert flags.value == 0
    Z[5, 7] = 1.0e5                                                          # gelu(1e5) = 1e5 > 65504
    ops.linear_dw(d(G), 128, [ops.Seg(d(Z))], M, a_op=1)
    torch.cuda.synchronize()
    lib.gfv_status_flags(C.byref(flags))
    assert flags.value & 1, "GFV_FLAG_DW_RANGE"
    lib.gfv_status_flags(C.byref(flags))
    assert flags.value == 0, "reading clears the word"


def test_chain_group_scales_feed_the_weight_gradient(dev, chain_mode):
    """The dX chain leaves one power of two per 16 rows for each gradient tensor it writes (gfv_rowtile_args_t.gscale);
    the weight-gradient launch that takes them (no pass over G) gives bit-identical results to the one that finds the slab
    maximum itself, since both arrive at the same slab scale."""
    if chain_mode != "f32":
        pytest.skip("one pass is enough (this test brings its own weight images)")
    from gfv import lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(21)
    M = 3001
    d = lambda t: t.to(dev).contiguous()
    G = d(torch.randn(M, 128, generator=g) * 10.0 ** (torch.arange(M).float()[:, None] // 500 * 3 - 9))
    z2, z1 = d(torch.randn(M, 128, generator=g)), d(torch.randn(M, 128, generator=g))
    W3t, W2t = d(torch.randn(128, 128, generator=g) * 0.1), d(torch.randn(128, 128, generator=g) * 0.1)
    wi = ops.WeightImages(dev, torch.maximum(W3t.abs().max(), W2t.abs().max()).reshape(1).clone())
    wi.static = [(0, 1 << 62)]
    gz2, gz1 = torch.empty(M, 128, device=dev), torch.empty(M, 128, device=dev)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    have = ops.rowtile_chain(M, [ops.Seg(G)], [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=z2),
                                              ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, aux=z1)], [gz1], wimg=wi, gscale=gs)
    assert have and lib.gfv_rowtile_last_path() >= 5
    torch.cuda.synchronize()
    for slot, t in ((0, G), (1, gz2), (2, gz1)):
        n16 = (M + 15) // 16
        pad = torch.zeros(n16 * 16, 128, device=dev)
        pad[:M] = t
        mx = pad.view(n16, 16 * 128).abs().max(1).values
        s = gs[slot, :n16]
        ok = (mx == 0) | ((s * mx >= 2.0 ** 13) & (s * mx < 2.0 ** 14))
        assert bool(ok.all()), (slot, int((~ok).sum()))
    A = d(torch.randn(M, 128, generator=g))
    for slot, t in ((0, G), (1, gz2), (2, gz1)):
        a, _ = ops.linear_dw(t, 128, [ops.Seg(A)], M, a_op=1)
        b, _ = ops.linear_dw(t, 128, [ops.Seg(A)], M, a_op=1, gscale=gs[slot])
        assert torch.equal(a, b), slot


def test_rowtile_unsupported_shape_is_an_error_and_launches_nothing(dev, chain_mode):
    """A shape no kernel family takes (include/gfv.h, "ACCEPTED SHAPES": the generic LDS kernel that used to take the rest was
    retired with ABI 2) returns GFV_ERR_ARG and leaves the output untouched: a LayerNorm-backward epilogue with an output row
    stride that is not a multiple of 4, and a GELU' epilogue on a ragged last layer."""
    from gfv import lib as L, ops
    from gfv.ops import LayerSpec, Seg
    if chain_mode != "f32":
        pytest.skip("one pass is enough")
    M = 256
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(M, 128, generator=g).to(dev)
    W = (0.05 * torch.randn(128, 128, generator=g)).to(dev)
    y = torch.randn(M, 128, generator=g).to(dev)
    gamma = torch.ones(128, device=dev)
    out = torch.full((M, 130), 7.0, device=dev)
    part = torch.zeros(ops.rowtile_tiles(M), 2, 128, device=dev)
    with pytest.raises(RuntimeError, match="gfv_rowtile_chain"):
        ops.rowtile_chain(M, [Seg(x)], [LayerSpec(W)], [(out, 130)], fin_op=L.FIN_LNBWD, fin_gamma=gamma, fin_aux=y, ln_partial=part)
    W2 = (0.05 * torch.randn(40, 128, generator=g)).to(dev)
    z = torch.randn(M, 40, generator=g).to(dev)
    out2 = torch.full((M, 40), 7.0, device=dev)
    with pytest.raises(RuntimeError, match="gfv_rowtile_chain"):
        ops.rowtile_chain(M, [Seg(x)], [LayerSpec(W2, None, L.OP_MUL_DGELU, aux=z)], [out2])
    torch.cuda.synchronize()
    assert bool((out == 7.0).all()) and bool((out2 == 7.0).all()) and bool((part == 0).all())


def test_weight_absmax_over_blocks_and_column_blocks(dev):
    """gfv_weight_absmax: max |W| over a set of weight blocks (whole matrices and a column block of a wider one) - a 4-byte fill +
    integer atomic maxima on the bit pattern: equal to torch, call after call."""
    from gfv import lib as L, ops
    lib = L.load()
    g = torch.Generator(device="cpu").manual_seed(21)
    mats = [torch.randn(128, 128, generator=g).to(dev) * 0.1, torch.randn(128, 384, generator=g).to(dev) * 0.2,
            torch.randn(3, 128, generator=g).to(dev), torch.randn(128, 15, generator=g).to(dev) * 0.5]
    blocks = [(m.data_ptr(), m.stride(0), m.shape[0], m.shape[1]) for m in mats[:1] + mats[2:]]
    blocks.append((mats[1][:, 128:256].data_ptr(), 384, 128, 128))        # a column block of the wide matrix
    desc, nd = ops.WeightImages._upload([(b, torch.empty(lib.gfv_weight_image_bytes(b[2], b[3]), dtype=torch.uint8, device=dev))
                                         for b in blocks], dev)
    for rep in range(3):
        want = max(float(mats[0].abs().max()), float(mats[2].abs().max()), float(mats[3].abs().max()), float(mats[1][:, 128:256].abs().max()))
        ref = torch.full((1,), float("nan"), device=dev)
        L.check(lib.gfv_weight_absmax(desc.data_ptr(), nd, ref.data_ptr(), L.stream_ptr()), "absmax")
        assert want == float(ref), (rep, want, float(ref))
        mats[3].mul_(3.0)                                                  # the next call sees other values
