"""GPU: the bf16 single-product form (gfv_set_f16split(3): BASELINE config 3's "bf16 MLP GEMMs on MFMA" to the letter -
v_mfma_f32_16x16x32_bf16 on bf16-rounded operands, fp32 accumulation, fp32 everywhere else).

What the form IS is pinned kernel family by kernel family: every GEMM of a launch equals the float64 product of its
bf16-ROUNDED operands (round to nearest even, as torch's .to(bfloat16)), to fp32 accuracy where the operands are given (one
Linear, the weight gradient) and to 2e-3 of scale where a launch rounds values it computed itself (a last-bit difference of an
fp32 intermediate can move its bf16 rounding by one unit = 2^-8 of that value).  How far the form is from the fp32 model is
the stated tolerance cases.BF16_TOL, asserted on the whole model against the fp32 oracle in tests/test_model_gpu.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_colchain_gpu import _images, _params, _ref, rel

pytestmark = pytest.mark.gpu

TIGHT = 2e-6      # operands given: fp32 accumulation of exactly representable products
CHAINED = 2e-3    # operands computed (and rounded) inside the launch


def bfr(t):
    return t.detach().float().to(torch.bfloat16).double().cpu()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    from gfv import lib
    lib.load()
    return torch.device("cuda:0")


@pytest.fixture()
def bf16_form():
    from gfv import lib as L
    lib = L.load()
    assert lib.gfv_f16split_enabled() == 1
    lib.gfv_set_f16split(3)
    yield lib
    lib.gfv_set_f16split(1)


def dgelu(z):
    return 0.5 * (1 + torch.erf(z / math.sqrt(2))) + z * torch.exp(-0.5 * z * z) / math.sqrt(2 * math.pi)


@pytest.mark.parametrize("M", [3000, 1100])
def test_one_linear_is_the_product_of_the_bf16_rounded_operands(dev, bf16_form, M):
    """The lean single-layer kernel and the row-owner chain kernel, 128 -> 128 with bias and residual and 128 -> 256 as two
    stacked blocks: the float64 product of the rounded operands to fp32 accuracy - and measurably not the fp32 product."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M)
    d = lambda t: t.to(dev).contiguous()
    x = torch.randn(M, 128, generator=g) * torch.logspace(-4, 2, M)[:, None]
    W = torch.randn(256, 128, generator=g) * 0.1
    b, r = torch.randn(256, generator=g), torch.randn(M, 128, generator=g)
    xd, Wd, bd, rd = d(x), d(W), d(b), d(r)
    wi = _images(dev, [W])
    want = bfr(x) @ bfr(W).T + b.double()
    exact = x.double() @ W.double().T + b.double()
    for fam, path in ((0, 5 + 32), (L.CHAIN_ROW_OWNER, 5)):
        o = torch.full((M, 128), float("nan"), device=dev)
        ops.rowtile_chain(M, [ops.Seg(xd)], [ops.LayerSpec(Wd[0:128], bd[0:128])], [o], res=[rd], wimg=wi, family=fam)
        assert bf16_form.gfv_rowtile_last_path() == path
        assert rel(o, want[:, 0:128] + r.double()) < TIGHT, (fam, rel(o, want[:, 0:128] + r.double()))
        assert rel(o, exact[:, 0:128] + r.double()) > 1e-4, "the bf16 switch did not reach the kernel"
        o2 = torch.full((M, 256), float("nan"), device=dev)
        ops.rowtile_chain(M, [ops.Seg(xd)], [ops.LayerSpec(Wd[0:128], bd[0:128], stack=Wd[128:256], bias2=bd[128:256])],
                          [(o2, 256), (o2.data_ptr() + 512, 256)], wimg=wi, family=fam)
        assert rel(o2, want) < TIGHT, (fam, rel(o2, want))


def test_images_follow_the_form_they_are_used_in(dev):
    """A WeightImages set built in the default form is rebuilt when the next launch runs in the bf16 form, and back (the
    images hold bf16 high parts in one and fp16 hi + lo parts in the other: include/gfv.h)."""
    from gfv import lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    M = 1500
    d = lambda t: t.to(dev).contiguous()
    x, W = torch.randn(M, 128, generator=g), torch.randn(128, 128, generator=g) * 0.1
    xd, Wd = d(x), d(W)
    wi = _images(dev, [W])
    exact, rounded = x.double() @ W.double().T, bfr(x) @ bfr(W).T
    try:
        for form, want, tol in ((1, exact, 1e-5), (3, rounded, TIGHT), (1, exact, 1e-5), (2, None, None), (3, rounded, TIGHT)):
            lib.gfv_set_f16split(form)
            o = torch.empty(M, 128, device=dev)
            ops.rowtile_chain(M, [ops.Seg(xd)], [ops.LayerSpec(Wd)], [o], wimg=wi)
            if want is not None:
                assert rel(o, want) < tol, (form, rel(o, want))
    finally:
        lib.gfv_set_f16split(1)


def test_library_refuses_images_of_the_other_form_class(dev):
    """The C-ABI side of the same rule (include/gfv.h, gfv_weight_images_form): the library remembers per `wmax` scalar which class
    of parts its images were built with, and a launch in the other class of product form is GFV_ERR_ARG, not a product of
    garbage - here the Python-side bookkeeping is told the stale fp16 images were bf16 ones."""
    from gfv import lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    M = 600
    x, W = torch.randn(M, 128, generator=g).to(dev), (torch.randn(128, 128, generator=g) * 0.1).to(dev)
    wi = _images(dev, [W.cpu()])
    o = torch.full((M, 128), 7.0, device=dev)
    try:
        ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W)], [o], wimg=wi)          # builds the fp16 (hi, lo) images
        assert lib.gfv_weight_images_form(wi.wmax.data_ptr()) == 0
        o.fill_(7.0)
        lib.gfv_set_f16split(3)
        wi._bf = True                                                                   # (suppresses the rebuild)
        with pytest.raises(RuntimeError, match="gfv_rowtile_chain"):
            ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W)], [o], wimg=wi)
        torch.cuda.synchronize()
        assert bool((o == 7.0).all())
        wi._bf = None                                                                   # the honest path: rebuilt as bf16 images
        ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W)], [o], wimg=wi)
        assert lib.gfv_weight_images_form(wi.wmax.data_ptr()) == 1
        assert rel(o, bfr(x.cpu()) @ bfr(W.cpu()).T) < TIGHT
    finally:
        lib.gfv_set_f16split(1)
    assert lib.gfv_weight_images_form(wi.wmax.data_ptr() + 4) == -1      # (an address no set of images was scaled with)


@pytest.mark.parametrize("M", [5000, 700])
def test_weight_gradient_is_the_product_of_the_bf16_rounded_operands(dev, bf16_form, M):
    from gfv import ops
    g = torch.Generator().manual_seed(M + 3)
    d = lambda t: t.to(dev).contiguous()
    G = torch.randn(M, 128, generator=g) * torch.logspace(-3, 0, M)[:, None]
    xa, xb = torch.randn(M, 128, generator=g), torch.randn(M, 64, generator=g)
    dW, db = ops.linear_dw(d(G), 128, [ops.Seg(d(xa)), ops.Seg(d(xb))], M)
    want = bfr(G).T @ torch.cat((bfr(xa), bfr(xb)), 1)
    exact = G.double().T @ torch.cat((xa, xb), 1).double()
    assert rel(dW, want) < TIGHT, rel(dW, want)
    assert rel(dW, exact) > 1e-4
    assert rel(db, G.double().sum(0)) < 1e-5          # (the bias gradient is an fp32 column sum in this kernel)


@pytest.mark.parametrize("M", [3000, 333])
def test_forward_chain_rounds_every_layer_input_to_bf16(dev, bf16_form, M):
    """Three Linear layers with GELU between them and LayerNorm behind (EPD.py:10-33): each layer = the product of its
    bf16-rounded input rows and bf16-rounded weights; the saved pre-activations and the output against that model."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + 11)
    d = lambda t: t.to(dev).contiguous()
    x = torch.randn(M, 128, generator=g)
    P = _params(g, 128)
    Pd = {k: d(v) for k, v in P.items()}
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    z1 = bfr(x) @ bfr(P["W1"]).T + P["b1"].double()
    z2 = bfr(F.gelu(z1)) @ bfr(P["W2"]).T + P["b2"].double()
    y3 = bfr(F.gelu(z2)) @ bfr(P["W3"]).T + P["b3"].double()
    out = F.layer_norm(y3, (128,), P["gamma"].double(), P["beta"].double(), 1e-5)
    z1d, z2d, o = (torch.full((M, 128), float("nan"), device=dev) for _ in range(3))
    ops.rowtile_chain(M, [ops.Seg(d(x))], [ops.LayerSpec(Pd["W1"], Pd["b1"], L.OP_BIAS_GELU, save=z1d),
                                            ops.LayerSpec(Pd["W2"], Pd["b2"], L.OP_BIAS_GELU, save=z2d),
                                            ops.LayerSpec(Pd["W3"], Pd["b3"])], [o], fin_op=L.FIN_LN, fin_gamma=Pd["gamma"],
                      fin_beta=Pd["beta"], wimg=wi)
    assert bf16_form.gfv_rowtile_last_path() & 4
    assert rel(z1d, z1) < TIGHT and rel(z2d, z2) < CHAINED and rel(o, out) < CHAINED, (rel(z1d, z1), rel(z2d, z2), rel(o, out))
    exact = _ref({k: v for k, v in P.items()}, x.double())[3]
    assert 1e-4 < rel(o, exact) < 5e-2, rel(o, exact)


@pytest.mark.parametrize("M", [4000, 97])
@pytest.mark.parametrize("rc", [False, True])
def test_column_owner_backward_in_the_bf16_form(dev, bf16_form, M, rc):
    """The persistent backward with fused weight gradients (csrc/colchain_kernel.h): input gradient, gz1, dW3, dW2, their bias
    gradients and the LayerNorm's against a float64 model that rounds every GEMM operand to bf16 where the kernel does (the
    gradient rows g3 / gz2 / gz1, the transposed weights, the activations a2 / a1 of the weight gradients; the bias gradients
    are column sums of the ROUNDED rows: they come off the matrix pipe).  Read form and recompute form."""
    from gfv import lib as L, ops
    g = torch.Generator().manual_seed(M + 1)
    e = torch.randn(M, 128, generator=g)
    P = _params(g, 128)
    Pq = {k: v.double() for k, v in P.items()}
    go = torch.randn(M, 128, generator=g) * torch.logspace(-5, 0, M)[:, None]
    if rc:      # what the forward of this form would have saved: z1 exact here, the rest rebuilt from it with rounded operands
        z1 = F.linear(e.double(), Pq["W1"], Pq["b1"])
        z2 = bfr(F.gelu(z1)) @ bfr(P["W2"]).T + Pq["b2"]
        y3 = bfr(F.gelu(z2)) @ bfr(P["W3"]).T + Pq["b3"]
    else:
        z1, z2, y3, _ = _ref(Pq, e.double())
    mean, rstd = y3.mean(1, keepdim=True), (y3.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    xhat = (y3 - mean) * rstd
    gg = go.double() * Pq["gamma"]
    g3 = rstd * (gg - gg.mean(1, keepdim=True) - xhat * (gg * xhat).mean(1, keepdim=True))
    gz2 = (bfr(g3) @ bfr(P["W3"])) * dgelu(z2)
    gz1 = (bfr(gz2) @ bfr(P["W2"])) * dgelu(z1)
    ge = bfr(gz1) @ bfr(P["W1"]) + go.double()
    want = dict(W3=bfr(g3).T @ bfr(F.gelu(z2)), b3=bfr(g3).sum(0), W2=bfr(gz2).T @ bfr(F.gelu(z1)), b2=bfr(gz2).sum(0),
                gamma=(go.double() * xhat).sum(0), beta=go.double().sum(0))
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    stats = d(torch.cat((mean, rstd), 1).float())
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    god = d(go)
    gz1d, ged = (torch.full((M, 128), float("nan"), device=dev) for _ in range(2))
    part = torch.full((L.load().gfv_rowtile_dw_partials_m(M), L.DW_FUSED_FLOATS), float("nan"), device=dev)
    kw = dict(rc=(Pd["W2"], Pd["b2"], Pd["W3"], Pd["b3"])) if rc else {}
    layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, aux=None if rc else d(z2.float())),
              ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, save=gz1d, aux=d(z1.float())),
              ops.LayerSpec(ops.transpose(Pd["W1"]))]
    ops.rowtile_chain(M, [ops.Seg(god)], layers, [ged], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=None if rc else d(y3.float()),
                      in_stats=stats, res=[god], dw_partial=part, wimg=wi, family=L.CHAIN_COLUMN_OWNER, **kw)
    assert bf16_form.gfv_rowtile_last_path() == 5 + 16
    assert rel(ged, ge) < CHAINED and rel(gz1d, gz1) < CHAINED, (rel(ged, ge), rel(gz1d, gz1))
    tot = part.double().sum(0).cpu()
    got = dict(W3=tot[:16384].view(128, 128), b3=tot[16384:16512], W2=tot[16512:16512 + 16384].view(128, 128),
               b2=tot[16512 + 16384:16512 + 16384 + 128], gamma=tot[2 * 16384 + 256:2 * 16384 + 384],
               beta=tot[2 * 16384 + 384:2 * 16384 + 512])
    for name in want:
        assert rel(got[name], want[name]) < (1e-5 if name in ("gamma", "beta") else CHAINED), (name, rel(got[name], want[name]))
    flags = L.C.c_int32(0)
    L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
    assert flags.value == 0
