import sys
sys.path.insert(0, 'gen-fvgn-steady_amd'); sys.path.insert(0, 'tests')
import torch
from gfv import lib as L, ops
from test_colchain_gpu import _params, _ref, _images
dev = torch.device('cuda:0')
for M, full in ((25251, False), (75499, False), (25251, True), (75499, True)):
    g = torch.Generator().manual_seed(1)
    P = _params(g, 128 if full else 16)
    x = torch.randn(M, 128 if full else 16, generator=g)
    z1, z2, y3, ln = _ref({k: v.double() for k, v in P.items()}, x.double())
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    z1d, z2d, y3d = d(z1.float()), d(z2.float()), d(y3.float())
    stats = d(torch.stack((y3.mean(1), (y3.var(1, unbiased=False) + 1e-5).rsqrt()), 1).float())
    wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
    go = d(torch.randn(M, 128, generator=g))
    gz1 = torch.empty(M, 128, device=dev); ge = torch.empty(M, 128, device=dev)
    part = torch.empty(L.load().gfv_rowtile_dw_partials(), L.DW_FUSED_FLOATS, device=dev)
    gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
    if full:
        layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, aux=z2d), ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, save=gz1, aux=z1d), ops.LayerSpec(ops.transpose(Pd["W1"]))]
        kw = dict(res=[go], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, dw_partial=part, wimg=wi, family=L.CHAIN_COLUMN_OWNER)
        outs = [ge]
    else:
        layers = [ops.LayerSpec(ops.transpose(Pd["W3"]), None, L.OP_MUL_DGELU, aux=z2d), ops.LayerSpec(ops.transpose(Pd["W2"]), None, L.OP_MUL_DGELU, aux=z1d)]
        kw = dict(in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, dw_partial=part, gscale=gs, wimg=wi, family=L.CHAIN_COLUMN_OWNER)
        outs = [gz1]
    for _ in range(3):
        ops.rowtile_chain(M, [ops.Seg(go)], layers, outs, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.rowtile_chain(M, [ops.Seg(go)], layers, outs, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} {'3-layer with dX' if full else '2-layer NOOUT (encoder)'}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us  path {L.load().gfv_rowtile_last_path()}")
