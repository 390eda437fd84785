"""Fused EdgeBlock MLP launch (M rows) in both families with output streams removed one by one: where the time goes."""
import sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec

dev = 'cuda'
wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
wi.static = [(0, 1 << 62)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator(device='cpu').manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 603992
e = torch.randn(M, 128, device=dev)
pab = torch.randn(M // 3 + 1, 256, device=dev)
s = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
r = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
z1, z2, y3, out, nores = (torch.empty(M, 128, device=dev) for _ in range(5))
W = [torch.randn(128, 128, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05,
     torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev),
     torch.ones(128, device=dev), torch.zeros(128, device=dev)]
for fam, name in ((L.CHAIN_ROW_OWNER, 'row-owner'), (L.CHAIN_COLUMN_OWNER, 'column-owner')):
    for label, sv1, sv2, sv3, nr, padd, res in (("all streams", z1, z2, y3, nores, True, True), ("no z1", None, z2, y3, nores, True, True),
                                               ("no z1 z2", None, None, y3, nores, True, True), ("no z1 z2 y3", None, None, None, nores, True, True),
                                               ("out only", None, None, None, None, True, True), ("out only, no padd", None, None, None, None, False, True),
                                               ("out only, no padd, no res", None, None, None, None, False, False)):
        layers = [LayerSpec(W[0], W[1], L.OP_BIAS_GELU, save=sv1), LayerSpec(W[2], W[3], L.OP_BIAS_GELU, save=sv2), LayerSpec(W[4], W[5])]
        kw = dict(padd=pab, padd_s=s, padd_r=r) if padd else {}
        fn = lambda: ops.rowtile_chain(M, [Seg(e)], layers, [out], fin_op=L.FIN_LN, fin_gamma=W[6], fin_beta=W[7], fin_presave=sv3,
                                       res=[e] if res else None, out_nores=nr, wimg=wi, family=fam, **kw)
        t = timeit(fn)
        print(f"M={M} {name:13s} {label:28s}: {t:7.1f} us", flush=True)
