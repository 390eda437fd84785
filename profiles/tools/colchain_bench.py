"""One fused-MLP launch in both kernel families of the split-fp16 form, EdgeBlock and NodeBlock shapes of the 50 k-cell bench
mesh (M = 75 499 / 25 479): microseconds per launch, HIP events around 30 launches on an otherwise idle GPU."""
import sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec

dev = 'cuda'
wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
wi.static = [(0, 1 << 62)]


def timeit(fn, n=30):
    for _ in range(5):
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
N = 25479
for M, kind in ((75499, 'edge'), (N, 'node'), (8 * 75499, 'edge'), (8 * N, 'node')):
    e = torch.randn(M, 128, device=dev)
    pab = torch.randn(M // 3 + 1, 256, device=dev)
    s = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
    r = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
    nbm = torch.randn(M, 64, device=dev)
    z1, z2, y3, out, nores = (torch.empty(M, 128, device=dev) for _ in range(5))
    K = 128 if kind == 'edge' else 192
    W = [torch.randn(128, K, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05,
         torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev),
         torch.ones(128, device=dev), torch.zeros(128, device=dev)]
    layers = [LayerSpec(W[0], W[1], L.OP_BIAS_GELU, save=z1), LayerSpec(W[2], W[3], L.OP_BIAS_GELU, save=z2), LayerSpec(W[4], W[5])]
    for fam, name in ((L.CHAIN_ROW_OWNER, 'row-owner'), (L.CHAIN_COLUMN_OWNER, 'column-owner')):
        if kind == 'edge':
            fn = lambda: ops.rowtile_chain(M, [Seg(e)], layers, [out], fin_op=L.FIN_LN, fin_gamma=W[6], fin_beta=W[7], fin_presave=y3,
                                           res=[e], out_nores=nores, padd=pab, padd_s=s, padd_r=r, wimg=wi, family=fam)
            by = M * (512 * 6 + 8 + 2 * 512) + 192 * 1024
        else:
            fn = lambda: ops.rowtile_chain(M, [Seg(nbm), Seg(e)], layers, [out], fin_op=L.FIN_LN, fin_gamma=W[6], fin_beta=W[7],
                                           fin_presave=y3, res=[e], wimg=wi, family=fam)
            by = M * (768 + 512 * 4 + 512) + 224 * 1024
        t = timeit(fn)
        print(f"{kind} M={M} {name}: {t:.1f} us  {by / t / 1e6:.2f} TB/s algorithmic (path {L.load().gfv_rowtile_last_path()})", flush=True)
