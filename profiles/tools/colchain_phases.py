"""Per-phase cycle counts of the column-owner forward kernel (a -DGFV_CC_TIMING build: GFV_LIB=.../libgfv_cctime.so)."""
import os, sys
os.environ.setdefault("GFV_COLCHAIN_LITE", "0")
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec
dev = 'cuda'
wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
wi.static = [(0, 1 << 62)]
g = torch.Generator(device='cpu').manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 603992
variant = sys.argv[2] if len(sys.argv) > 2 else "all"
e = torch.randn(M, 128, device=dev)
pab = torch.randn(M // 3 + 1, 256, device=dev)
s = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
r = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
z1, z2, y3, out, nores = (torch.empty(M, 128, device=dev) for _ in range(5))
W = [torch.randn(128, 128, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05,
     torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev),
     torch.ones(128, device=dev), torch.zeros(128, device=dev)]
full = variant == "all"
layers = [LayerSpec(W[0], W[1], L.OP_BIAS_GELU, save=z1 if full else None), LayerSpec(W[2], W[3], L.OP_BIAS_GELU, save=z2 if full else None),
          LayerSpec(W[4], W[5])]
dbg = torch.zeros(512 * 8 * 12, dtype=torch.int64, device=dev)
kw = dict(padd=pab, padd_s=s, padd_r=r) if full else {}
for _ in range(3):
    ops.rowtile_chain(M, [Seg(e)], layers, [out], fin_op=L.FIN_LN, fin_gamma=W[6], fin_beta=W[7], fin_presave=y3 if full else None,
                      res=[e] if full else None, out_nores=nores if full else None, wimg=wi, family=L.CHAIN_COLUMN_OWNER,
                      in_aux=dbg.view(torch.float32), **kw)
torch.cuda.synchronize()
d = dbg.view(-1, 8, 12)[:256].double()
tiles = ((M + 15) // 16) / 256 / 8
names = ["P4+P0 work", "bar0 wait", "P1 work", "bar1 wait", "P2 work", "bar2 wait", "P3 work", "bar3 wait", "last P4"]
print(f"M={M} variant={variant}: ~{tiles:.1f} tiles per workgroup; cycles per tile (mean over workgroups; min / max over the 8 waves of the per-wave means)")
tot = 0
for k, n in enumerate(names):
    per = d[:, :, k] / tiles
    tot += per.mean().item()
    print(f"  {n:12s} {per.mean().item():9.0f}   wave means {per.mean(0).min().item():9.0f} .. {per.mean(0).max().item():9.0f}")
print(f"  total        {tot:9.0f}")
