"""A few launches of the fused EdgeBlock MLP in one kernel family (argv[1]: row | col | collite; argv[2]: rows) - the
program rocprofv3 --pmc profiles (profiles/tools/chain_pmc.sh)."""
import os, sys
fam = sys.argv[1]
os.environ["GFV_COLCHAIN_LITE"] = "1" if fam == "collite" else "0"
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gen-fvgn-steady_amd"))
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec
dev = 'cuda'
wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
wi.static = [(0, 1 << 62)]
g = torch.Generator(device='cpu').manual_seed(0)
M = int(sys.argv[2]) if len(sys.argv) > 2 else 603992
e = torch.randn(M, 128, device=dev)
pab = torch.randn(M // 3 + 1, 256, device=dev)
s = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
r = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
z1, z2, y3, out, nores = (torch.empty(M, 128, device=dev) for _ in range(5))
W = [torch.randn(128, 128, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05,
     torch.zeros(128, device=dev), torch.randn(128, 128, generator=g).to(dev) * 0.05, torch.zeros(128, device=dev),
     torch.ones(128, device=dev), torch.zeros(128, device=dev)]
layers = [LayerSpec(W[0], W[1], L.OP_BIAS_GELU, save=z1), LayerSpec(W[2], W[3], L.OP_BIAS_GELU, save=z2), LayerSpec(W[4], W[5])]
family = L.CHAIN_ROW_OWNER if fam == "row" else L.CHAIN_COLUMN_OWNER
for _ in range(4):
    ops.rowtile_chain(M, [Seg(e)], layers, [out], fin_op=L.FIN_LN, fin_gamma=W[6], fin_beta=W[7], fin_presave=y3, res=[e],
                      out_nores=nores, padd=pab, padd_s=s, padd_r=r, wimg=wi, family=family)
torch.cuda.synchronize()
