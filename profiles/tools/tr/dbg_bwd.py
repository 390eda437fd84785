import sys
sys.path.insert(0, 'gen-fvgn-steady_amd'); sys.path.insert(0, 'tests')
import torch, torch.nn.functional as F
from gfv import lib as L, ops
from test_colchain_gpu import _params, _ref, _images, rel
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
M = 256
e = torch.randn(M, 128, generator=g)
P = _params(g, 128)
Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
X = e.double().requires_grad_(True)
z1, z2, y3, ln = _ref(Pg, X)
go = torch.randn(M, 128, generator=g)
(ln * go.double()).sum().backward(retain_graph=True)
g3_ref = torch.autograd.grad((ln * go.double()).sum(), y3, retain_graph=True)[0]
gz2_ref = torch.autograd.grad((ln * go.double()).sum(), z2, retain_graph=True)[0]
gz1_ref = torch.autograd.grad((ln * go.double()).sum(), z1, retain_graph=True)[0]
d = lambda t: t.to(dev).contiguous()
Pd = {k: d(v) for k, v in P.items()}
z1d, z2d, y3d = d(z1.detach().float()), d(z2.detach().float()), d(y3.detach().float())
stats = d(torch.stack((y3.detach().mean(1), (y3.detach().var(1, unbiased=False) + 1e-5).rsqrt()), 1).float())
W3t, W2t, W1t = ops.transpose(Pd["W3"]), ops.transpose(Pd["W2"]), ops.transpose(Pd["W1"])
wi = _images(dev, [P["W1"], P["W2"], P["W3"]])
god = d(go)
g3, gz2, gz1, ge = (torch.full((M, 128), float("nan"), device=dev) for _ in range(4))
part = torch.zeros(L.load().gfv_rowtile_dw_partials(), L.DW_FUSED_FLOATS, device=dev)
layers = [ops.LayerSpec(W3t, None, L.OP_MUL_DGELU, save=gz2, aux=z2d), ops.LayerSpec(W2t, None, L.OP_MUL_DGELU, save=gz1, aux=z1d), ops.LayerSpec(W1t)]
ops.rowtile_chain(M, [ops.Seg(god)], layers, [ge], in_op=L.IN_LNBWD, in_gamma=Pd["gamma"], in_aux=y3d, in_stats=stats, in_save=g3,
                  res=[god], dw_partial=part, wimg=wi, family=L.CHAIN_COLUMN_OWNER)
print("path", L.load().gfv_rowtile_last_path())
print("g3", rel(g3, g3_ref), "gz2", rel(gz2, gz2_ref), "gz1", rel(gz1, gz1_ref), "ge", rel(ge, X.grad + go.double()))
print("g3 nan rows", int(torch.isnan(g3).any(1).sum()), "ge nan rows", int(torch.isnan(ge).any(1).sum()))
print(g3[:2, :6].cpu(), g3_ref[:2, :6])
tot = part.double().sum(0).cpu()
dgam, dbet = tot[2 * 16384 + 256:2 * 16384 + 384], tot[2 * 16384 + 384:2 * 16384 + 512]
print("dgamma", rel(dgam, Pg["gamma"].grad), "dbeta", rel(dbet, Pg["beta"].grad))
err = (g3.double().cpu() - g3_ref).abs()
print("g3 err by 16-column block:", [round(float(err[:, 16 * w:16 * w + 16].max()), 4) for w in range(8)])
print("g3 err by 16-row group  :", [round(float(err[16 * q:16 * q + 16].max()), 4) for q in range(16)])
print("dbeta mine/ref", dbet[:4], Pg["beta"].grad[:4])
