"""Per-phase cycle counts of the column-owner backward kernel (a -DGFV_CC_TIMING build: GFV_LIB=.../libgfv_cctime.so).
    python profiles/tools/colchain_bwd_phases.py [M] [rc]      rc: the recompute form (z2, LayerNorm input rebuilt from z1)"""
import os, sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec
dev = 'cuda'
wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
wi.static = [(0, 1 << 62)]
ops.set_weight_images(wi)
g = torch.Generator(device='cpu').manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 603992
e, G, y3, z1, z2 = (torch.randn(M, 128, device=dev) for _ in range(5))
if "alias" in sys.argv:      # every input array is the same memory: one read stream instead of five (is the launch memory bound?)
    G = y3 = z1 = z2 = e
if "small" in sys.argv:      # ... and only 4 MB of it (L2 / Infinity-Cache resident): no HBM read traffic at all
    pass
stats = torch.stack((y3.mean(1), (y3.var(1, unbiased=False) + 1e-5).rsqrt()), 1).contiguous()
gagg = torch.randn(M // 3 + 1, 64, device=dev)
s = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
r = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
Wt = [torch.randn(128, 128, generator=g).to(dev) * 0.05 for _ in range(3)]
gam = torch.ones(128, device=dev)
gz1, ge = (torch.empty(M, 128, device=dev) for _ in range(2))
nwg = L.load().gfv_rowtile_dw_partials()
dwp = torch.empty(nwg, L.DW_FUSED_FLOATS, device=dev)
dbg = torch.zeros(512 * 8 * 16, dtype=torch.int64, device=dev)
rc = "rc" in sys.argv
Wf = [torch.randn(128, 128, generator=g).to(dev) * 0.05 for _ in range(2)]
bf = [torch.randn(128, generator=g).to(dev) * 0.05 for _ in range(2)]
kw = dict(rc=(Wf[0], bf[0], Wf[1], bf[1])) if rc else {}
for _ in range(3):
    ops.rowtile_chain(M, [Seg(G)], [LayerSpec(Wt[0], None, L.OP_MUL_DGELU, aux=None if rc else z2),
                                    LayerSpec(Wt[1], None, L.OP_MUL_DGELU, save=gz1, aux=z1),
                                    LayerSpec(Wt[2])], [ge], res=[G], in_op=L.IN_LNBWD, in_gamma=gam, in_aux=None if rc else y3,
                      in_stats=stats, gadd=gagg, gadd_s=s, gadd_r=r, dw_partial=dwp, family=L.CHAIN_COLUMN_OWNER,
                      fin_aux=dbg.view(torch.float32), **kw)
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
d = dbg.view(-1, 8, 16)[:256].double()
tiles = ((M + 15) // 16) / 256 / 4
names = ["(R3 +) P0", "barA wait", "P0b", "barB wait", "P3", "barC wait", "dW3 (+ dW2)", "barD wait", "P2", "barE wait", "P1", "dW2 | prefetch",
         "R1", "barR1 wait", "R2", "barR2 wait"]
print(f"M={M}: ~{tiles:.1f} tiles of 64 rows per workgroup; cycles per tile (mean over workgroups; min / max of the per-wave means)")
tot = 0
for k, n in enumerate(names):
    per = d[:, :, k] / tiles
    tot += per.mean().item()
    print(f"  {n:14s} {per.mean().item():9.0f}   wave means {per.mean(0).min().item():9.0f} .. {per.mean(0).max().item():9.0f}")
print(f"  total          {tot:9.0f}")
