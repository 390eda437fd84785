"""The single-layer launches at node level (25 251 rows = 198 blocks of 128 rows on 256 CUs: one block's latency is the launch's
time) in isolation: microseconds per launch, back-to-back.   python profiles/tools/lin1_node_bench.py [M]"""
import sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25251
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
x, x2, res, z = r(M, 128), r(M, 128), r(M, 128), r(M, 128)
W = r(128, 128) * 0.1; Wb = r(128, 128) * 0.1; W256 = r(128, 256) * 0.1; b = r(128); b2 = r(128)
gam, bet = r(128), r(128)
wi = ops.WeightImages(dev, torch.full((1,), 0.5, device=dev)); wi.static = [(0, 1 << 62)]
o, o2 = torch.empty(M, 128, device=dev), torch.empty(M, 256, device=dev)
stats_part = torch.empty((M + 63) // 64, 256, device=dev)
y = r(M, 128)
cases = {
    "128 -> 128 + bias + residual": lambda: ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W, b)], [o], res=[res], wimg=wi),
    "128 -> 256 (two stacked blocks)": lambda: ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W, b, stack=Wb, bias2=b2)], [(o2, 256), (o2.data_ptr() + 512, 256)], wimg=wi),
    "LayerNorm -> 128 -> 256": lambda: ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W, b, stack=Wb, bias2=b2)], [(o2, 256), (o2.data_ptr() + 512, 256)], in_op=L.IN_LN, in_gamma=gam, in_beta=bet, wimg=wi),
    "GELU(256) -> 128": lambda: ops.rowtile_chain(M, [ops.Seg(x), ops.Seg(x2)], [ops.LayerSpec(W256, b)], [o], in_op=L.IN_GELU, wimg=wi),
    "128 -> 256 x GELU'(z)": lambda: ops.rowtile_chain(M, [ops.Seg(x)], [ops.LayerSpec(W, None, L.OP_MUL_DGELU, aux=o2, stack=Wb)], [(o2, 256), (o2.data_ptr() + 512, 256)], wimg=wi) if False else None,
    "256 -> 128 -> LayerNorm backward (+ residual)": lambda: ops.rowtile_chain(M, [ops.Seg(x), ops.Seg(x2)], [ops.LayerSpec(W256)], [o], res=[res], fin_op=L.FIN_LNBWD, fin_aux=y, fin_gamma=gam, ln_partial=stats_part, wimg=wi),
}
for name, fn in cases.items():
    if fn() is None and "GELU'" in name:
        continue
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    path = L.load().gfv_rowtile_last_path()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"M={M} {name:48s} {e0.elapsed_time(e1) * 10:6.1f} us  (path {path})")
