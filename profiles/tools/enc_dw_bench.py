"""The encoders' first-layer weight gradient (gz1^T x with x = edge_attr [E,15] / node inputs [N,12]; the launches that end the
backward on the main queue): microseconds in isolation.   python profiles/tools/enc_dw_bench.py"""
import sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
for M, w, ld in ((75499, 15, 16), (25251, 12, 12), (603992, 15, 16), (202008, 12, 12), (75499, 128, 128)):
    G = (torch.randn(M, 128, generator=g) * 1e-3).to(dev)
    x = torch.randn(M, ld, generator=g).to(dev)
    seg = ops.Seg(x, width=w, ld=ld)
    for _ in range(3):
        dW, db = ops.linear_dw(G, 128, [seg], M, a_op=L.DW_COLSCALE if w <= 16 else 0) if 'a_op' in ops.linear_dw.__code__.co_varnames else ops.linear_dw(G, 128, [seg], M)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.linear_dw(G, 128, [seg], M)
    e1.record(); torch.cuda.synchronize()
    ref = G.double().T @ x[:, :w].double()
    err = float((dW.double() - ref).abs().max() / ref.abs().max())
    print(f"M={M:7d} width {w:3d}: {e0.elapsed_time(e1) * 1e3 / 50:7.1f} us per weight gradient (launch + reduction)   rel err {err:.1e}")
