"""Backward of one fused EdgeBlock MLP (M rows): the row-owner dX chain + the weight-gradient launch (3 tiles) + reduction
against the column-owner chain with fused weight gradients + the one-tile weight-gradient launch + reduction."""
import sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec
from gfv.engine import Engine, GradStore

dev = 'cuda'
wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
wi.static = [(0, 1 << 62)]
ops.set_weight_images(wi)


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
for M in (75499, 603992):
    e, G, y3, z1, z2 = (torch.randn(M, 128, device=dev) for _ in range(5))
    stats = torch.stack((y3.mean(1), (y3.var(1, unbiased=False) + 1e-5).rsqrt()), 1).contiguous()
    gagg = torch.randn(M // 3 + 1, 64, device=dev)
    s = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
    r = (torch.arange(M, device=dev) // 3 + torch.randint(0, 40, (M,), device=dev)).clamp_(max=M // 3).int()
    Wt = [torch.randn(128, 128, generator=g).to(dev) * 0.05 for _ in range(3)]
    gam = torch.ones(128, device=dev)
    g3, gz2, gz1, ge = (torch.empty(M, 128, device=dev) for _ in range(4))
    tiles = ops.rowtile_tiles(M)
    part = torch.empty(tiles, 2, 128, device=dev)
    gs = torch.empty(3, ops.gscale_ld(M), device=dev)
    nwg = L.load().gfv_rowtile_dw_partials()
    dwp = torch.empty(nwg, L.DW_FUSED_FLOATS, device=dev)
    dwp1 = torch.empty(nwg, L.DW_FUSED_FLOATS_IN, device=dev)

    def chain_row():
        ops.rowtile_chain(M, [Seg(G)], [LayerSpec(Wt[0], None, L.OP_MUL_DGELU, save=gz2, aux=z2),
                                        LayerSpec(Wt[1], None, L.OP_MUL_DGELU, save=gz1, aux=z1), LayerSpec(Wt[2])],
                          [ge], res=[G], in_op=L.IN_LNBWD, in_gamma=gam, in_aux=y3, in_save=g3, ln_partial=part, gadd=gagg,
                          gadd_s=s, gadd_r=r, gscale=gs, family=L.CHAIN_ROW_OWNER)

    def chain_col():
        ops.rowtile_chain(M, [Seg(G)], [LayerSpec(Wt[0], None, L.OP_MUL_DGELU, aux=z2),
                                        LayerSpec(Wt[1], None, L.OP_MUL_DGELU, save=gz1, aux=z1), LayerSpec(Wt[2])],
                          [ge], res=[G], in_op=L.IN_LNBWD, in_gamma=gam, in_aux=y3, in_stats=stats, gadd=gagg, gadd_s=s,
                          gadd_r=r, dw_partial=dwp, family=L.CHAIN_COLUMN_OWNER)

    def chain_col1():
        ops.rowtile_chain(M, [Seg(G)], [LayerSpec(Wt[0], None, L.OP_MUL_DGELU, aux=z2),
                                        LayerSpec(Wt[1], None, L.OP_MUL_DGELU, save=gz1, aux=z1), LayerSpec(Wt[2])],
                          [ge], res=[G], in_op=L.IN_LNBWD, in_gamma=gam, in_aux=y3, in_stats=stats, gadd=gagg, gadd_s=s,
                          gadd_r=r, dw_partial=dwp1, dw_in=e, family=L.CHAIN_COLUMN_OWNER)

    eng = Engine()
    grads = GradStore(["W1c", "b1", "W2", "b2", "W3", "b3"], [(128, 128), (128,), (128, 128), (128,), (128, 128), (128,)], dev)

    def dw3():
        eng._dw_block(grads, [("W1c", "b1", 1), ("W2", "b2", 1), ("W3", "b3", 1)],
                      [eng._tile(gz1, 128, Seg(e), gscale=gs[2]), eng._tile(gz2, 128, Seg(z1), a_op=1, gscale=gs[1]),
                       eng._tile(g3, 128, Seg(z2), a_op=1, gscale=gs[0])], M)

    def dw1():
        eng._dw_block(grads, [("W1c", "b1", 1)], [eng._tile(gz1, 128, Seg(e))], M)

    def red():
        ops.reduce_multi([dict(partial=dwp, out=grads.view("W3"), n_chunks=nwg, chunk_stride=L.DW_FUSED_FLOATS, rows=1, cols=16384 + 128),
                          dict(partial=dwp.data_ptr() + 4 * (16384 + 128), out=grads.view("W2"), n_chunks=nwg,
                               chunk_stride=L.DW_FUSED_FLOATS, rows=1, cols=16384 + 128)])

    chain_row()
    t = [timeit(f) for f in (chain_row, dw3, chain_col, dw1, red, chain_col1)]
    print(f"M={M}: row-owner dX chain {t[0]:.1f} + dW (3 tiles, with reduction) {t[1]:.1f} = {t[0] + t[1]:.1f} us | "
          f"column-owner fused {t[2]:.1f} + dW1 {t[3]:.1f} + reduce {t[4]:.1f} = {t[2] + t[3] + t[4]:.1f} us | "
          f"with dW1 fused too {t[5]:.1f} + reduce ~{1.5 * t[4]:.1f} = {t[5] + 1.5 * t[4]:.1f} us", flush=True)
