"""Where does one tile of the row-owner chain spend its time?  Per-phase cycle counters of a -DGFV_TIMING build of the library
(profiles/tools/build_variant.sh tm "-DGFV_TIMING"; run with GFV_LIB=profiles/tools/variants/libgfv_tm.so): the NodeBlock MLP
forward ([nbm 64 | x 128] -> 128 -> 128 -> 128, LayerNorm, residual, saves) at several row counts.
   python profiles/tools/tchain_phases.py [M ...]
Counter slots (csrc/tchain_kernel.h TS): 0 first weight slice staged + barrier, 1 input rows loaded / scaled / split, 2 prefetch
issue, 3 MFMAs of a slice, 4 prefetched slice -> LDS (waits for its global load), 5 slice barrier, 6 hidden epilogues (bias, save,
GELU, split; all memory waited for), 7 final epilogue (LayerNorm, stores), 8 whole wave, 9 start time."""
import sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
dev = torch.device('cuda:0')
Ms = [int(a) for a in sys.argv[1:]] or [1089, 5184, 25479]
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
names = ["slice0+barrier", "input rows", "prefetch issue", "MFMA slices", "slice -> LDS", "slice barrier", "hidden epilogues",
         "final epilogue", "wave total"]
for M in Ms:
    nbm, x = r(M, 64), r(M, 128)
    W1, W2, W3 = r(128, 192) * 0.05, r(128, 128) * 0.05, r(128, 128) * 0.05
    b1, b2, b3, gam, bet = r(128), r(128), r(128), r(128), r(128)
    wi = ops.WeightImages(dev, torch.full((1,), 0.5, device=dev)); wi.static = [(0, 1 << 62)]
    z1, z2, y3, out = (torch.empty(M, 128, device=dev) for _ in range(4))
    tiles = (M + 63) // 64
    dbg = torch.zeros(tiles * 4 * 10 + 64, dtype=torch.int64, device=dev)
    def run():
        ops.rowtile_chain(M, [ops.Seg(nbm), ops.Seg(x)],
                          [ops.LayerSpec(W1, b1, L.OP_BIAS_GELU, save=z1), ops.LayerSpec(W2, b2, L.OP_BIAS_GELU, save=z2), ops.LayerSpec(W3, b3)],
                          [out], fin_op=L.FIN_LN, fin_gamma=gam, fin_beta=bet, fin_presave=y3, res=[x], wimg=wi, ln_partial=dbg.view(torch.float32))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 20
    t = dbg[:tiles * 4 * 10].view(tiles * 4, 10).double().cpu()
    t = t[t[:, 8] > 0]
    start = t[:, 9] - t[:, 9].min()
    end = start + t[:, 8]
    print(f"M={M}: {tiles} tiles, {us:.1f} us per launch (timing build), path {L.load().gfv_rowtile_last_path()}; "
          f"span first start -> last end {end.max():.0f} cycles, last start at {start.max():.0f}")
    for k, n in enumerate(names):
        print(f"   {n:18s} mean {t[:, k].mean():9.0f}  min {t[:, k].min():9.0f}  max {t[:, k].max():9.0f} cycles")
