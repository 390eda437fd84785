"""The Transolver's linear_post adjoint launch (one Linear 128 -> 256 with the GELU' epilogue, csrc/lin1.hip) in isolation against
its plain sibling (same shapes, no epilogue operand):  python profiles/tools/lin1_dgelu_bench.py [rows]"""
import sys
sys.path.insert(0, 'gen-fvgn-steady_amd')
import torch
from gfv import lib as L, ops
from gfv.ops import Seg, LayerSpec
dev = 'cuda'
N = int(sys.argv[1]) if len(sys.argv) > 1 else 203832
wi = ops.WeightImages(torch.device(dev), torch.full((1,), 0.25, device=dev))
wi.static = [(0, 1 << 62)]
ops.set_weight_images(wi)
g = torch.randn(N, 128, device=dev)
z = torch.randn(N, 256, device=dev)
out = torch.empty(N, 256, device=dev)
Wt = torch.randn(256, 128, device=dev) * 0.05


def run(dgelu):
    ly = LayerSpec(Wt, None, L.OP_MUL_DGELU, aux=z) if dgelu else LayerSpec(Wt, None)
    ops.rowtile_chain(N, [Seg(g)], [ly], [(out, 256), (out.data_ptr() + 512, 256)])


for dgelu in (False, True):
    run(dgelu)
    path = L.load().gfv_rowtile_last_path()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(dgelu)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    by = 4.0 * N * (128 + 256 + (256 if dgelu else 0))
    print(f"rows {N}  GELU' epilogue {dgelu!s:5s}  path {path}  {best:8.1f} us  {by / best / 1e3:7.0f} GB/s algorithmic")
