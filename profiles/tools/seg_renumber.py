"""Would a locality renumbering of the nodes help the scatter-add?  The neighbour sum nb = sum_nbr x ([N,128] <- [N,128]) on the
bench mesh's own node -> node CSR table, as numbered by the mesh generator and renumbered along a Morton curve of the node
positions (per mesh; rows, columns and source rows permuted consistently - the same sums in a different row order):
microseconds back-to-back and cold (behind a 1 GiB fill), as profiles/tools/seg_bench.py times them.
    python profiles/tools/seg_renumber.py [meshes_per_gpu]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gen-fvgn-steady_amd"))
sys.path.insert(0, ROOT)
import torch
import bench
from gfv import ops
from gfv.plan import get_plan

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
graphs_cpu, sz = bench.build_workload("cylinder", 50000, B, 0, dev)
graphs = tuple(g.clone().to(dev) for g in graphs_cpu)
pl = get_plan(graphs)
N = pl.N
rowptr, col = pl.n_rowptr.cpu().long(), pl.n_col_node.cpu().long()
pos, batch = graphs_cpu[0].pos.double(), graphs_cpu[0].batch.long()


def morton(p, b):
    q = p - p.min(0).values
    q = (q / q.max(0).values * 65535).long()
    code = torch.zeros(p.shape[0], dtype=torch.long)
    for bit in range(16):
        code |= ((q[:, 0] >> bit) & 1) << (2 * bit)
        code |= ((q[:, 1] >> bit) & 1) << (2 * bit + 1)
    return torch.argsort(b * (1 << 34) + code, stable=True)     # meshes stay contiguous


def timed(src, rp, cl, label):
    rp, cl = rp.int().to(dev), cl.int().to(dev)
    out = ops.seg_gather_sum(src, rp, cl, N)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(30):
            ops.seg_gather_sum(src, rp, cl, N, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 30)
    junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    ts = []
    for _ in range(9):
        junk.fill_(1.0)
        e0.record()
        ops.seg_gather_sum(src, rp, cl, N, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    span = (cl.long().cpu() - torch.repeat_interleave(torch.arange(N), (rp[1:] - rp[:-1]).long().cpu())).abs().double()
    print(f"  {label:28s} back-to-back {best:6.2f} us   cold {ts[len(ts) // 2]:6.2f} us   |row - neighbour| mean {span.mean():9.0f} rows, "
          f"median {span.median():7.0f}, 95 % {span.quantile(0.95):9.0f}")
    return out


g = torch.Generator().manual_seed(1)
x = torch.randn(N, 128, generator=g)
o0 = timed(x.to(dev), rowptr, col, "generator numbering")
perm = morton(pos, batch)                    # new row i holds old row perm[i]
inv = torch.empty_like(perm)
inv[perm] = torch.arange(N)
deg = rowptr[1:] - rowptr[:-1]
rp2 = torch.zeros(N + 1, dtype=torch.long)
rp2[1:] = torch.cumsum(deg[perm], 0)
starts = rowptr[:-1][perm]
idx = torch.repeat_interleave(starts - rp2[:-1], deg[perm]) + torch.arange(int(rp2[-1]))
col2 = inv[col[idx]]
o1 = timed(x[perm].to(dev), rp2, col2, "Morton order of the positions")
print("  same sums:", bool(torch.equal(o1.cpu(), o0.cpu()[perm])))
