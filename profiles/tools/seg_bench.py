"""The step's segmented-reduce (scatter-add) launches in isolation, on the bench mesh's own CSR tables: microseconds per launch
(HIP events, back-to-back launches), algorithmic GB/s by DISTINCT source rows and the gathered-row rate the
L2 / Infinity-Cache side serves.  (GFV_SEG_FORM / SEG_FORMS: round 4 compared three kernel forms with this tool, each in a
child process, outputs checked bit for bit against form 0 - profiles/r04_seg_forms.txt; the product library has form 0 only.)
    python profiles/tools/seg_bench.py [meshes_per_gpu] [reps]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "gen-fvgn-steady_amd"))
sys.path.insert(0, ROOT)


def worker(B, reps, dump):
    import torch
    import bench
    from gfv import ops
    from gfv.plan import get_plan
    dev = torch.device("cuda", 0)
    graphs_cpu, sz = bench.build_workload("cylinder", 50000, B, 0, dev)
    graphs = tuple(g.clone().to(dev) for g in graphs_cpu)
    pl = get_plan(graphs)
    N, E = pl.N, pl.E
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(N, 128, generator=g).to(dev)
    e2 = torch.randn(2 * E, 64, generator=g).to(dev)
    agg = torch.randn(N, 64, generator=g).to(dev)
    gz1 = torch.randn(E, 128, generator=g).to(dev)
    shapes = [("nb   = sum_nbr x          [N,128] <- [N,128]", x, pl.n_rowptr, pl.n_col_node, N, None),
              ("agg  = sum_inc e'         [N, 64] <- [2E,64]", e2, pl.n_rowptr, pl.n_col_edge2, N, None),
              ("nbm  = mean_nbr agg       [N, 64] <- [N, 64]", agg, pl.n_rowptr, pl.n_col_node, N, pl.inv_deg),
              ("G_s  = sum_{s(e)=n} gz1   [N,128] <- [E,128]", gz1, pl.s_rowptr, pl.s_col, N, None)]
    outs = []
    for name, src, rp, col, R, scale in shapes:
        F = src.shape[1]
        out = ops.seg_gather_sum(src, rp, col, R, scale=scale)
        torch.cuda.synchronize()
        if os.environ.get("SEG_PMC"):      # counter passes (profiles/tools/seg_pmc.sh): 1 + 3 launches per shape, in this order
            for _ in range(3):
                ops.seg_gather_sum(src, rp, col, R, scale=scale, out=out)
            torch.cuda.synchronize()
            nnz = col.shape[0]
            print(f"SEGSHAPE {name} | distinct_bytes={4.0 * min(src.shape[0], nnz) * F + 4.0 * nnz + 4.0 * R * F + 4.0 * R:.0f} "
                  f"gathered_bytes={4.0 * nnz * F + 4.0 * nnz + 4.0 * R * F:.0f}")
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if os.environ.get("SEG_COLD"):
            # every timed launch behind a 1 GiB fill: L2 and the 256 MB Infinity Cache hold none of its source rows, as inside a
            # step of 8 meshes (the back-to-back form below re-reads what the previous launch left in the Infinity Cache)
            junk = torch.empty(1 << 28, dtype=torch.float32, device=dev)
            ts = []
            for _ in range(max(5, reps // 5)):
                junk.fill_(1.0)
                e0.record()
                ops.seg_gather_sum(src, rp, col, R, scale=scale, out=out)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            nnz = col.shape[0]
            distinct = 4.0 * min(src.shape[0], nnz) * F + 4.0 * nnz + 4.0 * R * F + 4.0 * R
            print(f"  {name}: cold {ts[len(ts) // 2]:7.2f} us median (min {ts[0]:.2f})   {distinct / ts[len(ts) // 2] / 1e3:7.0f} GB/s by distinct rows")
            del junk
            outs.append(out.cpu())
            continue
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(reps):
                ops.seg_gather_sum(src, rp, col, R, scale=scale, out=out)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        nnz = col.shape[0]
        distinct = 4.0 * min(src.shape[0], nnz) * F + 4.0 * nnz + 4.0 * R * F + 4.0 * R
        gathered = 4.0 * nnz * F + 4.0 * nnz + 4.0 * R * F
        print(f"  {name}: {best:7.2f} us   {distinct / best / 1e3:7.0f} GB/s by distinct rows ({distinct / 1e6:6.1f} MB) = "
              f"{distinct / best / 1e3 / 8000:.3f} of 8 TB/s   {gathered / best / 1e3:7.0f} GB/s gathered")
        outs.append(out.cpu())
    torch.save(outs, dump)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import torch
    ref = None
    for form in os.environ.get("SEG_FORMS", "0").split():
        dump = f"/tmp/seg_bench_{form}.pt"
        print(f"GFV_SEG_FORM={form}  ({B} mesh(es) per GPU)", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", str(B), str(reps), dump],
                       env=dict(os.environ, GFV_SEG_FORM=form), check=True)
        outs = torch.load(dump)
        if ref is None:
            ref = outs
        else:
            print("    bit-identical to form 0:", all(torch.equal(a, b) for a, b in zip(ref, outs)))
