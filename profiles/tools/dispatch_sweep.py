"""Which kernel family should take a launch of M rows - measured over MESH SIZES, not on the benchmark's two meshes only.

For every mesh size: the fused training step (TrainStep, command-list replay) with the dispatch limits as shipped, then with ONE
family at a time forced to its small-tile form ("small") and to its large-launch form ("large").  The shipped limit is right at a
size when the default equals the faster of the two.  Families and the switches that move them:

    cbwd    GFV_CBWD_MAX_M                         node-level 3-layer backward: cbwd.hip | the persistent colchain_bwd
    ctrans  GFV_CTRANS_MAX_M                       Transolver row chains: ctrans.hip | transmlp.hip
    lin1s   GFV_LIN1S_MAX_M, engine._csr1_max      single-layer launches: lin1s.hip (neighbour sum in its prologue) | lin1.hip
    cfwd    GFV_CFWD_MAX_M                         3-layer forward: cfwd.hip | the row-owner chain
            (until this sweep the encoders' narrow first layers and the decoder had a limit of their own, GFV_CFWD_RAG_MAX_M
            16 384: "rag" in profiles/r06_dispatch_sweep.txt; faster on cfwd.hip at every size, so the limit went)
    --engine: the host-side switches of gfv/engine.py (fused neighbour sums, which flushes end on the main queue, ...) one at a time

    python3 profiles/tools/dispatch_sweep.py [--cells 16000,24000,...] [--steps 60] > gpurun_out/dispatch_sweep.txt
"""
import argparse
import gc
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gen-fvgn-steady_amd"))

import torch  # noqa: E402

BIG = 1 << 30
FAMILIES = {
    "cbwd": (dict(GFV_CBWD_MAX_M=BIG), dict(GFV_CBWD_MAX_M=0), None),
    "ctrans": (dict(GFV_CTRANS_MAX_M=BIG), dict(GFV_CTRANS_MAX_M=0), None),
    "lin1s": (dict(GFV_LIN1S_MAX_M=BIG), dict(GFV_LIN1S_MAX_M=0), "_csr1_max"),
    "cfwd": (dict(GFV_CFWD_MAX_M=BIG), dict(GFV_CFWD_MAX_M=0), None),
}


ENGINE_SETTINGS = [   # (label, {engine attribute: value}) - the host-side switches of gfv/engine.py, one at a time
    ("default", {}),
    ("fuse_mask=0", dict(_fuse_mask=0)), ("fuse_mask=1", dict(_fuse_mask=1)), ("fuse_mask=2", dict(_fuse_mask=2)),
    ("fuse_mask=3", dict(_fuse_mask=3)), ("fuse_mask=5", dict(_fuse_mask=5)), ("fuse_mask=6", dict(_fuse_mask=6)),
    ("fuse_mask=7", dict(_fuse_mask=7)),
    ("tail=(1,0)", dict(_tail_env=True, _tail_main=1, _tail_split=0)), ("tail=(2,0)", dict(_tail_env=True, _tail_main=2, _tail_split=0)),
    ("tail=(2,1)", dict(_tail_env=True, _tail_main=2, _tail_split=1)), ("tail=(2,2)", dict(_tail_env=True, _tail_main=2, _tail_split=2)),
    ("tail=(3,0)", dict(_tail_env=True, _tail_main=3, _tail_split=0)), ("tail=(3,1)", dict(_tail_env=True, _tail_main=3, _tail_split=1)),
    ("tail=(3,2)", dict(_tail_env=True, _tail_main=3, _tail_split=2)),
    ("factor=0", dict(factor=False)), ("fvm_fuse=0", dict(_fvm_fuse=False)), ("reduce_merge=0", dict(_trans_reduce_merge=False)),
    ("agg_ln=0", dict(_agg_ln=False)), ("slice_fuse=0", dict(_slice_fuse=False)), ("trans_fuse=0", dict(_trans_fuse=False)), ("overlap=0", dict(overlap=False)),
    ("defer=0", dict(_defer_mode=False)), ("fuse_dw_min=0", dict(_fuse_dw_min=0)), ("fuse_dw_min=8192", dict(_fuse_dw_min=8192)),
    ("default", {}),
]


def time_setting(graphs, limits, engine_attr, small, steps):
    from gfv import lib as L
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    from FVMmodel.importer import NNmodel
    torch.manual_seed(0)
    with L.limits(**limits):
        model = NNmodel(default_params(dataset_size=1)).to("cuda")
        eng = model.engine()
        if engine_attr == "_csr1_max":          # neighbour sum in the single-layer launch's prologue up to this many rows
            eng._csr1_max = BIG if small else 0
        elif isinstance(engine_attr, dict):
            for k, v in engine_attr.items():
                assert hasattr(eng, k), k
                setattr(eng, k, v)
        ts = TrainStep(model, tuple(g.clone() for g in graphs), use_graph="list")
        for _ in range(12):
            ts.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                ts.step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / steps)
        flags = L.status_flags() if hasattr(L, "status_flags") else 0
    del ts, model
    gc.collect()
    torch.cuda.empty_cache()
    return 1e3 * best, flags


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", default="16000,24000,32000,40000,46000,54000,64000,80000")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--families", default="cbwd,ctrans,lin1s,cfwd")
    ap.add_argument("--engine", action="store_true", help="sweep the host-side switches of gfv/engine.py instead of the limits")
    ap.add_argument("--workload", default="cylinder")
    ap.add_argument("--cbwd-node", action="store_true",
                    help="the NODE-level 3-layer backward alone: persistent fused backward (limit 0) | small-tile cbwd for the node rows "
                         "only (limit N + 1) | for node and edge rows (limit 2^30)")
    ap.add_argument("--tail", action="store_true", help="only the (GFV_TAIL_MAIN, GFV_TAIL_SPLIT) pairs, as one table over the sizes")
    args = ap.parse_args()
    import bench
    from gfv import host
    host.pin_to_l3()
    fams = args.families.split(",")
    if args.cbwd_node:
        print("# ms per fused training step; GFV_CBWD_MAX_M = 0 (persistent backward at both levels) | N + 1 (cbwd for the node-level "
              "MLPs only) | 2^30 (cbwd at both levels); 'shipped' = 25 000")
        print("%8s %7s %7s %9s %9s %9s %9s" % ("cells", "N", "E", "shipped", "none", "node", "both"), flush=True)
        for cells in [int(c) for c in args.cells.split(",")]:
            graphs_cpu, sz = bench.build_workload(args.workload, cells, 1, 0, torch.device("cuda"))
            graphs = tuple(g.clone().to("cuda") for g in graphs_cpu)
            row = [time_setting(graphs, lim, None, False, args.steps)[0]
                   for lim in ({}, dict(GFV_CBWD_MAX_M=0), dict(GFV_CBWD_MAX_M=sz["N"] + 1), dict(GFV_CBWD_MAX_M=BIG))]
            print("%8d %7d %7d " % (sz["C"], sz["N"], sz["E"]) + " ".join("%9.4f" % v for v in row), flush=True)
            del graphs, graphs_cpu
            gc.collect()
            torch.cuda.empty_cache()
        return
    if args.tail:
        pairs = [(2, 0), (2, 1), (2, 2), (3, 0), (3, 1), (3, 2), (1, 0)]
        print("# ms per fused training step, (GFV_TAIL_MAIN, GFV_TAIL_SPLIT) forced; 'rule' = gfv/engine.py _tail_cfg as shipped")
        print("%-9s %7s %7s %8s " % ("workload", "cells", "E", "rule") + " ".join("%8s" % ("(%d,%d)" % p) for p in pairs), flush=True)
        for item in args.cells.split(","):
            wl, cells = (item.split(":") + [None])[:2] if ":" in item else (args.workload, item)
            graphs_cpu, sz = bench.build_workload(wl, int(cells), 1, 0, torch.device("cuda"))
            graphs = tuple(g.clone().to("cuda") for g in graphs_cpu)
            r0, _ = time_setting(graphs, {}, {}, False, args.steps)
            row = [time_setting(graphs, {}, dict(_tail_env=True, _tail_main=a, _tail_split=b), False, args.steps)[0] for a, b in pairs]
            r1, _ = time_setting(graphs, {}, {}, False, args.steps)
            print("%-9s %7d %7d %8.4f " % (wl, sz["C"], sz["E"], min(r0, r1)) + " ".join("%8.4f" % v for v in row), flush=True)
            del graphs, graphs_cpu
            gc.collect()
            torch.cuda.empty_cache()
        return
    if args.engine:
        print("# ms per fused training step (command-list replay, best of 3 x %d steps), one engine switch moved at a time" % args.steps)
        for cells in [int(c) for c in args.cells.split(",")]:
            graphs_cpu, sz = bench.build_workload(args.workload, cells, 1, 0, torch.device("cuda"))
            graphs = tuple(g.clone().to("cuda") for g in graphs_cpu)
            print("%s cells %d N %d E %d" % (args.workload, sz["C"], sz["N"], sz["E"]), flush=True)
            for label, attrs in ENGINE_SETTINGS:
                try:
                    ms, _ = time_setting(graphs, {}, dict(attrs), False, args.steps)
                    print("   %-18s %9.4f" % (label, ms), flush=True)
                except Exception as e:   # a switch that cannot be moved on this batch
                    print("   %-18s failed: %s" % (label, str(e)[:120]), flush=True)
            del graphs, graphs_cpu
            gc.collect()
            torch.cuda.empty_cache()
        return
    print("# ms per fused training step (command-list replay, best of 3 x %d steps); default = the shipped limits;" % args.steps)
    print("# <family>:small / :large = that family forced to its small-tile / large-launch kernels, everything else as shipped")
    hdr = "%8s %7s %7s %9s" % ("cells", "N", "E", "default")
    for f in fams:
        hdr += " %12s %12s" % (f + ":small", f + ":large")
    print(hdr, flush=True)
    for cells in [int(c) for c in args.cells.split(",")]:
        graphs_cpu, sz = bench.build_workload(args.workload, cells, 1, 0, torch.device("cuda"))
        graphs = tuple(g.clone().to("cuda") for g in graphs_cpu)
        d0, _ = time_setting(graphs, {}, None, False, args.steps)
        row = []
        for f in fams:
            small, large, attr = FAMILIES[f]
            a, _ = time_setting(graphs, small, attr, True, args.steps)
            b, _ = time_setting(graphs, large, attr, False, args.steps)
            row += [a, b]
        d1, _ = time_setting(graphs, {}, None, False, args.steps)
        line = "%8d %7d %7d %9.4f" % (sz["C"], sz["N"], sz["E"], min(d0, d1))
        for v in row:
            line += " %12.4f" % v
        line += "   (default again: %.4f)" % max(d0, d1)
        print(line, flush=True)
        del graphs, graphs_cpu
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
