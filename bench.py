#!/usr/bin/env python3
"""Headline benchmark: training iterations / second of the Gen-FVGN hot path (forward + loss + backward + Adam,
batch resident on the device) on the synthetic ~50k-cell cylinder mesh (BASELINE.json metric / configs[2]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; every rank owns `--meshes-per-gpu` meshes (weak scaling, graphs sharded by rank) and
the flat fp32 gradient (4.7 MB) is all-reduced with RCCL before the fused Adam step.  Rank 0 prints ONE JSON line.
`roofline` is measured live with HIP events around every launch of the dominant kernel (an instrumented pass of
the same step); `cpu_baseline` times the oracle (oracle/fvgn_oracle.py, the CPU restatement pinned to the reference)
on the host cores for a bounded number of steps of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gen-fvgn-steady_amd"))

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact fp32
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec


def build_workload(cells, meshes, rank, device):
    from gfv import meshgen
    from gfv.graph import build_batch
    nx, ny = meshgen.cylinder_grid_for_cells(cells)
    ms, fs = [], []
    for i in range(meshes):
        raw = meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, jitter=0.2, seed=1234 + rank * meshes + i)
        m = meshgen.finish_mesh(raw)
        ms.append(m)
        fs.append(meshgen.random_fields(m, seed=1 + rank * meshes + i))
    graphs = build_batch(ms, fs, device="cpu")
    sizes = dict(N=int(graphs[0].x.shape[0]), E=int(graphs[0].edge_index.shape[1]), C=int(graphs[3].pos.shape[0]),
                 Sigma=int(graphs[0].face.shape[0]), Ex=int(graphs[1].face_node_x.shape[1]), B=meshes)
    return graphs, sizes


def algorithmic_step_flops(sz, mp=3):
    # forward FLOPs per node / per edge (H=128, TransFVGN_v2), step = 3x forward
    return 3.0 * (1330944.0 * sz["N"] + 1052416.0 * sz["E"])


def cpu_baseline(graphs_cpu, budget_s, max_steps=3):
    from oracle import fvgn_oracle as O
    torch.manual_seed(0)
    P = O.init_parameters(0, perturb=False)
    buffers = O.new_normalizer_buffers()
    state = {}
    x0 = graphs_cpu[0].x.clone()
    times = []
    t_begin = time.time()
    for i in range(max_steps + 1):
        graphs_cpu[0].x = x0.clone()
        t0 = time.time()
        O.train_step(P, buffers, graphs_cpu, state, hyper={"dataset_size": 1})
        dt = time.time() - t0
        if i > 0:
            times.append(dt)
        if time.time() - t_begin > budget_s and times:
            break
    return float(np.median(times)), len(times)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--meshes-per-gpu", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="hipGraph replay of the step; auto = time a few warm-up steps both ways and keep the faster "
                         "(eager launches overlap the side stream better, graph replay needs no host time)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=3)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)  # one rank per GPU on the 8-GPU node; ranks share GPU 0 only in the 1-GPU self-test
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GFV_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; "gloo" only for the self-test
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from gfv import lib as L
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    from FVMmodel.importer import NNmodel
    lib = L.load()

    graphs_cpu, sz = build_workload(args.cells, args.meshes_per_gpu, rank, device)
    graphs = tuple(g.clone().to(device) for g in graphs_cpu)
    torch.manual_seed(0)  # identical initial weights on every rank (data parallel replicas)
    # dataset_size=1: the solve-script regime (solve_with_grad_GPU.py), where the online Normalizer never accumulates
    # and is the identity (utils/normalization.py:39); a single mesh has constant conditioning columns.
    model = NNmodel(default_params(dataset_size=1)).to(device)
    graph_mode = "off" if args.no_graph else args.graph
    ts = TrainStep(model, graphs, world_size=world, use_graph=(graph_mode != "off"))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step()
    if graph_mode == "auto":
        # both ways are the same launches in the same order; which one is faster depends on the host (eager needs ~270
        # launches per step from Python) - decide on this box, all ranks alike (rank 0's measurement is broadcast)
        cal = {}
        for mode in (True, False):
            ts.use_graph = mode
            for _ in range(3):
                ts.step()
            barrier()
            tc = time.perf_counter()
            for _ in range(10):
                ts.step()
            barrier()
            cal[mode] = time.perf_counter() - tc
        pick = torch.tensor([1 if cal[True] <= cal[False] else 0], device=device)
        if world > 1:
            dist.broadcast(pick, src=0)
        ts.use_graph = bool(pick.item())
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(ts.loss.item())

    # ---- roofline leg: same step, eager, HIP events around every launch of the main kernels -----------------
    # every rank runs the instrumented steps (they contain the gradient all-reduce); rank 0 reports.  The timed region
    # above runs the weight-gradient kernels on a side stream, concurrently with the dX chains; here every kernel is
    # launched on ONE stream so that its duration is its own (concurrent kernels share the CUs and stretch each other).
    roof, roof_all = None, []
    ts_use_graph, eng_overlap = ts.use_graph, ts.engine.overlap
    ts.use_graph, ts.engine.overlap = False, False
    lib.gfv_profile_reset()
    lib.gfv_profile_enable(1)
    args.profile_steps = max(1, args.profile_steps)
    for _ in range(args.profile_steps):
        ts.step()
    torch.cuda.synchronize()
    lib.gfv_profile_enable(0)
    ts.use_graph, ts.engine.overlap = ts_use_graph, eng_overlap
    executed_flops = 0.0
    if rank == 0:
        out = (ctypes.c_double * 4)()
        # names as rocprofv3 prints them (profiles/*_kernel_stats.csv)
        # the chain kernels run their fp32 products as 3 f16 MFMAs per product group (include/gfv.h): their matrix
        # roofline is the f16 MFMA peak / 3 in fp32-equivalent flops; every kernel is priced against BOTH rooflines
        # (algorithmic flops and algorithmic bytes over the measured duration) and reported on the one it sits closer to
        h = ", true>" if ts.engine.f16split else ", false>"
        chain_peak = PEAK_F16_MFMA_TFLOPS / 3.0 if ts.engine.f16split else PEAK_F32_MFMA_TFLOPS
        spec = {7: ("tchain_kernel<1, 0, false" + h, chain_peak), 8: ("tchain_kernel<1, 1, false" + h, chain_peak),
                9: ("tchain_kernel<1, 2, false" + h, chain_peak), 10: ("tchain_kernel<1, 0, true" + h, chain_peak),
                1: ("rowtile_chain_kernel", PEAK_F32_MFMA_TFLOPS),
                2: ("dw_multi_h_kernel" if ts.engine.f16split else "dw_multi_kernel", chain_peak),
                3: ("seg_gather_sum_vec", None)}
        for kind, (kname, mfma_peak) in spec.items():
            lib.gfv_profile_collect(kind, out)
            n, ms, fl, by = out[0], out[1], out[2], out[3]
            if n == 0:
                continue
            tf, gbs = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
            f_mfma = tf / mfma_peak if mfma_peak else 0.0
            f_hbm = gbs / PEAK_HBM_GBS
            if mfma_peak:
                executed_flops += fl / args.profile_steps
            if f_mfma >= f_hbm:
                bound, ach, peak, unit = "mfma", tf, mfma_peak, "TFLOP/s"
            else:
                bound, ach, peak, unit = "hbm", gbs, PEAK_HBM_GBS, "GB/s"
            roof_all.append({"kernel": kname, "bound": bound, "achieved": round(ach, 3), "peak": round(peak, 1), "unit": unit,
                             "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": n / args.profile_steps,
                             "avg_launch_us": round(1e3 * ms / n, 2), "ms_per_step": round(ms / args.profile_steps, 4),
                             "fp32_equiv_tflops": round(tf, 3), "algorithmic_gbs": round(gbs, 1),
                             "frac_mfma": round(f_mfma, 4), "frac_hbm": round(f_hbm, 4)})
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        # the committed PMC figures were collected on the default workload (one 50 k-cell mesh per GPU)
        if os.path.exists(pmc) and args.meshes_per_gpu == 1 and args.cells == 50000:
            traffic = json.load(open(pmc))
            for r in roof_all:
                r["traffic"] = traffic.get(r["kernel"])
        if roof_all:
            roof = max(roof_all, key=lambda r: r["ms_per_step"])

    lib.gfv_profile_reset()
    if world > 1:
        dist.barrier()

    # ---- the same step with every GEMM product on the fp32 MFMA (the form without the fp16 split), for the record ----
    fp32_form = None
    if ts.engine.f16split:
        ts_use_graph = ts.use_graph
        ts.use_graph, ts.engine.f16split = False, False
        lib.gfv_set_f16split(0)
        for _ in range(3):
            ts.step()
        barrier()
        t0 = time.perf_counter()
        nf = max(5, min(20, args.steps))
        for _ in range(nf):
            ts.step()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        lib.gfv_set_f16split(1)
        ts.use_graph, ts.engine.f16split = ts_use_graph, True
        fp32_form = {"value": round(world * args.meshes_per_gpu * nf / el, 3), "ms_per_step": round(1e3 * el / nf, 4),
                     "steps": nf, "note": "eager launches, all products on v_mfma_f32_16x16x4_f32 (GFV_F16SPLIT=0)"}

    cpu = None
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        # 16 threads is the fastest setting for this launch-bound eager workload on the GPU box's host
        # (measured 8/16/32/64/256 threads: 4.3 / 3.5 / 3.6 / 5.2 / 188 s per step); override with GFV_CPU_THREADS
        ncores = int(os.environ.get("GFV_CPU_THREADS", min(16, os.cpu_count() or 1)))
        torch.set_num_threads(ncores)
        sec, nst = cpu_baseline(graphs_cpu, args.cpu_budget)
        cpu = {"value": round(args.meshes_per_gpu / sec, 5), "unit": "train-iters/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": f"{nst} timed steps (after 1 warm-up) of the same {sz['C']}-cell mesh batch, "
               f"oracle/fvgn_oracle.py fp32 eager PyTorch, median {sec:.3f} s/step"}

    if rank == 0:
        total_meshes = world * args.meshes_per_gpu
        value = total_meshes * args.steps / elapsed
        line = {
            "metric": "training iters/sec, 50k-cell cylinder mesh (fwd + loss + bwd + Adam, batch resident in HBM)",
            "value": round(value, 3), "unit": "mesh-train-iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": ("fp32 values end to end; the products of the fused GEMM chains run as 3 f16 MFMAs on exact (hi, lo) "
                           "fp16 splits of the fp32 operands with fp32 accumulation (error <= the f32 MFMA's, parity tests at 1e-5)"
                           if ts.engine.f16split else "fp32 MFMA"),
            "config": {"workload": "cylinder_flow tri mesh, TransFVGN_v2 (hidden 128, mp 3), 2nd-order WLSQ, conserved form",
                       "cells": sz["C"], "nodes": sz["N"], "faces": sz["E"], "meshes_per_gpu": args.meshes_per_gpu,
                       "global_batch": total_meshes, "parallelism": f"dp{world}", "hip_graph": bool(ts.use_graph),
                       "final_loss": round(final_loss, 6)},
            "roofline": ({k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | {"kernel": roof["kernel"]})
            if roof else None,
            "roofline_kernels": roof_all,
            "fp32_mfma_form": fp32_form,
            # reference algorithm vs what the launches execute (EdgeBlock first layer factored through
            # the nodes, gfv/engine.py): the fraction of the fp32 MFMA peak is quoted on the EXECUTED flops
            "algorithmic_step_tflops": round(algorithmic_step_flops(sz) / 1e12, 4),
            "executed_step_tflops": round(executed_flops / 1e12, 4),
            "step_mfma_frac": round(executed_flops / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "roofline_note": "per-kernel durations: HIP events, one stream, eager; value: hipGraph replay with dW on a side stream",
            "cpu_baseline": cpu,
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
