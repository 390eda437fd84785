#!/usr/bin/env python3
"""Headline benchmark: training iterations / second of the Gen-FVGN hot path (forward + loss + backward + Adam,
batch resident on the device) on the synthetic ~50k-cell cylinder mesh (BASELINE.json metric / configs[2]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One process per GPU; every rank owns `--meshes-per-gpu` meshes (weak scaling, graphs sharded by rank) and
the flat fp32 gradient (4.7 MB) is all-reduced with RCCL before the fused Adam step.  Rank 0 prints ONE JSON line.

What the line carries (DESIGN.md 5 explains every figure):
  value / ms_per_step   K-step timed loops bracketed by barrier + synchronize, repeated until `--min-time` seconds have
                        been timed (so the device is visibly busy); value = all timed steps / all timed seconds, no picking
  step_modes            the same step timed as eager launches AND as hipGraph replay (the faster one runs the timed loops)
  roofline              the dominant kernel, measured live with HIP events around every launch (single stream, eager)
  roofline_kernels      every kernel class of the step (chains, weight gradients, segmented reduce, slice attention,
                        finite-volume, reductions, weight images, input preparation / Adam) with algorithmic flops / bytes
  roofline_step         SURVEY.md 8(d) COMPULSORY bytes of the whole step over the step time and 8 TB/s (frac_compulsory),
                        per class and for the step, and the share of the single-stream step the priced launches cover
  cpu_baseline          the oracle (oracle/fvgn_oracle.py, the CPU restatement pinned to the reference) on the host cores
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

# (The runtime's default of 4 hardware queues stays: streams are multiplexed onto them round-robin and RCCL's own streams can
# push the step's two streams onto one queue - gfv/engine.py pick_concurrent_stream checks its choice at run time instead.
# GPU_MAX_HW_QUEUES=8 was tried as a default: the same eager / command-list step time, hipGraph replay 7.3 instead of 4.7 ms.)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gen-fvgn-steady_amd"))

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact fp32
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec
N_PARAMS = 1181539             # TransFVGN_v2, hidden 128, mp 3


def polygon_example_mesh():
    """BASELINE.json configs[4] stand-in that can travel: the reference's polygon example mesh (cylinder_flow_poly, Tecplot
    FEPolygon, 17 436 cells of 3 ... 9 nodes) from the reader arrays committed as test data (tests/golden/poly_cylinder.npz,
    made by tests/golden/make_golden_poly.py), through the product's own ingest (gfv.ingest.tecplot_to_raw)."""
    import json
    from gfv import ingest, meshgen
    fx = np.load(os.path.join(ROOT, "tests", "golden", "poly_cylinder.npz"))
    tec = {"pos": fx["tec.pos"], "face_node": fx["tec.face_node"].astype(np.int64), "left": fx["tec.left"].astype(np.int64),
           "right": fx["tec.right"].astype(np.int64), "boundary_pos": fx["tec.boundary_pos"]}
    bcd = json.loads(str(fx["raw.bc"]))
    bc_json = {"stencil|khops": bcd["stencil|khops"], "sigma": bcd["sigma"], "inlet_type": bcd["inlet_type"],
               "theta_PDE": dict(bcd["theta_PDE"], inlet=[bcd["U"]], rho=[bcd["rho"]], mu=[bcd["mu"]], source=[bcd["source"]],
                                 aoa=[bcd["aoa"]], dt=bcd["dt"], L=bcd["L"])}
    raw = ingest.tecplot_to_raw(tec, bc_json)
    raw["bc"] = bcd
    return meshgen.finish_mesh(raw), fx["field"]


def build_workload(workload, cells, meshes, rank, device):
    from gfv import meshgen
    from gfv.graph import build_batch
    ms, fs = [], []
    if workload == "poly":
        m, f = polygon_example_mesh()
        ms, fs = [m] * meshes, [f] * meshes
    for i in range(meshes if workload != "poly" else 0):
        seed = 1234 + rank * meshes + i
        if workload == "cavity":      # BASELINE.json configs[1]: lid-driven cavity, n x n quads (71 x 71 = 5 041 cells)
            n = max(2, int(round(cells ** 0.5)))
            raw = meshgen.raw_quad_cavity(n=n, jitter=0.0, seed=seed)
        else:                         # configs[2]: triangulated channel with a cylinder, ~cells cells
            nx, ny = meshgen.cylinder_grid_for_cells(cells)
            raw = meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, jitter=0.2, seed=seed)
        # several meshes per GPU (BASELINE config 4: 8 per GPU): the two heavy set-up steps (k-hop stencil, WLSQ moments) run as
        # HIP kernels (gfv.device_prep) - 8 ranks x 8 meshes of host set-up would take minutes before the first step
        m = meshgen.finish_mesh(raw, device=device if meshes > 1 else None)
        ms.append(m)
        fs.append(meshgen.random_fields(m, seed=1 + rank * meshes + i))
    graphs = build_batch(ms, fs, device="cpu")
    sizes = dict(N=int(graphs[0].x.shape[0]), E=int(graphs[0].edge_index.shape[1]), C=int(graphs[3].pos.shape[0]),
                 Sigma=int(graphs[0].face.shape[0]), Ex=int(graphs[1].face_node_x.shape[1]), B=meshes)
    return graphs, sizes


def algorithmic_step_flops(sz, mp=3):
    # forward FLOPs per node / per edge (H=128, TransFVGN_v2), step = 3x forward
    return 3.0 * (1330944.0 * sz["N"] + 1052416.0 * sz["E"])


def compulsory_step_bytes(sz):
    """SURVEY.md 8(d) convention: every distinct input element read once, every output element written once, int32
    indices; a step = forward + ~2x backward.  Per class (bytes per STEP):
      gnn   6 GnBlocks x (1584 E + 3584 N) forward (8d), + encoders (read x 48 N, edge_attr 60 E; write 512 N + 512 E),
            decoder (512 N in, 12 N out), 2 Transolver blocks as Linear layers only (in/out 512 N each per Linear pair:
            3 passes of 1024 N) - all x3 for the step
      slice 2 blocks x (x_mid 512 N in + slice weights 1024 N out/in + out 512 N) x3
      fvm   WLSQ 28 Ex + 408 N, interpolation / flux 100 Sigma (8d: ~15 MB at 50 k cells), x3
      misc  Adam 28 B / parameter (8d), input preparation 108 N + 72 E once"""
    N, E, Ex, Sg = sz["N"], sz["E"], sz["Ex"], sz["Sigma"]
    gnn = 3.0 * (6 * (1584.0 * E + 3584.0 * N) + (48.0 + 512.0) * N + (60.0 + 512.0) * E + 524.0 * N + 2 * 3 * 1024.0 * N)
    slc = 3.0 * 2 * 2048.0 * N
    fvm = 3.0 * (28.0 * Ex + 408.0 * N + 100.0 * Sg)
    misc = 28.0 * N_PARAMS + 108.0 * N + 72.0 * E
    return {"gnn": gnn, "slice": slc, "fvm": fvm, "misc": misc}


def measured_copy_rate_gbs(device, nbytes=1 << 30, reps=6):
    """HBM copy rate of THIS device in THIS run: a 1 GiB device-to-device copy (read + write = 2 GiB of traffic, ~0.35 ms ...
    repeated until ~20 ms have been timed), HIP events on the current stream, best of `reps`."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device=device).fill_(1.0)
    dst = torch.empty_like(src)
    dst.copy_(src)
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 8 * 2.0 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return best


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(graphs_cpu, budget_s, max_steps=6, min_steps=1):
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
        if time.time() - t_begin > budget_s and len(times) >= min_steps:
            break
    return float(np.median(times)), len(times)


def drop_in_leg(device, graphs_cpu, ts_ms, steps, min_time=1.0):
    """The path north_star names - the reference's driver UNCHANGED: the literal sequence of pre_train_Adam.py:158-191 /
    solve_with_grad_GPU.py:133-181 (`optimizer.zero_grad(); out = model(...); loss = mean(log(...)); loss.backward();
    optimizer.step()`) on a fresh `NNmodel` and the same batch, node state restored and norm flags re-armed every iteration as the
    solve loop does.  Timed with stock `torch.optim.Adam` and with `gfv.optim.Adam` (one import changed), each with the recorded
    launch lists (default) and with every launch issued eagerly (GFV_DROPIN_REPLAY=0: what rounds 1 - 5 shipped)."""
    from FVMmodel.importer import NNmodel
    from gfv.optim import Adam as GfvAdam
    from gfv.params import default_params
    out = {}
    for name, opt_cls, replay in (("torch_adam", torch.optim.Adam, True), ("gfv_adam", GfvAdam, True),
                                  ("torch_adam_eager", torch.optim.Adam, False)):
        torch.manual_seed(0)
        params = default_params(dataset_size=1)
        model = NNmodel(params).to(device)
        model._replay.enabled = replay
        graphs = tuple(g.clone().to(device) for g in graphs_cpu)
        gn = graphs[0]
        backup = gn.x.clone()
        opt = opt_cls(model.parameters(), lr=params.lr)

        def it():
            gn.x.copy_(backup)
            gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
            opt.zero_grad()
            lc, lx, ly, lp, un, uc = model(graph_node=graphs[0], graph_node_x=graphs[1], graph_edge=graphs[2],
                                           graph_cell=graphs[3], graph_Index=graphs[4], is_training=True)
            loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lx + params.loss_mom * ly))
            loss.backward()
            opt.step()
            return loss
        for _ in range(8):
            it()
        torch.cuda.synchronize()
        n, el = 0, 0.0
        while el < min_time and n < 50 * steps:
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = it()
            torch.cuda.synchronize()
            el += time.perf_counter() - t0
            n += steps
        out[name] = {"ms_per_step": round(1e3 * el / n, 4), "timed_steps": n, "final_loss": round(float(loss.detach()), 6),
                     "replayed_forward_calls": model._replay.replays, "over_trainstep": round(1e3 * el / n / ts_ms, 3)}
        del model, opt, graphs
        torch.cuda.empty_cache()
    out["note"] = ("the reference driver's own call sequence on FVMmodel.importer.NNmodel (autograd node, loss in torch ops, "
                   "torch.optim.Adam over 159 tensors) against gfv.trainer.TrainStep's fused step (`ms_per_step` of this line, "
                   "over_trainstep = the ratio); same mesh, same launch sequence inside the model.  Single process, no collectives")
    return out


def self_launch_command(gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` runs when it is started WITHOUT a launcher: the driver's own multi-GPU line
    (one rank per GPU over RCCL, rendezvous on 127.0.0.1), with this invocation's arguments passed through unchanged."""
    if port is None:
        port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(gpus, argv):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    env.pop("GFV_BENCH_SELF_LAUNCH", None)
    cmd = self_launch_command(gpus, argv)
    print("bench.py: no launcher in the environment, starting " + " ".join(cmd[1:7]) + " ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)   # stdout is inherited: rank 0's JSON line is this process's line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--inner-steps", type=int, default=0,
                    help="unsteady solve loop (solve_with_grad_GPU.py:133-197): after every N training iterations the predicted "
                         "field becomes the next time step's state (TrainStep.advance_time); 0 = steady (default; 20 with "
                         "--workload poly)")
    ap.add_argument("--workload", choices=("cylinder", "cavity", "poly"), default="cylinder",
                    help="cylinder: BASELINE configs[2] (~50k-cell tri mesh, the headline); cavity: configs[1] (use --cells 5041)")
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--meshes-per-gpu", type=int, default=1)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--graph", choices=("auto", "on", "off", "list"), default="auto",
                    help="how the step is launched: off = eager Python launches, on = hipGraph replay, list = command-list "
                         "replay (gfv/cmdlist.py); auto = time all three (reported as step_modes) and keep the fastest")
    ap.add_argument("--min-time", type=float, default=2.0,
                    help="repeat the timed K-step loop until this many seconds have been timed (0: exactly one loop)")
    ap.add_argument("--cpu-budget", type=float, default=45.0, help="seconds of CPU work for the cpu_baseline leg (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--skip-fp32-form", action="store_true",
                    help="do not time the all-fp32-MFMA form as well (profiling runs: every executed step is then the same form)")
    ap.add_argument("--skip-copy-rate", action="store_true",
                    help="do not measure the device's copy rate (profiling runs: the 1 GiB copies would be counted as step traffic)")
    ap.add_argument("--no-pin-host", action="store_true",
                    help="leave the process's threads wherever the scheduler puts them (default: gfv.host.pin_to_l3() - the loop's "
                         "host threads confined to the CPUs of one L3; the cpu_baseline leg runs un-pinned either way)")
    ap.add_argument("--skip-drop-in", action="store_true",
                    help="do not time the reference driver's call sequence on NNmodel (profiling runs)")
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="self-test only: let several ranks share one GPU (the JSON then says so in distinct_gpus)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("GFV_BENCH_SELF_LAUNCH") == "1"):
        # `python bench.py --gpus N` by itself: start the N ranks as a CHILD process group (nothing in this process has touched
        # the GPU yet - importing torch does not - and it never will: no exec, the child's exit code is ours)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    # host threads of the loop (this one, PyTorch's autograd thread, RCCL's proxies) on the CPUs of ONE L3: the host-bound legs (drop-in
    # on small meshes) run 20 - 30 % faster than with the threads placed freely over the box's 256 CPUs (gfv/host.py)
    from gfv import host as gfv_host
    # (several ranks on one node: every rank takes an L3 group of its own - on its GPU's NUMA node where sysfs tells it, else the
    # rank-th group of the host - so no two ranks share one; device PROPERTIES only are queried for that, no context is created)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.no_pin_host:
        pinned_prev = None
    elif world > 1:
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        nd = torch.cuda.device_count()
        pinned_prev = gfv_host.pin_to_l3(rank=lr, gpu_nodes=gfv_host.gpu_numa_nodes(), device=lr % nd if nd > 0 else None)
    else:
        pinned_prev = gfv_host.pin_to_l3()
    pinned_cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} under WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    ndev = torch.cuda.device_count()   # (counting devices does not initialise the GPU)
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if world > ndev and not args.allow_shared_gpu:
        raise SystemExit(f"WORLD_SIZE={world} > {ndev} visible GPUs: one rank per GPU is the contract "
                         "(--allow-shared-gpu for the single-GPU self-test)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    if world > 1 or os.environ.get("GFV_DIST_FORCE") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GFV_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; "gloo" only for the self-test
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    dist_on = backend is not None

    from gfv import lib as L
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    from FVMmodel.importer import NNmodel
    lib = L.load()

    graphs_cpu, sz = build_workload(args.workload, args.cells, args.meshes_per_gpu, rank, device)
    graphs = tuple(g.clone().to(device) for g in graphs_cpu)
    torch.manual_seed(0)  # identical initial weights on every rank (data parallel replicas)
    # dataset_size=1: the solve-script regime (solve_with_grad_GPU.py), where the online Normalizer never accumulates
    # and is the identity (utils/normalization.py:39); a single mesh has constant conditioning columns.
    model = NNmodel(default_params(dataset_size=1)).to(device)
    graph_mode = "off" if args.no_graph else args.graph
    ts = TrainStep(model, graphs, world_size=world, use_graph=False, distributed=dist_on)
    executed = [0]
    _step = ts.step

    def counted_step():
        executed[0] += 1
        return _step()
    ts.step = counted_step

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    inner = args.inner_steps if args.inner_steps > 0 else (20 if args.workload == "poly" else 0)
    since_advance = [0]
    own_times = []   # this rank's own wall time of every timed() loop

    def timed(n):
        barrier()
        tc = time.perf_counter()
        for _ in range(n):
            ts.step()
            if inner:
                since_advance[0] += 1
                if since_advance[0] >= inner:   # time advance of the unsteady solve loop: inside the timed region
                    ts.advance_time()
                    since_advance[0] = 0
        barrier()
        el = time.perf_counter() - tc
        own_times.append(el)
        if dist_on:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        ts.step()
    # the same launches in the same order, as eager launches and as one hipGraph replay: both are measured and reported;
    # the timed loops use the faster (all ranks alike: rank 0's decision is broadcast)
    modes = {}
    cal_steps = max(5, min(20, args.steps))
    flags = {"eager": False, "hip_graph": True, "cmd_list": "list"}
    want = {"auto": ("eager", "hip_graph", "cmd_list"), "off": ("eager",), "on": ("hip_graph",), "list": ("cmd_list",)}[graph_mode]
    for name in want:
        ts.use_graph = flags[name]
        for _ in range(4):
            ts.step()
        modes[name] = timed(cal_steps) / cal_steps
    order = sorted(modes, key=modes.get)
    # the command list costs the host 1.5 us per launch instead of ~18: within 1 % of the fastest mode it is the one to run
    # (8 ranks on one host share its cores; an eager step that wins by a hair on an idle host does not there)
    if "cmd_list" in modes and modes["cmd_list"] <= 1.01 * modes[order[0]]:
        order.remove("cmd_list")
        order.insert(0, "cmd_list")
    pick = torch.tensor([list(flags).index(order[0])], device=device)
    if dist_on:
        dist.broadcast(pick, src=0)
    used = list(flags)[int(pick.item())]
    ts.use_graph = flags[used]
    for _ in range(4):
        ts.step()

    reps, elapsed = [], 0.0
    own_times.clear()
    while True:
        el = timed(args.steps)           # EXACTLY K steps between barrier + synchronize on both sides
        reps.append(el)
        elapsed += el
        if elapsed >= args.min_time or len(reps) >= 200:
            break
    timed_steps = args.steps * len(reps)
    ms_per_step = 1e3 * elapsed / timed_steps
    final_loss = float(ts.loss.item())
    # every rank's own clock over the same timed loops (the line's ms_per_step is the MAX over ranks, loop by loop)
    rank_ms = [1e3 * sum(own_times) / timed_steps]
    if dist_on:
        t = torch.zeros(world, dtype=torch.float64, device=device)
        t[rank] = rank_ms[0]
        dist.all_reduce(t)
        rank_ms = [float(v) for v in t.tolist()]
    # what the gradient exchange costs the step: the same step, same launch mode, with the collectives taken out (every rank
    # then steps on its own gradient: a timing leg, so the replicas' state is put back afterwards)
    exposed_us = None
    if dist_on:
        snap = [b.clone() for b in (ts.flat_p, ts.flat_m, ts.flat_v, ts.adam_state)]
        ts.dist_on = ts.engine.dist_force = False
        for _ in range(6):
            ts.step()
        local_ms = 1e3 * timed(cal_steps) / cal_steps
        ts.dist_on = ts.engine.dist_force = True
        for dst, src in zip((ts.flat_p, ts.flat_m, ts.flat_v, ts.adam_state), snap):
            dst.copy_(src)
        for _ in range(4):
            ts.step()
        with_ms = 1e3 * timed(cal_steps) / cal_steps
        exposed_us = round(1e3 * (with_ms - local_ms), 1)

    # ---- the drop-in path: the reference driver's literal call sequence on the same batch, beside the fused TrainStep ----
    drop_in = None
    if rank == 0 and world == 1 and not args.skip_drop_in:
        drop_in = drop_in_leg(device, graphs_cpu, ms_per_step, max(5, min(20, args.steps)))

    # ---- roofline leg: same step, eager, HIP events around every launch --------------------------------------------
    # every rank runs the instrumented steps (they contain the gradient all-reduce); rank 0 reports.  The timed region
    # above runs the weight-gradient kernels on a side stream, concurrently with the dX chains; here every kernel is
    # launched on ONE stream so that its duration is its own (concurrent kernels share the CUs and stretch each other).
    roof, roof_all = None, []
    copy_rate = measured_copy_rate_gbs(device) if (rank == 0 and not args.skip_copy_rate) else 0.0
    do_roofline = args.profile_steps > 0   # (--profile-steps 0: a profiling run that executes the timed form of the step only)
    single_stream_ms = float("nan")
    if do_roofline:
        ts_use_graph, eng_overlap = ts.use_graph, ts.engine.overlap
        ts.use_graph, ts.engine.overlap = False, False
        for _ in range(2):
            ts.step()
        single_stream_ms = 1e3 * timed(5) / 5          # the un-instrumented single-stream eager step the classes must add up to
        lib.gfv_profile_reset()
        lib.gfv_profile_set_sizes(float(2 * sz["Ex"] + 2 * sz["B"]), float(sz["Sigma"]))
        lib.gfv_profile_enable(1)
        for _ in range(args.profile_steps):
            ts.step()
        torch.cuda.synchronize()
        lib.gfv_profile_enable(0)
        ts.use_graph, ts.engine.overlap = ts_use_graph, eng_overlap
    executed_flops = 0.0
    step_roof = None
    traffic_source = None
    if rank == 0 and do_roofline:
        out = (ctypes.c_double * 4)()
        # names as rocprofv3 prints them (profiles/*_kernel_stats.csv).  The chain kernels run their fp32 products as 3 f16
        # MFMAs per product group (include/gfv.h): their matrix roofline is the f16 MFMA peak / 3 in fp32-equivalent
        # flops; every kernel is priced against BOTH rooflines (algorithmic flops and algorithmic bytes over the measured
        # duration) and reported on the one it sits closer to
        h = "true" if ts.engine.f16split else "false"
        chain_peak = PEAK_F16_MFMA_TFLOPS / 3.0 if ts.engine.f16split else PEAK_F32_MFMA_TFLOPS
        # (the 7th template argument - element ops fixed at compile time, 0 / 1 / 2 - is folded into the class: `*`)
        tc = lambda lnm, rag, csr="false": f"tchain_kernel<1, {lnm}, {rag}, {h}, 4, {csr}, *>"
        spec = {7: (tc(0, "false"), chain_peak, "gnn"), 8: (tc(1, "false"), chain_peak, "gnn"),
                9: (tc(2, "false"), chain_peak, "gnn"), 10: (tc(0, "true"), chain_peak, "gnn"),
                13: (tc(0, "false", "true"), chain_peak, "gnn"),
                14: ("colchain_bwd_kernel", chain_peak, "gnn"),
                15: ("cfwd_kernel", chain_peak, "gnn"),   # column-owner small-tile forward of the 3-layer launches up to 100 k rows (csrc/cfwd.hip)
                16: ("lin1_kernel / lin1_lnbwd_kernel / lin1_csr_kernel (single-layer launches)", chain_peak, "gnn"),
                2: ("dw_multi_h_kernel" if ts.engine.f16split else "dw_multi_kernel", chain_peak, "gnn"),
                3: ("seg_gather_sum_vec", None, "gnn"),
                11: ("reduce_multi_kernel / reduce_partials_*", None, "gnn"),
                4: ("slice_* / deslice (Transolver slice attention)", None, "slice"),
                5: ("wlsq_* / face_* / cell_* / node_bwd / graph_loss (finite volume)", None, "fvm"),
                12: ("wimg / wabsmax / transpose_batch (per-step weight images)", None, "misc"),
                6: ("input preparation, train_loss, adam", None, "misc")}
        class_ms = {}
        for kind, (kname, mfma_peak, cls) in spec.items():
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
            class_ms[cls] = class_ms.get(cls, 0.0) + ms / args.profile_steps
            roof_all.append({"kernel": kname, "class": cls, "bound": bound, "achieved": round(ach, 3), "peak": round(peak, 1),
                             "unit": unit, "frac": round(ach / peak, 4), "traffic": None,
                             "launches_per_step": n / args.profile_steps,
                             "avg_launch_us": round(1e3 * ms / n, 2), "ms_per_step": round(ms / args.profile_steps, 4),
                             "fp32_equiv_tflops": round(tf, 3) if mfma_peak else None, "algorithmic_gbs": round(gbs, 1),
                             "frac_mfma": round(f_mfma, 4), "frac_hbm": round(f_hbm, 4),
                             "algorithmic_bytes_per_launch": round(by / n), "frac_isolated": round(ach / peak, 4)})
            if copy_rate:
                # against what THIS box's memory side delivers (a 1 GiB device-to-device copy in this run: 4.8 - 5.2 TB/s = 0.60 - 0.65 of
                # the 8 TB/s spec on this pool): the fraction a bandwidth-bound kernel can be held to
                roof_all[-1]["frac_of_copy_rate"] = round(gbs / copy_rate, 4)
            if kind == 3:   # gathers: priced by DISTINCT source rows; what L2 / Infinity Cache serve beside it
                roof_all[-1]["l2_side_gbs"] = round(fl / (ms * 1e-3) / 1e9, 1)
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        traffic_source, step_traffic = None, None
        # the committed PMC figures were collected on the default workload (one 50 k-cell mesh per GPU)
        if os.path.exists(pmc) and args.meshes_per_gpu == 1 and args.cells == 50000 and args.workload == "cylinder":
            traffic = json.load(open(pmc))
            traffic_source = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this "
                              "command (profiles/collect.sh), bytes per launch; NOT measured in this run")
            for r in roof_all:
                r["traffic"] = traffic.get(r["kernel"])
            step_traffic = traffic.get("__step_total__")
        # the same kernels INSIDE the timed step (two streams: the weight gradients run beside the dX chains and stretch them):
        # average durations of a rocprofv3 --kernel-trace --stats run of this command (profiles/collect.sh -> the committed
        # profiles/kernel_stats_in_step.json), against the same algorithmic bytes / flops per launch
        instep_path = os.path.join(ROOT, "profiles", "kernel_stats_in_step.json")
        instep_note, dominant = None, None
        if os.path.exists(instep_path) and args.meshes_per_gpu == 1 and args.cells == 50000 and args.workload == "cylinder":
            instep = json.load(open(instep_path))
            built_from = instep.get("__lib_srchash__")
            here = None
            try:
                here = open(L.LIB_PATH + ".srchash").read().strip()
            except OSError:
                pass
            if built_from is None or built_from != here:
                # the committed averages were taken from another build of the kernels: not this run's, not quoted
                instep_note = "profiles/kernel_stats_in_step.json belongs to another build of libgfv.so (source hash differs or absent): ignored"
            else:
                instep_note = ("profiles/kernel_stats_in_step.json: rocprofv3 --kernel-trace --stats of `bench.py --graph list "
                               "--skip-fp32-form --profile-steps 0 --cpu-budget 0` (profiles/collect.sh), same library build; "
                               "NOT measured in this run")
                for r in roof_all:
                    us = instep.get(r["kernel"])
                    if us:
                        r["avg_launch_us_in_step"] = round(us, 2)
                        r["frac_in_step"] = round(r["frac_isolated"] * r["avg_launch_us"] / us, 4)
                        r["ms_per_step_in_step"] = round(us * r["launches_per_step"] / 1e3, 4)
                timed_in = [r for r in roof_all if r.get("ms_per_step_in_step")]
                if timed_in:
                    d = max(timed_in, key=lambda r: r["ms_per_step_in_step"])
                    dominant = {"kernel": d["kernel"], "ms_per_step_in_step": d["ms_per_step_in_step"],
                                "avg_launch_us_in_step": d["avg_launch_us_in_step"], "frac_in_step": d["frac_in_step"],
                                "bound": d["bound"], "note": "the class with the largest summed duration inside the timed two-stream "
                                "step (its queue may differ from the main one: weight gradients run on the side queue)"}
        if roof_all:
            roof = max(roof_all, key=lambda r: r["ms_per_step"])
        comp = compulsory_step_bytes(sz)
        priced_ms = sum(class_ms.values())
        step_roof = {
            "compulsory_bytes_per_step": {k: round(v) for k, v in comp.items()} | {"total": round(sum(comp.values()))},
            "frac_compulsory": round(sum(comp.values()) / (ms_per_step * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
            "frac_compulsory_by_class": {k: round(comp[k] / (class_ms[k] * 1e-3) / (PEAK_HBM_GBS * 1e9), 4)
                                         for k in comp if class_ms.get(k)},
            "class_ms_per_step_single_stream": {k: round(v, 4) for k, v in class_ms.items()},
            "priced_ms_per_step": round(priced_ms, 4), "single_stream_eager_ms_per_step": round(single_stream_ms, 4),
            "priced_share_of_single_stream_step": round(priced_ms / single_stream_ms, 4),
            "priced_launch_records_per_step": sum(r["launches_per_step"] for r in roof_all),
            "pmc_traffic_bytes_per_step": step_traffic, "pmc_traffic_source": traffic_source,
            "convention": "SURVEY.md 8(d): distinct inputs read once, outputs written once, int32 indices; step = 3 x forward",
        }

    lib.gfv_profile_reset()
    if dist_on:
        dist.barrier()

    # ---- the same step with every GEMM product on the fp32 MFMA (the form without the fp16 split), for the record ----
    fp32_form = None
    if ts.engine.f16split and not args.skip_fp32_form:
        ts_use_graph = ts.use_graph
        ts.engine.f16split = False      # (same launch mode as `value`: a form change re-records the list, Engine.capture_signature)
        lib.gfv_set_f16split(0)
        for _ in range(5):
            ts.step()
        nf = max(5, min(20, args.steps))
        el = timed(nf)
        lib.gfv_set_f16split(1)
        ts.use_graph, ts.engine.f16split = ts_use_graph, True
        fp32_form = {"value": round(world * args.meshes_per_gpu * nf / el, 3), "ms_per_step": round(1e3 * el / nf, 4),
                     "steps": nf, "note": f"launch mode {used}, all products on v_mfma_f32_16x16x4_f32 (GFV_F16SPLIT=0)"}

    # ---- and with ONE fp16 x fp16 product per term (the reduced-precision form, include/gfv.h gfv_set_f16split(2)): the
    # counterpart of the reference's autocast runs (BASELINE configs 3 / 5), reported beside the headline, never as it ----
    f16_form = None
    if ts.engine.f16split and not args.skip_fp32_form:
        ts_use_graph = ts.use_graph
        lib.gfv_set_f16split(2)
        for _ in range(5):
            ts.step()
        nf = max(5, min(20, args.steps))
        el = timed(nf)
        lib.gfv_set_f16split(1)
        ts.use_graph = ts_use_graph
        f16_form = {"value": round(world * args.meshes_per_gpu * nf / el, 3), "ms_per_step": round(1e3 * el / nf, 4),
                    "steps": nf, "note": f"launch mode {used}, single fp16 x fp16 products with fp32 accumulation (GFV_F16SPLIT=2): "
                                         "NOT the form `value` is measured on.  Stated tolerances against the fp32 oracle "
                                         "(tests/golden/cases.py LOWP_TOL; asserted by tests/test_model_gpu.py and, on this "
                                         "mesh, tests/test_fullsize_gpu.py): fields 2e-4, residual losses 2e-3, log-loss 1e-5, "
                                         "gradients norm-wise 2e-3; measured on this mesh 4.1e-5 / 1.0e-4 / 1.6e-6 / 1.8e-3 "
                                         "(profiles/r03_parity_fp64.txt)",
                    "tolerance_vs_fp32_oracle": {"field": 2e-4, "losses": 2e-3, "logloss": 1e-5, "grad_norm": 2e-3}}

    # ---- and on bf16 operands (gfv_set_f16split(3): v_mfma_f32_16x16x32_bf16 - BASELINE config 3's "bf16 MLP GEMMs on MFMA"
    # to the letter; same kernels, bf16-rounded operands, fp32 accumulation and fp32 everywhere else) ----
    bf16_form = None
    if ts.engine.f16split and not args.skip_fp32_form:
        ts_use_graph = ts.use_graph
        lib.gfv_set_f16split(3)
        for _ in range(5):
            ts.step()
        nf = max(5, min(20, args.steps))
        el = timed(nf)
        lib.gfv_set_f16split(1)
        ts.use_graph = ts_use_graph
        for _ in range(5):
            ts.step()   # (back in the default form: the images are rebuilt, the list re-recorded)
        bf16_form = {"value": round(world * args.meshes_per_gpu * nf / el, 3), "ms_per_step": round(1e3 * el / nf, 4),
                     "steps": nf, "note": f"launch mode {used}, single bf16 x bf16 products with fp32 accumulation (GFV_F16SPLIT=3, "
                                          "v_mfma_f32_16x16x32_bf16): NOT the form `value` is measured on.  Stated tolerances against "
                                          "the fp32 oracle (tests/golden/cases.py BF16_TOL; asserted by tests/test_model_gpu.py; what "
                                          "the form computes is pinned per kernel family by tests/test_bf16_form_gpu.py)",
                     "tolerance_vs_fp32_oracle": {"field": 2e-3, "losses": 2e-2, "logloss": 1e-4, "grad_norm": 2e-2}}

    cpu = None
    gfv_host.restore(pinned_prev)   # the CPU baseline's intra-op threads (created now) go wherever the scheduler puts them
    if rank == 0 and world == 1 and args.cpu_budget > 0:
        # 16 threads is the fastest setting for this launch-bound eager workload on the GPU box's host
        # (measured 8/16/32/64/256 threads: 4.3 / 3.5 / 3.6 / 5.2 / 188 s per step); override with GFV_CPU_THREADS
        total = os.cpu_count() or 1
        ncores = int(os.environ.get("GFV_CPU_THREADS", min(16, total)))
        torch.set_num_threads(ncores)
        sec, nst = cpu_baseline(graphs_cpu, 0.6 * args.cpu_budget, max_steps=6, min_steps=5)   # >= 5 timed steps
        cpu = {"value": round(args.meshes_per_gpu / sec, 5), "unit": "train-iters/s", "cores": torch.get_num_threads(),
               "kind": "port", "cpu_model": cpu_model(), "host_logical_cpus": total,
               "sample": f"{nst} timed steps (after 1 warm-up) of the same {sz['C']}-cell mesh batch, "
               f"oracle/fvgn_oracle.py fp32 eager PyTorch, median {sec:.3f} s/step at {ncores} threads (the fastest of "
               "8 / 16 / 32 / 64 / 256 on this host)"}
        # the reference's own setting: torch.set_num_threads(os.cpu_count() // 2) (pre_train_Adam.py:38), beside the best one
        half = max(1, total // 2)
        if half != ncores and "GFV_CPU_THREADS" not in os.environ:
            torch.set_num_threads(half)
            sec2, nst2 = cpu_baseline(graphs_cpu, 0.4 * args.cpu_budget, max_steps=5, min_steps=1)
            cpu["at_half_the_cpus_indicative"] = {
                "value": round(args.meshes_per_gpu / sec2, 5), "cores": half, "timed_steps": nst2, "median_s_per_step": round(sec2, 3),
                "note": "torch.set_num_threads(os.cpu_count() // 2) as the reference sets it (pre_train_Adam.py:38).  INDICATIVE "
                        "only: at ~10 s per step the budget of the default run allows as few as ONE timed step here, fewer than the "
                        "five SURVEY.md 8(d) asks for - no ratio is quoted on it; `value` above (>= 5 timed steps at the fastest "
                        "thread count) is the baseline"}
            torch.set_num_threads(ncores)

    torch.cuda.synchronize()
    sf = ctypes.c_int32(0)
    lib.gfv_status_flags(ctypes.byref(sf))
    status_flags = int(sf.value) | int(L.status_mirror()[0])
    if rank == 0:
        total_meshes = world * args.meshes_per_gpu
        value = total_meshes * timed_steps / elapsed
        wl = {"cylinder": "cylinder_flow tri mesh", "cavity": "lid_driven_cavity quad mesh",
              "poly": f"cylinder_flow_poly polygon mesh (reference example, cells of 3 ... 9 nodes), unsteady: time advance "
                      f"every {inner} iterations"}[args.workload] + \
            ", TransFVGN_v2 (hidden 128, mp 3), 2nd-order WLSQ, conserved form"
        line = {
            "metric": f"training iters/sec, {sz['C'] // 1000}k-cell {args.workload} mesh (fwd + loss + bwd + Adam, batch resident in HBM)",
            "value": round(value, 3), "unit": "mesh-train-iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if args.workload != "poly" else "reference example mesh (tests/golden/poly_cylinder.npz), synthetic fields",
            "timed": {"loops": len(reps), "steps_per_loop": args.steps, "timed_steps": timed_steps,
                      "timed_seconds": round(elapsed, 4), "min_time": args.min_time,
                      "loop_ms_per_step_min_max": [round(1e3 * min(reps) / args.steps, 4), round(1e3 * max(reps) / args.steps, 4)]},
            "step_modes": {k + "_ms_per_step": round(1e3 * v, 4) for k, v in modes.items()} | {"used": used},
            "dtype_note": ("fp32 values end to end; the products of the fused GEMM chains run as 3 f16 MFMAs on exact (hi, lo) "
                           "fp16 splits of the fp32 operands with fp32 accumulation (error <= the f32 MFMA's, parity tests at 1e-5)"
                           if ts.engine.f16split else "fp32 MFMA"),
            "parity_note": ("fields, residual losses and the scalar loss are within 1e-5 of the float64 value of the reference's "
                            "algorithm and of exactly pooled fp32 residuals; against the REFERENCE'S OWN fp32 outputs the pooled "
                            "residual losses are held to 1e-4, not 1e-5: its sequential fp32 index_add_ pooling is itself 3e-5 ... 1e-4 "
                            "from the exact sum of its own terms (DESIGN.md 2).  The reduced-precision form reported beside the "
                            "headline are f16_products_form (single fp16 x fp16 products) and bf16_products_form (single bf16 x bf16 "
                            "products on v_mfma_f32_16x16x32_bf16: what BASELINE config 3 names); neither is the form of `value`"),
            "config": {"workload": wl, "cells": sz["C"], "nodes": sz["N"], "faces": sz["E"],
                       "meshes_per_gpu": args.meshes_per_gpu, "global_batch": total_meshes, "parallelism": f"dp{world}",
                       "hip_graph": ts.use_graph is True, "launch_mode": used, "final_loss": round(final_loss, 6)},
            "rccl_ranks": (dist.get_world_size() if dist_on else 0), "dist_backend": (dist.get_backend() if dist_on else None),
            "distinct_gpus": min(world, ndev),
            "rank_ms_per_step": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4),
                                 "note": "each rank's own clock over the timed loops; ms_per_step is the max over ranks per loop"},
            "allreduce_exposed_us": exposed_us,   # step with the gradient exchange minus the same step without it (same run, same mode)
            "roofline": ({k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
                         | {"kernel": roof["kernel"], "traffic_source": traffic_source,
                            "frac_isolated": roof["frac_isolated"], "frac_in_step": roof.get("frac_in_step"),
                            "frac_note": "frac = frac_isolated: HIP events around the launches of a single-stream eager step in this "
                                         "run; frac_in_step: the same bytes over the kernel's average duration inside the two-stream "
                                         "step (rocprofv3 run of this command, profiles/kernel_stats_in_step.json)",
                            "kernel_in_step_dominant": dominant, "in_step_source": instep_note,
                            "hbm_copy_rate_measured_gbs": round(copy_rate, 1) if copy_rate else None,
                            "hbm_copy_rate_note": "1 GiB device-to-device copy (read + write bytes / time), HIP events, THIS run",
                            "hbm_rates_committed_gbs": {"spec": PEAK_HBM_GBS, "copy (guide)": 6290.0,
                                                        "2 reads + 5 writes, runs >= 128 B": 5800.0,
                                                        "2 reads + 5 writes, 64-B runs": 4200.0,
                                                        "source": "constants, NOT measured in this run: MI355X_MICROARCH.md (copy); "
                                                                  "profiles/r03_stream_run.txt (mixes, profiles/tools/stream)"}})
                        if roof else None,
            "roofline_step": step_roof,
            "roofline_kernels": roof_all,
            "fp32_mfma_form": fp32_form,
            "f16_products_form": f16_form,
            "bf16_products_form": bf16_form,
            # reference algorithm vs what the launches execute (EdgeBlock first layer factored through
            # the nodes, gfv/engine.py): the fraction of the fp32 MFMA peak is quoted on the EXECUTED flops
            "algorithmic_step_tflops": round(algorithmic_step_flops(sz) / 1e12, 4),
            "executed_step_tflops": round(executed_flops / 1e12, 4),
            "step_mfma_frac": round(executed_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "roofline_note": ("per-kernel durations: HIP events around every launch, ONE stream, eager (profile leg); value: "
                              + {"eager": "eager launches from Python", "hip_graph": "hipGraph replay",
                                 "cmd_list": "command-list replay of the eager launch sequence (gfv/cmdlist.py)"}[used]
                              + (" with the weight gradients on a side stream" if ts.engine.overlap else "")),
            "cpu_baseline": cpu,
            "drop_in": drop_in,
            "host_threads": {"pinned_to_one_l3": pinned_prev is not None, "cpus": pinned_cpus,
                             "note": "gfv.host.pin_to_l3() at start-up (--no-pin-host: off); the cpu_baseline leg runs un-pinned"},
            "status_flags": status_flags,   # device status word at the end of the run (0: no kernel left its fp16 window / range)
            "steps_executed": executed[0],   # every training step this process ran (all legs): profiles divide by it
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
