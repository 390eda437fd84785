"""Where does a data-parallel step lose time against the single-process step?  One rank, backend nccl (= RCCL), command-list
mode: host time and device time of replay / all-reduce / Adam, per step.  Run under torch.distributed.run --nproc-per-node 1."""
import os, sys, time
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd")]
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import bench
from gfv.params import default_params
from gfv.trainer import TrainStep
from FVMmodel.importer import NNmodel
graphs_cpu, sz = bench.build_workload("cylinder", 50000, 1, 0, "cuda")
graphs = tuple(g.clone().to("cuda") for g in graphs_cpu)
torch.manual_seed(0)
model = NNmodel(default_params(dataset_size=1)).cuda()
for distributed in (False, True):
    ts = TrainStep(model, graphs, world_size=1, use_graph="list", distributed=distributed)
    for _ in range(8):
        ts.step()
    torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"distributed={distributed}: {1e3 * t_all / n:.3f} ms/step, host issue {1e3 * t_issue / n:.3f} ms/step")
    if distributed:
        # pieces, host side
        cl = next(v for k, v in ts._graphs.items() if k[0] == "list")[0]
        hs = {"replay": 0.0, "allreduce": 0.0, "adam": 0.0}
        torch.cuda.synchronize()
        for _ in range(n):
            a = time.perf_counter(); cl.replay(); b = time.perf_counter(); ts._allreduce(); c = time.perf_counter(); ts._adam(); d = time.perf_counter()
            hs["replay"] += b - a; hs["allreduce"] += c - b; hs["adam"] += d - c
        torch.cuda.synchronize()
        print("  host ms/step:", {k: round(1e3 * v / n, 3) for k, v in hs.items()})
        # device time of the all-reduce alone
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            dist.all_reduce(ts.flat_g)
        e1.record(); torch.cuda.synchronize()
        print(f"  20 back-to-back all-reduces of {ts.flat_g.numel() * 4 / 1e6:.1f} MB: {e0.elapsed_time(e1) / 20:.3f} ms each (device)")
dist.destroy_process_group()
