"""Host cost of replaying the step's launch list before and after an RCCL process group exists in the process."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd")]
torch.cuda.set_device(0)
if os.environ.get("EARLY") and "RANK" in os.environ:
    import torch.distributed as dist
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    if os.environ["EARLY"] == "2":
        t = torch.zeros(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
import bench
from gfv.params import default_params
from gfv.trainer import TrainStep
from FVMmodel.importer import NNmodel
graphs_cpu, sz = bench.build_workload(os.environ.get("LC_WORKLOAD", "cylinder"), int(os.environ.get("LC_CELLS", "50000")), 1, 0, "cuda")
graphs = tuple(g.clone().to("cuda") for g in graphs_cpu)
torch.manual_seed(0)
model = NNmodel(default_params(dataset_size=1)).cuda()
ts = TrainStep(model, graphs, use_graph="list")
for _ in range(8):
    ts.step()
def measure(tag, n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ts.step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{tag}: {1e3 * t_all / n:.3f} ms/step, host issue {1e3 * t_issue / n:.3f} ms/step", flush=True)
print("env:", {k: v for k, v in os.environ.items() if k.startswith(("OMP", "HSA", "HIP", "NCCL", "RCCL", "TORCH", "GPU"))})
measure("no process group" if not os.environ.get("EARLY") else "nccl group created first (EARLY=%s)" % os.environ["EARLY"])
if "RANK" in os.environ and not os.environ.get("EARLY"):
    import torch.distributed as dist
    dist.init_process_group("gloo")
    measure("gloo group")
    dist.destroy_process_group()
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    measure("nccl group (no collective yet)")
    t = torch.zeros(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    measure("nccl group (after a collective)")
    dist.destroy_process_group()
    measure("after destroy")
