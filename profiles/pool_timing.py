"""Row f1 measurement: dataset-style training (a different mesh every step) on the 50 k-cell bench meshes.

  rebuilt : what the reference's loop does per step (pre_train_Adam.py:146-156): the batch's five graph objects are
            copied host -> device and the per-batch tables are rebuilt (here: gfv.plan.build_plan), then the step runs
  pooled  : gfv.pool.DevicePool - meshes, per-mesh plans and fields stay in HBM, the batch is assembled by one launch
Both run the same eager (un-captured) HIP training step.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gen-fvgn-steady_amd"))
import torch  # noqa: E402

from bench import build_workload  # noqa: E402
from FVMmodel.importer import NNmodel  # noqa: E402
from gfv import meshgen  # noqa: E402
from gfv.graph import build_batch  # noqa: E402
from gfv.params import default_params  # noqa: E402
from gfv.pool import DevicePool  # noqa: E402
from gfv.trainer import TrainStep  # noqa: E402

K, STEPS = 4, 24
nx, ny = meshgen.cylinder_grid_for_cells(50000)
meshes, fields = [], []
for i in range(K):
    m = meshgen.finish_mesh(meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, jitter=0.2, seed=1234 + i))
    meshes.append(m)
    fields.append(meshgen.random_fields(m, seed=1 + i))
cpu_batches = [build_batch([meshes[i]], [fields[i]], device="cpu") for i in range(K)]
for g in cpu_batches:
    for d in g:
        for k, v in list(d.__dict__.items()):
            if torch.is_tensor(v):
                setattr(d, k, v.pin_memory())
torch.manual_seed(0)
model = NNmodel(default_params(dataset_size=1)).cuda()
pool = DevicePool(meshes, fields)
g0, _ = pool.batch([0])
ts = TrainStep(model, g0, use_graph=False)


def h2d(graphs):
    """pinned host tensors -> device, no host-side copy (the cheapest form of the reference's per-step transfer)"""
    from gfv.graph import Data
    return tuple(Data(**{k: (v.to("cuda", non_blocking=True) if torch.is_tensor(v) else v) for k, v in d.__dict__.items()})
                 for d in graphs)


def run(mode):
    def one(i):
        if mode == "pooled":
            graphs, _ = pool.batch([i % K])
        else:
            graphs = h2d(cpu_batches[i % K])
        ts.set_batch(graphs)
        ts.step()
    for i in range(4):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        one(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / STEPS


def assemble_only(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        if mode == "pooled":
            pool.batch([i % K])
        else:
            from gfv.plan import build_plan
            build_plan(*h2d(cpu_batches[i % K]))
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / STEPS


res = {"workload": "4 x 50 020-cell cylinder meshes, one mesh per step, eager step", "steps": STEPS,
       "ms_per_step_rebuilt": round(run("rebuilt"), 3), "ms_per_step_pooled": round(run("pooled"), 3),
       "ms_batch_rebuilt (H2D + build_plan)": round(assemble_only("rebuilt"), 3),
       "ms_batch_pooled (one launch)": round(assemble_only("pooled"), 3)}
print(json.dumps(res))
