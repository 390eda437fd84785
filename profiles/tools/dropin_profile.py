#!/usr/bin/env python3
"""Where the host time of the drop-in training iteration goes (round 6): the reference driver's call sequence on NNmodel under
cProfile, on the 5 k-cell cavity (where the iteration is host-bound) - `python profiles/tools/dropin_profile.py [cells] [gfv|torch]`."""
import cProfile
import os
os.environ.setdefault("GFV_DROPIN_TIMING", "1")
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd")]
from gfv import host

if os.environ.get("GFV_PROFILE_PIN", "1") == "1":   # the loop's two host threads on one L3, as bench.py runs it (gfv/host.py)
    host.pin_to_l3()
import torch

import bench
from FVMmodel.importer import NNmodel
from gfv.optim import Adam as GfvAdam
from gfv.params import default_params

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 5041
which = sys.argv[2] if len(sys.argv) > 2 else "gfv"
dev = torch.device("cuda", 0)
graphs_cpu, sz = bench.build_workload("cavity" if cells < 20000 else "cylinder", cells, 1, 0, dev)
params = default_params(dataset_size=1)
model = NNmodel(params).to(dev)
graphs = tuple(g.clone().to(dev) for g in graphs_cpu)
gn = graphs[0]
backup = gn.x.clone()
opt = (GfvAdam if which == "gfv" else torch.optim.Adam)(model.parameters(), lr=params.lr)


def it():
    gn.x.copy_(backup)
    gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
    opt.zero_grad()
    lc, lx, ly, lp, un, uc = model(*graphs)
    loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lx + params.loss_mom * ly))
    loss.backward()
    opt.step()


for _ in range(10):
    it()
torch.cuda.synchronize()
for per_sync in (200, 20, 5, 20, 200):   # iterations between two device synchronisations
    t0 = time.perf_counter()
    for k in range(200):
        it()
        if (k + 1) % per_sync == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(f"{cells} cells, {which} Adam, host threads on {len(os.sched_getaffinity(0))} CPUs: {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per iteration "
          f"(un-profiled, a device synchronisation every {per_sync} iterations)")
from gfv import functions as GF
if GF.TIMING:
    T = GF.TIMING
    n = max(T["bwd_calls"], 1)
    print("autograd node, host ms per call: forward %.3f (list replay %.3f) | backward %.3f (list replay %.3f, gradient views %.3f)"
          % (1e3 * T["fwd_total"] / max(T["fwd_calls"], 1), 1e3 * T["fwd_replay"] / max(T["fwd_calls"], 1), 1e3 * T["bwd_total"] / n,
             1e3 * T["bwd_replay"] / n, 1e3 * T["bwd_views"] / n))
# sections of the iteration, host wall time with a device synchronisation behind each (so that a section's launches are charged to it)
import collections
sec = collections.OrderedDict()


def timed_it():
    def mark(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sec[name] = sec.get(name, 0.0) + (t1 - t0)
        return t1
    t = time.perf_counter()
    gn.x.copy_(backup)
    gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
    opt.zero_grad()
    t = mark("restore + zero_grad", t)
    lc, lx, ly, lp, un, uc = model(*graphs)
    t = mark("model(...)", t)
    loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lx + params.loss_mom * ly))
    t = mark("loss", t)
    loss.backward()
    t = mark("loss.backward()", t)
    opt.step()
    t = mark("optimizer.step()", t)


for _ in range(100):
    timed_it()
print("sections, ms per iteration WITH a device synchronisation behind each: " + ", ".join(f"{k} {1e3 * v / 100:.3f}" for k, v in sec.items()))
# host time only: the same loop WITHOUT waiting for the device at the end of each iteration
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    it()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
