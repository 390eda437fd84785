"""Worker of tests/test_operators_gpu.py::test_data_parallel_trainstep_two_ranks (launched with torch.distributed.run,
2 ranks sharing GPU 0, gloo) and of tests/test_rccl_gpu.py (N ranks, one GPU each, RCCL): sharded TrainStep; prints whether the
ranks hold identical parameters / Normalizer buffers and the deviation from a single-process run over the global batch.
GFV_TEST_MODE=eager (default): an ACCUMULATING Normalizer (dataset_size 5) for 3 steps - every step exchanges the statistics
inside the forward and runs eager.  GFV_TEST_MODE=list: dataset_size 2, 7 steps with use_graph="list" - two accumulating
steps, two warm-up steps, the recording and two REPLAYS of a command list that carries the early gradient bucket (the fork to
the communication stream and its all-reduce) inside it."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)


def main():
    backend = os.environ.get("GFV_TEST_BACKEND", "gloo")
    rank = int(os.environ["RANK"])
    # gloo: both ranks share GPU 0 (the collectives go through host memory).  nccl (= RCCL): one GPU per rank when the box has
    # them; on a one-GPU box both ranks name GPU 0 and RCCL itself decides whether it takes that (it refuses duplicate devices)
    dev = rank % max(torch.cuda.device_count(), 1) if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    try:
        if backend == "nccl":
            import datetime
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=datetime.timedelta(seconds=90))
            probe = torch.ones(4, device="cuda")
            dist.all_reduce(probe)      # communicator creation happens here
            torch.cuda.synchronize()
        else:
            dist.init_process_group("gloo")
    except Exception as exc:   # noqa: BLE001 - reported to the parent test, which decides
        print(f"DPUNSUPPORTED rank={rank} devices={torch.cuda.device_count()} {type(exc).__name__}: {str(exc)[:600]}", flush=True)
        os._exit(0)
    world = dist.get_world_size()
    import cases
    from oracle import fvgn_oracle as O
    from FVMmodel.importer import NNmodel
    from gfv import meshgen
    from gfv.graph import build_batch
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    specs = cases.CASES["cyl_b3"]   # three meshes: rank r trains on mesh r % 3
    meshes = [meshgen.finish_mesh(getattr(meshgen, fac)(**kw), U=U) for fac, kw, U, _ in specs]
    fields = [meshgen.random_fields(m, seed=s[3]) for m, s in zip(meshes, specs)]
    P = O.init_parameters(cases.WEIGHT_SEED)

    mode = os.environ.get("GFV_TEST_MODE", "eager")
    nsteps, dsize, use_graph = (7, 2, "list") if mode == "list" else (3, 5, False)
    nm = len(meshes)

    def make(ids, ws):
        model = NNmodel(default_params(dataset_size=dsize))
        sd = model.state_dict()
        for k, v in P.items():
            sd[k].copy_(v)
        model.load_state_dict(sd)
        model = model.cuda()
        g = build_batch([meshes[i] for i in ids], [fields[i] for i in ids], device="cuda")
        return model, TrainStep(model, g, world_size=ws, use_graph=use_graph)

    model, ts = make([rank % nm], world)
    for _ in range(nsteps):
        ts.step()
    torch.cuda.synchronize()
    # the gradient exchange ran in two buckets: the upper one (last processor + decoder) was started from inside the backward
    split = ts._bucket_split()
    assert 0 < split < ts.flat_g.numel() and ts._comm is not None and not ts._early, (split, ts._comm, ts._early)
    if mode == "list":
        (cl, _, early), = ts._graphs.values()
        assert early and sum(1 for c in cl.cmds if c[2] is not None and c[2] == ts._comm) == 1, "the early bucket is not part of the command list"
    mine = torch.cat([ts.P[k].reshape(-1) for k in ts.P] + [model.node_norm.acc_sum.reshape(-1),
                                                          model.node_norm.acc_sum_squared.reshape(-1),
                                                          model.node_norm.acc_count.reshape(-1)]).cpu()
    if backend == "nccl":
        mine_d = mine.cuda()
        gathered = [torch.zeros_like(mine_d) for _ in range(world)]
        dist.all_gather(gathered, mine_d)
        gathered = [t.cpu() for t in gathered]
    else:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        ref_model, ref_ts = make([r % nm for r in range(world)], 1)
        for _ in range(nsteps):
            ref_ts.step()
        torch.cuda.synchronize()
        ref = torch.cat([ref_ts.P[k].reshape(-1) for k in ref_ts.P] + [ref_model.node_norm.acc_sum.reshape(-1),
                                                                     ref_model.node_norm.acc_sum_squared.reshape(-1),
                                                                     ref_model.node_norm.acc_count.reshape(-1)]).cpu()
        n = sum(v.numel() for v in ts.P.values())
        perr = float((mine[:n] - ref[:n]).abs().max() / ref[:n].abs().max())
        nerr = float((mine[n:] - ref[n:]).abs().max() / ref[n:].abs().max())
        print(f"DPRESULT same={int(same)} param_err={perr:.3e} norm_err={nerr:.3e} backend={dist.get_backend()} "
              f"gpus={torch.cuda.device_count()} world={world} mode={mode}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
