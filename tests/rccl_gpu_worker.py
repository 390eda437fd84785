"""Worker of tests/test_rccl_gpu.py (launched with torch.distributed.run --nproc-per-node 1, backend nccl = RCCL):
TrainStep(distributed=True) on a one-rank RCCL group.  Every collective of the data-parallel step executes under the
backend it was written for - the early gradient bucket on the communication stream from inside the backward
(`_bucket_ready`), the late bucket + join (`_allreduce`), the Normalizer statistics exchange inside the forward
(`allreduce_normalizer`, accumulating Normalizer) - and, a one-rank all-reduce being the identity, the parameters,
Adam moments and Normalizer buffers must be BIT-IDENTICAL to the non-distributed TrainStep on the same batch.
Eager, hipGraph replay (the exchange stays outside the graph) and command-list replay (the early bucket is part of the list) are
checked."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)


NSTEPS = 7   # two accumulating steps (always eager), two warm-up steps, the recording, two replays of the command list


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    import cases
    from oracle import fvgn_oracle as O
    from FVMmodel.importer import NNmodel
    from gfv import meshgen
    from gfv.graph import build_batch
    from gfv.params import default_params
    from gfv.trainer import TrainStep
    specs = cases.CASES["cyl_b3"][:2]
    meshes = [meshgen.finish_mesh(getattr(meshgen, fac)(**kw), U=U) for fac, kw, U, _ in specs]
    fields = [meshgen.random_fields(m, seed=s[3]) for m, s in zip(meshes, specs)]
    P = O.init_parameters(cases.WEIGHT_SEED)

    def run(distributed, use_graph):
        model = NNmodel(default_params(dataset_size=3))   # accumulates on the first two steps, then frozen
        sd = model.state_dict()
        for k, v in P.items():
            sd[k].copy_(v)
        model.load_state_dict(sd)
        model = model.cuda()
        g = build_batch(meshes, fields, device="cuda")
        ts = TrainStep(model, g, world_size=1, use_graph=use_graph, distributed=distributed)
        calls = {"bucket": 0, "allreduce": 0}
        if distributed:
            b0, a0 = ts._bucket_ready, ts._allreduce

            def bucket():
                calls["bucket"] += 1
                return b0()

            def allred(early=None):
                calls["allreduce"] += 1
                # the early bucket must have gone out on the communication stream when the backward returns - in eager steps
                # (issued by the hook) and in command-list steps (recorded with the list, re-issued by every replay)
                calls["early_pending"] = calls.get("early_pending", 0) + (1 if (ts._early if early is None else early) else 0)
                return a0(early)
            ts._bucket_ready, ts._allreduce = bucket, allred
        for _ in range(NSTEPS):
            ts.step()
        torch.cuda.synchronize()
        nn_ = model.node_norm
        ns = ts.named_state()   # (named tensors only: the alignment padding of the flat buffers is not state)
        parts = dict(p=torch.cat([v[0].reshape(-1) for v in ns.values()]), m=torch.cat([v[1].reshape(-1) for v in ns.values()]),
                     v=torch.cat([v[2].reshape(-1) for v in ns.values()]), t=ts.adam_state[0:1], acc_sum=nn_.acc_sum.reshape(-1),
                     acc_sq=nn_.acc_sum_squared.reshape(-1), acc_count=nn_.acc_count.reshape(-1), loss=ts.loss.reshape(-1))
        state = {k: v.detach().cpu().clone() for k, v in parts.items()}
        return state, calls, ts

    ok = True
    for use_graph in (False, True, "list"):
        ref, _, _ = run(False, use_graph)
        got, calls, ts = run(True, use_graph)
        diff = {k: float((ref[k].double() - got[k].double()).abs().max()) for k in ref if not torch.equal(ref[k], got[k])}
        same = not diff
        if diff:
            print("RCCLDIFF", use_graph, diff)
        ok = ok and same and calls["allreduce"] == NSTEPS
        if use_graph is False:
            # eager: every step started the upper bucket from inside the backward, on the communication stream
            ok = ok and calls["bucket"] == NSTEPS and calls.get("early_pending", 0) == NSTEPS and ts._comm is not None
        if use_graph == "list":
            # command list: the two accumulating steps run eager (2 hook calls), two warm-up steps (2), the recording (1): the
            # replays carry the early bucket INSIDE the list (no hook call), and every step's late exchange joins an early one
            ok = ok and calls.get("early_pending", 0) == NSTEPS and calls["bucket"] == 5 and ts._comm is not None
            n_comm = sum(1 for c in next(iter(ts._graphs.values()))[0].cmds if c[2] is not None and c[2] == ts._comm)
            ok = ok and n_comm == 1
            print("RCCLLIST recorded_on_comm_stream", n_comm)
        print(f"RCCLRESULT graph={ {False: 0, True: 1, 'list': 2}[use_graph] } same={int(same)} bucket={calls['bucket']} allreduce={calls['allreduce']} "
              f"early_pending={calls.get('early_pending', 0)} backend={dist.get_backend()} world={dist.get_world_size()}")
    print(f"RCCLOK {int(ok)}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
