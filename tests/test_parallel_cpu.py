"""CPU, world_size 2 over gloo: the data-parallel exchange of the hot path (gfv/parallel.py).  Graphs are sharded by
rank; the all-reduced (summed, then 1/world-scaled) flat gradient must equal the gradient of the global-batch loss and
every rank must hold identical parameters after the optimiser step."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    import cases
    from gfv import meshgen
    from gfv.graph import build_batch
    from gfv.parallel import allreduce_flat_grad, flat_pack, shard_range
    from oracle import fvgn_oracle as O
    specs = cases.CASES["cyl_cavity_b2"] + cases.CASES["cyl_b3"][:1] + cases.CASES["cavity_mixed_b1"]  # 4 graphs
    meshes, fields = [], []
    for fac, kw, U, fseed in specs:
        m = meshgen.finish_mesh(getattr(meshgen, fac)(**kw), U=U)
        meshes.append(m)
        fields.append(meshgen.random_fields(m, seed=fseed))
    mine = list(shard_range(len(meshes), rank, world))
    hyper = {"dataset_size": 1}
    P = O.init_parameters(0)
    names = list(P)

    def grads_of(ids):
        g = build_batch([meshes[i] for i in ids], [fields[i] for i in ids])
        Pg = {k: v.detach().requires_grad_(True) for k, v in P.items()}
        out = O.model_forward(Pg, O.new_normalizer_buffers(), g, hyper)
        loss = O.training_loss(out, hyper)
        gl = torch.autograd.grad(loss, [Pg[k] for k in names], allow_unused=True)
        return [torch.zeros_like(P[k]) if t is None else t for k, t in zip(names, gl)]

    flat = flat_pack(grads_of(mine))
    scale = allreduce_flat_grad(flat, world)
    flat *= scale
    full = flat_pack(grads_of(list(range(len(meshes)))))
    err = float((flat - full).abs().max() / full.abs().max())
    # identical Adam step on every rank
    flat_p = flat_pack([P[k] for k in names])
    m, v = torch.zeros_like(flat_p), torch.zeros_like(flat_p)
    m.mul_(0.9).add_(flat, alpha=0.1)
    v.mul_(0.999).addcmul_(flat, flat, value=0.001)
    flat_p -= 5e-5 / (1 - 0.9) * m / ((v.sqrt() / (1 - 0.999) ** 0.5) + 1e-8)
    gathered = [torch.zeros_like(flat_p) for _ in range(world)]
    dist.all_gather(gathered, flat_p)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    # Normalizer statistics: each rank accumulates its own shard, the exchange must leave every rank with
    # the statistics of the global batch, bit-identical across ranks
    from gfv.parallel import allreduce_normalizer, snapshot_normalizer
    feats = lambda ids: build_batch([meshes[i] for i in ids], [fields[i] for i in ids])[0].x[:, 3:]
    nbuf = O.new_normalizer_buffers()
    nbuf["acc_sum"] += 0.5                      # pretend earlier steps have been accumulated already
    nbuf["acc_count"] += 100.0
    full_buf = {k: v.clone() for k, v in nbuf.items()}
    before = snapshot_normalizer(nbuf)
    O.normalizer_forward(nbuf, feats(mine), max_accumulations=10)
    allreduce_normalizer(nbuf, before, world)
    O.normalizer_forward(full_buf, feats(list(range(len(meshes)))), max_accumulations=10)
    nerr = max(float((nbuf[k] - full_buf[k]).abs().max() / (full_buf[k].abs().max() + 1e-30))
               for k in ("acc_sum", "acc_sum_squared", "acc_count", "num_accumulations"))
    packed = snapshot_normalizer(nbuf)
    gathered_n = [torch.zeros_like(packed) for _ in range(world)]
    dist.all_gather(gathered_n, packed)
    nsame = all(torch.equal(gathered_n[0], t) for t in gathered_n)
    if rank == 0:
        ret["err"], ret["same"], ret["nerr"], ret["nsame"] = err, same, nerr, nsame
    dist.destroy_process_group()


def test_sharded_gradient_allreduce_equals_global_batch_gradient():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["same"], "ranks diverged after the optimiser step"
    assert ret["err"] < 1e-5, ret["err"]
    assert ret["nsame"], "Normalizer buffers differ across ranks after the exchange"
    assert ret["nerr"] < 1e-6, ret["nerr"]
