"""Data-parallel exchange of the hot path: ONE all-reduce(SUM) of the flat fp32 gradient per step, scaled by 1/world
in the fused Adam.  Backend "nccl" (= RCCL over xGMI on the MI355X node), "gloo" in the CPU tests.
Graphs are sharded by rank with equal counts, so the mean of the per-rank gradients of mean_b log(.) equals the gradient
of the global-batch loss (pre_train_Adam.py:184)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, equal shards (rank r takes graphs r*k ... r*k+k-1)."""
    if n_items % world:
        raise ValueError("equal shards are required for the gradient mean to equal the global-batch gradient")
    k = n_items // world
    return range(rank * k, (rank + 1) * k)


def allreduce_flat_grad(flat_grad, world, group=None):
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world  # factor the optimiser applies


def flat_pack(tensors, pad=4):
    """Pack tensors (None -> zeros) into one flat buffer with every tensor aligned to `pad` floats."""
    sizes = [((t.numel() + pad - 1) // pad) * pad for t in tensors]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=tensors[0].device)
    off = 0
    for t, s in zip(tensors, sizes):
        flat[off:off + t.numel()] = t.reshape(-1)
        off += s
    return flat


def snapshot_normalizer(buffers):
    """19 floats: (acc_sum[9], acc_sum_squared[9], acc_count) before this rank's batch is accumulated."""
    return torch.cat((buffers["acc_sum"].reshape(-1), buffers["acc_sum_squared"].reshape(-1),
                      buffers["acc_count"].reshape(-1))).clone()


def allreduce_normalizer(buffers, before, world, group=None, force=False):
    """the online Normalizer (utils/normalization.py:32-85) accumulates sum x, sum x^2 and the row count of
    every batch it sees; with the graphs sharded by rank each replica would keep different statistics and the replicas
    would drift.  `buffers` already hold this rank's batch, `before` is `snapshot_normalizer` taken before it: the
    buffers become  before + all-reduce(sum)(this rank's delta)  - the statistics of the GLOBAL batch, and identical
    bit for bit on every rank (the same two operands are added everywhere)."""
    if world <= 1:
        if force:
            # a one-rank group (the single-GPU RCCL self-test): the collective is the identity; the statistics go through
            # it unchanged, so the result stays bit-identical to the non-distributed step
            cur = snapshot_normalizer(buffers)
            dist.all_reduce(cur, op=dist.ReduceOp.SUM, group=group)
            _write_normalizer(buffers, cur)
        return
    delta = snapshot_normalizer(buffers) - before
    dist.all_reduce(delta, op=dist.ReduceOp.SUM, group=group)
    _write_normalizer(buffers, before + delta)


def _write_normalizer(buffers, new):
    n = buffers["acc_sum"].numel()
    buffers["acc_sum"].copy_(new[0:n].view_as(buffers["acc_sum"]))
    buffers["acc_sum_squared"].copy_(new[n:2 * n].view_as(buffers["acc_sum_squared"]))
    buffers["acc_count"].copy_(new[2 * n:2 * n + 1].view_as(buffers["acc_count"]))
