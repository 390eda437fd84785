"""Training-step driver: forward + loss + backward (+ RCCL gradient all-reduce) + fused Adam on the HIP engine.

Reproduces the call sequence of the reference's drivers for ONE batch kept on the device
(solve_with_grad_GPU.py:133-181: restore `x` from a backup and re-arm the norm flags every inner step;
pre_train_Adam.py:158-191: zero_grad, forward, `mean(log(weighted residuals))`, backward, Adam), without the autograd
tape: parameters, gradients and Adam moments live in flat fp32 buffers, so the optimiser is one kernel and the
data-parallel exchange is ONE all-reduce of 4.7 MB.  The fixed launch sequence can be captured into
a hipGraph (torch.cuda.CUDAGraph) to remove host launch overhead.
"""
from __future__ import annotations

import torch

from . import lib as L
from .engine import GradStore
from .functions import unused_param_names
from .plan import get_plan


class TrainStep:
    def __init__(self, model, graphs, *, lr=None, betas=(0.9, 0.999), eps=1e-8, loss_weights=None, world_size=1,
                 process_group=None, use_graph=False, want_outputs=True):
        self.model = model
        self.graphs = graphs
        self.plan = get_plan(graphs)
        self.engine = model.engine()
        p = model.params
        self.lr = p.lr if lr is None else lr
        self.betas, self.eps = betas, eps
        self.w_cont, self.w_mom, self.w_press = loss_weights or (p.loss_cont, p.loss_mom, p.loss_press)
        self.world_size, self.pg = world_size, process_group
        self.engine.dist_world, self.engine.dist_group = world_size, process_group
        self.use_graph = use_graph
        self._split, self._comm, self._work = None, None, None
        self.want_outputs = want_outputs
        dev = graphs[0].x.device
        self.dev = dev

        # flat parameter / gradient / moment buffers; the module's parameters become views of the flat buffer
        names, tensors = model.param_names_tensors()
        skip = unused_param_names(names)
        self.flat_g = None
        self.G = GradStore(names, [t.shape for t in tensors], dev, skip=skip)
        self.flat_g = self.G.flat
        total = self.G.total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self.P = {}
        for n, t in zip(names, tensors):
            off, k = self.G.off[n], t.numel()
            view = self.flat_p[off:off + k].view(t.shape)
            view.copy_(t.data)
            t.data = view
            self.P[n] = view
        assert all(v.data_ptr() % 16 == 0 for v in self.P.values())
        self.n_params = total
        self.x = graphs[0].x
        self.x_backup = self.x.clone()
        B = self.plan.B
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gloss = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        self.losses = None
        self.uvp_node = None
        self._graphs = {}

    def set_batch(self, graphs):
        """Switch to another batch (dataset training: a new batch every step, pre_train_Adam.py:146-156).  Captured
        hipGraphs belong to the tensors of one batch and are dropped; with a gfv.pool.DevicePool the switch costs one
        small launch."""
        self.graphs = graphs
        self.plan = get_plan(graphs)
        self.x = graphs[0].x
        self.x_backup = self.x.clone()
        if self.gloss.shape[0] != self.plan.B:
            self.gloss = torch.zeros((self.plan.B, 4), dtype=torch.float32, device=self.dev)
        self._graphs = {}

    # one un-captured step body ------------------------------------------------------------------------------
    def _body(self, accumulate, with_adam=True):
        lib = L.load()
        st = L.stream_ptr()
        self.x.copy_(self.x_backup)  # solve_with_grad_GPU.py:143 (fresh, un-normalised node state every step)
        losses, uvp_node, uvp_cell, _, ctx = self.engine.forward(
            self.P, self.model.node_norm.buffers_dict(), self.x, self.plan, norm_global=True, accumulate=accumulate,
            want_outputs=self.want_outputs, want_edge_attr15=False)
        L.check(lib.gfv_train_loss(losses.data_ptr(), self.plan.B, self.w_cont, self.w_mom, self.w_press,
                                   self.loss.data_ptr(), self.gloss.data_ptr(), st), "train_loss")
        self.engine.backward(self.P, ctx, self.gloss, self.G, self.plan)
        self.losses, self.uvp_node, self.uvp_cell = losses, uvp_node, uvp_cell
        if with_adam:
            self._adam()

    def _adam(self):
        lib = L.load()
        L.check(lib.gfv_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                  self.flat_v.data_ptr(), self.n_params, self.step_t.data_ptr(), self.lr, self.betas[0],
                                  self.betas[1], self.eps, 1.0 / self.world_size, L.stream_ptr()), "adam_step")

    # data-parallel exchange: the flat gradient is reduced in two buckets.  The upper one (last processor + decoder:
    # their backward runs first) goes out on a communication stream as soon as its last gradient kernel is launched and
    # overlaps the backward of the first processor and the encoders; the lower one follows the backward.
    def _bucket_split(self):
        if self._split is None:
            names = [n for n in self.G.off if ".processpr_list." in n]
            last = max((int(n.split(".processpr_list.")[1].split(".")[0]) for n in names), default=-1)
            offs = [self.G.off[n] for n in names if f".processpr_list.{last}." in n]
            self._split = min(offs) if (last > 0 and offs) else 0
        return self._split

    def _bucket_ready(self):
        import torch.distributed as dist
        split = self._bucket_split()
        if split <= 0 or torch.cuda.is_current_stream_capturing():
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        self._comm.wait_stream(torch.cuda.current_stream())
        if self.engine._side is not None:
            self._comm.wait_stream(self.engine._side)      # the weight-gradient kernels run there
        with torch.cuda.stream(self._comm):
            self._work = dist.all_reduce(self.flat_g[split:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _allreduce(self):
        import torch.distributed as dist
        if self._work is not None:
            dist.all_reduce(self.flat_g[:self._split], op=dist.ReduceOp.SUM, group=self.pg)
            self._work.wait()                              # the current stream waits for the early bucket
            torch.cuda.current_stream().wait_stream(self._comm)
            self._work = None
        else:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)

    def step(self):
        """One training iteration.  Returns the (device) scalar loss tensor of this rank's batch."""
        acc = self.model.node_norm.should_accumulate()
        dist_on = self.world_size > 1
        if not self.use_graph or (acc and dist_on):
            # (an accumulating data-parallel step exchanges the Normalizer statistics inside the forward: not captured)
            self.engine.bucket_hook = self._bucket_ready if dist_on else None
            try:
                self._body(acc, with_adam=not dist_on)
            finally:
                self.engine.bucket_hook = None
            if dist_on:
                self._allreduce()
                self._adam()
        else:
            key = (acc, dist_on)
            g = self._graphs.get(key)
            if g is None:
                # warm the allocator on a side stream, then capture (PyTorch CUDA-graph recipe)
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    self._snapshot()
                    # twice: the first pass notes the weight blocks of the chain launches, the second allocates and
                    # builds their split-fp16 images - the capture then records the launch sequence of a steady step
                    for _ in range(2 if self.engine.f16split else 1):
                        self._body(acc, with_adam=not dist_on)
                        self._restore()
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._body(acc, with_adam=not dist_on)
                self._graphs[key] = g
                self._restore()  # capture does not execute; state is as before this step
            g.replay()
            if dist_on:
                self._allreduce()
                self._adam()
        if acc:
            self.model.node_norm.note_accumulated()
        return self.loss

    # state snapshot so the capture warm-up does not advance training ---------------------------------------
    def _snapshot(self):
        nb = self.model.node_norm
        self._snap = (self.flat_p.clone(), self.flat_m.clone(), self.flat_v.clone(), self.step_t.clone(),
                      nb.acc_count.clone(), nb.num_accumulations.clone(), nb.acc_sum.clone(), nb.acc_sum_squared.clone())

    def _restore(self):
        nb = self.model.node_norm
        for dst, src in zip((self.flat_p, self.flat_m, self.flat_v, self.step_t, nb.acc_count, nb.num_accumulations,
                             nb.acc_sum, nb.acc_sum_squared), self._snap):
            dst.copy_(src)
