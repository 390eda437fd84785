"""Training-step driver: forward + loss + backward (+ RCCL gradient all-reduce) + fused Adam on the HIP engine.

Reproduces the call sequence of the reference's drivers for ONE batch kept on the device
(solve_with_grad_GPU.py:133-181: restore `x` from a backup and re-arm the norm flags every inner step;
pre_train_Adam.py:158-191: zero_grad, forward, `mean(log(weighted residuals))`, backward, Adam), without the autograd
tape: parameters, gradients and Adam moments live in flat fp32 buffers, so the optimiser is one kernel and the
data-parallel exchange is ONE all-reduce of 4.7 MB.  The fixed launch sequence can be captured into
a hipGraph (torch.cuda.CUDAGraph) to remove host launch overhead.
"""
from __future__ import annotations

import os

import torch

from . import cmdlist
from . import lib as L
from .engine import GradStore
from .functions import unused_param_names
from .plan import _live_key, _refresh_live, get_plan


def _gather(src, index, out):
    torch.index_select(src, 0, index, out=out)


class TrainStep:
    def __init__(self, model, graphs, *, lr=None, betas=(0.9, 0.999), eps=1e-8, loss_weights=None, world_size=1,
                 process_group=None, use_graph=False, want_outputs=True, distributed=None):
        self.model = model
        self.graphs = graphs
        self.plan = get_plan(graphs)
        self._live = _live_key(graphs)
        self.engine = model.engine()
        p = model.params
        self._hyper_host = None
        self._lr = p.lr if lr is None else lr
        self._betas, self._eps = tuple(betas), eps
        self._lw = tuple(loss_weights or (p.loss_cont, p.loss_mom, p.loss_press))
        self.world_size, self.pg = world_size, process_group
        # distributed=True with world_size 1 sends the step through the same collectives (identity all-reduce): the
        # RCCL path can then be exercised on a single GPU (tests/test_rccl_gpu.py)
        self.dist_on = (world_size > 1) if distributed is None else bool(distributed)
        self.engine.dist_world, self.engine.dist_group = world_size, process_group
        self.engine.dist_force = self.dist_on
        self.use_graph = use_graph
        self._split, self._comm, self._early = None, None, False
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
        self.flat_p = torch.zeros(total + 4, dtype=torch.float32, device=dev)[:total + 1]   # (+ one zero the padding maps read)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        # Adam step counter + derived bias corrections, and the hyper-parameters: device resident (include/gfv.h,
        # gfv_adam_step_dev), so a captured step follows `ts.lr = ...` (lr_scheduler.step() in the reference drivers)
        # (adam_state[16] = {completed steps, the NEXT step's bias corrections, arrival counter, 1 - betas, running powers}: include/gfv.h)
        self.adam_state = torch.zeros(16, dtype=torch.float32, device=dev)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._sync_hyper()
        L.status_mirror()   # the fused Adam publishes the device status word from now on; step() reads it without a sync
        self.P = {}
        for n, t in zip(names, tensors):
            off, k = self.G.off[n], t.numel()
            view = self.flat_p[off:off + k].view(t.shape)
            view.copy_(t.data)
            t.data = view
            self.P[n] = view
        assert all(v.data_ptr() % 16 == 0 for v in self.P.values())
        self._setup_padding(names, tensors, skip)
        self._probes = [(names[i], tensors[i]) for i in sorted({0, len(names) // 2, len(names) - 1})]
        self.n_params = total
        self.x = graphs[0].x
        self.x_backup = self.x.clone()
        B = self.plan.B
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.gloss = torch.zeros((B, 4), dtype=torch.float32, device=dev)
        self.losses = None
        self.uvp_node = None
        self._graphs = {}
        self._list_warm = {}

    # hidden_size below 128 (FVMmodel/padding.py): parameters, moments and gradients of the TRUE shapes stay the state
    # (flat_p / flat_m / flat_v / flat_g, what Adam and the checkpoints see); every step gathers the parameters into the
    # kernels' 128-column shapes with one index_select, runs forward / backward on those, and gathers the gradients back
    # with another.  The two index maps come from padding the flat offsets themselves.
    def _setup_padding(self, names, tensors, skip):
        from FVMmodel.padding import pad_parameters
        h = int(getattr(self.model, "hidden_size", 128))
        self.padded = h != 128
        if not self.padded:
            self.P_run, self.G_run = self.P, self.G
            return
        dev, total = self.dev, self.G.total
        offs = [torch.arange(t.numel(), device=dev, dtype=torch.int64).view(t.shape) + (self.G.off[n] + 1)
                for n, t in zip(names, tensors)]                       # flat offset + 1 of every true element (0 = padding)
        padded = pad_parameters(names, offs, h)
        self.G_run = GradStore(names, [t.shape for t in padded], dev, skip=skip)
        fwd = torch.full((self.G_run.total,), total, dtype=torch.int64, device=dev)   # -> the zero behind flat_p
        back = torch.zeros(total, dtype=torch.int64, device=dev)
        for n, t in zip(names, padded):
            o = self.G_run.off[n]
            flat = t.reshape(-1)
            real = flat > 0
            fwd[o:o + flat.numel()] = torch.where(real, flat - 1, torch.full_like(flat, total))
            back[(flat[real] - 1)] = o + torch.nonzero(real).reshape(-1)
        self._pad_map, self._unpad_map = fwd, back
        self.flat_pp = torch.zeros(self.G_run.total, dtype=torch.float32, device=dev)
        self.P_run = {n: self.flat_pp[self.G_run.off[n]:self.G_run.off[n] + self.G_run.numel(n)].view(self.G_run.shape[n])
                      for n in names}

    # hyper-parameters: plain attributes on the host, mirrored into `self.hyper` on change --------------------------
    def _sync_hyper(self):
        vals = (self._lr, self._betas[0], self._betas[1], self._eps, 1.0 / self.world_size, *self._lw)
        if vals != self._hyper_host:
            betas_moved = self._hyper_host is None or vals[1:3] != self._hyper_host[1:3]
            self.hyper.copy_(torch.tensor(vals, dtype=torch.float32))
            self._hyper_host = vals
            if betas_moved:
                self._init_adam_state()

    def _init_adam_state(self, steps_done=None):
        """The device-side bias corrections of the next step, from the step count (a new object, a loaded checkpoint, new betas)."""
        t = float(self.adam_state[0]) if steps_done is None else float(steps_done)
        L.check(L.load().gfv_adam_state_init(self.adam_state.data_ptr(), float(self._betas[0]), float(self._betas[1]), t,
                                             L.stream_ptr()), "adam_state_init")

    lr = property(lambda self: self._lr)
    betas = property(lambda self: self._betas)
    eps = property(lambda self: self._eps)
    loss_weights = property(lambda self: self._lw)
    w_cont = property(lambda self: self._lw[0])
    w_mom = property(lambda self: self._lw[1])
    w_press = property(lambda self: self._lw[2])

    @lr.setter
    def lr(self, v):
        self._lr = float(v)
        self._sync_hyper()

    @betas.setter
    def betas(self, v):
        self._betas = (float(v[0]), float(v[1]))
        self._sync_hyper()

    @eps.setter
    def eps(self, v):
        self._eps = float(v)
        self._sync_hyper()

    @loss_weights.setter
    def loss_weights(self, v):
        self._lw = tuple(float(x) for x in v)
        self._sync_hyper()

    def set_lr(self, lr):
        """What `lr_scheduler.step()` does to the reference's optimizer; eager and hipGraph steps both follow it."""
        self.lr = lr

    @property
    def step_t(self):
        return self.adam_state[0:1]

    # optimizer state in the reference's checkpoint slot (importer.py:292-313 saves `optimizer{i}`) -------------------
    def state_dict(self):
        """Same nesting as torch.optim.Adam.state_dict(): per-parameter step / exp_avg / exp_avg_sq keyed by position in
        `model.parameters()` order, one param group.  `NNmodel.save_checkpoint(path, optimizer=ts)` stores it under
        `optimizer0`, `load_checkpoint(optimizer=ts, ...)` restores it."""
        names = list(self.G.off)
        t = self.adam_state[0:1].detach().cpu().clone().reshape(())
        state = {}
        for i, n in enumerate(names):
            off, k = self.G.off[n], self.G.numel(n)
            if n in self.G.skip:
                continue   # parameters without a gradient have no Adam state in torch either
            state[i] = {"step": t.clone(), "exp_avg": self.flat_m[off:off + k].view(self.G.shape[n]).detach().cpu().clone(),
                        "exp_avg_sq": self.flat_v[off:off + k].view(self.G.shape[n]).detach().cpu().clone()}
        group = {"lr": self._lr, "betas": self._betas, "eps": self._eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "gfv_param_names": names, "gfv_loss_weights": self._lw}

    def load_state_dict(self, sd):
        names = list(self.G.off)
        if "gfv_param_names" in sd and list(sd["gfv_param_names"]) != names:
            raise ValueError("optimizer state belongs to a different parameter set")
        step = None
        self.flat_m.zero_()
        self.flat_v.zero_()
        for i, st in sd["state"].items():
            n = names[int(i)]
            off, k = self.G.off[n], self.G.numel(n)
            self.flat_m[off:off + k].copy_(st["exp_avg"].reshape(-1))
            self.flat_v[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            step = float(st["step"]) if step is None else step
            if float(st["step"]) != step:
                raise ValueError("per-parameter step counts differ: not a state this fused Adam can resume")
        g = sd["param_groups"][0]
        self._lr, self._betas, self._eps = float(g["lr"]), (float(g["betas"][0]), float(g["betas"][1])), float(g["eps"])
        if "gfv_loss_weights" in sd:
            self._lw = tuple(float(x) for x in sd["gfv_loss_weights"])
        self._sync_hyper()
        self._init_adam_state(0.0 if step is None else step)

    def named_state(self):
        """{name: (parameter, exp_avg, exp_avg_sq)} views of the flat buffers.  (The alignment padding between tensors is not
        state: its gradient slots take whatever the shared slab workspace held, see gfv_dw_multi in include/gfv.h.)"""
        out = {}
        for n in self.G.off:
            off, k, sh = self.G.off[n], self.G.numel(n), self.G.shape[n]
            out[n] = tuple(b[off:off + k].view(sh) for b in (self.flat_p, self.flat_m, self.flat_v))
        return out

    def advance_time(self):
        """Time advance of the reference's solve loop (solve_with_grad_GPU.py:180-197): after the inner iterations of a time
        step the predicted node field becomes the next step's input state, the conditioning columns stay -
        `graph_node.x = cat(uvp_node_new.detach(), backup[:, 3:])`.  In place on the persistent backup (captured steps keep
        reading the same tensor)."""
        if self.uvp_node is None:
            raise RuntimeError("advance_time() needs the prediction of a step (want_outputs=True)")
        self.x_backup[:, 0:3].copy_(self.uvp_node)
        self.x.copy_(self.x_backup)

    def sync_from_model(self):
        """Call after loading a checkpoint into the model: `load_state_dict` copies into the parameter views of the flat
        buffer in place, so the values are already there; this re-checks the aliasing and drops captured steps (their
        weight images belong to the old values' step, the next step rebuilds them anyway)."""
        for n, t in zip(*self.model.param_names_tensors()):
            off = self.G.off[n]
            if t.data_ptr() != self.flat_p.data_ptr() + 4 * off:
                view = self.flat_p[off:off + t.numel()].view(t.shape)
                view.copy_(t.data)
                t.data = view
                self.P[n] = view
        self.model.node_norm._host_num_acc = None

    def set_batch(self, graphs):
        """Switch to another batch (dataset training: a new batch every step, pre_train_Adam.py:146-156).  Captured
        hipGraphs belong to the tensors of one batch and are dropped; with a gfv.pool.DevicePool the switch costs one
        small launch."""
        self.graphs = graphs
        self.plan = get_plan(graphs)
        self._live = _live_key(graphs)
        self.x = graphs[0].x
        self.x_backup = self.x.clone()
        if self.gloss.shape[0] != self.plan.B:
            self.gloss = torch.zeros((self.plan.B, 4), dtype=torch.float32, device=self.dev)
        self._graphs = {}
        self._list_warm = {}

    # one un-captured step body ------------------------------------------------------------------------------
    def _body(self, accumulate, with_adam=True):
        lib = L.load()
        st = L.stream_ptr()
        if self.padded:
            cmdlist.call(_gather, self.flat_p, self._pad_map, self.flat_pp)
        # solve_with_grad_GPU.py:143: fresh, un-normalised node state every step.  Round 6: no restore copy - the input
        # preparation reads the raw rows from the persistent backup and writes the normalised ones into `x` (Engine.prep_fwd)
        with self.engine.model_width():
            losses, uvp_node, uvp_cell, _, ctx = self.engine.forward(
                self.P_run, self.model.node_norm.buffers_dict(), self.x, self.plan, norm_global=True, accumulate=accumulate,
                want_outputs=self.want_outputs, want_edge_attr15=False, x_raw=self.x_backup,
                train_loss=(self.hyper, self.loss, self.gloss))   # loss + its gradient behind the residual norms, same launch
            self.engine.backward(self.P_run, ctx, self.gloss, self.G_run, self.plan)
        if self.padded:
            cmdlist.call(_gather, self.G_run.flat, self._unpad_map, self.flat_g)
        self.losses, self.uvp_node, self.uvp_cell = losses, uvp_node, uvp_cell
        if with_adam:
            self._adam()

    def _adam(self):
        # ONE launch (round 6): its last workgroup advances the step count, forms the next step's bias corrections and publishes
        # the device status word into the pinned mirror `step()` reads (include/gfv.h gfv_adam_step_dev, gfv_status_mirror)
        L.check(L.load().gfv_adam_step_dev(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                                           self.flat_v.data_ptr(), self.n_params, self.adam_state.data_ptr(),
                                           self.hyper.data_ptr(), L.stream_ptr()), "adam_step")

    # data-parallel exchange: the flat gradient is reduced in two buckets.  The upper one (last processor + decoder:
    # their backward runs first) goes out on a communication stream as soon as its last gradient kernel is launched and
    # overlaps the backward of the first processor and the encoders; the lower one follows the backward.  The fork to the
    # communication stream and the early all-reduce go through cmdlist.call: a RECORDED step replays them at the same
    # point of the backward (round 4: the command list is the default launch mode - the exchange of its first bucket used
    # to be exposed behind the replay).  A hipGraph capture keeps the exchange outside the graph, in one piece.
    def _bucket_split(self):
        if self._split is None:
            names = [n for n in self.G.off if ".processpr_list." in n]
            last = max((int(n.split(".processpr_list.")[1].split(".")[0]) for n in names), default=-1)
            offs = [self.G.off[n] for n in names if f".processpr_list.{last}." in n]
            self._split = min(offs) if (last > 0 and offs) else 0
        return self._split

    def _allreduce_upper(self):
        import torch.distributed as dist
        # (the blocking form: under `torch.cuda.stream(comm)` it is the communication STREAM that waits for the collective and
        # nothing has to be kept alive for a replay.  The overlap with the rest of the backward is an RCCL property: under gloo,
        # or with TORCH_NCCL_BLOCKING_WAIT set, the HOST waits here and the early bucket is merely correct, not overlapped -
        # the gloo tests assert numerics only)
        dist.all_reduce(self.flat_g[self._split:], op=dist.ReduceOp.SUM, group=self.pg)

    def _bucket_ready(self):
        split = self._bucket_split()
        if split <= 0 or torch.cuda.is_current_stream_capturing():
            return
        if self._comm is None:
            self._comm = torch.cuda.Stream()
        L.stream_wait(self._comm, torch.cuda.current_stream())
        for sd in self.engine._sides:
            L.stream_wait(self._comm, sd)        # the weight-gradient kernels run there
        with torch.cuda.stream(self._comm):
            cmdlist.call(self._allreduce_upper)
        self._early = True

    def _allreduce(self, early=None):
        """The rest of the exchange, behind the backward: the lower bucket + the join with the communication stream when the
        upper one went out early (`early`: default = what the step that just ran did), else the whole buffer."""
        import torch.distributed as dist
        early = self._early if early is None else early
        self._early = False
        if early:
            dist.all_reduce(self.flat_g[:self._split], op=dist.ReduceOp.SUM, group=self.pg)
            torch.cuda.current_stream().wait_stream(self._comm)
        else:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)

    def _hooked_body(self, acc, dist_on):
        # (a padded model's true-shape gradient exists only after the gather at the end of the backward: one all-reduce)
        self._early = False
        self.engine.bucket_hook = self._bucket_ready if (dist_on and not self.padded) else None
        try:
            self._body(acc, with_adam=not dist_on)
        finally:
            self.engine.bucket_hook = None
        return self._early

    def _eager(self, acc, dist_on):
        self._hooked_body(acc, dist_on)
        if dist_on:
            self._allreduce()
            self._adam()

    def _check_aliasing(self):
        """The module's parameters must still be the views of the flat buffer this object made them (a `model.to()`, a
        `load_state_dict(assign=True)` or a second TrainStep on the same model re-points them): captured steps hold raw
        pointers, and a replay does not run the host code that would notice.  Three probes per step; on a mismatch the
        views are re-established from the module's current values and every capture is dropped."""
        for n, t in self._probes:
            if t.data_ptr() != self.flat_p.data_ptr() + 4 * self.G.off[n]:
                self.sync_from_model()
                self._graphs.clear()
                self._list_warm.clear()
                return

    def step(self):
        """One training iteration.  Returns the (device) scalar loss tensor of this rank's batch.
        `use_graph`: False = eager launches, True = hipGraph replay, "list" = command-list replay."""
        self._check_aliasing()
        # what the kernels of the steps BEFORE the last one raised in the device status word (a hidden activation outside the
        # fixed-scale fp16 window, a weight-gradient operand beyond fp16): read from the pinned mirror the fused Adam publishes
        # into - no synchronisation; the step that overflowed has long finished when its successor is issued
        L.raise_on_status("TrainStep.step")
        # caller-owned data the plan holds copies of (Dirichlet targets, node / face types, theta_PDE, sigma, uvp_dim, dt_graph):
        # an in-place edit between two steps reaches the plan's tensors here - copied in place, so the pointers a captured step
        # holds stay valid (the reference re-reads graph_node.y / graph_Index on every forward, importer.py:141-154,168).
        # Batches of a DevicePool carry their plan with them and are refreshed the same way.
        live = _live_key(self.graphs)
        if live != self._live:
            _refresh_live(self.plan, self.graphs)
            self._live = live
        acc = self.model.node_norm.should_accumulate()
        dist_on = self.dist_on
        if not self.use_graph or (acc and dist_on):
            # (an accumulating data-parallel step exchanges the Normalizer statistics inside the forward: not captured)
            self._eager(acc, dist_on)
        elif self.use_graph == "list":
            # command-list replay (gfv/cmdlist.py): the eager launch sequence of one step, recorded at the C ABI and
            # replayed with one ctypes call per launch - same two streams, same event edges, no per-launch Python work
            key = ("list", acc, dist_on)
            entry = self._graphs.get(key)
            sig = self.engine.capture_signature()
            if entry is not None and entry[1] != sig:
                self._graphs.pop(key)
                self._list_warm.pop(key, None)
                entry = None
            if entry is None:
                warm = self._list_warm.get(key, 0)
                if warm < 2:
                    # the first steps settle the weight-image set and the persistent workspaces (allocated outside the pool)
                    self._list_warm[key] = warm + 1
                    self._eager(acc, dist_on)
                else:
                    with cmdlist.record() as cl:
                        early = self._hooked_body(acc, dist_on)   # (the early bucket's fork + all-reduce are part of the list)
                    self._graphs[key] = (cl, self.engine.capture_signature(), early)
                    if dist_on:
                        self._allreduce(early)
                        self._adam()
            else:
                entry[0].replay()
                if dist_on:
                    self._allreduce(entry[2])
                    self._adam()
        else:
            key = (acc, dist_on)
            g = self._graphs.get(key)
            if g is not None and g[1] != self.engine.capture_signature():
                # the captured launches hold raw pointers into the weight-image set / descriptor tables of the engine:
                # a changed set (model.to(), re-flattened parameters, a new weight block) makes every capture stale
                self._graphs.clear()
                self._list_warm.clear()   # (command lists were dropped with it: they re-record only after new warm-up steps)
                g = None
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
                # (thread_local: the RCCL watchdog thread of a process group polls events while this thread captures; under
                # the default global mode that poll invalidates the capture and aborts the process)
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._body(acc, with_adam=not dist_on)
                g = (g, self.engine.capture_signature())
                self._graphs[key] = g
                self._restore()  # capture does not execute; state is as before this step
            g[0].replay()
            if dist_on:
                self._allreduce()
                self._adam()
        if acc:
            self.model.node_norm.note_accumulated()
        return self.loss

    # state snapshot so the capture warm-up does not advance training ---------------------------------------
    def _snapshot(self):
        nb = self.model.node_norm
        self._snap = (self.flat_p.clone(), self.flat_m.clone(), self.flat_v.clone(), self.adam_state.clone(),
                      nb.acc_count.clone(), nb.num_accumulations.clone(), nb.acc_sum.clone(), nb.acc_sum_squared.clone())

    def _restore(self):
        nb = self.model.node_norm
        for dst, src in zip((self.flat_p, self.flat_m, self.flat_v, self.adam_state, nb.acc_count, nb.num_accumulations,
                             nb.acc_sum, nb.acc_sum_squared), self._snap):
            dst.copy_(src)
