"""torch.autograd glue: each Function runs a hand-orchestrated forward of gfv.engine.Engine and, on backward, the
matching hand-written adjoint.  Parameters are passed as explicit tensor arguments (so autograd tracks them) and
re-keyed by name for the engine."""
from __future__ import annotations

import torch

from .engine import Engine, GradStore
from .plan import get_plan


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("Gen-FVGN MI355X path: tensors must live on the GPU (HIP kernels only, no CPU fallback)")


def _alloc_grads(names, tensors, skip=()):
    return GradStore(names, [t.shape for t in tensors], tensors[0].device, skip=skip)


def unused_param_names(names):
    """Parameters that exist in the reference state_dict but never receive a gradient."""
    return {n for n in names if ".ln_1." in ("." + n) or n.endswith("Attn.temperature")}


class GnBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, names, x, e, *params):
        require_gpu(x)
        P = dict(zip(names, (p.detach() for p in params)))
        x_out, e_out, sv = engine.gn_fwd(P, "blk", x.detach().contiguous(), e.detach().contiguous(), plan)
        ctx.engine, ctx.plan, ctx.names, ctx.sv, ctx.P = engine, plan, names, sv, P
        return x_out, e_out

    @staticmethod
    def backward(ctx, g_x, g_e):
        P = ctx.P
        grads = _alloc_grads(ctx.names, [P[n] for n in ctx.names])
        dev = next(iter(P.values())).device
        if g_x is None:
            g_x = torch.zeros((ctx.plan.N, 128), device=dev)
        if g_e is None:
            g_e = torch.zeros((ctx.plan.E, 128), device=dev)
        gx, ge = ctx.engine.gn_bwd(P, ctx.sv, g_x.contiguous(), g_e.contiguous(), grads, ctx.plan)
        ctx.engine.join()
        return (None, None, None, gx, ge) + tuple(grads.view(n) for n in ctx.names)


class Mlp3Fn(torch.autograd.Function):
    """Plain fused MLP on rows of x (Encoder / Decoder)."""

    @staticmethod
    def forward(ctx, engine, names, ln, width, x, *params):
        require_gpu(x)
        from .ops import Seg
        P = dict(zip(names, (p.detach() for p in params)))
        xd = x.detach().contiguous()
        M = xd.shape[0]
        out, _, sv = engine.mlp3_fwd(P, "mlp", M, [Seg(xd, width=width, ld=xd.stride(0))], ln=ln)
        ctx.engine, ctx.names, ctx.sv, ctx.P, ctx.need_dx = engine, names, sv, P, x.requires_grad
        ctx.in_width = xd.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):
        P = ctx.P
        grads = _alloc_grads(ctx.names, [P[n] for n in ctx.names])
        gx = None
        if ctx.need_dx:
            gx = torch.empty((ctx.sv["M"], ctx.in_width), dtype=torch.float32, device=g.device)
            assert ctx.in_width == 128, "input gradients are only needed for 128-wide latent inputs"
            ctx.engine.mlp3_bwd(P, ctx.sv, g.contiguous(), grads, outs=[gx])
            ctx.engine.join()
        else:
            ctx.engine.mlp3_bwd(P, ctx.sv, g.contiguous(), grads)
            ctx.engine.join()
        return (None, None, None, None, gx) + tuple(grads.view(n) for n in ctx.names)


class TransolverFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, names, fx, *params):
        require_gpu(fx)
        P = dict(zip(names, (p.detach() for p in params)))
        out, sv = engine.trans_fwd(P, "tb", fx.detach().contiguous(), None, plan)
        ctx.engine, ctx.plan, ctx.names, ctx.sv, ctx.P = engine, plan, names, sv, P
        return out

    @staticmethod
    def backward(ctx, g):
        P = ctx.P
        skip = unused_param_names(ctx.names)
        grads = _alloc_grads(ctx.names, [P[n] for n in ctx.names], skip)
        gfx = ctx.engine.trans_bwd(P, ctx.sv, g.contiguous(), grads, ctx.plan)
        ctx.engine.join()
        return (None, None, None, gfx) + tuple(grads.view(n) for n in ctx.names)


class SimulatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, names, x, ea16, *params):
        require_gpu(x)
        P = dict(zip(names, (p.detach() for p in params)))
        dec, sv = engine.simulator_fwd(P, x.detach().contiguous(), ea16.detach().contiguous(), plan)
        ctx.engine, ctx.plan, ctx.names, ctx.sv, ctx.P = engine, plan, names, sv, P
        return dec

    @staticmethod
    def backward(ctx, g):
        P = ctx.P
        skip = unused_param_names(ctx.names)
        grads = _alloc_grads(ctx.names, [P[n] for n in ctx.names], skip)
        ctx.engine.simulator_bwd(P, ctx.sv, g.contiguous(), grads, ctx.plan)
        ctx.engine.join()
        return (None, None, None, None, None) + tuple(grads.view(n) for n in ctx.names)


class IntegratorFn(torch.autograd.Function):
    """dec [N,3] (raw decoder output) -> four per-graph residual losses + node / cell fields."""

    @staticmethod
    def forward(ctx, engine, plan, dec, uv_old):
        require_gpu(dec)
        losses, uvp_node, uvp_cell, sv = engine.fvm_fwd(dec.detach().contiguous(), uv_old.detach().contiguous(), plan)
        ctx.engine, ctx.plan, ctx.sv = engine, plan, sv
        ctx.mark_non_differentiable(uvp_node, uvp_cell)
        return losses, uvp_node, uvp_cell

    @staticmethod
    def backward(ctx, g_losses, _gn, _gc):
        gdec = ctx.engine.fvm_bwd(ctx.sv, g_losses.contiguous(), ctx.plan)
        return None, None, gdec, None


# (storage address of the flat gradient tensor the last ModelFn.backward handed out, positions of the parameters that got no
# gradient - their slots hold zeros): gfv.optim.Adam feeds that tensor to its fused launch as it is
LAST_FLAT = None
# (GFV_DROPIN_TIMING=1: host seconds spent in the sections of the replayed autograd node, summed - profiles/tools/dropin_profile.py)
TIMING = None
if __import__("os").environ.get("GFV_DROPIN_TIMING") == "1":
    TIMING = {"fwd_calls": 0, "fwd_total": 0.0, "fwd_replay": 0.0, "bwd_calls": 0, "bwd_total": 0.0, "bwd_replay": 0.0, "bwd_views": 0.0}


def _note_flat(flat, names, skip):
    global LAST_FLAT
    LAST_FLAT = (flat.untyped_storage().data_ptr(), frozenset(i for i, n in enumerate(names) if n in skip))


class _Replay:
    """One recorded (forward, backward) pair of command lists of NNmodel.forward for ONE batch, parameter set and flag
    combination (gfv/cmdlist.py).  The lists hold raw pointers: `x` (the node-state tensor of the recorded call), the saved
    activations `sv` and the outputs live in the forward list's private pool, the gradient rows in `grads` (allocated once,
    outside the pools), the incoming loss gradient is copied into `gloss`."""

    __slots__ = ("warm", "fwd", "bwd", "x", "outs", "sv", "sig", "gloss", "grads", "plan", "pending", "P", "split_sizes", "split_how")

    def __init__(self):
        self.warm, self.fwd, self.bwd, self.pending = 0, None, None, False
        self.x = self.outs = self.sv = self.sig = self.gloss = self.grads = self.plan = self.P = self.split_sizes = self.split_how = None


class ReplayCache:
    """Per-model cache of recorded forward / backward lists (round 6: the drop-in path `loss.backward(); optimizer.step()` of
    pre_train_Adam.py:158-191 / solve_with_grad_GPU.py:133-181 used to issue every launch eagerly from Python - 2.9 x the
    command-list step on the 5 k-cell cavity).  A call replays when the batch's plan, the parameter storage, the product form
    and the two norm flags are the ones of a recorded call; anything else runs eagerly, as before.  GFV_DROPIN_REPLAY=0 turns
    it off."""

    WARM = 2   # eager calls before recording: they settle the weight-image set and the engine's persistent workspaces

    def __init__(self):
        import os
        self.enabled = os.environ.get("GFV_DROPIN_REPLAY", "1") != "0"
        self.entries = {}
        self.replays = 0   # (diagnostics: how many forward calls were replayed)

    def lookup(self, engine, plan, key):
        if not self.enabled or engine.hidden != 128 or engine.dist_world > 1 or engine.dist_force:
            return None
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) >= 4:    # a training loop over many batches: keep the most recent few (a recorded entry holds
                                          # its activations' pool: ~1 GB per 50 k-cell mesh)
                self.entries.pop(next(iter(self.entries)))
            ent = self.entries[key] = _Replay()
            ent.plan = plan
        return ent

    def clear(self):
        self.entries.clear()


class ModelFn(torch.autograd.Function):
    """Whole NNmodel.forward (importer.py:156-240) as one autograd node."""

    @staticmethod
    def forward(ctx, engine, plan, names, buffers, x, flags, cache, *params):
        require_gpu(x)
        from . import cmdlist
        from . import lib as L
        t_in = __import__("time").perf_counter() if TIMING is not None else 0.0
        P = None   # name -> detached parameter: built only where launches are issued from Python (159 detach calls: ~65 us)
        ent = None
        if cache is not None and cmdlist.active() is None and not torch.cuda.is_current_stream_capturing():
            key = (id(plan), flags["norm_global"], flags["accumulate"], L.load().gfv_f16split_enabled(),
                   tuple(p.data_ptr() for p in params))
            ent = cache.lookup(engine, plan, key)

        def run():
            nonlocal P
            P = dict(zip(names, (p.detach() for p in params)))
            return engine.forward(P, buffers, x, plan, norm_global=flags["norm_global"], accumulate=flags["accumulate"])
        with engine.model_width():
            if ent is None or ent.pending:
                # (pending: the previous replayed forward's backward has not run - its saved rows must not be overwritten)
                ent = None
                losses, uvp_node, uvp_cell, ea15, sv = run()
            elif ent.fwd is None:
                if ent.warm < ReplayCache.WARM:
                    ent.warm += 1
                    ent = None
                    losses, uvp_node, uvp_cell, ea15, sv = run()
                else:
                    with cmdlist.record() as cl:
                        losses, uvp_node, uvp_cell, ea15, sv = run()
                    ent.fwd, ent.x, ent.sv, ent.P = cl, x, sv, P
                    ent.outs = (losses, uvp_node, uvp_cell, ea15)
                    ent.sig = engine.capture_signature()
            elif ent.sig != engine.capture_signature():
                # the engine's weight-image set / descriptor tables changed under the lists: drop them, run eagerly
                cache.entries = {k: v for k, v in cache.entries.items() if v is not ent}
                ent = None
                losses, uvp_node, uvp_cell, ea15, sv = run()
            else:
                foreign = x.data_ptr() != ent.x.data_ptr()
                if foreign:                   # another tensor than the recorded one carries the node state: through the recorded
                    ent.x.copy_(x)            # tensor and back (the reference normalises graph_node.x IN PLACE, importer.py:123-130)
                t_r = __import__("time").perf_counter() if TIMING is not None else 0.0
                ent.fwd.replay()
                if TIMING is not None:
                    TIMING["fwd_replay"] += __import__("time").perf_counter() - t_r
                if foreign:
                    x.copy_(ent.x)
                losses, uvp_node, uvp_cell, ea15 = ent.outs
                sv, P = ent.sv, ent.P
                cache.replays += 1
        if ent is not None:
            # the lists' own output tensors are overwritten by the next replay: hand out copies (4 small launches)
            losses, uvp_node, uvp_cell, ea15 = losses.clone(), uvp_node.clone(), uvp_cell.clone(), ea15.clone()
            ent.pending = any(ctx.needs_input_grad)   # a backward will read the saved rows: no replay into them until it has run
        ctx.engine, ctx.plan, ctx.names, ctx.sv, ctx.P, ctx.ent = engine, plan, names, sv, P, ent
        ctx.mark_non_differentiable(uvp_node, uvp_cell, ea15)
        if TIMING is not None:
            TIMING["fwd_calls"] += 1
            TIMING["fwd_total"] += __import__("time").perf_counter() - t_in
        return losses, uvp_node, uvp_cell, ea15

    @staticmethod
    def backward(ctx, g_losses, _gn, _gc, _ge):
        from . import cmdlist
        from . import lib as L
        t_in = __import__("time").perf_counter() if TIMING is not None else 0.0
        P, ent = ctx.P, ctx.ent
        skip = unused_param_names(ctx.names)
        tail = (None, None, None, None, None, None, None)
        if ent is None:
            grads = _alloc_grads(ctx.names, [P[n] for n in ctx.names], skip)
            with ctx.engine.model_width():
                ctx.engine.backward(P, ctx.sv, g_losses.contiguous(), grads, ctx.plan)
            L.status_publish()    # NNmodel.forward reads the mirror at its next call (no synchronisation)
            _note_flat(grads.flat, ctx.names, skip)
            return tail + tuple(grads.view(n) for n in ctx.names)
        with ctx.engine.model_width():
            if ent.bwd is None:
                ent.gloss = g_losses.contiguous().clone()
                ent.grads = G0 = _alloc_grads(ctx.names, [P[n] for n in ctx.names], skip)   # zeroed once; every step rewrites the same rows
                pad = lambda k: (k + 3) // 4 * 4
                ent.split_sizes = [pad(G0.numel(n)) for n in ctx.names]
                assert sum(ent.split_sizes) == G0.flat.numel()
                # per tensor: None = no gradient; True = the chunk as it is (1-D, unpadded); else (numel, shape)
                ent.split_how = [None if n in G0.skip else
                                 (True if (len(G0.shape[n]) == 1 and pad(G0.numel(n)) == G0.numel(n)) else (G0.numel(n), G0.shape[n]))
                                 for n in ctx.names]
                with cmdlist.record() as cl:
                    ctx.engine.backward(P, ent.sv, ent.gloss, ent.grads, ctx.plan)
                    L.status_publish()
                ent.bwd = cl
            else:
                ent.gloss.copy_(g_losses)
                t_r = __import__("time").perf_counter() if TIMING is not None else 0.0
                ent.bwd.replay()
                if TIMING is not None:
                    TIMING["bwd_replay"] += __import__("time").perf_counter() - t_r
        ent.pending = False
        t_v = __import__("time").perf_counter() if TIMING is not None else 0.0
        # autograd keeps what it is handed as `.grad` (or adds it to one): a fresh flat copy per step - one launch - so that a
        # later replay never rewrites a tensor the caller still holds; its views keep the flat layout gfv.optim.Adam recognises
        flat = ent.grads.flat.clone()
        G = ent.grads
        _note_flat(flat, ctx.names, G.skip)
        # the 159 views: ONE split at the (16-byte aligned) offsets, then a reshape only where the tensor is not 1-D / not padded
        # (slicing + viewing every tensor from Python was ~0.4 ms of a 2.6 ms iteration on the 5 k-cell cavity,
        # profiles/r06_dropin_host_profile.txt)
        out = tail + tuple(None if how is None else (c if how is True else c[:how[0]].view(how[1]))
                           for c, how in zip(flat.split_with_sizes(ent.split_sizes), ent.split_how))
        if TIMING is not None:
            now = __import__("time").perf_counter()
            TIMING["bwd_calls"] += 1
            TIMING["bwd_views"] += now - t_v
            TIMING["bwd_total"] += now - t_in
        return out


__all__ = ["Engine", "get_plan", "GnBlockFn", "Mlp3Fn", "TransolverFn", "SimulatorFn", "IntegratorFn", "ModelFn",
           "ReplayCache", "unused_param_names"]
