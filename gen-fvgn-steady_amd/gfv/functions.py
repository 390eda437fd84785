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


class ModelFn(torch.autograd.Function):
    """Whole NNmodel.forward (importer.py:156-240) as one autograd node."""

    @staticmethod
    def forward(ctx, engine, plan, names, buffers, x, flags, *params):
        require_gpu(x)
        P = dict(zip(names, (p.detach() for p in params)))
        with engine.model_width():
            losses, uvp_node, uvp_cell, ea15, sv = engine.forward(
                P, buffers, x, plan, norm_global=flags["norm_global"], accumulate=flags["accumulate"])
        ctx.engine, ctx.plan, ctx.names, ctx.sv, ctx.P = engine, plan, names, sv, P
        ctx.mark_non_differentiable(uvp_node, uvp_cell, ea15)
        return losses, uvp_node, uvp_cell, ea15

    @staticmethod
    def backward(ctx, g_losses, _gn, _gc, _ge):
        P = ctx.P
        skip = unused_param_names(ctx.names)
        grads = _alloc_grads(ctx.names, [P[n] for n in ctx.names], skip)
        with ctx.engine.model_width():
            ctx.engine.backward(P, ctx.sv, g_losses.contiguous(), grads, ctx.plan)
        return (None, None, None, None, None, None) + tuple(grads.view(n) for n in ctx.names)


__all__ = ["Engine", "get_plan", "GnBlockFn", "Mlp3Fn", "TransolverFn", "SimulatorFn", "IntegratorFn", "ModelFn",
           "unused_param_names"]
