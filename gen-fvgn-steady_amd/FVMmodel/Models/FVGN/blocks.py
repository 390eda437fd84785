"""EdgeBlock / NodeBlock with the reference's constructor and forward signatures (FVMmodel/Models/FVGN/blocks.py:7-120).

They are parameter containers with the reference module tree (so state_dict keys match) and standalone operators:
`forward(graph_node)` runs the segmented-reduce + fused-MLP HIP kernels of libgfv (via GnBlock's adjoint-complete
path when used inside a GnBlock; standalone they compose the same kernels through gfv.functions)."""
import torch
from FVMmodel.padding import require_native
import torch.nn as nn

from gfv.graph import Data
from gfv import functions as GF
from gfv import ops
from gfv.ops import Seg


def _mlp_param_list(net, prefix):
    names, tensors = [], []
    for n, p in net.named_parameters():
        names.append(f"{prefix}.{n}")
        tensors.append(p)
    return names, tensors


class _EdgeBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, names, x, e, *params):
        GF.require_gpu(x)
        P = dict(zip(names, (p.detach() for p in params)))
        xd, ed = x.detach().contiguous(), e.detach().contiguous()
        nb = ops.seg_gather_sum(xd, plan.n_rowptr, plan.n_col_node, plan.N)
        out, _, sv = engine.mlp3_fwd(P, "blk.net", plan.E, [Seg(nb, plan.es), Seg(nb, plan.er), Seg(ed)])
        ctx.engine, ctx.plan, ctx.names, ctx.sv, ctx.P = engine, plan, names, sv, P
        return out

    @staticmethod
    def backward(ctx, g):
        P, pl = ctx.P, ctx.plan
        grads = GF._alloc_grads(ctx.names, [P[n] for n in ctx.names])
        gnb2 = torch.empty((pl.E, 256), device=g.device)
        ge = torch.empty((pl.E, 128), device=g.device)
        ctx.engine.mlp3_bwd(P, ctx.sv, g.contiguous(), grads, outs=[(gnb2, 256), (gnb2.data_ptr() + 512, 256), ge])
        g_nb = ops.seg_gather_sum(gnb2.view(2 * pl.E, 128), pl.n_rowptr, pl.n_col_edge2, pl.N)
        gx = ops.seg_gather_sum(g_nb, pl.n_rowptr, pl.n_col_node, pl.N)
        return (None, None, None, gx, ge) + tuple(grads.view(n) for n in ctx.names)


class _NodeBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, names, x, e, *params):
        GF.require_gpu(x)
        P = dict(zip(names, (p.detach() for p in params)))
        xd, ed = x.detach().contiguous(), e.detach().contiguous()
        agg = ops.seg_gather_sum(ed.view(2 * plan.E, 64), plan.n_rowptr, plan.n_col_edge2, plan.N)
        nbm = ops.seg_gather_sum(agg, plan.n_rowptr, plan.n_col_node, plan.N, scale=plan.inv_deg)
        out, _, sv = engine.mlp3_fwd(P, "blk.net", plan.N, [Seg(nbm), Seg(xd)])
        ctx.engine, ctx.plan, ctx.names, ctx.sv, ctx.P = engine, plan, names, sv, P
        return out

    @staticmethod
    def backward(ctx, g):
        P, pl = ctx.P, ctx.plan
        dev = g.device
        grads = GF._alloc_grads(ctx.names, [P[n] for n in ctx.names])
        W1 = P["blk.net.0.0.weight"]
        W1t = torch.empty((192, 128), device=dev)
        ops.transpose(W1, out=W1t[0:128], col0=64, ncols=128)
        ops.transpose(W1, out=W1t[128:192], col0=0, ncols=64)
        gx, gnbm = torch.empty((pl.N, 128), device=dev), torch.empty((pl.N, 64), device=dev)
        ctx.engine.mlp3_bwd(P, ctx.sv, g.contiguous(), grads, outs=[gx, (gnbm, 64)], W1t=W1t)
        g_agg = ops.seg_gather_sum(gnbm, pl.n_rowptr, pl.n_col_node, pl.N, src_scale=pl.inv_deg)
        ge = ops.gather_pair(g_agg, pl.es, pl.er)
        return (None, None, None, gx, ge) + tuple(grads.view(n) for n in ctx.names)


def _graph_plan(graph_node):
    from gfv.plan import build_gnn_plan
    return build_gnn_plan(graph_node)


class NodeBlock(nn.Module):
    def __init__(self, input_size, custom_func=None):
        super().__init__()
        self.net = custom_func

    def forward(self, graph_node, graph_cell=None):
        require_native(self.net[0][0].out_features)
        names, tensors = _mlp_param_list(self.net, "blk.net")
        x = _NodeBlockFn.apply(GF.Engine(), _graph_plan(graph_node), names, graph_node.x, graph_node.edge_attr, *tensors)
        return Data(x=x, edge_attr=graph_node.edge_attr, edge_index=graph_node.edge_index, face=graph_node.face,
                    num_graphs=graph_node.num_graphs, batch=graph_node.batch)


class EdgeBlock(nn.Module):
    def __init__(self, input_size=None, custom_func=None):
        super().__init__()
        self.net = custom_func

    def forward(self, graph_node, graph_cell=None):
        require_native(self.net[0][0].out_features)
        names, tensors = _mlp_param_list(self.net, "blk.net")
        e = _EdgeBlockFn.apply(GF.Engine(), _graph_plan(graph_node), names, graph_node.x, graph_node.edge_attr, *tensors)
        return Data(x=graph_node.x, edge_attr=e, edge_index=graph_node.edge_index, face=graph_node.face,
                    num_graphs=graph_node.num_graphs, batch=graph_node.batch)
