"""`Intergrator` with the reference's signature (FVMmodel/FVdiscretization/FVscheme.py:23-30,618-724): WLSQ gradient
reconstruction + conserved-form finite-volume residuals, as HIP kernels with hand-written adjoints (csrc/fvm.hip).

Stand-alone operator form of what `NNmodel.forward` runs fused: takes the already clamped / BC-enforced node fields.
The node / cell field outputs are returned detached (the reference's drivers detach them)."""
import torch
import torch.nn as nn

from gfv import functions as GF
from gfv.plan import get_plan


class _IntegratorPhiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, plan, phi8):
        GF.require_gpu(phi8)
        # raw_outputs: the smoothed node field BEFORE the Dirichlet overwrite and both fields un-scaled, as the reference's
        # Intergrator returns them (FVscheme.py:253-262,718-724; importer.py:223-231 applies both afterwards)
        losses, uvp_node, uvp_cell, sv = engine.fvm_core_fwd(phi8.detach().contiguous(), plan, raw_outputs=True)
        ctx.engine, ctx.plan, ctx.sv = engine, plan, sv
        ctx.mark_non_differentiable(uvp_node, uvp_cell)
        return losses, uvp_node, uvp_cell

    @staticmethod
    def backward(ctx, g_losses, _gn, _gc):
        return None, None, ctx.engine.fvm_core_bwd(ctx.sv, g_losses.contiguous(), ctx.plan)


class Intergrator(nn.Module):
    def __init__(self):
        super().__init__()
        self.epoch = 0

    def forward(self, uvp_new_node=None, uv_hat_node=None, uv_old_node=None, graph_node=None, graph_node_x=None,
                graph_edge=None, graph_cell=None, graph_Index=None, params=None):
        plan = get_plan((graph_node, graph_node_x, graph_edge, graph_cell, graph_Index))
        n = uvp_new_node.shape[0]
        phi8 = torch.cat((uvp_new_node[:, 0:3], uv_hat_node[:, 0:2], uv_old_node[:, 0:2],
                          torch.zeros((n, 1), dtype=uvp_new_node.dtype, device=uvp_new_node.device)), dim=-1)
        eng = GF.Engine(ncn_smooth=getattr(params, "ncn_smooth", True),
                        conserved_form=bool(getattr(params, "conserved_form", True)),   # FVscheme.py:671-715
                        order=getattr(params, "order", "2nd"))                          # FVscheme.py:653
        losses, uvp_node, uvp_cell = _IntegratorPhiFn.apply(eng, plan, phi8)
        return (losses[:, 0:1], losses[:, 1:2], losses[:, 2:3], losses[:, 3:4], uvp_node, uvp_cell)
