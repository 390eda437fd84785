"""`Interplot` with the reference's method signatures (FVMmodel/FVdiscretization/FVInterpolation.py:27-265): the three
2nd-order interpolations of the finite-volume scheme as stand-alone differentiable operators, any channel count, on
libgfv's `gfv_interp2_fwd / gfv_interp2_bwd` (one generic CSR gather with the Taylor correction (x_target - x_source) . grad
and either mean or inverse-distance weights; hand-written adjoint over the transposed incidence, no atomics).

    node_to_cell_2nd_order   :36-109   cell value = mean over the cell's nodes of phi_n + (x_c - x_n) . grad_n
    node_to_face_2nd_order   :111-185  face value = average of the two end nodes' extrapolations (grad may be None)
    cell_to_node_2nd_order   :218-265  node value = inverse-distance weighted mean of the adjacent cells (+ cell_grad)

DELIBERATE DEVIATION (cell_to_node_2nd_order with `cell_grad`): the reference gathers the correction's gradient as
`cell_grad[cells_node]` (FVInterpolation.py:248) - a [C, .., 2] CELL array indexed by NODE ids, which pairs every
(cell, node) incidence with the gradient of an unrelated cell (and reads out of bounds when a node id exceeds the cell
count).  Here the gradient of the incidence's own cell is used, `cell_grad[cells_index]`: the first-order Taylor
correction phi_c + (x_n - x_c) . grad_c the method's name and its two siblings describe.  No caller in the reference
passes `cell_grad` (FVscheme.py:257, importer.py: cell_grad=None), so no result of the training / solve paths depends on
it; tests/test_operators_gpu.py::test_cell_to_node_with_cell_grad_uses_the_cells_own_gradient pins the behaviour.
All inputs are computed in float32 (float64 inputs are converted); positions receive no gradient (mesh geometry is
constant in every caller).

Inside `NNmodel` the same arithmetic runs fused with the fluxes (csrc/fvm.hip face_fwd / cell_fwd / cell_to_node); these are
the operator-API forms.  Hessian corrections (`node_hessian`) are not built: no caller of the reference passes one
(FVscheme.py:101-122 passes None)."""
import torch
from torch import nn

from gfv import functions as GF
from gfv import lib as L

_PLANS = {}


def _incidence(rows_of, cols_of, n_rows, n_src, tag):
    """CSR of the (row, source) incidence list + its transpose, cached on the identity of the index tensors."""
    key = (tag, rows_of.data_ptr(), cols_of.data_ptr(), rows_of._version, cols_of._version, int(rows_of.numel()), n_rows, n_src)
    hit = _PLANS.get(key)
    if hit is not None:
        return hit[2]
    r, c = rows_of.reshape(-1).to(torch.int64), cols_of.reshape(-1).to(torch.int64)
    order = torch.argsort(r, stable=True)
    rp = torch.zeros(n_rows + 1, dtype=torch.int64, device=r.device)
    rp[1:] = torch.cumsum(torch.bincount(r, minlength=n_rows), 0)
    torder = torch.argsort(c, stable=True)
    tp = torch.zeros(n_src + 1, dtype=torch.int64, device=r.device)
    tp[1:] = torch.cumsum(torch.bincount(c, minlength=n_src), 0)
    plan = dict(rowptr=rp.to(torch.int32), col=c[order].to(torch.int32).contiguous(),
                trow=tp.to(torch.int32), tidx=r[torder].to(torch.int32).contiguous())
    if len(_PLANS) > 32:
        _PLANS.pop(next(iter(_PLANS)))
    _PLANS[key] = (rows_of, cols_of, plan)
    return plan


class _Interp2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, phi, grad, srcpos, tgtpos, plan, mode):
        GF.require_gpu(phi)
        lib = L.load()
        S, C = phi.shape
        R = tgtpos.shape[0]
        f = lambda t: t.detach().to(torch.float32).contiguous()
        phi_, sp, tp = f(phi), f(srcpos), f(tgtpos)
        g_ = None if grad is None else f(grad)
        out = torch.empty((R, C), dtype=torch.float32, device=phi.device)
        wsum = torch.empty((R,), dtype=torch.float32, device=phi.device)
        L.check(lib.gfv_interp2_fwd(phi_.data_ptr(), None if g_ is None else g_.data_ptr(), sp.data_ptr(), tp.data_ptr(),
                                    plan["rowptr"].data_ptr(), plan["col"].data_ptr(), mode, out.data_ptr(), wsum.data_ptr(),
                                    R, C, L.stream_ptr()), "interp2_fwd")
        ctx.save = (sp, tp, wsum, plan, mode, S, C, grad is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        sp, tp, wsum, plan, mode, S, C, has_grad = ctx.save
        lib = L.load()
        go = gout.to(torch.float32).contiguous()
        gphi = torch.empty((S, C), dtype=torch.float32, device=go.device)
        ggrad = torch.empty((S, C, 2), dtype=torch.float32, device=go.device) if has_grad else None
        L.check(lib.gfv_interp2_bwd(go.data_ptr(), wsum.data_ptr(), sp.data_ptr(), tp.data_ptr(), plan["trow"].data_ptr(),
                                    plan["tidx"].data_ptr(), mode, gphi.data_ptr(), None if ggrad is None else ggrad.data_ptr(),
                                    S, C, L.stream_ptr()), "interp2_bwd")
        return gphi, ggrad, None, None, None, None


def _phi2d(phi):
    return phi[:, None] if phi.dim() == 1 else phi


class Interplot(nn.Module):
    def __init__(self, mesh_pos=None, centroid=None, cells_node=None, cells_index=None):
        super().__init__()
        self.plotted = False
        self.mesh_pos, self.centroid, self.cells_node, self.cells_index = mesh_pos, centroid, cells_node, cells_index

    def node_to_cell_2nd_order(self, node_phi=None, node_grad=None, node_hessian=None, graph_node=None, graph_cell=None,
                               cells_node=None, cells_index=None, mesh_pos=None, centroid=None):
        if node_hessian is not None:
            raise NotImplementedError("Hessian correction: no caller of the reference passes one (FVscheme.py:101-107)")
        if (cells_node is None) and (cells_index is None) and (mesh_pos is None) and (centroid is None):
            cells_node, cells_index = graph_node.face, graph_cell.face
            mesh_pos, centroid = graph_node.pos, graph_cell.pos
        plan = _incidence(cells_index, cells_node, centroid.shape[0], mesh_pos.shape[0], "n2c")
        return _Interp2.apply(_phi2d(node_phi), node_grad, mesh_pos, centroid, plan, 0)

    def node_to_face_2nd_order(self, node_phi=None, node_grad=None, node_hessian=None, graph_node=None, graph_edge=None):
        if node_hessian is not None:
            raise NotImplementedError("Hessian correction: no caller of the reference passes one (FVscheme.py:109-122)")
        ei = graph_node.edge_index
        E = ei.shape[1]
        key = ("n2f_idx", ei.data_ptr(), ei._version, E)
        hit = _PLANS.get(key)
        if hit is None:
            hit = (ei, torch.arange(E, device=ei.device).repeat(2), torch.cat((ei[0], ei[1])))
            _PLANS[key] = hit
        _, rows, cols = hit
        plan = _incidence(rows, cols, E, graph_node.pos.shape[0], "n2f")
        phi = node_phi
        if phi.dim() == 3:     # a gradient field [N, C, 2] interpolated without correction (FVscheme.py:117-122)
            N, C, D = phi.shape
            out = _Interp2.apply(phi.reshape(N, C * D), None, graph_node.pos, graph_edge.pos, plan, 0)
            return out.reshape(E, C, D)
        return _Interp2.apply(_phi2d(phi), node_grad, graph_node.pos, graph_edge.pos, plan, 0)

    def cell_to_node_2nd_order(self, cell_phi=None, cell_grad=None, cells_node=None, cells_index=None, centroid=None,
                               mesh_pos=None):
        if (cell_grad is not None) and (len(cell_grad.size()) < 3):
            raise ValueError("cell_grad must be 3 dim [N,C,2] N is the number of cells, C is the number of variables")
        plan = _incidence(cells_node, cells_index, mesh_pos.shape[0], centroid.shape[0], "c2n")
        return _Interp2.apply(_phi2d(cell_phi), cell_grad, centroid, mesh_pos, plan, 1)
