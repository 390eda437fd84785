"""`node_based_WLSQ` with the reference's signature (FVMmodel/FVdiscretization/FVgrad.py:235-244), both branches (moments
handed in, FVgrad.py:295-325, or built here from the node positions, :273-294), `rt_cond`, orders 1st .. 4th (M = 2 / 5 / 9 / 14 Taylor terms, FVorder.py:23-72): HIP kernels `gfv_wlsq_fwd_ex` /
`gfv_wlsq_bwd_ex` (CSR-ordered stencil gather, per-node M x M LU with partial pivoting on the row-normalised moment matrix,
transpose solve for the adjoint)."""
import torch

from gfv import functions as GF
from gfv import lib as L
from gfv.plan import MeshPlan, wlsq_part


class _WlsqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, phi):
        GF.require_gpu(phi)
        lib = L.load()
        N, C = phi.shape
        phi8 = torch.zeros((N, 8), dtype=torch.float32, device=phi.device)
        phi8[:, :C] = phi.detach()
        grad = torch.empty((N, 16), dtype=torch.float32, device=phi.device)
        full = torch.empty((N, 8, plan.M), dtype=torch.float32, device=phi.device)
        L.check(lib.gfv_wlsq_fwd_ex(phi8.data_ptr(), plan.x_rowptr.data_ptr(), plan.x_out.data_ptr(), plan.x_B.data_ptr(),
                                    plan.An.data_ptr(), plan.rn.data_ptr(), grad.data_ptr(), full.data_ptr(), N, plan.M,
                                    L.stream_ptr()), "wlsq_fwd_ex")
        ctx.plan, ctx.C = plan, C
        return full[:, :C, :].clone()

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        plan, C = ctx.plan, ctx.C
        N = g.shape[0]
        g8 = torch.zeros((N, 8, plan.M), dtype=torch.float32, device=g.device)
        g8[:, :C, :] = g
        gphi = torch.zeros((N, 8), dtype=torch.float32, device=g.device)
        ws = torch.empty((N, 8, plan.M), dtype=torch.float32, device=g.device)
        L.check(lib.gfv_wlsq_bwd_ex(None, g8.data_ptr(), plan.An.data_ptr(), plan.rn.data_ptr(), plan.xo_rowptr.data_ptr(),
                                    plan.xo_in.data_ptr(), plan.xo_B.data_ptr(), plan.sumB.data_ptr(), ws.data_ptr(),
                                    gphi.data_ptr(), N, plan.M, L.stream_ptr()), "wlsq_bwd_ex")
        return None, gphi[:, :C]


def node_based_WLSQ(phi_node=None, edge_index=None, extra_edge_index=None, mesh_pos=None, order=None,
                    precompute_Moments: list = None, periodic_idx=None, rt_cond=False):
    if (order is None) or (order not in ["1st", "2nd", "3rd", "4th"]):
        raise ValueError("order must be specified in [\"1st\", \"2nd\", \"3rd\", \"4th\"]")   # FVgrad.py:261-262
    if precompute_Moments is None:
        # FVgrad.py:273-294 -> compute_normal_matrix (:183-232) -> moments_order (FVorder.py:7-86): the moment matrices
        # from the node positions, on the device (gfv/device_prep.py: sorts + prefix sums, no atomics), then the same solve
        from gfv import device_prep
        if extra_edge_index is None:
            extra_edge_index = torch.zeros((2, 0), dtype=edge_index.dtype, device=edge_index.device)
        A64, B1, Bx = device_prep.wlsq_moments(mesh_pos.to(torch.float64), edge_index, extra_edge_index, order)
        precompute_Moments = [A64.to(torch.float32), B1.to(torch.float32), Bx.to(torch.float32)]
    terms = {"1st": 2, "2nd": 5, "3rd": 9, "4th": 14}[order]
    if precompute_Moments[0].shape[-1] != terms:
        raise ValueError(f"moment matrices are {precompute_Moments[0].shape[-1]} wide, order {order} needs {terms}")
    if phi_node.shape[1] > 7:
        raise NotImplementedError("at most 7 channels (FVscheme.py:643-646)")
    A, B1, Bx = precompute_Moments
    if extra_edge_index is None:
        extra_edge_index = torch.zeros((2, 0), dtype=edge_index.dtype, device=edge_index.device)
        Bx = Bx[:0]
    plan = wlsq_part(MeshPlan(), edge_index, extra_edge_index, A, B1, Bx, phi_node.shape[0])
    out = _WlsqFn.apply(plan, phi_node)
    if rt_cond:
        # FVgrad.py:363-364: the 2-norm condition number of the row-normalised moment matrices (a diagnostic)
        Af = A.to(torch.float32)
        An = Af / (torch.norm(Af, p=2, dim=2, keepdim=True) + 1e-8)
        return out, torch.linalg.cond(An)
    return out
