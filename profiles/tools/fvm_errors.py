"""Finite-volume stages alone: the float64 oracle's phi (network output + boundary values) is handed, rounded to fp32, to the
HIP kernels (Engine.fvm_core_fwd) and to the fp32 oracle; gradient, cell values, smoothed node field of both against float64.
usage: python profiles/tools/fvm_errors.py [real_naca0012 | real_cylinder | ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import cases  # noqa: E402
from oracle import fvgn_oracle as O  # noqa: E402
from test_fullsize_gpu import _model, graphs_to  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "real_naca0012"
    graphs = cases.real_mesh(name)[0]
    P = O.init_parameters(cases.WEIGHT_SEED)
    hyper = dict(O.DEFAULT_HYPER, dataset_size=1)
    Pg = {k: v.double() for k, v in P.items()}
    buf = {k: v.double() for k, v in O.new_normalizer_buffers().items()}
    with torch.no_grad():
        out, inter = O.model_forward(Pg, buf, graphs_to(graphs, torch.float64), hyper={"dataset_size": 1}, return_intermediates=True)
    uvp_new = inter["uvp_new"]
    G64 = O.graph_tensors(*graphs_to(graphs, torch.float64))
    nb = G64["node_batch"]
    x = graphs[0].x.double()
    uv_old = x[:, 0:2] / G64["uvp_dim"][nb, 0:2]
    uv_hat = (uv_old + uvp_new[:, 0:2]) / 2.0
    phi64 = torch.cat((uvp_new[:, 0:3], uv_hat, uv_old), -1)
    phi32 = phi64.float()

    def fv(phi, G):
        grad = O.node_based_WLSQ(phi, G["face_node_x"], G["support_edge"], G["A"], G["B1"], G["Bx"], "2nd")[:, :, 0:2]
        phic = O.node_to_cell_2nd_order(phi, grad, G["cells_node"], G["cells_index"], G["pos"], G["centroid"])
        sm = O.cell_to_node_2nd_order(phic[:, 0:3], G["cells_node"], G["cells_index"], G["centroid"], G["pos"])
        return grad, phic, sm

    g64, c64, s64 = fv(phi32.double(), G64)           # exact arithmetic on the SAME fp32 inputs
    G32 = O.graph_tensors(*graphs_to(graphs, torch.float32))
    g32, c32, s32 = fv(phi32, G32)
    model = _model(P)
    eng = model.engine()
    from gfv.plan import get_plan
    hg = tuple(g.clone().to("cuda") for g in graphs)
    pl = get_plan(hg)
    phi8 = torch.zeros((phi32.shape[0], 8), device="cuda")
    phi8[:, 0:7] = phi32.cuda()
    losses, uvp_node, uvp_cell, sv = eng.fvm_core_fwd(phi8, pl, True)
    torch.cuda.synchronize()
    gh = sv["grad"].reshape(-1, 8, 2)[:, 0:7]
    print(f"{name}: finite-volume stages from the same fp32 phi, distance to float64 arithmetic: fp32 oracle | HIP")
    print(f"  WLSQ gradient   {rel(g32, g64):.2e} | {rel(gh, g64):.2e}")
    for c in range(7):
        print(f"    channel {c}     {rel(g32[:, c], g64[:, c]):.2e} | {rel(gh[:, c], g64[:, c]):.2e}   max |grad| {float(g64[:, c].abs().max()):.3e}")
    print(f"  cell values     {rel(c32, c64):.2e} | {rel(sv['phic'][:, 0:7], c64):.2e}")
    print(f"  cell uvp        {rel(c32[:, 0:3], c64[:, 0:3]):.2e} | {rel(sv['phic'][:, 0:3], c64[:, 0:3]):.2e}")
    dimn = (G64["uvp_dim"][nb] * G64["sigma"][nb])
    sm64 = O.enforce_boundary_condition(s64, G64["node_type"], G64["y"]) * dimn
    sm32 = O.enforce_boundary_condition(s32, G32["node_type"], G32["y"]) * dimn.float()
    print(f"  smoothed nodes  {rel(sm32, sm64):.2e} | {rel(uvp_node, sm64):.2e}")
    # where is the worst cell?
    d = (sv["phic"][:, 0:3].double().cpu() - c64[:, 0:3]).abs().max(1).values
    k = int(d.argmax())
    print(f"  worst cell {k}: |diff| {float(d[k]):.3e}, cell area {float(G64['cells_area'].reshape(-1)[k]):.3e}, "
          f"values {c64[k, 0:3].tolist()}")
    cn, ci = G64["cells_node"], G64["cells_index"]
    nodes = cn[ci == k]
    for n in nodes.tolist():
        print(f"    node {n}: type {int(G64['node_type'][n])} grad err HIP {float((gh[n].double().cpu() - g64[n]).abs().max()):.3e} "
              f"oracle32 {float((g32[n].double() - g64[n]).abs().max()):.3e} |grad| {float(g64[n].abs().max()):.3e} "
              f"stencil {int((G64['face_node_x'] == n).sum())}")


if __name__ == "__main__":
    main()
