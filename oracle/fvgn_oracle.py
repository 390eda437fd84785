"""ORACLE - CPU restatement of the Gen-FVGN training-step hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file, and only
as the checker / the timed CPU baseline.  The product path (``gen-fvgn-steady_amd/``) never imports it and fails
loudly when its HIP extension is missing.

What it is: a plain fp32 PyTorch (core ops only: indexing, ``index_add_``, ``linalg.solve``, ``nn.functional``)
restatement of the reference's algorithm for SURVEY.md section 8 rows a-1 ... a-15, each function citing the
reference ``file:line`` it follows (paths relative to ``/root/reference/src``).  It is functional: weights come in
as a dict keyed by the reference's ``state_dict`` names, so a reference checkpoint drives it directly.  Gradients
come from torch autograd over these ops, which is exactly what the reference does (pre_train_Adam.py:188).

Parity pin: ``tests/golden/make_golden.py`` runs the *reference itself* (imported through stubs in the build
container) and this restatement on identical meshes/weights and commits the reference's outputs as golden vectors
(``tests/golden/*.npz``); ``tests/test_oracle_golden.py`` re-checks the oracle against them on every run.
The arithmetic that lives in un-vendored third-party wheels (torch_scatter, torch_geometric.global_add_pool) is
restated from those libraries' published semantics (scatter-sum / scatter-mean with count clamp >= 1).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

# NodeType, utils/utilities.py:7-13
NORMAL, INFLOW, OUTFLOW, WALL_BOUNDARY, PRESS_POINT, IN_WALL = 0, 1, 2, 3, 4, 5

DEFAULT_HYPER = dict(  # utils/get_param.py:37-75
    net="TransFVGN_v2", message_passing_num=3, hidden_size=128, node_input_size=12, node_output_size=3,
    node_phi_size=3, dataset_size=100, integrator="imex", order="2nd", conserved_form=True, ncn_smooth=True,
    loss_cont=6e4, loss_mom=5e4, loss_press=1.0, lr=5e-5, heads=8, slice_num=32,
)


# --------------------------------------------------------------------------------------------------------------
# third-party primitives restated (torch_scatter / torch_geometric published semantics)
# --------------------------------------------------------------------------------------------------------------
def scatter_add(src, index, dim_size):
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def scatter_mean(src, index, dim_size):
    s = scatter_add(src, index, dim_size)
    cnt = torch.zeros((dim_size,), dtype=src.dtype, device=src.device).index_add_(
        0, index, torch.ones_like(index, dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return s / cnt.view(-1, *([1] * (src.dim() - 1)))


def global_add_pool(x, batch, size):
    return scatter_add(x, batch, size)


# --------------------------------------------------------------------------------------------------------------
# a-2 / a-3 / a-4: input normalisation and edge features
# --------------------------------------------------------------------------------------------------------------
def normalize_graph_features(x, batch, num_graphs):
    """importer.py:80-93 - per-graph mean / population std, eps added to std."""
    mean = scatter_mean(x, batch, num_graphs)
    residual = x - mean[batch]
    var = scatter_mean(residual ** 2, batch, num_graphs)
    std = torch.sqrt(var)
    return residual / (std[batch] + 1e-8)


def normalizer_forward(buffers, x, max_accumulations, accumulate=True):
    """utils/normalization.py:32-85.  buffers: dict acc_count, num_accumulations, acc_sum, acc_sum_squared (in place)."""
    if accumulate and float(buffers["num_accumulations"]) < max_accumulations:
        buffers["acc_sum"] += x.sum(dim=0)
        buffers["acc_sum_squared"] += (x ** 2).sum(dim=0)
        buffers["acc_count"] += float(x.shape[0])
        buffers["num_accumulations"] += 1
    safe = torch.clamp(buffers["acc_count"], min=1.0)
    mean = buffers["acc_sum"] / safe
    std = torch.sqrt(buffers["acc_sum_squared"] / safe - mean ** 2)
    std = torch.where(std < 1e-8, torch.ones_like(std), std)
    return (x - mean) / std


def relative_edge_attr(x, pos, edge_index):
    """importer.py:54-78."""
    s, r = edge_index[0], edge_index[1]
    dpos = pos[s] - pos[r]
    return torch.cat((x[s] - x[r], dpos, torch.norm(dpos, p=2, dim=-1, keepdim=True)), dim=-1)


# --------------------------------------------------------------------------------------------------------------
# a-5 ... a-11: the GNN
# --------------------------------------------------------------------------------------------------------------
def mlp3(P, prefix, x, layer_norm=True):
    """build_mlp, EPD.py:10-33: Linear-GELU(erf)-Linear-GELU-Linear [-LayerNorm(eps 1e-5)]."""
    lin = prefix + ".0" if layer_norm else prefix
    h = F.linear(x, P[f"{lin}.0.weight"], P[f"{lin}.0.bias"])
    h = F.gelu(h)
    h = F.linear(h, P[f"{lin}.2.weight"], P[f"{lin}.2.bias"])
    h = F.gelu(h)
    h = F.linear(h, P[f"{lin}.4.weight"], P[f"{lin}.4.bias"])
    if layer_norm:
        h = F.layer_norm(h, (h.shape[-1],), P[f"{prefix}.1.weight"], P[f"{prefix}.1.bias"], 1e-5)
    return h


def edge_block(P, prefix, x, e, edge_index):
    """blocks.py:71-120 - neighbour SUM over both directions, then MLP on [nb[s] | nb[r] | e]."""
    s, r = edge_index[0], edge_index[1]
    indeg, outdeg = torch.cat((s, r)), torch.cat((r, s))
    nb = scatter_add(x[outdeg], indeg, x.shape[0])
    return mlp3(P, prefix + ".net", torch.cat((nb[s], nb[r], e), dim=1))


def node_block(P, prefix, x, e, edge_index):
    """blocks.py:13-63 - first half of the edge channels to senders, second half to receivers; then
    neighbour MEAN (count clamp >= 1) of those aggregates; MLP on [nbm | x]."""
    s, r = edge_index[0], edge_index[1]
    indeg, outdeg = torch.cat((s, r)), torch.cat((r, s))
    twoway = torch.cat(torch.chunk(e, 2, dim=-1), dim=0)
    agg = scatter_add(twoway, indeg, x.shape[0])
    nbm = scatter_mean(agg[outdeg], indeg, x.shape[0])
    return mlp3(P, prefix + ".net", torch.cat((nbm, x), dim=1))


def gn_block(P, prefix, x, e, edge_index):
    """EPD.py:177-195 - EdgeBlock, NodeBlock (sees the NEW edges), residuals."""
    e_new = edge_block(P, prefix + ".eb_module", x, e, edge_index)
    x_new = node_block(P, prefix + ".nb_module", x, e_new, edge_index)
    return x + x_new, e + e_new


def physics_attention(P, prefix, x, batch, num_graphs, heads=8):
    """GraphTransolver.py:48-95 (Graph_Physics_Attention_1D.graph_forward)."""
    n, dim = x.shape
    dh = dim // heads
    fx_mid = F.linear(x, P[f"{prefix}.in_project_fx.weight"], P[f"{prefix}.in_project_fx.bias"]).view(n, heads, dh)
    x_mid = F.linear(x, P[f"{prefix}.in_project_x.weight"], P[f"{prefix}.in_project_x.bias"]).view(n, heads, dh)
    logits = F.linear(x_mid, P[f"{prefix}.in_project_slice.weight"], P[f"{prefix}.in_project_slice.bias"])
    w = torch.softmax(logits / P[f"{prefix}.graph_temperature"], dim=-1)          # [n,H,G]
    slice_norm = scatter_add(w, batch, num_graphs)                                 # [B,H,G]
    token = scatter_add(fx_mid.unsqueeze(-2) * w.unsqueeze(-1), batch, num_graphs)  # [B,H,G,dh]
    token = token / (slice_norm.unsqueeze(-1) + 1e-5)
    q = F.linear(token, P[f"{prefix}.to_q.weight"])
    k = F.linear(token, P[f"{prefix}.to_k.weight"])
    v = F.linear(token, P[f"{prefix}.to_v.weight"])
    dots = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)
    attn = torch.softmax(dots, dim=-1)
    out_token = torch.matmul(attn, v)                                              # [B,H,G,dh]
    out_x = torch.sum(out_token[batch] * w.unsqueeze(-1), dim=-2).reshape(n, dim)  # [n, H*dh]
    return F.linear(out_x, P[f"{prefix}.to_out.0.weight"], P[f"{prefix}.to_out.0.bias"])


def transolver_block(P, prefix, fx, batch, num_graphs, heads=8):
    """GraphTransolver.py:163-169 with in_layernorm=False (ln_1 unused), MLP n_layers=0 (:98-128)."""
    fx = physics_attention(P, prefix + ".Attn", fx, batch, num_graphs, heads) + fx
    h = F.layer_norm(fx, (fx.shape[-1],), P[f"{prefix}.ln_2.weight"], P[f"{prefix}.ln_2.bias"], 1e-5)
    h = F.gelu(F.linear(h, P[f"{prefix}.mlp.linear_pre.0.weight"], P[f"{prefix}.mlp.linear_pre.0.bias"]))
    h = F.linear(h, P[f"{prefix}.mlp.linear_post.weight"], P[f"{prefix}.mlp.linear_post.bias"])
    return h + fx


def decoder(P, prefix, x):
    """EPD.py:198-219 - build_mlp_from_num_layer(num_layer=2, lay_norm=False): 128-128-128-3."""
    return mlp3(P, prefix + ".node_decode_module", x, layer_norm=False)


def processor_prefixes(net, prefix="simulator"):
    """Parameter prefixes of the (GnBlocks + Transolver block) processors: TransFVGN_v2 has two AttnProcessors
    (TransFVGN_v2.py:69-76), TransFVGN_v1 one, whose modules hang directly off the simulator (TransFVGN_v1.py:30-46)."""
    if net in ("TransFVGN_v2", "TransFVGN"):
        return [f"{prefix}.processpr_list.0", f"{prefix}.processpr_list.1"]
    if net == "TransFVGN_v1":
        return [prefix]
    raise NotImplementedError(net)


def simulator_v2(P, x, edge_attr, edge_index, batch, num_graphs, mp=3, heads=8, prefix="simulator",
                 return_intermediates=False, net="TransFVGN_v2"):
    """TransFVGN_v2.py:54-105 - Encoder, 2 x AttnProcessor(mp GnBlocks + Transolver block), Decoder;
    net="TransFVGN_v1" (TransFVGN_v1.py:53-74): the same with ONE processor."""
    inter = {}
    xn = mlp3(P, f"{prefix}.encoder.nb_encoder", x)            # EPD.py:118
    en = mlp3(P, f"{prefix}.encoder.eb_encoder", edge_attr)    # EPD.py:119
    inter["enc_x"], inter["enc_e"] = xn, en
    for ip, pp in enumerate(processor_prefixes(net, prefix)):
        emb = xn
        for ig in range(mp):
            xn, en = gn_block(P, f"{pp}.GN_block_list.{ig}", xn, en, edge_index)
            inter[f"p{ip}.gn{ig}.x"], inter[f"p{ip}.gn{ig}.e"] = xn, en
        xn = transolver_block(P, f"{pp}.TransBlock", xn + emb, batch, num_graphs, heads)
        inter[f"p{ip}.trans.x"] = xn
    out = decoder(P, f"{prefix}.decoder", xn)
    return (out, inter) if return_intermediates else out


def enforce_boundary_condition(uvp, node_type, y):
    """importer.py:141-154 (functional: returns a new tensor)."""
    dirichlet = ((node_type == WALL_BOUNDARY) | (node_type == INFLOW) | (node_type == PRESS_POINT)
                 | (node_type == IN_WALL))
    press = node_type == PRESS_POINT
    uv = torch.where(dirichlet.unsqueeze(1), y[:, 0:2], uvp[:, 0:2])
    p = torch.where(press.unsqueeze(1), torch.zeros_like(uvp[:, 2:3]), uvp[:, 2:3])
    return torch.cat((uv, p), dim=1)


# --------------------------------------------------------------------------------------------------------------
# a-12: WLSQ gradient reconstruction, precomputed-moments branch
# --------------------------------------------------------------------------------------------------------------
def wlsq_stencil(face_node_x, support_edge):
    """FVgrad.py:264-271: directed stencil = [fx, fx.flip(0), support_edge]; returns (out_idx, in_idx)."""
    comp = torch.cat((face_node_x, face_node_x.flip(0), support_edge), dim=1)
    return comp[0], comp[1]


def wlsq_full_B(B1, Bextra, order=2):
    """FVgrad.py:299-312: reverse-direction rows have the odd-order moments negated."""
    Brev = B1.clone()
    Brev[:, 0:2] *= -1
    if order >= 3:
        Brev[:, 5:9] *= -1
    return torch.cat((B1, Brev, Bextra), dim=0)


def node_based_WLSQ(phi, face_node_x, support_edge, A, B1, Bextra, order="2nd"):
    """FVgrad.py:295-325,335-359: rhs = scatter_add(B * (phi[out]-phi[in])); row-normalise; solve; transpose."""
    out_idx, in_idx = wlsq_stencil(face_node_x, support_edge)
    Bfull = wlsq_full_B(B1, Bextra, int(order[0]))
    diff = Bfull * (phi[out_idx] - phi[in_idx]).unsqueeze(1)          # [S,5,C]
    rhs = scatter_add(diff, in_idx, phi.shape[0])                     # [N,5,C]
    row_norms = torch.norm(A, p=2, dim=2, keepdim=True)
    A_n = A / (row_norms + 1e-8)
    rhs_n = rhs / (row_norms + 1e-8)
    return torch.linalg.solve(A_n, rhs_n).transpose(1, 2)             # [N,C,5]


# --------------------------------------------------------------------------------------------------------------
# a-13: interpolation
# --------------------------------------------------------------------------------------------------------------
def node_to_cell_2nd_order(phi, grad, cells_node, cells_index, pos, centroid):
    """FVInterpolation.py:36-109 + utilities.py:16-35 (mean over the cell's nodes)."""
    r = centroid[cells_index] - pos[cells_node]                        # [S,2]
    val = phi[cells_node] + (grad[cells_node] * r.unsqueeze(1)).sum(-1)
    return scatter_mean(val, cells_index, centroid.shape[0])


def node_to_face_2nd_order(phi, grad, edge_index, pos, face_pos):
    """FVInterpolation.py:111-185: average of the two end-node extrapolations (grad may be None)."""
    two = torch.cat((edge_index[0], edge_index[1]), dim=0)
    val = phi[two]
    if grad is not None:
        r = face_pos.repeat(2, 1) - pos[two]
        val = val + (grad[two] * r.unsqueeze(1)).sum(-1)
    ne = edge_index.shape[1]
    return (val[:ne] + val[ne:]) / 2.0


def cell_to_node_2nd_order(cell_phi, cells_node, cells_index, centroid, pos):
    """FVInterpolation.py:218-265 (cell_grad=None): inverse-distance weighted mean of the adjacent cells."""
    d = pos[cells_node] - centroid[cells_index]
    w = 1.0 / torch.norm(d, dim=-1, keepdim=True)
    num = scatter_add(cell_phi[cells_index] * w, cells_node, pos.shape[0])
    den = scatter_add(w, cells_node, pos.shape[0])
    return num / den


# --------------------------------------------------------------------------------------------------------------
# a-14: integrator, conserved form
# --------------------------------------------------------------------------------------------------------------
def fix_face_flux_bc(face_uv, face_type, y_face):
    """FVscheme.py:32-48 (functional)."""
    inflow = (face_type == INFLOW).unsqueeze(1)
    wall = (face_type == WALL_BOUNDARY).unsqueeze(1)
    out = torch.where(inflow, y_face[:, 0:2], face_uv)
    return torch.where(wall, torch.zeros_like(out), out)


def integrator_conserved(uvp_new, uv_hat, uv_old, G, hyper, return_intermediates=False):
    """FVscheme.py:618-724 (forward) -> conserved_form (:50-274).  G = dict of the batched graph tensors."""
    phi = torch.cat((uvp_new[:, 0:3], uv_hat[:, 0:2], uv_old[:, 0:2]), dim=-1)           # :643-646
    grad_l = node_based_WLSQ(phi, G["face_node_x"], G["support_edge"], G["A"], G["B1"], G["Bx"], hyper["order"])
    grad = grad_l[:, :, 0:2]                                                             # :658

    cells_node, cells_face, cells_index = G["cells_node"], G["cells_face"], G["cells_index"]
    edge_index, face_type = G["edge_index"], G["face_type"]
    B = G["num_graphs"]
    C = G["centroid"].shape[0]
    theta_c = G["theta_PDE"][G["cell_batch"]]
    cells_area = G["cells_area"].view(-1, 1)
    Svec = G["cells_face_unv"].view(-1, 2) * G["face_area"].view(-1, 1)[cells_face]      # :89
    unsteady, conv_c, gradp_c, diff_c = theta_c[:, 0:1], theta_c[:, 2:3], theta_c[:, 3:4], theta_c[:, 4:5]
    source = theta_c[:, 5:6] * cells_area
    dt_cell = G["dt_graph"][G["cell_batch"], :]

    phi_cell = node_to_cell_2nd_order(phi, grad, cells_node, cells_index, G["pos"], G["centroid"])   # :101
    phi_face = node_to_face_2nd_order(phi[:, 0:5], grad[:, 0:5], edge_index, G["pos"], G["face_pos"])  # :109
    grad_face = node_to_face_2nd_order(grad[:, 0:5], None, edge_index, G["pos"], G["face_pos"])        # :117

    y_face = (G["y"][edge_index[0]] + G["y"][edge_index[1]]) / 2.0
    uv_face_new = fix_face_flux_bc(phi_face[:, 0:2], face_type, y_face)                  # :125
    uv_face_hat = fix_face_flux_bc(phi_face[:, 3:5], face_type, y_face)                  # :128
    p_face_new = phi_face[:, 2:3]
    uvp_cell_new, uv_cell_old = phi_cell[:, 0:3], phi_cell[:, 5:7]
    nabla_uvp_face, nabla_uv_face_hat = grad_face[:, 0:3], grad_face[:, 3:5]

    # pressure outlet (:145-167)
    out_mask = face_type[cells_face] == OUTFLOW
    if bool(out_mask.any()):
        visc = diff_c[cells_index] * torch.matmul(nabla_uvp_face[cells_face, 0:2], Svec.unsqueeze(2)).squeeze(2)
        surface_p = p_face_new[cells_face, :] * Svec
        lp = (visc - surface_p)[out_mask]
        loss_press = torch.sqrt(global_add_pool(lp ** 2, G["edge_batch"][cells_face[out_mask]], B)
                                .sum(dim=-1, keepdim=True))
    else:
        loss_press = torch.zeros((B, 1), dtype=phi.dtype, device=phi.device)

    unsteady_cell = ((uvp_cell_new[:, 0:2] - uv_cell_old) / dt_cell) * cells_area       # :170
    div = scatter_add((uv_face_new[cells_face] * Svec).sum(-1), cells_index, C).view(-1, 1)  # :174-183
    loss_cont = torch.sqrt(global_add_pool(div ** 2, G["cell_batch"], B)) * G["theta_PDE"][:, 1:2]

    uu = uv_face_hat.unsqueeze(2) * uv_face_hat.unsqueeze(1)                             # :195 [E,2,2]
    conv_flux = uu[cells_face] * conv_c[cells_index].unsqueeze(1)
    vis_flux = nabla_uv_face_hat[cells_face] * diff_c[cells_index, None]
    P_flux = torch.diag_embed(p_face_new[cells_face].expand(-1, 2)) * gradp_c[cells_index, None]
    J = torch.matmul(conv_flux + P_flux - vis_flux, Svec.unsqueeze(-1)).squeeze(-1)      # :227
    total_rhs = scatter_add(J, cells_index, C) - source                                  # :232-240
    mom = unsteady * unsteady_cell + total_rhs
    loss_mom = torch.sqrt(global_add_pool(mom ** 2, G["cell_batch"], B)) * G["sigma"][:, 0:2]  # :243-247

    if hyper["ncn_smooth"]:
        rt = cell_to_node_2nd_order(uvp_cell_new[:, 0:3], cells_node, cells_index, G["centroid"], G["pos"])
    else:
        rt = uvp_new
    res = (loss_cont, loss_mom[:, 0:1], loss_mom[:, 1:2], loss_press, rt, uvp_cell_new)
    if return_intermediates:
        return res, dict(grad=grad, phi_cell=phi_cell, phi_face=phi_face, grad_face=grad_face, div=div, mom=mom)
    return res


def integrator_non_conserved(uvp_new, uv_hat, uv_old, G, hyper, return_intermediates=False):
    """FVscheme.py:618-724 (forward) -> non_conserved_form (:276-511) with hessian_phi = None, as the reference passes
    it (:660-669): gradient-based continuity, convection and pressure terms from the CELL-averaged node gradients,
    divergence-form diffusion and the pressure-outlet condition from the face values as in the conserved form."""
    phi = torch.cat((uvp_new[:, 0:3], uv_hat[:, 0:2], uv_old[:, 0:2]), dim=-1)
    grad_l = node_based_WLSQ(phi, G["face_node_x"], G["support_edge"], G["A"], G["B1"], G["Bx"], hyper["order"])
    grad = grad_l[:, :, 0:2]
    cells_node, cells_face, cells_index = G["cells_node"], G["cells_face"], G["cells_index"]
    edge_index, face_type = G["edge_index"], G["face_type"]
    B = G["num_graphs"]
    C = G["centroid"].shape[0]
    theta_c = G["theta_PDE"][G["cell_batch"]]
    cells_area = G["cells_area"].view(-1, 1)
    Svec = G["cells_face_unv"].view(-1, 2) * G["face_area"].view(-1, 1)[cells_face]
    unsteady, conv_c, gradp_c, diff_c = theta_c[:, 0:1], theta_c[:, 2:3], theta_c[:, 3:4], theta_c[:, 4:5]
    source = theta_c[:, 5:6] * cells_area
    dt_cell = G["dt_graph"][G["cell_batch"], :]

    phi_cell = node_to_cell_2nd_order(phi, grad, cells_node, cells_index, G["pos"], G["centroid"])       # :326-332
    uvp_cell_new, uv_cell_hat, uv_cell_old = phi_cell[:, 0:3], phi_cell[:, 3:5], phi_cell[:, 5:7]
    phi_face = node_to_face_2nd_order(phi[:, 0:5], grad[:, 0:5], edge_index, G["pos"], G["face_pos"])     # :338-344
    p_face_new = phi_face[:, 2:3]
    grad_face = node_to_face_2nd_order(grad[:, 0:5], None, edge_index, G["pos"], G["face_pos"])           # :349-354
    grad_cell = scatter_mean(grad[:, 0:5][cells_node], cells_index, C)                                   # :356-361
    nabla_uvp_face, nabla_uvp_cell = grad_face[:, 0:3], grad_cell[:, 0:3]
    nabla_uv_face_hat, nabla_uv_cell_hat = grad_face[:, 3:5], grad_cell[:, 3:5]

    out_mask = face_type[cells_face] == OUTFLOW                                                         # :375-398
    if bool(out_mask.any()):
        visc = diff_c[cells_index] * torch.matmul(nabla_uvp_face[cells_face, 0:2], Svec.unsqueeze(2)).squeeze(2)
        surface_p = p_face_new[cells_face, :] * Svec
        lp = (visc - surface_p)[out_mask]
        loss_press = torch.sqrt(global_add_pool(lp ** 2, G["edge_batch"][cells_face[out_mask]], B)
                                .sum(dim=-1, keepdim=True))
    else:
        loss_press = torch.zeros((B, 1), dtype=phi.dtype, device=phi.device)

    unsteady_cell = ((uvp_cell_new[:, 0:2] - uv_cell_old) / dt_cell) * cells_area                        # :401
    div = (nabla_uvp_cell[:, 0:1, 0] + nabla_uvp_cell[:, 1:2, 1]) * cells_area                           # :405-408
    loss_cont = torch.sqrt(global_add_pool(div ** 2, G["cell_batch"], B)) * G["theta_PDE"][:, 1:2]
    conv = torch.matmul(nabla_uv_cell_hat, uv_cell_hat.unsqueeze(2)).squeeze(2) * cells_area             # :447-450
    gradp = nabla_uvp_cell[:, 2] * cells_area                                                            # :454
    visc_face = torch.matmul(nabla_uv_face_hat[cells_face, 0:2], Svec.unsqueeze(2)).squeeze(2)           # :458-461
    visc_force = scatter_add(visc_face, cells_index, C)                                                  # :463-469
    mom = unsteady * unsteady_cell + conv_c * conv + gradp_c * gradp - diff_c * visc_force - source       # :472-478
    loss_mom = torch.sqrt(global_add_pool(mom ** 2, G["cell_batch"], B)) * G["sigma"][:, 0:2]

    if hyper["ncn_smooth"]:
        rt = cell_to_node_2nd_order(uvp_cell_new[:, 0:3], cells_node, cells_index, G["centroid"], G["pos"])
    else:
        rt = uvp_new
    res = (loss_cont, loss_mom[:, 0:1], loss_mom[:, 1:2], loss_press, rt, uvp_cell_new)
    if return_intermediates:
        return res, dict(grad=grad, phi_cell=phi_cell, grad_cell=grad_cell, div=div, mom=mom)
    return res


# --------------------------------------------------------------------------------------------------------------
# a-1: model forward, a-15: loss / Adam
# --------------------------------------------------------------------------------------------------------------
def graph_tensors(graph_node, graph_node_x, graph_edge, graph_cell, graph_Index):
    """Flatten the five graph objects into the dict the oracle functions use."""
    return dict(
        edge_index=graph_node.edge_index, cells_node=graph_node.face, pos=graph_node.pos,
        node_type=graph_node.node_type, y=graph_node.y, node_batch=graph_node.batch,
        face_node_x=graph_node_x.face_node_x, support_edge=graph_node_x.support_edge,
        A=graph_node_x.A_node_to_node, B1=graph_node_x.single_B_node_to_node, Bx=graph_node_x.extra_B_node_to_node,
        face_type=graph_edge.face_type, face_area=graph_edge.face_area, cells_face=graph_edge.face,
        face_pos=graph_edge.pos, edge_batch=graph_edge.batch,
        cells_face_unv=graph_cell.cells_face_unv, cells_area=graph_cell.cells_area, centroid=graph_cell.pos,
        cells_index=graph_cell.face, cell_batch=graph_cell.batch,
        theta_PDE=graph_Index.theta_PDE, sigma=graph_Index.sigma, uvp_dim=graph_Index.uvp_dim,
        dt_graph=graph_Index.dt_graph, num_graphs=int(graph_Index.theta_PDE.shape[0]),
    )


def model_forward(P, buffers, graphs, hyper=None, norm_uvp=True, norm_global=True, return_intermediates=False):
    """importer.py:156-240 (training branch).  P: parameters, buffers: Normalizer state (updated in place).

    Like the reference, writes the normalised features into ``graph_node.x`` and sets ``graph_node.edge_attr``."""
    hyper = {**DEFAULT_HYPER, **(hyper or {})}
    graph_node, graph_node_x, graph_edge, graph_cell, graph_Index = graphs
    G = graph_tensors(*graphs)
    B = G["num_graphs"]
    nb = G["node_batch"]
    x = graph_node.x
    uv_old = x[:, 0:2] / G["uvp_dim"][nb, 0:2]                                            # :168-170
    if not norm_uvp:
        raise ValueError("The graph node features have already been normalized")          # :123-124
    nphi = hyper["node_phi_size"]
    x_phi = normalize_graph_features(x[:, :nphi], nb, B)                                  # :121
    x_rest = x[:, nphi:]
    if norm_global:
        x_rest = normalizer_forward(buffers, x_rest, hyper["dataset_size"])               # :127
    xin = torch.cat((x_phi, x_rest), dim=1)
    graph_node.x = xin
    edge_attr = relative_edge_attr(xin, G["pos"], G["edge_index"])                        # :178
    graph_node.edge_attr = edge_attr
    sim = simulator_v2(P, xin, edge_attr, G["edge_index"], nb, B, hyper["message_passing_num"], hyper["heads"],
                       return_intermediates=return_intermediates, net=hyper["net"])
    dec, inter = sim if return_intermediates else (sim, {})
    uvp_new = torch.tanh(dec / 10) * 10                                                   # :187
    uvp_new = enforce_boundary_condition(uvp_new, G["node_type"], G["y"])                 # :189
    if hyper["integrator"] == "explicit":
        uv_hat = uv_old
    elif hyper["integrator"] == "implicit":
        uv_hat = uvp_new[:, 0:2]
    else:
        uv_hat = (uv_old + uvp_new[:, 0:2]) / 2.0                                         # :198-201
    integ = integrator_conserved if hyper["conserved_form"] else integrator_non_conserved
    res = integ(uvp_new, uv_hat, uv_old, G, hyper, return_intermediates)
    (lc, lmx, lmy, lp, smoothed, uvp_cell), finter = res if return_intermediates else (res, {})
    smoothed = enforce_boundary_condition(smoothed, G["node_type"], G["y"])               # :223
    uvp_node_dim = smoothed * G["uvp_dim"][nb] * G["sigma"][nb]                           # :228-231
    uvp_cell_dim = uvp_cell * G["uvp_dim"][G["cell_batch"]] * G["sigma"][G["cell_batch"]]
    out = (lc, lmx, lmy, lp, uvp_node_dim, uvp_cell_dim)
    if return_intermediates:
        inter.update(finter)
        inter.update(dec=dec, uvp_new=uvp_new, edge_attr=edge_attr, x_norm=xin)
        return out, inter
    return out


def training_loss(outputs, hyper=None):
    """pre_train_Adam.py:177-184."""
    hyper = {**DEFAULT_HYPER, **(hyper or {})}
    lc, lmx, lmy, lp = outputs[0:4]
    batch = hyper["loss_press"] * lp + hyper["loss_cont"] * lc + hyper["loss_mom"] * lmx + hyper["loss_mom"] * lmy
    return torch.mean(torch.log(batch))


def new_normalizer_buffers(size=9, device="cpu"):
    """utils/normalization.py:25-31 (acc_count and num_accumulations start at 1.0)."""
    return dict(acc_count=torch.tensor(1.0, device=device), num_accumulations=torch.tensor(1.0, device=device),
                acc_sum=torch.zeros(size, device=device), acc_sum_squared=torch.zeros(size, device=device))


# --------------------------------------------------------------------------------------------------------------
# parameter construction (names/shapes = the reference state_dict)
# --------------------------------------------------------------------------------------------------------------
def parameter_shapes(hyper=None):
    hyper = {**DEFAULT_HYPER, **(hyper or {})}
    H, mp = hyper["hidden_size"], hyper["message_passing_num"]
    nin, nout, heads, G = hyper["node_input_size"], hyper["node_output_size"], hyper["heads"], hyper["slice_num"]
    dh = H // heads
    shapes = {}

    def mlp(prefix, kin, out=H, ln=True):
        lin = prefix + ".0" if ln else prefix
        shapes[f"{lin}.0.weight"], shapes[f"{lin}.0.bias"] = (H, kin), (H,)
        shapes[f"{lin}.2.weight"], shapes[f"{lin}.2.bias"] = (H, H), (H,)
        shapes[f"{lin}.4.weight"], shapes[f"{lin}.4.bias"] = (out, H), (out,)
        if ln:
            shapes[f"{prefix}.1.weight"], shapes[f"{prefix}.1.bias"] = (H,), (H,)

    mlp("simulator.encoder.eb_encoder", nin + 3)
    mlp("simulator.encoder.nb_encoder", nin)
    for pp in processor_prefixes(hyper["net"]):
        for ig in range(mp):
            g = f"{pp}.GN_block_list.{ig}"
            mlp(f"{g}.nb_module.net", H + H // 2)
            mlp(f"{g}.eb_module.net", 3 * H)
        t = f"{pp}.TransBlock"
        shapes[f"{t}.ln_1.weight"], shapes[f"{t}.ln_1.bias"] = (H,), (H,)
        shapes[f"{t}.Attn.temperature"] = (1, heads, 1, 1)
        shapes[f"{t}.Attn.graph_temperature"] = (1, heads, 1)
        for nm in ("in_project_x", "in_project_fx"):
            shapes[f"{t}.Attn.{nm}.weight"], shapes[f"{t}.Attn.{nm}.bias"] = (H, H), (H,)
        shapes[f"{t}.Attn.in_project_slice.weight"], shapes[f"{t}.Attn.in_project_slice.bias"] = (G, dh), (G,)
        for nm in ("to_q", "to_k", "to_v"):
            shapes[f"{t}.Attn.{nm}.weight"] = (dh, dh)
        shapes[f"{t}.Attn.to_out.0.weight"], shapes[f"{t}.Attn.to_out.0.bias"] = (H, H), (H,)
        shapes[f"{t}.ln_2.weight"], shapes[f"{t}.ln_2.bias"] = (H,), (H,)
        shapes[f"{t}.mlp.linear_pre.0.weight"], shapes[f"{t}.mlp.linear_pre.0.bias"] = (2 * H, H), (2 * H,)
        shapes[f"{t}.mlp.linear_post.weight"], shapes[f"{t}.mlp.linear_post.bias"] = (H, 2 * H), (H,)
    mlp("simulator.decoder.node_decode_module", H, out=nout, ln=False)
    return shapes


def init_parameters(seed=0, hyper=None, perturb=True):
    """Deterministic, platform-independent weights (numpy PCG64), reference init law importer.py:45-52:
    Linear weights trunc_normal(std .02, cut at +-2 [absolute]), biases 0, LayerNorm 1/0, temperatures 0.5.

    ``perturb=True`` additionally jitters biases / LayerNorm affine / temperatures so parity tests exercise them
    (a freshly initialised reference model has all of those at their trivial values)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    P = {}
    for name, shape in parameter_shapes(hyper).items():
        if name.endswith("temperature"):
            v = np.full(shape, 0.5) + (rng.uniform(-0.1, 0.1, size=shape) if perturb else 0.0)
        elif ".ln_" in name or name.endswith(".1.weight") or name.endswith(".1.bias"):
            base = 1.0 if name.endswith("weight") else 0.0
            v = np.full(shape, base) + (rng.uniform(-0.1, 0.1, size=shape) if perturb else 0.0)
        elif name.endswith("bias"):
            v = rng.uniform(-0.05, 0.05, size=shape) if perturb else np.zeros(shape)
        else:
            std = 0.02
            if perturb and any(t in name for t in (".to_q.", ".to_k.", ".to_v.", ".in_project_slice.")):
                std = 0.3  # make the slice attention non-degenerate so its gradients are not rounding noise
            v = np.clip(rng.standard_normal(size=shape) * std, -2.0, 2.0)
        P[name] = torch.from_numpy(np.asarray(v, dtype=np.float32))
    return P


def adam_step(P, grads, state, lr=5e-5, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (pre_train_Adam.py:79), restated; parameters with grad None are skipped."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, p in P.items():
        g = grads.get(k)
        if g is None:
            continue
        m = state.setdefault(("m", k), torch.zeros_like(p))
        v = state.setdefault(("v", k), torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.data.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
    return P


def train_step(P, buffers, graphs, adam_state, hyper=None):
    """One training iteration = forward + loss + backward + Adam (SURVEY.md 8d metric definition)."""
    hyper = {**DEFAULT_HYPER, **(hyper or {})}
    Pg = {k: v.detach().requires_grad_(True) for k, v in P.items()}
    out = model_forward(Pg, buffers, graphs, hyper)
    loss = training_loss(out, hyper)
    names = list(Pg.keys())
    gl = torch.autograd.grad(loss, [Pg[k] for k in names], allow_unused=True)
    grads = dict(zip(names, gl))
    adam_step(P, grads, adam_state, lr=hyper["lr"])
    return loss.detach(), out, grads
