"""Golden vectors for SURVEY.md row f4 (WLSQ reconstruction orders 1st / 3rd / 4th): the REFERENCE ITSELF with
``params.order = <order>`` on the ``cyl_cavity_b2`` case whose moment matrices are built for that order
(tests/golden/order_<order>_cyl_cavity_b2.npz), with the oracle-vs-reference deviations printed, and the moment matrices of
gfv.meshgen checked against the reference's ``moments_order`` (FVorder.py:7-86).
Build container only (needs /root/reference):  python tests/golden/make_golden_orders.py"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refstubs"))
sys.path.insert(0, HERE)

import cases  # noqa: E402
import ref_import  # noqa: E402

sys.path.insert(0, cases.ROOT)
from oracle import fvgn_oracle as O  # noqa: E402
from make_golden import rel, to_ref_graphs  # noqa: E402

torch.set_num_threads(8)
NAME = "cyl_cavity_b2"


def one(R, order):
    graphs = cases.make_graphs(NAME, order=order)
    # moments of the first mesh against the reference's own construction
    from FVMmodel.FVdiscretization.FVorder import moments_order
    mesh = cases.make_meshes(NAME, order)[0][0]
    fx, sup = torch.from_numpy(mesh["face_node_x"]), torch.from_numpy(mesh["support_edge"])
    comp = torch.cat((fx, fx.flip(0), sup), 1)
    pos = torch.from_numpy(mesh["node|pos"])
    A_ref, B_ref = moments_order(order=order, mesh_pos_diff_on_edge=pos[comp[0]] - pos[comp[1]], indegree_node_index=comp[1])
    print(f"== order {order}: moment matrices vs reference moments_order: A rel {rel(torch.from_numpy(mesh['A_node_to_node']), A_ref):.2e}, "
          f"B rel {rel(torch.from_numpy(mesh['single_B_node_to_node']), B_ref[:fx.shape[1]]):.2e}")
    P0 = O.init_parameters(cases.WEIGHT_SEED)
    params = R.get_param.params()
    params.order = order
    model = R.importer.NNmodel(params)
    sd = model.state_dict()
    for k, v in P0.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    model.train()
    gn, gx, ge, gc, gi = to_ref_graphs(graphs)
    gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
    lc, lmx, lmy, lp, uvp_node, uvp_cell = model(graph_node=gn, graph_node_x=gx, graph_edge=ge, graph_cell=gc,
                                                 graph_Index=gi, is_training=True)
    loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lmx + params.loss_mom * lmy))
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters()}

    Pg = {k: v.detach().requires_grad_(True) for k, v in P0.items()}
    oout = O.model_forward(Pg, O.new_normalizer_buffers(), tuple(g.clone() for g in graphs), hyper={"order": order})
    oloss = O.training_loss(oout)
    names = list(Pg)
    ograds = dict(zip(names, torch.autograd.grad(oloss, [Pg[k] for k in names], allow_unused=True)))
    print(f"   loss ref {float(loss.detach()):.8f} oracle {float(oloss.detach()):.8f}")
    for nm, a, b in (("loss_cont", oout[0], lc), ("loss_mom_x", oout[1], lmx), ("loss_mom_y", oout[2], lmy),
                     ("loss_press", oout[3], lp), ("uvp_node", oout[4], uvp_node), ("uvp_cell", oout[5], uvp_cell)):
        print(f"   oracle vs reference {nm:12s} rel {rel(a, b):.3e}")
    worst = 0.0
    for k in names:
        if grads[k] is None:
            assert ograds[k] is None, k
            continue
        worst = max(worst, rel(ograds[k], grads[k]))
    print(f"   oracle vs reference worst parameter-gradient rel {worst:.3e}")
    out = {"loss": np.float64(loss.item()), "loss_cont": lc.detach().numpy(), "loss_mom_x": lmx.detach().numpy(),
           "loss_mom_y": lmy.detach().numpy(), "loss_press": lp.detach().numpy(), "uvp_node": uvp_node.detach().numpy(),
           "uvp_cell": uvp_cell.detach().numpy(), "param_names": np.array(names),
           "grad_fp": np.stack([cases.fingerprint(grads[k].numpy()) if grads[k] is not None else np.full(3, np.nan)
                                for k in names])}
    np.savez_compressed(os.path.join(HERE, f"order_{order}_{NAME}.npz"), **out)
    print("   wrote", f"order_{order}_{NAME}.npz")


def main():
    ref_import.install()
    R = ref_import.reference_modules()
    for order in ("1st", "3rd", "4th"):
        one(R, order)


if __name__ == "__main__":
    main()
