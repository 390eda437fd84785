"""Golden vector for ``--hidden_size 64`` (utils/get_param.py:69): runs the REFERENCE ITSELF with ``hidden_size=64`` on the
``cyl_cavity_b2`` case of ``cases.py`` and commits its outputs (tests/golden/hidden64_cyl_cavity_b2.npz), printing the
oracle-vs-reference deviations.  Build container only (needs /root/reference):  python tests/golden/make_golden_hidden.py"""
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
HYPER = {"hidden_size": 64}


def main():
    ref_import.install()
    R = ref_import.reference_modules()
    graphs = cases.make_graphs(NAME)
    P0 = O.init_parameters(cases.WEIGHT_SEED, hyper=HYPER)
    params = R.get_param.params()
    params.hidden_size = 64
    model = R.importer.NNmodel(params)
    sd = model.state_dict()
    ref_keys = [k for k in sd if not k.startswith("node_norm.")]
    assert ref_keys == list(P0), (set(ref_keys) ^ set(P0))
    for k, v in P0.items():
        assert sd[k].shape == v.shape, k
        sd[k].copy_(v)
    model.load_state_dict(sd)
    model.train()
    gn, gx, ge, gc, gi = to_ref_graphs(graphs)
    gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
    out = model(graph_node=gn, graph_node_x=gx, graph_edge=ge, graph_cell=gc, graph_Index=gi, is_training=True)
    lc, lmx, lmy, lp, uvp_node, uvp_cell = out
    loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lmx + params.loss_mom * lmy))
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters()}

    Pg = {k: v.detach().requires_grad_(True) for k, v in P0.items()}
    og = tuple(g.clone() for g in graphs)
    oout = O.model_forward(Pg, O.new_normalizer_buffers(), og, hyper=HYPER)
    oloss = O.training_loss(oout)
    names = list(Pg)
    ograds = dict(zip(names, torch.autograd.grad(oloss, [Pg[k] for k in names], allow_unused=True)))
    print(f"== hidden_size 64 {NAME}: {len(names)} tensors, {sum(v.numel() for v in P0.values())} parameters; "
          f"loss ref {float(loss):.8f} oracle {float(oloss):.8f}")
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
    fx = {"loss": np.float64(loss.item()), "loss_cont": lc.detach().numpy(), "loss_mom_x": lmx.detach().numpy(),
          "loss_mom_y": lmy.detach().numpy(), "loss_press": lp.detach().numpy(), "uvp_node": uvp_node.detach().numpy(),
          "uvp_cell": uvp_cell.detach().numpy(), "param_names": np.array(names),
          "grad_fp": np.stack([cases.fingerprint(grads[k].numpy()) if grads[k] is not None else np.full(3, np.nan)
                               for k in names])}
    np.savez_compressed(os.path.join(HERE, f"hidden64_{NAME}.npz"), **fx)
    print("wrote", f"hidden64_{NAME}.npz")


if __name__ == "__main__":
    main()
