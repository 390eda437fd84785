"""Where does the HIP forward lose accuracy?  Per-stage distance to the float64 oracle (encoder, every GnBlock, every
Transolver block, decoder, uvp_node, uvp_cell) of the HIP path in each product form and of the fp32 oracle.
usage: python profiles/tools/stage_errors.py [real_naca0012 | real_cylinder | real_cavity101 | bench]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "gen-fvgn-steady_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import cases  # noqa: E402
from oracle import fvgn_oracle as O  # noqa: E402
from test_fullsize_gpu import _mesh, _model, graphs_to  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "real_naca0012"
    if name == "bench":
        from gfv.graph import build_batch
        m, f = _mesh(50000, 1234)
        graphs = build_batch([m], [f])
    else:
        graphs = cases.real_mesh(name)[0]
    P = O.init_parameters(cases.WEIGHT_SEED)
    runs = {}
    for dt in (torch.float64, torch.float32):
        Pg = {k: v.to(dt) for k, v in P.items()}
        buf = {k: v.to(dt) for k, v in O.new_normalizer_buffers().items()}
        with torch.no_grad():
            out, inter = O.model_forward(Pg, buf, graphs_to(graphs, dt), hyper={"dataset_size": 1}, return_intermediates=True)
        runs[dt] = (out, inter)
    keys = ["enc_x", "enc_e"]
    for ip in range(2):
        for ig in range(3):
            keys += [f"p{ip}.gn{ig}.x", f"p{ip}.gn{ig}.e"]
        keys.append(f"p{ip}.trans.x")
    from gfv import lib as L
    from gfv.engine import Engine
    lib = L.load()
    rows = {}
    for form in (0, 1, 2):
        lib.gfv_set_f16split(form)
        rec = {}
        model = _model(P)
        orig_gn, orig_mlp, orig_tr = Engine.gn_fwd, Engine.mlp3_fwd, Engine.trans_fwd
        cnt = dict(gn=0, tr=0)

        def gn(self, P_, prefix, xn, en, pl, _o=orig_gn):
            r = _o(self, P_, prefix, xn, en, pl)
            i = cnt["gn"]
            rec[f"p{i // 3}.gn{i % 3}.x"], rec[f"p{i // 3}.gn{i % 3}.e"] = r[0].clone(), r[1].clone()
            cnt["gn"] += 1
            return r

        def mlp(self, P_, prefix, M, segs, *a, _o=orig_mlp, **k):
            r = _o(self, P_, prefix, M, segs, *a, **k)
            if prefix.endswith("nb_encoder"):
                rec["enc_x"] = r[0].clone()
            if prefix.endswith("eb_encoder"):
                rec["enc_e"] = r[0].clone()
            if prefix.endswith("node_decode_module"):
                rec["dec"] = r[0].clone()
            return r

        def tr(self, P_, prefix, xn, emb, pl, _o=orig_tr):
            r = _o(self, P_, prefix, xn, emb, pl)
            rec[f"p{cnt['tr']}.trans.x"] = r[0].clone()
            cnt["tr"] += 1
            return r

        Engine.gn_fwd, Engine.mlp3_fwd, Engine.trans_fwd = gn, mlp, tr
        try:
            hg = tuple(g.clone().to("cuda") for g in graphs)
            hg[0].norm_uvp, hg[0].norm_global = True, True
            with torch.no_grad():
                out = model(*hg)
            torch.cuda.synchronize()
        finally:
            Engine.gn_fwd, Engine.mlp3_fwd, Engine.trans_fwd = orig_gn, orig_mlp, orig_tr
        rows[form] = (rec, out)
    lib.gfv_set_f16split(1)
    o64, i64 = runs[torch.float64]
    o32, i32 = runs[torch.float32]
    print(f"{name}: distance to the float64 oracle (max |diff| / max |ref|)")
    print(f"{'stage':16s} {'fp32 oracle':>12s} {'HIP fp32':>12s} {'HIP split':>12s} {'HIP lowp':>12s}")
    for k in keys:
        print(f"{k:16s} {rel(i32[k], i64[k]):12.2e} " + " ".join(f"{rel(rows[f][0][k][:, :i64[k].shape[1]], i64[k]):12.2e}" for f in (0, 1, 2)))
    for i, k in enumerate(("loss_cont", "loss_mom_x", "loss_mom_y", "loss_press", "uvp_node", "uvp_cell")):
        print(f"{k:16s} {rel(o32[i], o64[i]):12.2e} " + " ".join(f"{rel(rows[f][1][i], o64[i]):12.2e}" for f in (0, 1, 2)))


if __name__ == "__main__":
    main()
