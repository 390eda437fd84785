"""The row-local Linear chains of a Transolver block as one launch each (csrc/transmlp.hip, include/gfv.h gfv_trans_mlp_fwd /
gfv_trans_mlp_bwd) against float64 and against the three single-layer launches they replace (to_out + residual, ln_2 +
linear_pre, GELU + linear_post + residual; and the adjoints).  Reference: GraphTransolver.py:93-95,163-169
(/root/reference/src/FVMmodel/Models/GraphTransolver/GraphTransolver.py)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _setup(M, seed):
    g = torch.Generator().manual_seed(seed)
    s = lambda *sh: torch.randn(*sh, generator=g)
    P = {"Wout": s(128, 128) * 0.09, "bout": s(128) * 0.1, "gamma": 1 + 0.1 * s(128), "beta": 0.1 * s(128),
         "Wpre": s(256, 128) * 0.09, "bpre": s(256) * 0.1, "Wpost": s(128, 256) * 0.06, "bpost": s(128) * 0.1}
    x = s(M, 128) * torch.logspace(-2, 1, M)[:, None]      # rows over three decades
    res = s(M, 128)
    return P, x, res, g


def _ref_fwd(P, x, res):
    fx1 = x @ P["Wout"].T + P["bout"] + res
    z = F.layer_norm(fx1, (128,), P["gamma"], P["beta"], 1e-5) @ P["Wpre"].T + P["bpre"]
    out = F.gelu(z) @ P["Wpost"].T + P["bpost"] + fx1
    return fx1, z, out


@pytest.mark.parametrize("M,small", [(3000, 1), (129, 1), (33, 1), (5184, 1), (3000, 0), (129, 0), (25479, 1)])
def test_transolver_chain_forward_and_backward_in_one_launch_each(M, small, gfv_limits):
    """small = 1: launches of up to 16 384 rows run the forward chain on its small-tile form (csrc/ctrans.hip); 0: the 128-row-block
    kernel of transmlp.hip at every size (the switch is a dispatch limit: gfv_set_limit)."""
    from gfv import lib as L, ops
    gfv_limits(GFV_CTRANS=small)
    dev = torch.device("cuda")
    P, x, res, g = _setup(M, M)
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    Pg = {k: v.double().requires_grad_(True) for k, v in P.items()}
    xg, rg = x.double().requires_grad_(True), res.double().requires_grad_(True)
    fx1, z, out = _ref_fwd(Pg, xg, rg)
    wmax = torch.stack([v.abs().max() for k, v in P.items() if k.startswith("W")]).max().reshape(1).to(dev)
    wi = ops.WeightImages(dev, wmax)
    wi.static = [(0, 1 << 62)]
    prev = ops.set_weight_images(wi)
    try:
        new = lambda *s: torch.full(s, float("nan"), device=dev)
        # ---- forward: fused against float64 and against the three launches ----
        xd, rd = d(x), d(res)
        f1, zz, oo = new(M, 128), new(M, 256), new(M, 128)
        assert ops.trans_mlp_fwd(xd, rd, Pd["Wout"], Pd["bout"], Pd["gamma"], Pd["beta"], Pd["Wpre"], Pd["bpre"], Pd["Wpost"],
                                 Pd["bpost"], f1, zz, oo)
        f1b, zzb, oob = new(M, 128), new(M, 256), new(M, 128)
        ops.rowtile_chain(M, [ops.Seg(xd)], [ops.LayerSpec(Pd["Wout"], Pd["bout"])], [f1b], res=[rd])
        ops.rowtile_chain(M, [ops.Seg(f1b)], [ops.LayerSpec(Pd["Wpre"], Pd["bpre"])], [(zzb, 256), (zzb.data_ptr() + 512, 256)],
                          in_op=L.IN_LN, in_gamma=Pd["gamma"], in_beta=Pd["beta"])
        ops.rowtile_chain(M, [ops.Seg(zzb, width=128, ld=256), ops.Seg(zzb, width=128, ld=256, offset=128)],
                          [ops.LayerSpec(Pd["Wpost"], Pd["bpost"])], [oob], in_op=L.IN_GELU, res=[f1b])
        for mine, other, ref, name in ((f1, f1b, fx1, "fx1"), (zz, zzb, z, "z"), (oo, oob, out, "out")):
            assert rel(mine, ref) < TOL, (name, rel(mine, ref))
            assert rel(mine, other) < 4e-6, (name, rel(mine, other))
        # ---- backward ----
        go = torch.randn(M, 128, generator=g) * torch.logspace(-4, 0, M)[:, None]
        ga = torch.randn(M, 128, generator=g) * 1e-2
        gtot = (go + ga).double()
        (out * gtot).sum().backward()
        # reference intermediates: g_z = d/dz, g_fx1 = d/dfx1 (both paths), g_out_x = d/d(out_x W^T) = g_fx1 W_out
        fx1r = fx1.detach().requires_grad_(True)
        zr = F.layer_norm(fx1r, (128,), Pg["gamma"].detach(), Pg["beta"].detach(), 1e-5) @ Pg["Wpre"].detach().T + Pg["bpre"].detach()
        zr.retain_grad()
        outr = F.gelu(zr) @ Pg["Wpost"].detach().T + Pg["bpost"].detach() + fx1r
        (outr * gtot).sum().backward()
        gz_ref, gfx1_ref = zr.grad, fx1r.grad
        Wpost_t, Wpre_t, Wout_t = ops.transpose(Pd["Wpost"]), ops.transpose(Pd["Wpre"]), ops.transpose(Pd["Wout"])
        god, gad = d(go), d(ga)
        gsum, gz, gfx1, gox = new(M, 128), new(M, 256), new(M, 128), new(M, 128)
        tiles = ops.rowtile_tiles(M)
        part = new(ops.ln_rows(M), 2, 128)
        nfill = L.load().gfv_trans_mlp_ln_rows(M)   # one row per 64 rows; per 32 from the small-tile form
        assert nfill == ((M + 31) // 32 if (small and M <= 16384) else tiles)
        gs = torch.zeros(3, ops.gscale_ld(M), device=dev)
        assert ops.trans_mlp_bwd(god, gad, gsum, zz, f1, Wpost_t, Wpre_t, Wout_t, Pd["gamma"], gz, gfx1, gox, part, gs[0])
        torch.cuda.synchronize()
        assert rel(gsum, gtot) < 1e-6
        assert rel(gz, gz_ref) < TOL, rel(gz, gz_ref)
        assert rel(gfx1, gfx1_ref) < TOL, rel(gfx1, gfx1_ref)
        assert rel(gox, xg.grad) < TOL, rel(gox, xg.grad)
        assert bool(torch.isnan(part[nfill:]).all()) and not bool(torch.isnan(part[:nfill]).any())
        part = part[:nfill]
        dgb = part.double().sum(0).cpu()
        assert rel(dgb[0], Pg["gamma"].grad) < TOL and rel(dgb[1], Pg["beta"].grad) < TOL
        # the group scales of g: s * max|g| over 16 rows in [2^13, 2^14)
        grp = gsum.abs().amax(1).cpu()[: (M // 16) * 16].view(-1, 16).amax(1)
        sc = gs[0, : M // 16].cpu()
        assert bool(((sc * grp >= 2.0 ** 12.99) & (sc * grp < 2.0 ** 14.01)).all())
        # ... and against the three separate launches
        gzb, gfx1b, goxb, gsb = new(M, 256), new(M, 128), new(M, 128), new(M, 128)
        partb = new(tiles, 2, 128)
        ops.rowtile_chain(M, [ops.Seg(god)], [ops.LayerSpec(Wpost_t, None, L.OP_MUL_DGELU, aux=zz)],
                          [(gzb, 256), (gzb.data_ptr() + 512, 256)], in_add=gad, in_save=gsb)
        ops.rowtile_chain(M, [ops.Seg(gzb, width=128, ld=256), ops.Seg(gzb, width=128, ld=256, offset=128)], [ops.LayerSpec(Wpre_t)],
                          [gfx1b], fin_op=L.FIN_LNBWD, fin_gamma=Pd["gamma"], fin_aux=f1, ln_partial=partb, res=[gsb])
        ops.rowtile_chain(M, [ops.Seg(gfx1b)], [ops.LayerSpec(Wout_t)], [goxb])
        assert rel(gz, gzb) < 4e-6 and rel(gfx1, gfx1b) < 4e-6 and rel(gox, goxb) < 4e-6
        assert rel(part.sum(0), partb.sum(0)) < 4e-6
        flags = L.C.c_int32(0)
        L.check(L.load().gfv_status_flags(L.C.byref(flags)), "gfv_status_flags")
        assert flags.value == 0
    finally:
        ops.set_weight_images(prev)


def test_transolver_chain_refuses_the_fp32_mfma_form():
    """Under gfv_set_f16split(0) there is no fused launch: the wrapper says so and the caller keeps the three launches."""
    from gfv import lib as L, ops
    lib = L.load()
    dev = torch.device("cuda")
    P, x, res, _ = _setup(256, 1)
    d = lambda t: t.to(dev).contiguous()
    Pd = {k: d(v) for k, v in P.items()}
    wi = ops.WeightImages(dev, torch.ones(1, device=dev))
    wi.static = [(0, 1 << 62)]
    prev = ops.set_weight_images(wi)
    lib.gfv_set_f16split(0)
    try:
        o = torch.empty(256, 128, device=dev)
        assert not ops.trans_mlp_fwd(d(x), d(res), Pd["Wout"], Pd["bout"], Pd["gamma"], Pd["beta"], Pd["Wpre"], Pd["bpre"],
                                     Pd["Wpost"], Pd["bpost"], o.clone(), torch.empty(256, 256, device=dev), o)
    finally:
        lib.gfv_set_f16split(1)
        ops.set_weight_images(prev)
