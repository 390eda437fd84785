"""The Transolver slice kernels (csrc/slice.hip) against FLOAT64 restatements of GraphTransolver.py:64-92 (1e-5): the one-pass
adjoint behind the attention (gfv_slice_post_bwd) - also against the four launches it replaces (gfv_slice_gw, gfv_deslice,
gfv_slice_gw accumulate, gfv_slice_softmax_bwd: the same terms in the same order, equal to rounding) -, the fused softmax +
token sums, the matrix-core token sums and de-slice.  (Every assertion that carries parity here is the float64 one.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(N, sizes, seed):
    g = torch.Generator().manual_seed(seed)
    B = len(sizes)
    batch = torch.repeat_interleave(torch.arange(B), torch.tensor(sizes)).int()
    assert batch.numel() == N
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).cuda()
    w = torch.softmax(torch.randn(N, 8, 32, generator=g) * 2, -1).cuda().contiguous()
    return dict(xmid=r(N, 128), Ws=r(32, 16, scale=0.3), bs=r(32, scale=0.1), temp=(0.5 + torch.rand(8, generator=g)).cuda(),
                w=w, gox=r(N, 128), T1=r(B, 8, 32, 16), fxm=r(N, 128), T2=r(B, 8, 32, 16, scale=0.2), gn=r(B, 8, 32, scale=0.1),
                batch=batch.cuda())


@pytest.mark.parametrize("N,sizes", [(5000, [1700, 2100, 1200]), (77, [77]), (1000, [3, 500, 497]), (32, [32])])
def test_one_pass_slice_adjoint_against_float64_and_the_four_launches(N, sizes):
    from gfv import lib as L
    lib = L.load()
    st = L.stream_ptr()
    c = _case(N, sizes, N)
    p = lambda t: t.data_ptr()
    nblk = lib.gfv_slice_softmax_bwd_blocks(N)
    # four launches
    gw = torch.empty(N, 256, device="cuda")
    L.check(lib.gfv_slice_gw(p(c["gox"]), p(c["T1"]), None, p(c["batch"]), p(gw), N, 0, st), "gw1")
    gfx0 = torch.empty(N, 128, device="cuda")
    L.check(lib.gfv_deslice(p(c["w"]), p(c["T2"]), p(c["batch"]), p(gfx0), N, 0, st), "deslice")
    L.check(lib.gfv_slice_gw(p(c["fxm"]), p(c["T2"]), p(c["gn"]), p(c["batch"]), p(gw), N, 1, st), "gw2")
    gx0, sp0 = torch.empty(N, 128, device="cuda"), torch.empty(nblk, 552, device="cuda")
    L.check(lib.gfv_slice_softmax_bwd(p(c["xmid"]), p(c["Ws"]), p(c["bs"]), p(c["temp"]), p(c["w"]), p(gw), p(gx0), p(sp0), N, st),
            "softmax_bwd")
    # one pass
    gx1, gfx1 = torch.full((N, 128), float("nan"), device="cuda"), torch.full((N, 128), float("nan"), device="cuda")
    sp1 = torch.full((nblk, 552), float("nan"), device="cuda")
    L.check(lib.gfv_slice_post_bwd(p(c["xmid"]), p(c["Ws"]), p(c["bs"]), p(c["temp"]), p(c["w"]), p(c["gox"]), p(c["T1"]),
                                   p(c["fxm"]), p(c["T2"]), p(c["gn"]), p(c["batch"]), p(gx1), p(gfx1), p(sp1), N, len(sizes), st), "post_bwd")
    torch.cuda.synchronize()
    # the same terms in the same order; the compiler contracts a * b + c into fused multiply-adds where it sees them (slice.hip
    # is built with -ffp-contract=fast), not necessarily at the same places in both forms: equal to rounding, not to the bit
    close = lambda a, b: float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    assert close(gfx1, gfx0) and close(gx1, gx0) and close(sp1[:, :544].sum(0), sp0[:, :544].sum(0))
    assert close(sp1[:, 544:].sum(0), sp0[:, 544:].sum(0))
    # and the float64 statement of the same adjoint
    d = lambda t: t.double().cpu()
    b = c["batch"].long().cpu()
    w, x = d(c["w"]), d(c["xmid"]).view(N, 8, 16)
    gwr = (torch.einsum("nhc,nhgc->nhg", d(c["gox"]).view(N, 8, 16), d(c["T1"])[b])
           + torch.einsum("nhc,nhgc->nhg", d(c["fxm"]).view(N, 8, 16), d(c["T2"])[b]) + d(c["gn"])[b])
    gz = w * (gwr - (w * gwr).sum(-1, keepdim=True))
    gl = gz / d(c["temp"]).view(1, 8, 1)
    gx_ref = torch.einsum("nhg,gc->nhc", gl, d(c["Ws"])).reshape(N, 128)
    gfx_ref = torch.einsum("nhg,nhgc->nhc", w, d(c["T2"])[b]).reshape(N, 128)
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())
    assert rel(gx1, gx_ref) < 1e-5 and rel(gfx1, gfx_ref) < 1e-5
    dWs = torch.einsum("nhg,nhc->gc", gl, x)
    assert rel(sp1[:, :512].sum(0).view(32, 16), dWs) < 1e-5
    assert rel(sp1[:, 512:544].sum(0), gl.sum((0, 1))) < 1e-5


@pytest.mark.parametrize("N,sizes", [(5000, [1700, 2100, 1200]), (77, [77]), (1000, [3, 500, 497])])
def test_matrix_core_token_and_deslice_kernels_against_float64(N, sizes):
    """gfv_slice_softmax_token (softmax + per-chunk token sums in one pass), the matrix-core form of gfv_slice_token_partial
    and of gfv_deslice against float64 statements of GraphTransolver.py:64-73,90-92."""
    from gfv import lib as L
    lib = L.load()
    st = L.stream_ptr()
    c = _case(N, sizes, N + 1)
    p = lambda t: t.data_ptr()
    # chunks of 64 nodes inside each graph (gfv/plan.py _batch_part)
    cb, ce, start = [], [], 0
    for n in sizes:
        for s0 in range(start, start + n, 64):
            cb.append(s0)
            ce.append(min(s0 + 64, start + n))
        start += n
    cbd, ced = torch.tensor(cb, dtype=torch.int32).cuda(), torch.tensor(ce, dtype=torch.int32).cuda()
    nch = len(cb)
    w1 = torch.full((N, 256), float("nan"), device="cuda")
    part1 = torch.full((nch, 256, 17), float("nan"), device="cuda")
    L.check(lib.gfv_slice_softmax_token(p(c["xmid"]), p(c["Ws"]), p(c["bs"]), p(c["temp"]), p(c["fxm"]), p(cbd), p(ced), nch,
                                        p(w1), p(part1), st), "softmax_token")
    part2 = torch.full((nch, 256, 17), float("nan"), device="cuda")
    L.check(lib.gfv_slice_token_partial(p(w1), p(c["gox"]), p(cbd), p(ced), nch, p(part2), st), "token_partial")
    out = torch.full((N, 128), float("nan"), device="cuda")
    L.check(lib.gfv_deslice(p(w1), p(c["T1"]), p(c["batch"]), p(out), N, 4 if len(sizes) == 1 else 0, st), "deslice")
    torch.cuda.synchronize()
    d = lambda t: t.double().cpu()
    x = d(c["xmid"]).view(N, 8, 16)
    logits = (torch.einsum("nhc,gc->nhg", x, d(c["Ws"])) + d(c["bs"])) / d(c["temp"]).view(1, 8, 1)
    wref = torch.softmax(logits, -1)
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())
    assert rel(w1.view(N, 8, 32), wref) < 1e-5
    wd = d(w1).view(N, 8, 32)
    for part, a in ((part1, c["fxm"]), (part2, c["gox"])):
        av = d(a).view(N, 8, 16)
        for k in range(nch):
            sl = slice(cb[k], ce[k])
            tok = torch.einsum("nhg,nhc->hgc", wd[sl], av[sl]).reshape(256, 16)
            assert rel(part[k, :, :16], tok) < 1e-5, k
            assert rel(part[k, :, 16], wd[sl].sum(0).reshape(256)) < 1e-5, k
    b = c["batch"].long().cpu()
    ref = torch.einsum("nhg,nhgc->nhc", wd, d(c["T1"])[b]).reshape(N, 128)
    assert rel(out, ref) < 1e-5
