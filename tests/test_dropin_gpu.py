"""Round 6: the drop-in path of the reference's drivers (pre_train_Adam.py:158-191, solve_with_grad_GPU.py:133-181) replayed from
recorded launch lists (gfv/functions.py ReplayCache), `gfv.optim.Adam` (torch.optim.Adam over flat buffers + the fused launch),
the status word a host loop reads without a synchronisation (include/gfv.h gfv_status_mirror), and the record API by itself."""
import ctypes as C

import pytest
import torch

import cases
from oracle import fvgn_oracle as O

pytestmark = pytest.mark.gpu


def _model(seed=cases.WEIGHT_SEED, dataset_size=1):
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    params = default_params(dataset_size=dataset_size)
    P0 = O.init_parameters(seed)
    model = NNmodel(params)
    sd = model.state_dict()
    for k, v in P0.items():
        sd[k].copy_(v)
    model.load_state_dict(sd)
    return model.to("cuda"), params


def _driver_steps(model, params, optimizer, graphs, steps, fresh_x=False):
    """The reference's inner loop on ONE batch: restore the node state, re-arm the norm flags, zero_grad, forward, log-loss,
    backward, step.  fresh_x: a NEW tensor carries the node state every iteration (a loader that hands out new batches of the
    same mesh) instead of the one the first call saw."""
    gn = graphs[0]
    backup = gn.x.clone()
    losses, fields = [], []
    for _ in range(steps):
        if fresh_x:
            gn.x = backup.clone()
        else:
            gn.x.copy_(backup)
        gn.norm_uvp, gn.norm_global = params.norm_uvp, params.norm_global
        optimizer.zero_grad()
        lc, lx, ly, lp, un, uc = model(*graphs)
        loss = torch.mean(torch.log(params.loss_press * lp + params.loss_cont * lc + params.loss_mom * lx + params.loss_mom * ly))
        loss.backward()
        optimizer.step()
        losses.append(loss.detach().clone())
        fields.append((un, uc, gn.x.clone(), gn.edge_attr))
    return losses, fields


@pytest.mark.parametrize("fresh_x", [False, True])
def test_replayed_drop_in_steps_equal_eager_ones(fresh_x):
    """Eight driver iterations with the recorded lists (two warm-up calls, the recording call, five replays) against the same
    eight issued eagerly: the same kernels on the same data in the same order - losses, fields, the normalised node state, the
    edge features and every parameter BIT-identical; outputs of earlier iterations are not overwritten by later replays."""
    out = {}
    for replay in (False, True):
        model, params = _model()
        model._replay.enabled = replay
        graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))
        opt = torch.optim.Adam(model.parameters(), lr=params.lr)
        losses, fields = _driver_steps(model, params, opt, graphs, 8, fresh_x=fresh_x)
        torch.cuda.synchronize()
        out[replay] = (losses, fields, [p.detach().clone() for p in model.parameters()], model._replay.replays)
    assert out[True][3] == 5 and out[False][3] == 0
    for a, b in zip(out[False][0], out[True][0]):
        assert torch.equal(a, b)
    for fa, fb in zip(out[False][1], out[True][1]):
        for ta, tb in zip(fa, fb):
            assert torch.equal(ta, tb)
    for a, b in zip(out[False][2], out[True][2]):
        assert torch.equal(a, b)
    # the fields handed out by the replayed iterations are tensors of their own (iteration 6's is not iteration 7's)
    assert out[True][1][6][0].data_ptr() != out[True][1][7][0].data_ptr()
    assert not torch.equal(out[True][1][6][0], out[True][1][7][0])


def test_second_forward_before_the_backward_does_not_touch_the_saved_rows():
    """forward(x1), forward(x2), backward-of-the-first: the second call must not replay into the rows the first one's backward
    reads (it runs eagerly); the gradients are those of forward(x1), backward alone - and differ from forward(x2)'s."""
    model, params = _model()
    graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cavity_mixed_b1"))
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    gn = graphs[0]
    raw = gn.x.clone()
    _driver_steps(model, params, opt, graphs, 4)      # warm-up + recording + one replay
    other = raw.clone()
    other[:, 0:3] = other[:, 0:3].flip(0) * 1.7 + 0.3   # another node state on the same mesh

    xa = gn.x   # the tensor the recorded lists know

    def fwd(state, tensor=None):
        # (a second forward must carry its node state in ANOTHER tensor: the first call's backward reads the normalised rows of
        # its own - overwriting them in place in between is an error in the reference as well, autograd's version check)
        gn.x = xa if tensor is None else tensor
        gn.x.copy_(state)
        gn.norm_uvp, gn.norm_global = True, True
        o = model(*graphs)
        return torch.mean(torch.log(o[3] + 6e4 * o[0] + 5e4 * o[1] + 5e4 * o[2]))

    def grads_of(state):
        opt.zero_grad()
        fwd(state).backward()
        return [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    want, want_other = grads_of(raw), grads_of(other)
    assert any(w is not None and not torch.equal(w, v) for w, v in zip(want, want_other))
    before = model._replay.replays
    opt.zero_grad()
    first = fwd(raw)                                  # replayed
    second = fwd(other, torch.empty_like(raw))        # the first call's backward is pending: issued eagerly, nothing of the lists is touched
    assert model._replay.replays == before + (1 if model._replay.enabled else 0)   # (GFV_DROPIN_REPLAY=0: both eager)
    first.backward()
    for p, w in zip(model.parameters(), want):
        assert (p.grad is None) == (w is None)
        if w is not None:
            assert torch.equal(p.grad, w)
    opt.zero_grad()
    second.backward()
    for p, w in zip(model.parameters(), want_other):
        if w is not None:
            assert torch.equal(p.grad, w)


def test_gfv_adam_equals_torch_adam_and_exchanges_state():
    """`gfv.optim.Adam(model.parameters(), lr)` in the driver loop against `torch.optim.Adam`: six iterations, parameters within
    2e-6 of scale (fp32 Adam arithmetic in another association), the flat-gradient fast path taken; its state_dict loads into
    torch's Adam and back, and a run resumed from it continues exactly."""
    from gfv.optim import Adam
    res = {}
    for which in ("torch", "gfv"):
        model, params = _model()
        graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))
        opt = (torch.optim.Adam if which == "torch" else Adam)(model.parameters(), lr=params.lr)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=3, gamma=0.5)
        losses = []
        for _ in range(2):
            l, _ = _driver_steps(model, params, opt, graphs, 3)
            losses += l
            sched.step()
        res[which] = (model, opt, losses, graphs, params)
    mt, ot, lt, _, _ = res["torch"]
    mg, og, lg, graphs, params = res["gfv"]
    for a, b in zip(lt, lg):
        assert abs(float(a) - float(b)) < 1e-5 * abs(float(a))
    for (n, a), b in zip(mt.named_parameters(), mg.parameters()):
        # (Adam's normalised step is ~lr per iteration whatever the gradient's size: 2 % of one step after six)
        assert float((a.detach() - b.detach()).abs().max()) < 0.02 * params.lr, n
    assert float(og.adam_state[0]) == 6.0
    # the gradients of the last backward sit in ONE flat tensor the optimiser recognises
    flat = og._flat_grad()
    assert flat is not og.flat_g and flat.numel() == og.G.total
    # state_dict nesting is torch's
    sd = og.state_dict()
    st = ot.state_dict()
    assert set(sd["param_groups"][0]) >= {"lr", "betas", "eps", "params"} and len(sd["param_groups"][0]["params"]) == len(st["param_groups"][0]["params"])
    m_scale = max(float(s["exp_avg"].abs().max()) for s in st["state"].values())
    for i, s in st["state"].items():
        # (six chaotic Adam steps apart: the moments agree as well as the gradients of the two runs do; tensors whose gradient is
        # rounding noise of the whole computation - 1e-10 here - are held to 1e-6 of the largest moment)
        assert float((sd["state"][i]["exp_avg"] - s["exp_avg"].cpu()).abs().max()) < 2e-3 * float(s["exp_avg"].abs().max()) + 1e-6 * m_scale
        assert float(sd["state"][i]["step"]) == float(s["step"]) == 6.0
    # resume: a new optimiser loaded from it continues exactly like the original
    m2, _ = _model()
    m2.load_state_dict(mg.state_dict())
    o2 = Adam(m2.parameters(), lr=1.0)
    o2.load_state_dict(sd)
    g1 = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))   # (fresh batches for both: `graphs` holds the
    g2 = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))   # normalised node state of its last iteration)
    _driver_steps(mg, params, og, g1, 2)
    _driver_steps(m2, params, o2, g2, 2)
    for a, b in zip(mg.parameters(), m2.parameters()):
        assert torch.equal(a, b)


def test_gfv_adam_gathers_gradients_that_are_not_the_flat_tensor():
    """A frozen layer (`p.grad = None` after the backward) and hand-made gradients: the optimiser must not use the backward's flat
    tensor as it is (the frozen slot holds a live value) - it gathers, and the frozen parameter keeps its value."""
    from gfv.optim import Adam
    model, params = _model()
    graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cavity_mixed_b1"))
    opt = Adam(model.parameters(), lr=1e-3)
    gn = graphs[0]
    gn.norm_uvp, gn.norm_global = True, True
    o = model(*graphs)
    torch.mean(torch.log(o[3] + 6e4 * o[0] + 5e4 * o[1] + 5e4 * o[2])).backward()
    frozen = model.simulator.decoder.node_decode_module[0].weight if hasattr(model.simulator.decoder, "node_decode_module") else list(model.parameters())[-2]
    before = frozen.detach().clone()
    other = list(model.parameters())[0]
    other_before = other.detach().clone()
    frozen.grad = None
    assert opt._flat_grad() is opt.flat_g
    opt.step()
    assert torch.equal(frozen, before)
    assert not torch.equal(other, other_before)


@pytest.mark.parametrize("mode", [False, True, "list"])
def test_status_word_reaches_the_training_loop(mode):
    """Weights scaled so that a hidden activation of the column-owner chain kernels leaves their fixed-scale fp16 window: the
    kernels raise GFV_FLAG_CHAIN_RANGE, the fused Adam publishes it, and the NEXT `TrainStep.step` raises FloatingPointError -
    in eager, hipGraph and command-list mode, without the loop ever synchronising by itself."""
    from gfv import lib as L
    from gfv.trainer import TrainStep
    model, params = _model()
    graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))
    ts = TrainStep(model, graphs, use_graph=mode, lr=0.0)
    for _ in range(4):
        ts.step()
    torch.cuda.synchronize()
    L.raise_on_status("test")    # a healthy run raises nothing
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "GN_block_list.0.eb_module.net.0.2.weight" in n:   # second Linear of an EdgeBlock MLP: its output feeds a hidden GELU
                p.mul_(3.0e5)
    with pytest.raises(FloatingPointError, match="GFV_FLAG"):
        for _ in range(6):
            ts.step()
            torch.cuda.synchronize()   # (the test makes "the step after" deterministic; the loop itself never waits)
    # the word was cleared by the error path: the device word and the mirror are both zero again
    flags = C.c_int32(0)
    L.check(L.load().gfv_status_flags(C.byref(flags)), "gfv_status_flags")
    assert int(L.status_mirror()[0]) == 0


def test_status_word_reaches_the_drop_in_loop():
    from gfv import lib as L
    model, params = _model()
    graphs = tuple(g.clone().to("cuda") for g in cases.make_graphs("cyl_cavity_b2"))
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    _driver_steps(model, params, opt, graphs, 2)
    torch.cuda.synchronize()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "GN_block_list.0.eb_module.net.0.2.weight" in n:
                p.mul_(3.0e5)
    with pytest.raises(FloatingPointError, match="GFV_FLAG"):
        for _ in range(4):
            _driver_steps(model, params, opt, graphs, 1)
            torch.cuda.synchronize()
    assert int(L.status_mirror()[0]) == 0


def test_record_api_replays_ranges_and_stream_edges():
    """include/gfv.h gfv_record_*: launches of two streams with a wait edge between them, recorded once; replaying [0, n) and the
    two halves [0, k) + [k, n) reproduces the buffers; a second begin on the same thread and a replay inside a recording are
    errors; a failed begin leaves gfv.cmdlist usable."""
    from gfv import cmdlist, ops
    from gfv import lib as L
    lib = L.load(raw=True)
    dev = torch.device("cuda")
    n, F = 4096, 64
    src = torch.randn(n, F, device=dev)
    rowptr = torch.arange(0, n + 1, dtype=torch.int32, device=dev)
    col = torch.randperm(n, device=dev).to(torch.int32)
    a, b = torch.zeros(n, F, device=dev), torch.zeros(n, F, device=dev)
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def body():
        # a = gather(src) on the side stream; b = gather(a) on the main stream behind a wait edge
        L.stream_wait(side, main)
        with torch.cuda.stream(side):
            ops.seg_gather_sum(src, rowptr, col, n, out=a)
        L.stream_wait(main, side)
        ops.seg_gather_sum(a, rowptr, col, n, out=b)
    body()
    torch.cuda.synchronize()
    want_a, want_b = a.clone(), b.clone()
    assert lib.gfv_record_begin() == 0
    assert lib.gfv_record_begin() != 0          # a second begin on the same thread
    body()
    cnt = lib.gfv_record_count()
    h = lib.gfv_record_end()
    assert cnt == lib.gfv_record_length(h) == 4  # wait, launch, wait, launch
    for ranges in ([(0, cnt)], [(0, 2), (2, cnt)]):
        a.zero_()
        b.zero_()
        torch.cuda.synchronize()
        for lo, hi in ranges:
            assert lib.gfv_record_replay(h, lo, hi) == 0
        torch.cuda.synchronize()
        assert torch.equal(a, want_a) and torch.equal(b, want_b)
    assert lib.gfv_record_replay(h, 3, 2) != 0 and lib.gfv_record_replay(h, 0, cnt + 1) != 0
    assert lib.gfv_record_begin() == 0
    assert lib.gfv_record_replay(h, 0, cnt) != 0   # inside a recording
    # gfv.cmdlist on a thread whose library-level recording is already open: begin fails, nothing is left half-entered
    # (GFV_CMDLIST_NATIVE=0 keeps its list in Python and never opens a library-level recording: nothing to collide with)
    try:
        if cmdlist.NATIVE:
            with pytest.raises(RuntimeError):
                with cmdlist.record():
                    pass
        assert cmdlist.active() is None
    finally:
        lib.gfv_record_free(lib.gfv_record_end())
    with cmdlist.record() as cl:
        body()
    torch.cuda.synchronize()
    assert len(cl) == 4
    assert lib.gfv_record_free(h) == 0 and lib.gfv_record_replay(h, 0, 1) != 0
