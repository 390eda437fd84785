"""the device-resident state pool.  A batch assembled on the device by one concat-with-offsets launch
must be tensor-for-tensor the plan that `build_plan(build_batch(...))` builds from scratch, give the same model outputs
and gradients, and `payback` must update the pooled fields (Graph_loader.py:370-396,405-480)."""
import pytest
import torch

import cases
from oracle import fvgn_oracle as O

pytestmark = pytest.mark.gpu


def _meshes():
    from gfv import meshgen
    ms, fs = [], []
    for fac, kw, U, seed in (("raw_tri_channel_cylinder", dict(nx=30, ny=6, quad_fraction=0.0, seed=21), 0.15, 5),
                             ("raw_quad_cavity", dict(n=7, jitter=0.1, tri_fraction=0.3, seed=13), 1.0, 3),
                             ("raw_tri_channel_cylinder", dict(nx=36, ny=7, quad_fraction=0.3, seed=22), 0.25, 6),
                             ("raw_poisson_cavity", dict(n=6, seed=14), None, 4)):
        m = meshgen.finish_mesh(getattr(meshgen, fac)(**kw), U=U)
        ms.append(m)
        fs.append(meshgen.random_fields(m, seed=seed))
    return ms, fs


@pytest.mark.parametrize("indices", [[2, 0], [1], [3, 1, 0, 2], [0, 0]])
def test_pooled_batch_equals_rebuilt_batch(indices):
    from gfv.graph import build_batch
    from gfv.plan import build_plan
    from gfv.pool import DevicePool
    ms, fs = _meshes()
    pool = DevicePool(ms, fs)
    graphs, plan = pool.batch(indices)
    ref_graphs = build_batch([ms[i] for i in indices], [fs[i] for i in indices], device="cuda")
    ref = build_plan(*ref_graphs)
    torch.cuda.synchronize()
    assert torch.equal(graphs[0].x, ref_graphs[0].x)
    checked = 0
    for k, v in vars(ref).items():
        if torch.is_tensor(v):
            mine = getattr(plan, k)
            assert mine.shape == v.shape and mine.dtype == v.dtype, (k, mine.shape, v.shape)
            assert torch.equal(mine, v), k
            checked += 1
        elif isinstance(v, int):
            assert getattr(plan, k) == v, k
    assert checked >= 45


def test_pooled_batch_through_the_model_and_payback():
    from FVMmodel.importer import NNmodel
    from gfv.graph import build_batch
    from gfv.params import default_params
    from gfv.pool import DevicePool
    ms, fs = _meshes()
    pool = DevicePool(ms, fs)
    P = O.init_parameters(cases.WEIGHT_SEED)

    def model():
        m = NNmodel(default_params(dataset_size=1))
        sd = m.state_dict()
        for k, v in P.items():
            sd[k].copy_(v)
        m.load_state_dict(sd)
        return m.cuda()

    idx = [2, 0, 3]
    hp = O.DEFAULT_HYPER
    outs = []
    for pooled in (True, False):
        if pooled:
            graphs, _ = pool.batch(idx)
        else:
            graphs = build_batch([ms[i] for i in idx], [fs[i] for i in idx], device="cuda")
            graphs[0].norm_uvp, graphs[0].norm_global = True, True
        m = model()
        out = m(*graphs)
        loss = torch.mean(torch.log(hp["loss_press"] * out[3] + hp["loss_cont"] * out[0] + hp["loss_mom"] * out[1]
                                    + hp["loss_mom"] * out[2]))
        loss.backward()
        outs.append(([o.detach().clone() for o in out], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    for a, b in zip(*[o[0] for o in outs]):
        assert torch.equal(a, b)
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
    # payback: the prediction replaces (u, v, p) of every mesh of the batch, theta columns untouched
    uvp = outs[0][0][4]
    before = [pool.x[i].clone() for i in range(4)]
    pool.payback(idx, uvp)
    o = 0
    for i in idx:
        n = pool.x[i].shape[0]
        assert torch.equal(pool.x[i][:, 0:3], uvp[o:o + n]) and torch.equal(pool.x[i][:, 3:], before[i][:, 3:])
        o += n
    assert torch.equal(pool.x[1], before[1])


def test_device_preprocessing_matches_host():
    """k-hop stencil and WLSQ moments computed by HIP kernels on the GPU vs the host (numpy) code of
    gfv.meshgen, which is validated against the reference's pipeline: stencil pairs identical, float64 moments to 1e-10
    (prefix-sum differences; the consumers use them in fp32)."""
    import numpy as np
    from gfv import device_prep, meshgen
    nx, ny = meshgen.cylinder_grid_for_cells(12000)
    m = meshgen.derive_geometry(meshgen.raw_tri_channel_cylinder(nx=nx, ny=ny, quad_fraction=0.2, seed=8))
    pos, fn = m["node|pos"], m["face|face_node"]
    n = pos.shape[0]
    host_pairs = meshgen.k_hop_pairs(fn, n, 2)
    dev_pairs = device_prep.k_hop_pairs(torch.from_numpy(fn).cuda(), n, 2)
    assert np.array_equal(dev_pairs.cpu().numpy(), host_pairs)
    # 1 and 3 hops, and a face list with repeated faces (the HIP kernels build their own adjacency: csrc/prep.hip)
    sx, sy = meshgen.cylinder_grid_for_cells(2500)
    ms = meshgen.derive_geometry(meshgen.raw_tri_channel_cylinder(nx=sx, ny=sy, quad_fraction=0.3, seed=3))
    fs, ns = ms["face|face_node"], ms["node|pos"].shape[0]
    for k in (1, 3):
        assert np.array_equal(device_prep.k_hop_pairs(torch.from_numpy(fs).cuda(), ns, k).cpu().numpy(), meshgen.k_hop_pairs(fs, ns, k))
    dup = np.concatenate((fs, fs[:, ::7], fs[::-1, ::5]), axis=1)
    assert np.array_equal(device_prep.k_hop_pairs(torch.from_numpy(dup).cuda(), ns, 2).cpu().numpy(), meshgen.k_hop_pairs(fs, ns, 2))
    fx = np.concatenate((m["face_node_x_base"], host_pairs), axis=1)
    sup = np.array([[0, 1], [1, 0]], dtype=np.int64)
    for order in ("2nd", "1st", "3rd", "4th"):
        A, B1, Bx = meshgen.wlsq_moments(pos, fx, sup, order)
        dA, dB1, dBx = device_prep.wlsq_moments(torch.from_numpy(pos).cuda(), torch.from_numpy(fx).cuda(),
                                                torch.from_numpy(sup).cuda(), order)
        for mine, ref in ((dA, A), (dB1, B1), (dBx, Bx)):
            ref = torch.from_numpy(np.asarray(ref))
            assert mine.shape == ref.shape
            # per column pair: the higher-order columns are many orders of magnitude smaller than the first
            scale = ref.abs().amax(dim=0, keepdim=True) + 1e-300
            err = float(((mine.cpu() - ref).abs() / scale).max())
            assert err < 1e-9, (order, err)


def test_reset_env_on_the_device_equals_host_transform_mesh():
    """SURVEY.md rows f1 / f2 (`Data_Pool.reset_env` -> `transform_mesh`, Graph_loader.py:154-229, Load_mesh.py:524-565):
    re-selecting a mesh's boundary condition in the pool (new inlet velocity, viscosity, source, angle of attack, time
    step) must leave the batch exactly as if the mesh had been re-finished on the host with those values - PDE
    coefficients, scales and the plan bit for bit, targets and the restarted field to fp32 round-off of the float64
    velocity profile."""
    import numpy as np
    from gfv import meshgen
    from gfv.graph import build_batch
    from gfv.plan import build_plan
    from gfv.pool import DevicePool
    specs = (("raw_tri_channel_cylinder", dict(nx=30, ny=6, quad_fraction=0.0, seed=21), 0.15),
             ("raw_quad_cavity", dict(n=7, jitter=0.1, tri_fraction=0.3, seed=13), 1.0))
    raws = [getattr(meshgen, fac)(**kw) for fac, kw, _ in specs]
    ms = [meshgen.finish_mesh(r, U=U) for r, (_, _, U) in zip(raws, specs)]
    pool = DevicePool(ms)                                   # fields: each mesh's own initial state
    new = dict(U=0.31, mu=2.0e-3, source=0.05, aoa=3.0, dt=0.02)
    pool.reset_env(0, **new)
    graphs, plan = pool.batch([0, 1])
    raw0 = dict(raws[0])
    raw0["bc"] = dict(raws[0]["bc"], **new)
    ref_ms = [meshgen.finish_mesh(raw0), ms[1]]
    ref_graphs = build_batch(ref_ms, device="cuda")
    ref = build_plan(*ref_graphs)
    torch.cuda.synchronize()
    assert float(ref.theta[0, 6]) != float(build_plan(*build_batch(ms, device="cuda")).theta[0, 6])   # it did change
    for k in ("theta", "dt", "uvp_dim", "sigma"):
        assert torch.equal(getattr(plan, k), getattr(ref, k)), k
    for k, v in vars(ref).items():                          # everything geometric is untouched
        if torch.is_tensor(v) and k not in ("theta", "dt", "uvp_dim", "y"):
            assert torch.equal(getattr(plan, k), v), k
    assert torch.equal(graphs[0].x[:, 3:12], ref_graphs[0].x[:, 3:12])
    for mine, want in ((graphs[0].x[:, 0:3], ref_graphs[0].x[:, 0:3]), (plan.y, ref.y)):
        assert float((mine - want).abs().max()) <= 2e-7 * float(want.abs().max())
    n0 = ms[0]["node|pos"].shape[0]
    assert float(graphs[0].x[:n0, 0].abs().max()) > 0 and float(plan.y[:n0, 0].max()) > 1.0   # a parabolic inlet profile


def test_trainstep_on_pooled_batches_set_batch_and_step():
    """`TrainStep` over batches of a `DevicePool` (`set_batch` + `step`, the path gfv/pool.py advertises and
    profiles/pool_timing.py times).  Pooled batches carry their plan; the stub graph objects have no `y` / `node_type` /
    `face_type` for the live-data check to read.  The steps must equal those of a TrainStep over the rebuilt batches, and a
    `reset_env` edit of the pooled plan must reach the next step (its tensors are the state)."""
    from FVMmodel.importer import NNmodel
    from gfv.graph import build_batch
    from gfv.params import default_params
    from gfv.pool import DevicePool
    from gfv.trainer import TrainStep
    ms, fs = _meshes()
    P = O.init_parameters(cases.WEIGHT_SEED)

    def model():
        m = NNmodel(default_params(dataset_size=1))
        sd = m.state_dict()
        for k, v in P.items():
            sd[k].copy_(v)
        m.load_state_dict(sd)
        return m.cuda()

    batches = ([2, 0], [1, 3], [2, 0])
    results = []
    for pooled in (True, False):
        pool = DevicePool(ms, fs)
        mk = (lambda idx: pool.batch(idx)[0]) if pooled else (
            lambda idx: build_batch([ms[i] for i in idx], [fs[i] for i in idx], device="cuda"))
        g0 = mk(batches[0])
        if not pooled:
            g0[0].norm_uvp, g0[0].norm_global = True, True
        ts = TrainStep(model(), g0, lr=1e-3, use_graph="list" if pooled else False)
        losses = []
        for k, idx in enumerate(batches):
            if k:
                g = mk(idx)
                if not pooled:
                    g[0].norm_uvp, g[0].norm_global = True, True
                ts.set_batch(g)
            for _ in range(4):     # (command list: two warm-up steps, the recording, one replay)
                losses.append(float(ts.step()))
        results.append((losses, ts.flat_p.clone()))
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])
