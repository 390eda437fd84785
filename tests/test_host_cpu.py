"""CPU: host logic of the product path - the C ABI library loads and exports every symbol include/gfv.h declares,
the model wrapper has the reference's state_dict layout, the per-batch plan (CSR tables) reproduces the reference's
scatter semantics, the mesh generator obeys the geometric invariants the reference asserts, and the product path
refuses to run without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest
import torch

import cases
from oracle import fvgn_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gfv import lib
    handle = lib.load()  # raises if libgfv.so is missing or a bound symbol is absent
    header = open(os.path.join(ROOT, "include", "gfv.h")).read()
    declared = set(re.findall(r"\b(gfv_[a-z0-9_]+)\s*\(", header))
    declared -= {"gfv_seg_t", "gfv_layer_t", "gfv_rowtile_args_t", "gfv_dw_tile_t", "gfv_wimg_desc_t", "gfv_reduce_piece_t"}
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/gfv.h but not exported by libgfv.so"
    assert declared == set(lib.declared_symbols()), declared ^ set(lib.declared_symbols())
    assert handle.gfv_abi_version() == 3


def test_ctypes_structs_match_header_layout():
    import ctypes as C
    from gfv import lib
    assert C.sizeof(lib.Seg) == 48 and C.sizeof(lib.Layer) == 64
    assert C.sizeof(lib.DwTile) == 6 * 8 + 6 * 4 + 2 * 8 + 8   # (+ gscale)
    assert C.sizeof(lib.RowtileArgs) % 8 == 0
    handle = lib.load()   # the library reports the sizes it was compiled with
    for which, st in enumerate((lib.Seg, lib.Layer, lib.RowtileArgs, lib.WimgDesc, lib.DwTile, lib.ReducePiece, lib.PlanDesc, lib.TransMlp,
                                lib.TransMlpBwd)):
        assert handle.gfv_struct_size(which) == C.sizeof(st), (which, st)


def test_plan_handle_entry_points_reject_bad_arguments_without_a_gpu():
    """gfv_plan_create / destroy / table: argument checks run before anything touches a device."""
    import ctypes as C
    import re
    from gfv import lib
    handle = lib.load()
    plan = C.c_void_p()
    assert handle.gfv_plan_create(None, C.byref(plan), None) == -1
    d = lib.PlanDesc(n_nodes=0)
    assert handle.gfv_plan_create(C.byref(d), C.byref(plan), None) == -1 and not plan.value     # no nodes
    d = lib.PlanDesc(n_nodes=10, n_faces=5)                                                      # faces without edge_index
    assert handle.gfv_plan_create(C.byref(d), C.byref(plan), None) == -1 and not plan.value
    d = lib.PlanDesc(n_nodes=2 ** 31)                                                            # beyond int32 tables
    assert handle.gfv_plan_create(C.byref(d), C.byref(plan), None) == -1
    assert handle.gfv_plan_destroy(None) == 0
    ptr, cnt = C.c_void_p(), C.c_int64()
    assert handle.gfv_plan_table(None, 0, C.byref(ptr), C.byref(cnt)) == -1
    # the binding's table order is the header's enum
    header = open(os.path.join(cases.ROOT, "include", "gfv.h")).read()
    enum = re.search(r"enum \{\s*GFV_PLAN_ES = 0(.*?)GFV_PLAN_TABLE_COUNT", header, re.S).group(0)
    names = re.findall(r"GFV_PLAN_([A-Z0-9_]+)", enum)[:-1]
    assert tuple(names) == lib.PLAN_TABLES


def test_state_dict_layout_matches_reference():
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    m = NNmodel(default_params())
    sd = m.state_dict()
    shapes = O.parameter_shapes()  # = the reference's keys/shapes (golden generator loads them into the reference)
    assert len(sd) == 164 and sum(p.numel() for p in m.parameters()) == 1181539
    assert [k for k in sd if not k.startswith("node_norm")] == list(shapes)
    for k, sh in shapes.items():
        assert tuple(sd[k].shape) == tuple(sh), k
    for k in ("acc_count", "num_accumulations", "acc_sum", "acc_sum_squared"):
        assert f"node_norm.{k}" in sd
    assert float(sd["node_norm.acc_count"]) == 1.0 and float(sd["node_norm.num_accumulations"]) == 1.0


def test_product_path_fails_loudly_on_cpu():
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    graphs = cases.make_graphs("poisson_b1")
    graphs[0].norm_uvp, graphs[0].norm_global = True, True
    m = NNmodel(default_params())
    with pytest.raises(RuntimeError, match="GPU"):
        m(*graphs)
    from gfv import lib
    saved, lib._lib, lib.LIB_PATH = lib._lib, None, "/nonexistent/libgfv.so"
    try:
        with pytest.raises(RuntimeError, match="no fallback"):
            lib.load()
    finally:
        lib._lib, lib.LIB_PATH = saved, os.path.join(os.path.dirname(lib.__file__), "libgfv.so")


def _emulate_seg(src, rowptr, col, scale=None, src_scale=None):
    out = torch.zeros((rowptr.numel() - 1, src.shape[1]), dtype=torch.float64)
    rp, cl = rowptr.tolist(), col.tolist()
    for r in range(len(rp) - 1):
        for k in range(rp[r], rp[r + 1]):
            v = src[cl[k]].double()
            out[r] += v * (float(src_scale[cl[k]]) if src_scale is not None else 1.0)
        if scale is not None:
            out[r] *= float(scale[r])
    return out


@pytest.mark.parametrize("name", ["cavity_mixed_b1", "cyl_cavity_b2"])
def test_plan_reproduces_reference_scatter_semantics(name):
    """The CSR plan + (emulated) segmented reduce must equal the reference's scatter formulation (blocks.py:25-51,84-99)."""
    from gfv.plan import build_plan
    graphs = cases.make_graphs(name)
    pl = build_plan(*graphs)
    N, E = pl.N, pl.E
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, 8, generator=g)
    e = torch.randn(E, 16, generator=g)
    s, r = graphs[0].edge_index
    indeg, outdeg = torch.cat((s, r)), torch.cat((r, s))
    nb_ref = O.scatter_add(x[outdeg], indeg, N)
    assert torch.allclose(_emulate_seg(x, pl.n_rowptr, pl.n_col_node).float(), nb_ref, atol=1e-5)
    agg_ref = O.scatter_add(torch.cat(torch.chunk(e, 2, dim=-1), dim=0), indeg, N)
    assert torch.allclose(_emulate_seg(e.reshape(2 * E, 8), pl.n_rowptr, pl.n_col_edge2).float(), agg_ref, atol=1e-5)
    nbm_ref = O.scatter_mean(agg_ref[outdeg], indeg, N)
    assert torch.allclose(_emulate_seg(agg_ref, pl.n_rowptr, pl.n_col_node, scale=pl.inv_deg).float(), nbm_ref, atol=1e-5)
    # WLSQ stencil: receiver-ordered CSR reproduces the rhs of FVgrad.py:314-325
    G = O.graph_tensors(*graphs)
    out_idx, in_idx = O.wlsq_stencil(G["face_node_x"], G["support_edge"])
    Bfull = O.wlsq_full_B(G["B1"], G["Bx"]).reshape(-1, 5)
    phi = torch.randn(N, 1, generator=g)
    rhs_ref = O.scatter_add(Bfull * (phi[out_idx] - phi[in_idx]), in_idx, N)
    rp, oc = pl.x_rowptr.tolist(), pl.x_out.tolist()
    rhs = torch.zeros(N, 5)
    for i in range(N):
        for k in range(rp[i], rp[i + 1]):
            rhs[i] += pl.x_B[k] * (phi[oc[k]] - phi[i])
    assert torch.allclose(rhs, rhs_ref, rtol=1e-4, atol=1e-5)
    # both orderings list the same multiset of directed stencil entries; cell / face / node incidence tables are consistent
    assert pl.x_rowptr[-1] == pl.xo_rowptr[-1] == pl.S
    assert int(pl.crow[-1]) == int(pl.frow[-1]) == int(pl.nrow[-1]) == pl.Sg
    assert torch.equal(torch.sort(pl.kface[pl.fk.long()])[0], torch.sort(pl.kface)[0])
    assert pl.gnode_ptr.tolist()[-1] == N and pl.gcell_ptr.tolist()[-1] == pl.C
    assert pl.chunk_end.tolist()[-1] == N and all(b - a <= 128 for a, b in zip(pl.chunk_beg.tolist(), pl.chunk_end.tolist()))


def test_mesh_generator_invariants():
    """The reference's own inline validations (parse_to_h5.py:413-414,437-438,457-472) on the synthetic meshes."""
    for name in cases.CASES:
        meshes, _ = cases.make_meshes(name)
        for m in meshes:
            C = m["cell|centroid"].shape[0]
            surf = m["unit_norm_v"] * m["face|face_area"][m["cells_face"]]
            closure = np.zeros((C, 2))
            np.add.at(closure, m["cells_index"], surf)
            assert np.allclose(closure, 0, rtol=1e-5, atol=1e-8)
            assert np.isfinite(m["unit_norm_v"]).all() and (m["cell|cells_area"] > 0).all()
            fn = m["face|face_node"]
            assert (fn[0] < fn[1]).all() and np.unique(fn, axis=1).shape == fn.shape
            assert m["face_node_x"].shape[1] > m["face|face_node"].shape[1]  # k-hop stencil with duplicated 1-hop pairs
            assert np.array_equal(m["support_edge"], np.array([[0, 1], [1, 0]]))


def test_edge_cases_empty_outflow_and_ragged_cells():
    """Poisson case: sigma=[1,0,0], no OUTFLOW face -> press / continuity / y-momentum losses are exactly zero."""
    graphs = cases.make_graphs("poisson_b1")
    out = O.model_forward(O.init_parameters(0), O.new_normalizer_buffers(), graphs)
    assert float(out[0]) == 0.0 and float(out[2]) == 0.0 and float(out[3]) == 0.0 and float(out[1]) > 0
    graphs = cases.make_graphs("cyl_cavity_b2")  # ragged: tri + quad cells, one graph with OUTFLOW and one without
    counts = torch.bincount(graphs[3].face)
    assert set(counts.tolist()) == {3, 4}
    out = O.model_forward(O.init_parameters(0), O.new_normalizer_buffers(), graphs)
    assert float(out[3][0]) > 0 and float(out[3][1]) == 0.0


def test_state_dict_layout_transfvgn_v1():
    from FVMmodel.importer import NNmodel
    from gfv.params import default_params
    m = NNmodel(default_params(net="TransFVGN_v1"))
    sd = {k: v for k, v in m.state_dict().items() if not k.startswith("node_norm.")}
    shapes = O.parameter_shapes({"net": "TransFVGN_v1"})   # = the reference's keys / order (tests/golden/make_golden_v1.py)
    assert list(sd) == list(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


@pytest.mark.parametrize("hidden", [16, 64, 112])
def test_padding_maps_place_every_parameter_once(hidden):
    """FVMmodel/padding.py (`--hidden_size` below 128): the padded tensors have exactly the shapes of the 128 model, every true
    element lands in exactly one padded slot (the rest is zero), and gradients flow back to the true parameters."""
    import torch
    from FVMmodel.importer import NNmodel
    from FVMmodel.padding import pad_parameters
    from gfv.params import default_params
    for net in ("TransFVGN_v2", "TransFVGN_v1"):
        model = NNmodel(default_params(hidden_size=hidden, net=net))
        names, tensors = model.param_names_tensors()
        ref_names, ref_tensors = NNmodel(default_params(net=net)).param_names_tensors()
        assert names == ref_names
        marks = [torch.arange(1, t.numel() + 1, dtype=torch.float64).view(t.shape) for t in tensors]
        padded = pad_parameters(names, marks, hidden)
        for n, p, m, r in zip(names, padded, marks, ref_tensors):
            assert p.shape == r.shape, n
            nz = p[p != 0]
            assert nz.numel() == m.numel() and torch.equal(torch.sort(nz).values, m.reshape(-1)), n
        out = pad_parameters(names, tensors, hidden)
        sum((q * q).sum() for q in out).backward()
        assert all(t.grad is not None and torch.allclose(t.grad, 2 * t.detach()) for t in tensors)
    with pytest.raises(NotImplementedError):
        NNmodel(default_params(hidden_size=72))
    with pytest.raises(NotImplementedError):
        NNmodel(default_params(hidden_size=256))


def test_bench_polygon_workload_builds_on_the_host():
    """`bench.py --workload poly`: the reference's polygon example mesh from the committed reader arrays through the product's
    ingest - the batch the GPU box will time (sizes pinned; no GPU needed to build it)."""
    import importlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    graphs, sz = bench.build_workload("poly", 0, 1, 0, "cpu")
    assert (sz["C"], sz["N"], sz["E"]) == (17436, 27778, 45214) and sz["B"] == 1
    assert graphs[0].x.shape == (27778, 12)


def test_bench_starts_its_own_ranks_when_called_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the 1-GPU driver line reads with N > 1): the parent
    touches no GPU, starts `torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a child process and exits with
    the child's code.  Checked here: the command it builds, that the arguments pass through, that the child is a child (no
    exec), and that a rank started by a launcher does not start ranks again."""
    import importlib
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], port=29611)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    seen = {}

    def fake_call(c, env=None):
        seen["cmd"], seen["env"] = c, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7                                      # the child's exit code is the parent's
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # under a launcher (WORLD_SIZE set) the same invocation goes on as a rank: here it stops at the GPU check, not at a spawn
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert not seen and "MI355X" in str(ex.value.code)


def test_host_thread_placement_helper(tmp_path):
    """gfv.host: CPU-list parsing, the L3 group of a CPU, pin / restore round trip (a no-op where the process already sits on one
    L3 - this container -, never an error)."""
    import os
    from gfv import host
    assert host._parse_cpu_list("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert host._parse_cpu_list("") == set()
    if not hasattr(os, "sched_getaffinity"):
        return
    before = os.sched_getaffinity(0)
    grp = host.l3_group(min(before))
    assert grp == set() or min(before) in grp
    prev = host.pin_to_l3()
    now = os.sched_getaffinity(0)
    assert now <= before and len(now) >= 1
    if prev is not None:
        assert prev == before and now == (host.l3_group(os.sched_getcpu()) & before or now)
    host.restore(prev)
    assert os.sched_getaffinity(0) == before
    groups = host.l3_groups(before)
    assert set().union(*groups) == set(before) and sum(len(g) for g in groups) == len(before)   # a partition of the allowed CPUs
    prev = host.pin_to_l3(rank=3)
    if prev is not None:
        assert os.sched_getaffinity(0) == groups[3 % len(groups)]
    host.restore(prev)
    assert os.sched_getaffinity(0) == before


def test_rank_placement_follows_the_gpus_numa_node(monkeypatch):
    """gfv.host.rank_l3_group on a made-up two-socket host (2 NUMA nodes x 4 L3 groups of 4 CPUs, GPUs 0 - 3 on node 0 and 4 - 7 on
    node 1): every rank gets a group of its own on its GPU's node; unknown topology falls back to the rank-th group."""
    from gfv import host
    cpus = set(range(32))
    monkeypatch.setattr(host, "l3_group", lambda c: set(range(4 * (c // 4), 4 * (c // 4) + 4)))
    monkeypatch.setattr(host, "numa_cpus", lambda n: set(range(16 * n, 16 * n + 16)) if n in (0, 1) else set())
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    got = [host.rank_l3_group(r, cpus, nodes, r) for r in range(8)]
    assert len({frozenset(g) for g in got}) == 8                       # no two ranks share a group
    for r, g in enumerate(got):
        assert g <= host.numa_cpus(nodes[r])                            # ... and each sits on its GPU's node
    # GPUs interleaved over the nodes: the k-th GPU of a node takes that node's k-th group
    inter = [0, 1, 0, 1, 0, 1, 0, 1]
    got = [host.rank_l3_group(r, cpus, inter, r) for r in range(8)]
    assert len({frozenset(g) for g in got}) == 8 and all(g <= host.numa_cpus(inter[r]) for r, g in enumerate(got))
    # nothing known about the GPUs (or about this one): the rank-th group of the host
    groups = host.l3_groups(cpus)
    assert host.rank_l3_group(5, cpus, [], 5) == groups[5] and host.rank_l3_group(5, cpus, [-1] * 8, 5) == groups[5]
    assert host.rank_l3_group(9, cpus, None, None) == groups[9 % 8]
    # a node whose CPUs are not among the allowed ones: fallback again
    assert host.rank_l3_group(2, set(range(16)), [3] * 8, 2) == host.l3_groups(set(range(16)))[2]
