"""Per-batch mesh plan: CSR tables, permuted WLSQ moments and narrowed (int32) indices on the device.

Built ONCE per batch of graphs with torch tensor ops on the batch's own device (host-side plumbing, not the per-step
hot path) and cached on ``graph_node._gfv_plan``.  Everything the HIP kernels index with lives here, so the hot
path has no host synchronisation (the reference syncs on ``mask.any()``, ``batch.max()+1``: FVscheme.py:148,162,
GraphTransolver.py:51).

Index conventions follow SURVEY.md 8(a-0): two-way node adjacency = "all forward edges, then all reverse edges"
(blocks.py:25-31), directed WLSQ stencil = [fx, fx.flip(0), support_edge] (FVgrad.py:264-271), reverse-direction
moment vectors with the odd-order terms negated (FVgrad.py:299-312).  Stable sorts keep the reference's summation
order inside every segment.
"""
from __future__ import annotations

import torch

SLICE_CHUNK = int(__import__("os").environ.get("GFV_SLICE_CHUNK", "64"))  # nodes per partial-token chunk


def _csr(index, n_rows):
    order = torch.argsort(index, stable=True)
    counts = torch.bincount(index, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=index.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32), order


def _ptr_from_batch(batch, B):
    counts = torch.bincount(batch, minlength=B)
    ptr = torch.zeros(B + 1, dtype=torch.int64, device=batch.device)
    ptr[1:] = torch.cumsum(counts, 0)
    return ptr


class MeshPlan:
    pass


# The index tables come from the C ABI's plan handle (gfv_plan_create, csrc/plan.hip) when the batch is on the GPU; the
# torch-op construction below is the same algorithm (stable sorts by destination row) and stays as the CPU-side builder
# (host tests, gloo ranks) - tests/test_plan_gpu.py holds the two bit-identical.  GFV_NATIVE_PLAN=0 forces the torch form.
NATIVE = __import__("os").environ.get("GFV_NATIVE_PLAN", "1") != "0"


class _DevArray:
    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def native_tables(N, C, edge_index, cells_node=None, cells_face=None, cells_index=None, face_node_x=None, support_edge=None):
    """{table name: tensor} from gfv_plan_create on the current stream (tables copied out, handle destroyed)."""
    import ctypes as Ct
    from . import lib as L
    lib = L.load(raw=True)
    dev = edge_index.device
    i64 = lambda t: None if t is None else t.to(torch.int64).contiguous()
    ei, cn, cf, ci, fx, sup = map(i64, (edge_index, cells_node, cells_face, cells_index, face_node_x, support_edge))
    ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
    d = L.PlanDesc(n_nodes=N, n_faces=ei.shape[1], n_cells=C, n_incidences=0 if ci is None else ci.numel(),
                   n_stencil_pairs=0 if fx is None else fx.shape[1], n_support_pairs=0 if sup is None else sup.shape[1],
                   edge_index=ptr(ei), cells_node=ptr(cn), cells_face=ptr(cf), cells_index=ptr(ci), face_node_x=ptr(fx),
                   support_edge=ptr(sup))
    handle = Ct.c_void_p()
    with torch.cuda.device(dev):
        rc = lib.gfv_plan_create(Ct.byref(d), Ct.byref(handle), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"gfv_plan_create failed ({rc}): an index tensor holds an entry outside its range" if rc == -1
                               else f"gfv_plan_create failed ({rc})")
        try:
            out = {}
            for which, name in enumerate(L.PLAN_TABLES):
                p_, n_ = Ct.c_void_p(), Ct.c_int64()
                assert lib.gfv_plan_table(handle, which, Ct.byref(p_), Ct.byref(n_)) == 0
                if n_.value == 0 or not p_.value:
                    out[name] = torch.empty(0, dtype=torch.float32 if name == "INV_DEG" else torch.int32, device=dev)
                    continue
                view = torch.as_tensor(_DevArray(p_.value, n_.value, "<f4" if name == "INV_DEG" else "<i4"), device=dev)
                out[name] = view.clone()
            torch.cuda.current_stream().synchronize()
        finally:
            lib.gfv_plan_destroy(handle)
    return out


def _gnn_part(p, edge_index, N):
    """Two-way node adjacency in CSR order of the receiving node (blocks.py:24-31,82-90)."""
    dev = edge_index.device
    E = edge_index.shape[1]
    i32 = lambda t: t.to(torch.int32).contiguous()
    p.N, p.E, p.device = N, E, dev
    if NATIVE and dev.type == "cuda":
        t = getattr(p, "_native", None) or native_tables(N, 0, edge_index)
        p.es, p.er, p.n_rowptr, p.n_col_node, p.n_col_edge2 = t["ES"], t["ER"], t["N_ROWPTR"], t["N_COL_NODE"], t["N_COL_EDGE2"]
        p.inv_deg, p.s_rowptr, p.s_col, p.r_rowptr, p.r_col = t["INV_DEG"], t["S_ROWPTR"], t["S_COL"], t["R_ROWPTR"], t["R_COL"]
        return p
    s, r = edge_index[0], edge_index[1]
    p.es, p.er = i32(s), i32(r)
    indeg = torch.cat((s, r))
    other = torch.cat((r, s))
    ar = torch.arange(E, device=dev)
    eid2 = torch.cat((2 * ar, 2 * ar + 1))
    p.n_rowptr, order = _csr(indeg, N)
    p.n_col_node = i32(other[order])
    p.n_col_edge2 = i32(eid2[order])
    deg = torch.bincount(indeg, minlength=N).to(torch.float32)
    p.inv_deg = (1.0 / deg.clamp(min=1.0)).contiguous()
    # edges by sender / by receiver (adjoint of the per-side gathers x[s], x[r] of the factored EdgeBlock first layer)
    p.s_rowptr, so = _csr(s, N)
    p.s_col = i32(so)
    p.r_rowptr, ro = _csr(r, N)
    p.r_col = i32(ro)
    return p


def _batch_part(p, nb, B=None):
    """Per-graph node ranges and the node chunks of the slice-token reduction (GraphTransolver.py:64-73)."""
    dev = nb.device
    if B is None:
        B = int(nb.max().item()) + 1 if nb.numel() else 0
    p.B = B
    p.N = int(nb.shape[0])
    p.batch = nb.to(torch.int32).contiguous()
    gnode_ptr = _ptr_from_batch(nb, B)
    p.gnode_ptr = gnode_ptr.to(torch.int32).contiguous()
    gp = gnode_ptr.tolist()
    cb, ce, gcp = [], [], [0]
    for b in range(B):
        for st in range(gp[b], gp[b + 1], SLICE_CHUNK):
            cb.append(st)
            ce.append(min(st + SLICE_CHUNK, gp[b + 1]))
        gcp.append(len(cb))
    p.chunk_beg = torch.tensor(cb, dtype=torch.int32, device=dev)
    p.chunk_end = torch.tensor(ce, dtype=torch.int32, device=dev)
    p.gchunk_ptr = torch.tensor(gcp, dtype=torch.int32, device=dev)
    p.n_chunks = len(cb)
    p.gunit_ptr = torch.arange(B + 1, dtype=torch.int32, device=dev)  # one pre-reduced partial row per graph
    return p


def build_gnn_plan(graph_node):
    """GNN-only plan for the stand-alone block operators, cached on the graph object."""
    key = (graph_node.edge_index.data_ptr(), int(graph_node.x.shape[0]))
    cached = getattr(graph_node, "_gfv_gnn_plan", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    full = getattr(graph_node, "_gfv_plan", None)
    p = full[1] if full is not None else _gnn_part(MeshPlan(), graph_node.edge_index, int(graph_node.x.shape[0]))
    try:
        graph_node._gfv_gnn_plan = (key, p)
    except AttributeError:
        pass
    return p


_batch_plan_cache = {}


def build_batch_plan(batch, plan=None):
    key = (batch.data_ptr(), int(batch.shape[0]), str(batch.device))
    if plan is None:
        hit = _batch_plan_cache.get(key)
        if hit is not None:
            return hit
        plan = MeshPlan()
    if not hasattr(plan, "chunk_beg"):
        _batch_part(plan, batch.reshape(-1))
    if len(_batch_plan_cache) > 64:
        _batch_plan_cache.clear()
    _batch_plan_cache[key] = plan
    return plan


def wlsq_part(p, face_node_x, support_edge, A, B1, Bx, N):
    """Directed stencil [fx, fx.flip(0), support_edge] (FVgrad.py:264-271) in CSR order of the receiving node and of the
    sending node, moment vectors permuted alongside (reverse direction: odd-order terms negated, FVgrad.py:299-312),
    row-normalised A (FVgrad.py:335-337)."""
    dev = face_node_x.device
    i32 = lambda t: t.to(torch.int32).contiguous()
    fx, sup = face_node_x, support_edge
    out_idx = torch.cat((fx[0], fx[1], sup[0]))
    in_idx = torch.cat((fx[1], fx[0], sup[1]))
    M = int(A.shape[-1])          # Taylor terms of the reconstruction order: 2 / 5 / 9 / 14 (FVorder.py:23-72)
    B1 = B1.reshape(-1, M)
    Brev = B1.clone()
    Brev[:, 0:2] *= -1
    if M >= 9:                    # 3rd-order terms are odd too (FVgrad.py:309-310)
        Brev[:, 5:9] *= -1
    Bfull = torch.cat((B1, Brev, Bx.reshape(-1, M)), 0).to(torch.float32)
    t = getattr(p, "_native", None)
    if t is not None:
        p.x_rowptr, p.x_out, o_in = t["X_ROWPTR"], t["X_OUT"], t["X_ORDER"].long()
        p.xo_rowptr, p.xo_in, o_out = t["XO_ROWPTR"], t["XO_IN"], t["XO_ORDER"].long()
    else:
        p.x_rowptr, o_in = _csr(in_idx, N)
        p.x_out = i32(out_idx[o_in])
        p.xo_rowptr, o_out = _csr(out_idx, N)
        p.xo_in = i32(in_idx[o_out])
    p.x_B = Bfull[o_in].contiguous()
    p.xo_B = Bfull[o_out].contiguous()
    # per-receiver sum of the moment vectors: exact fixed-point (2^-40) prefix sum in int64 and a difference per CSR row.
    # Integer arithmetic is associative and wraps modulo 2^64, so the row sums are exact whatever precedes them:
    # deterministic (index_add_ on the device uses float atomics - every gradient would differ in the last bit from run
    # to run) and independent of where the mesh sits in a batch (gfv.pool assembles plans mesh by mesh).
    fx = torch.round(p.x_B.to(torch.float64) * float(2 ** 40)).to(torch.int64).t().contiguous()      # [M, S]
    cs = torch.zeros((M, fx.shape[1] + 1), dtype=torch.int64, device=dev)
    cs[:, 1:] = torch.cumsum(fx, 1)
    rp = p.x_rowptr.to(torch.int64)
    p.sumB = ((cs[:, rp[1:]] - cs[:, rp[:-1]]).to(torch.float64) * float(2.0 ** -40)).t().to(torch.float32).contiguous()
    A = A.to(torch.float32)
    row_norms = torch.norm(A, p=2, dim=2, keepdim=True)          # FVgrad.py:335
    p.rn = (row_norms + 1e-8).reshape(N, M).contiguous()
    # the kernels divide A by rn themselves, in double (FVgrad.py:336; csrc/fvm.hip load_An): `An` holds A as stored
    p.An = A.reshape(N, M * M).contiguous()
    p.S, p.M = int(in_idx.shape[0]), M
    return p


def build_plan(graph_node, graph_node_x, graph_edge, graph_cell, graph_Index):
    p = MeshPlan()
    dev = graph_node.x.device
    i32 = lambda t: t.to(torch.int32).contiguous()
    f32 = lambda t: t.to(torch.float32).contiguous()
    N = graph_node.x.shape[0]
    E = graph_node.edge_index.shape[1]
    C = graph_cell.pos.shape[0]
    B = int(graph_Index.theta_PDE.shape[0])
    p.N, p.E, p.C, p.B, p.device = N, E, C, B, dev
    if NATIVE and dev.type == "cuda":
        p._native = native_tables(N, C, graph_node.edge_index, graph_node.face, graph_edge.face, graph_cell.face,
                                  graph_node_x.face_node_x, graph_node_x.support_edge)
    _gnn_part(p, graph_node.edge_index, N)
    _batch_part(p, graph_node.batch.reshape(-1), B)

    # ---- node data ---------------------------------------------------------------------------------------------
    p.node_type = i32(graph_node.node_type.reshape(-1))
    p.y = f32(graph_node.y[:, 0:2])
    p.pos = f32(graph_node.pos)

    # ---- WLSQ stencil --------------------------------------------------------------------------------------------
    wlsq_part(p, graph_node_x.face_node_x, graph_node_x.support_edge, graph_node_x.A_node_to_node,
              graph_node_x.single_B_node_to_node, graph_node_x.extra_B_node_to_node, N)

    # ---- faces -------------------------------------------------------------------------------------------------
    p.ftype = i32(graph_edge.face_type.reshape(-1))
    p.fpos = f32(graph_edge.pos)
    face_area = graph_edge.face_area.to(torch.float32).reshape(-1, 1)

    # ---- cells: (cell, face, node) incidences -----------------------------------------------------------------
    cells_node, cells_face, cells_index = graph_node.face, graph_edge.face, graph_cell.face
    Svec = graph_cell.cells_face_unv.to(torch.float32).reshape(-1, 2) * face_area[cells_face]  # FVscheme.py:89
    t = getattr(p, "_native", None)
    if t is not None:
        p.crow, oc, p.kface, p.knode, p.kcell = t["CROW"], t["K_ORDER"].long(), t["KFACE"], t["KNODE"], t["KCELL"]
        p.frow, p.fk, p.nrow, p.ncell = t["FROW"], t["FK"], t["NROW"], t["NCELL"]
        del p._native
    else:
        p.crow, oc = _csr(cells_index, C)
        kface, knode, kcell = cells_face[oc], cells_node[oc], cells_index[oc]
        p.kface, p.knode, p.kcell = i32(kface), i32(knode), i32(kcell)
        p.frow, of = _csr(kface, E)
        p.fk = i32(of)
        p.nrow, on = _csr(knode, N)
        p.ncell = i32(kcell[on])
    p.kS = Svec[oc].contiguous()
    p.Sg = int(cells_index.shape[0])
    p.centroid = f32(graph_cell.pos)
    p.area = f32(graph_cell.cells_area.reshape(-1))
    cbatch = graph_cell.batch
    p.cbatch = i32(cbatch)
    p.gcell_ptr = i32(_ptr_from_batch(cbatch, B))

    # ---- per-graph scalars ----------------------------------------------------------------------------------------
    p.theta = f32(graph_Index.theta_PDE)
    p.sigma = f32(graph_Index.sigma)
    p.uvp_dim = f32(graph_Index.uvp_dim)
    p.dt = f32(graph_Index.dt_graph.reshape(-1))
    return p


def _live_tensors(graphs):
    """Caller-owned data the plan holds copies (or aliases) of: boundary targets, node / face types, per-graph PDE
    coefficients.  The reference re-reads them on every forward, so an in-place edit must reach the plan."""
    gn, ge, gi = graphs[0], graphs[2], graphs[4]
    return (gn.y, gn.node_type, ge.face_type, gi.theta_PDE, gi.sigma, gi.uvp_dim, gi.dt_graph)


def _is_pooled(graphs):
    """Batches assembled by gfv.pool.DevicePool carry their plan; its tensors (y, node / face types, PDE coefficients) ARE the
    state - `reset_env` edits them in place - and the stub graph objects hold no copies of them to follow."""
    return getattr(graphs[0], "_gfv_pool_plan", None) is not None


def _live_key(graphs):
    if _is_pooled(graphs):
        return ()
    return tuple((t.data_ptr(), t._version) for t in _live_tensors(graphs))


def _refresh_live(p, graphs):
    """Re-read the live data into the plan's own tensors IN PLACE (pointers held by captured hipGraphs stay valid)."""
    if _is_pooled(graphs):
        return
    y, node_type, face_type, theta, sigma, uvp_dim, dt = _live_tensors(graphs)
    for dst, src in ((p.y, y[:, 0:2]), (p.node_type, node_type.reshape(-1)), (p.ftype, face_type.reshape(-1)),
                     (p.theta, theta), (p.sigma, sigma), (p.uvp_dim, uvp_dim), (p.dt, dt.reshape(-1))):
        if dst.data_ptr() != src.data_ptr() or dst.dtype != src.dtype:
            dst.copy_(src.to(dst.dtype).reshape(dst.shape))


def get_plan(graphs):
    """Plan cached on graph_node.  The structural part is keyed by the identity of the index tensors; the copied
    caller-owned data (Dirichlet targets `y`, node / face types, theta_PDE, sigma, uvp_dim, dt_graph) by data pointer +
    in-place version counter: editing a boundary condition or a PDE coefficient of a reused batch in place refreshes the
    plan's copies on the next call (the reference re-reads graph_node.y / graph_Index on every forward)."""
    graph_node = graphs[0]
    pooled = getattr(graph_node, "_gfv_pool_plan", None)   # batch assembled by gfv.pool.DevicePool: plan comes with it
    if pooled is not None:
        return pooled
    key = (graph_node.edge_index.data_ptr(), graph_node.face.data_ptr(), graphs[1].face_node_x.data_ptr(),
           graph_node.x.shape[0], str(graph_node.x.device))
    cached = getattr(graph_node, "_gfv_plan", None)
    if cached is not None and cached[0] == key:
        live = _live_key(graphs)
        if cached[2] != live:
            _refresh_live(cached[1], graphs)
            graph_node._gfv_plan = (key, cached[1], live)
        return cached[1]
    plan = build_plan(*graphs)
    graph_node._gfv_plan = (key, plan, _live_key(graphs))
    return plan
