"""The C ABI's mesh plan handle (gfv_plan_create / gfv_plan_table / gfv_plan_destroy, csrc/plan.hip) against
the torch-op plan builder of gfv/plan.py: every table bit-identical (int32 index work), on synthetic batches, on a polygon
mesh and on the reference's NACA0012 mesh; index validation; repeated create / destroy."""
import ctypes as C

import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _plans(graphs):
    from gfv import plan as PL
    hg = tuple(g.clone().to("cuda") for g in graphs)
    out = []
    for native in (True, False):
        PL.NATIVE = native
        try:
            out.append(PL.build_plan(*hg))
        finally:
            PL.NATIVE = True
    return out


def _same(a, b):
    names = sorted(k for k, v in vars(b).items() if torch.is_tensor(v))
    assert names == sorted(k for k, v in vars(a).items() if torch.is_tensor(v))
    for k in names:
        x, y = getattr(a, k), getattr(b, k)
        assert x.dtype == y.dtype and x.shape == y.shape, (k, x.dtype, y.dtype, x.shape, y.shape)
        assert torch.equal(x, y), k
    return names


@pytest.mark.parametrize("case", ["cyl_cavity_b2", "cyl_b3", "poisson_b1"])
def test_native_plan_equals_torch_plan_on_synthetic_batches(case):
    a, b = _plans(cases.make_graphs(case))
    names = _same(a, b)
    for k in ("n_rowptr", "n_col_node", "n_col_edge2", "s_col", "r_col", "x_rowptr", "x_out", "x_B", "xo_in", "xo_B", "sumB",
              "crow", "kface", "knode", "kcell", "kS", "frow", "fk", "nrow", "ncell", "inv_deg"):
        assert k in names, k


def test_native_plan_equals_torch_plan_on_polygon_and_airfoil_meshes(golden_dir):
    graphs, _fx, _mesh = cases.poly_cylinder(golden_dir)
    _same(*_plans(graphs))
    graphs, _fx, _mesh = cases.real_mesh("real_naca0012", golden_dir)
    a, b = _plans(graphs)
    _same(a, b)
    assert a.E == 47545 and a.Sg == int(graphs[3].face.numel())


def test_plan_handle_through_ctypes_validates_and_can_be_recreated():
    from gfv import lib as L
    lib = L.load()
    graphs = tuple(g.to("cuda") for g in cases.make_graphs("cavity_mixed_b1"))
    gn, gx, ge, gc, _ = graphs
    N, Cn = gn.x.shape[0], gc.pos.shape[0]
    st = torch.cuda.current_stream().cuda_stream

    def desc(edge_index):
        return L.PlanDesc(n_nodes=N, n_faces=edge_index.shape[1], n_cells=Cn, n_incidences=gc.face.numel(),
                          n_stencil_pairs=gx.face_node_x.shape[1], n_support_pairs=gx.support_edge.shape[1],
                          edge_index=edge_index.data_ptr(), cells_node=gn.face.data_ptr(), cells_face=ge.face.data_ptr(),
                          cells_index=gc.face.data_ptr(), face_node_x=gx.face_node_x.data_ptr(),
                          support_edge=gx.support_edge.data_ptr())

    ei = gn.edge_index.contiguous()
    handles = []
    for _ in range(3):
        h = C.c_void_p()
        assert lib.gfv_plan_create(C.byref(desc(ei)), C.byref(h), st) == 0 and h.value
        handles.append(h)
    sizes = (C.c_int64 * 5)()
    assert lib.gfv_plan_sizes(handles[0], sizes) == 0
    assert list(sizes) == [N, ei.shape[1], Cn, gc.face.numel(), 2 * gx.face_node_x.shape[1] + gx.support_edge.shape[1]]
    ptr, cnt = C.c_void_p(), C.c_int64()
    assert lib.gfv_plan_table(handles[0], L.PLAN_TABLES.index("N_ROWPTR"), C.byref(ptr), C.byref(cnt)) == 0
    assert cnt.value == N + 1 and ptr.value
    assert lib.gfv_plan_table(handles[0], len(L.PLAN_TABLES), C.byref(ptr), C.byref(cnt)) == -1
    for h in handles:
        assert lib.gfv_plan_destroy(h) == 0
    # an edge that names a node outside the mesh: refused, no handle
    bad = ei.clone()
    bad[1, 3] = N
    h = C.c_void_p()
    assert lib.gfv_plan_create(C.byref(desc(bad)), C.byref(h), st) == -1 and not h.value
    from gfv import plan as PL
    with pytest.raises(RuntimeError, match="outside its range"):
        PL.native_tables(N, 0, bad)
