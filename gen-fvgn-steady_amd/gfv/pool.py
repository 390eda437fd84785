"""Device-resident state pool.

The reference keeps its meshes and current fields in a CPU ``Data_Pool`` and, every training step, batches five PyG views on
the host and copies them to the device (Load_mesh/Graph_loader.py:131-152,405-480,830-1006; pre_train_Adam.py:150-156),
then copies the prediction back (``payback``, Graph_loader.py:370-396).  Here every mesh lives in HBM together with its
own ``MeshPlan`` (CSR tables, permuted WLSQ moments: built ONCE per mesh); a batch of any meshes of the pool is assembled
on the device by ONE launch of ``gfv_concat_offsets`` - a batch is block diagonal, so every batched plan tensor is the
concatenation of the per-mesh tensors with the node / face / cell / incidence offset of the mesh added to its indices
(the ``__inc__`` rules of ``CustomGraphData``) - and predictions are written back in place.  The assembled plan is
tensor-for-tensor equal to ``build_plan(build_batch(meshes))`` (tests/test_pool_gpu.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import lib as L
from .graph import Data, build_batch
from .plan import MeshPlan, SLICE_CHUNK, build_plan

# attribute -> (offset kind); offsets: n node, e face, e2 2*face, c cell, k incidence, s stencil entry, - none
_INT_ATTRS = dict(es="n", er="n", n_col_node="n", x_out="n", xo_in="n", knode="n", s_col="e", r_col="e", kface="e",
                  n_col_edge2="e2", kcell="c", ncell="c", fk="k", node_type="-", ftype="-")
# rowptr attribute -> (row count kind, nnz offset kind)
_ROWPTRS = dict(n_rowptr=("n", "e2"), s_rowptr=("n", "e"), r_rowptr=("n", "e"), x_rowptr=("n", "s"),
                xo_rowptr=("n", "s"), crow=("c", "k"), frow=("e", "k"), nrow=("n", "k"))
_FLOAT_ATTRS = ("inv_deg", "y", "pos", "x_B", "xo_B", "sumB", "rn", "An", "fpos", "kS", "centroid", "area", "theta", "sigma",
                "uvp_dim", "dt")
_FILL = dict(batch="n", cbatch="c")   # graph id per node / cell

_DESC = np.dtype([("src", "<u8"), ("dst", "<u8"), ("n", "<i8"), ("kind", "<i4"), ("add", "<i4")])


class DevicePool:
    def __init__(self, meshes, fields=None, device="cuda"):
        self.device = torch.device(device)
        self.n = len(meshes)
        self.plans, self.x, self.sizes = [], [], []
        self.bc = [dict(m["bc"]) if "bc" in m else None for m in meshes]       # sampled PDE parameters per mesh
        self.pos64 = [torch.from_numpy(np.ascontiguousarray(m["node|pos"])).to(self.device) for m in meshes]
        for i, m in enumerate(meshes):
            g = build_batch([m], None if fields is None else [fields[i]], device=self.device)
            p = build_plan(*g)
            self.plans.append(p)
            self.x.append(g[0].x.contiguous())          # [N, 3 + 9]: (u, v, p) state + theta_PDE (datapreprocessing)
            self.sizes.append(dict(n=p.N, e=p.E, e2=2 * p.E, c=p.C, k=p.Sg, s=p.S, nchunk=p.n_chunks))
        self._src = {}   # attr -> (ptr per mesh, words per mesh, words per row)
        for a in list(_INT_ATTRS) + list(_ROWPTRS) + list(_FLOAT_ATTRS):
            ts = [getattr(p, a) for p in self.plans]
            assert all(t.is_contiguous() and t.element_size() == 4 for t in ts), a
            self._src[a] = (np.array([t.data_ptr() for t in ts], dtype=np.uint64),
                            np.array([t.numel() for t in ts], dtype=np.int64), tuple(ts[0].shape[1:]), ts[0].dtype)
        self._src["x"] = (np.array([t.data_ptr() for t in self.x], dtype=np.uint64),
                          np.array([t.numel() for t in self.x], dtype=np.int64), tuple(self.x[0].shape[1:]), torch.float32)

    # ------------------------------------------------------------------------------------------------------------
    def batch(self, indices):
        """-> (graphs, plan): the five graph objects (carrying what the HIP model reads: ``graph_node.x`` and the plan)
        for the meshes `indices` of the pool, assembled on the device."""
        idx = [int(i) for i in indices]
        B = len(idx)
        sz = [self.sizes[i] for i in idx]
        off = {k: np.concatenate(([0], np.cumsum([s[k] for s in sz]))).astype(np.int64) for k in ("n", "e", "e2", "c", "k", "s")}
        off["-"] = np.zeros(B + 1, dtype=np.int64)
        dev = self.device
        descs = []
        out = {}

        def alloc(a, words, dtype, row_shape):
            t = torch.empty((words,), dtype=dtype, device=dev)
            out[a] = t.view((-1,) + row_shape) if row_shape else t
            return t.data_ptr()

        def add_pieces(a, kind, adds, extra_last=0):
            ptrs, words, row_shape, dtype = self._src[a]
            w = words[idx].copy()
            if extra_last:                     # rowptr: n_i entries per mesh, n_last + 1 for the last one
                w -= 1
                w[-1] += 1
            base = alloc(a, int(w.sum()), dtype, row_shape)
            starts = np.concatenate(([0], np.cumsum(w)[:-1]))
            d = np.zeros(B, dtype=_DESC)
            d["src"], d["dst"], d["n"], d["kind"], d["add"] = ptrs[idx], base + 4 * starts.astype(np.uint64), w, kind, adds
            descs.append(d)

        for a, k in _INT_ATTRS.items():
            add_pieces(a, 0 if k == "-" else 1, off[k][:B])
        for a, (_, nnz) in _ROWPTRS.items():
            add_pieces(a, 1, off[nnz][:B], extra_last=1)
        for a in _FLOAT_ATTRS:
            add_pieces(a, 0, 0)
        add_pieces("x", 0, 0)
        for a, k in _FILL.items():
            counts = np.array([s[k] for s in sz], dtype=np.int64)
            base = alloc(a, int(counts.sum()), torch.int32, ())
            d = np.zeros(B, dtype=_DESC)
            d["dst"], d["n"], d["kind"], d["add"] = base + 4 * off[k][:B].astype(np.uint64), counts, 2, np.arange(B)
            descs.append(d)
        # small per-graph pointer arrays and the slice-token chunks: computed on the host, one upload
        chunk_beg, chunk_end, gcp = [], [], [0]
        for b in range(B):
            n0, n1 = int(off["n"][b]), int(off["n"][b + 1])
            st = np.arange(n0, n1, SLICE_CHUNK)
            chunk_beg.append(st)
            chunk_end.append(np.minimum(st + SLICE_CHUNK, n1))
            gcp.append(gcp[-1] + len(st))
        small = dict(gnode_ptr=off["n"], gcell_ptr=off["c"], gchunk_ptr=np.array(gcp), gunit_ptr=np.arange(B + 1),
                     chunk_beg=np.concatenate(chunk_beg), chunk_end=np.concatenate(chunk_end))
        blob = np.concatenate([np.concatenate(descs).view(np.int32)] + [v.astype(np.int32) for v in small.values()])
        dblob = torch.from_numpy(blob).to(dev, non_blocking=True)
        ndesc = sum(len(d) for d in descs)
        L.check(L.load().gfv_concat_offsets(dblob.data_ptr(), ndesc, 48, L.stream_ptr()), "gfv_concat_offsets")

        p = MeshPlan()
        for a, t in out.items():
            if a != "x":
                setattr(p, a, t)
        pos = ndesc * (_DESC.itemsize // 4)
        for a, v in small.items():
            setattr(p, a, dblob[pos:pos + len(v)])
            pos += len(v)
        p.N, p.E, p.C, p.B = int(off["n"][B]), int(off["e"][B]), int(off["c"][B]), B
        p.S, p.Sg, p.n_chunks, p.device = int(off["s"][B]), int(off["k"][B]), gcp[-1], dev
        p.M = self.plans[idx[0]].M                         # WLSQ Taylor terms (all meshes of a pool share the order)
        p._keep = dblob                                    # the small arrays are views of the upload buffer
        graph_node = Data(x=out["x"], batch=p.batch, pos=p.pos, num_graphs=B, norm_uvp=True, norm_global=True)
        graph_node._gfv_pool_plan = p
        graphs = (graph_node, Data(num_graphs=B), Data(num_graphs=B), Data(num_graphs=B),
                  Data(theta_PDE=p.theta, sigma=p.sigma, uvp_dim=p.uvp_dim, dt_graph=p.dt.view(-1, 1), num_graphs=B))
        self._last = (idx, off["n"])
        return graphs, p

    # ------------------------------------------------------------------------------------------------------------
    def reset_env(self, i, **sampled):
        """Re-select the boundary condition of mesh `i` and restart its field, in place on the device
        (Data_Pool.reset_env -> CFDdatasetBase.transform_mesh, Graph_loader.py:154-229, Load_mesh.py:82-130,134-246,
        524-565).  `sampled`: any of U, rho, mu, source, aoa, dt, L - the values the reference draws from the ranges of
        BC.json (select_PDE_coef; the draw itself stays with the caller).  Geometry, stencil and moment matrices do not
        depend on them and stay as they are (the reference rebuilds identical copies); what changes is theta_PDE, dt, the
        dimensional scales, the Dirichlet targets and the initial field: 31 scalars from the host, the per-node part
        (inlet velocity profile -> target, initial state) with torch ops on the device."""
        from . import meshgen
        if self.bc[i] is None:
            raise ValueError("the mesh was given without its 'bc' record (gfv.meshgen.finish_mesh keeps it)")
        bc = self.bc[i]
        bc.update({k: float(v) for k, v in sampled.items()})
        theta, dt_graph, uvp_dim = meshgen.pde_coefficients(bc)
        p, dev = self.plans[i], self.device
        small = torch.from_numpy(np.concatenate((theta.reshape(-1), dt_graph.reshape(-1), uvp_dim.reshape(-1)))).to(dev)
        p.theta.copy_(small[0:9].view(1, 9))
        p.dt.copy_(small[9:10])
        p.uvp_dim.copy_(small[10:13].view(1, 3))
        U = float(bc["U"])
        pos, nt = self.pos64[i], p.node_type
        inlet = (nt == meshgen.INFLOW) | (nt == meshgen.IN_WALL) | (nt == meshgen.PRESS_POINT)

        def profile(pp):   # gfv.meshgen.velocity_profile in float64
            u = torch.zeros(pp.shape[0], dtype=torch.float64, device=dev)
            if pp.shape[0] == 0:
                return u
            if bc["inlet_type"] == "parabolic":
                yy = pp[:, 1] - pp[:, 1].min()
                ymax, ymin = yy.max(), yy.min()
                u = 6 * U * yy * (((ymax - ymin) - yy) / (ymax - ymin) ** 2)
            elif bc["inlet_type"] == "uniform":
                u = u + U
            else:
                raise ValueError(bc["inlet_type"])
            return u

        u = profile(pos).to(torch.float32)
        u[inlet] = profile(pos[inlet]).to(torch.float32)
        u = torch.where(nt == meshgen.WALL, torch.zeros_like(u), u)
        u = torch.where(nt == meshgen.IN_WALL, u / 2.0, u)
        x = self.x[i]
        x[:, 0] = u
        x[:, 1:3] = 0.0
        x[:, 3:12] = p.theta
        p.y[:, 0] = u / np.float32(U)
        p.y[:, 1] = 0.0

    # ------------------------------------------------------------------------------------------------------------
    def payback(self, indices, uvp_node):
        """Write the batch's predicted (u, v, p) back into the pool (Data_Pool.payback, Graph_loader.py:370-396)."""
        idx = [int(i) for i in indices]
        o = 0
        for i in idx:
            n = self.sizes[i]["n"]
            self.x[i][:, 0:3].copy_(uvp_node[o:o + n, 0:3])
            o += n
