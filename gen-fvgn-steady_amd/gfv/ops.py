"""Thin tensor-level wrappers over the C ABI (no autograd here; see gfv/functions.py).

Every function launches hand-written HIP kernels from libgfv.so on torch's current stream.  PyTorch is used only
for device memory (output allocation)."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def _p(t):
    return None if t is None else t.data_ptr()


def seg_gather_sum(src, rowptr, col, n_rows, scale=None, src_scale=None, out=None, accumulate=False):
    """out[r] = scale[r] * sum_{k in row r} src_scale[col[k]] * src[col[k]]   (src [n_src, F] contiguous fp32)."""
    lib = L.load()
    F = src.shape[-1]
    L.f32c(src)
    if out is None:
        assert not accumulate
        out = torch.empty((n_rows, F), dtype=torch.float32, device=src.device)
    rc = lib.gfv_seg_gather_sum_ex(_p(src), _p(L.i32c(rowptr)), _p(L.i32c(col)), _p(scale), _p(src_scale), _p(out),
                                   n_rows, F, 1 if accumulate else 0, col.shape[0], src.shape[0], L.stream_ptr())
    L.check(rc, "gfv_seg_gather_sum")
    return out


def seg_gather_sum_ln(y, stats, gamma, beta, rowptr, col, n_rows, out=None):
    """out[r] = sum_{k in row r} LayerNorm(y)[half-row col[k]]  (y [E,128] pre-LayerNorm rows + their saved (mean, 1 / std); a
    half-row c is columns 64 (c & 1) .. of row c >> 1): include/gfv.h gfv_seg_gather_sum_ln."""
    lib = L.load()
    assert y.dim() == 2 and y.shape[1] == 128 and y.is_contiguous() and stats.shape == (y.shape[0], 2) and stats.is_contiguous()
    if out is None:
        out = torch.empty((n_rows, 64), dtype=torch.float32, device=y.device)
    rc = lib.gfv_seg_gather_sum_ln(_p(L.f32c(y)), _p(L.f32c(stats)), _p(L.f32c(gamma)), _p(L.f32c(beta)), _p(L.i32c(rowptr)),
                                   _p(L.i32c(col)), _p(out), n_rows, col.shape[0], 2 * y.shape[0], L.stream_ptr())
    L.check(rc, "gfv_seg_gather_sum_ln")
    return out


def gather_pair(a, s, r, base=None, out=None):
    lib = L.load()
    F = a.shape[-1]
    E = s.shape[0]
    if out is None:
        out = torch.empty((E, 2 * F), dtype=torch.float32, device=a.device)
    rc = lib.gfv_gather_pair(_p(L.f32c(a)), _p(L.i32c(s)), _p(L.i32c(r)), _p(base), _p(out), E, F, L.stream_ptr())
    L.check(rc, "gfv_gather_pair")
    return out


def transpose(w, out=None, col0=0, ncols=None):
    """out [ncols, rows] = w[:, col0:col0+ncols]^T   (w contiguous [rows, cols])."""
    lib = L.load()
    rows, cols = w.shape
    ncols = cols - col0 if ncols is None else ncols
    if out is None:
        out = torch.empty((ncols, rows), dtype=torch.float32, device=w.device)
    rc = lib.gfv_transpose(L.f32c(w).data_ptr() + 4 * col0, cols, _p(out), rows, ncols, L.stream_ptr())
    L.check(rc, "gfv_transpose")
    return out


def reduce_partials(partial, n_chunks, n, out=None, accumulate=False):
    lib = L.load()
    if out is None:
        out = torch.empty((n,), dtype=torch.float32, device=partial.device)
    rc = lib.gfv_reduce_partials(_p(partial), n_chunks, n, _p(out), 1 if accumulate else 0, L.stream_ptr())
    L.check(rc, "gfv_reduce_partials")
    return out


def reduce_multi(pieces):
    """pieces: list of dicts (partial ptr / tensor, out ptr / tensor, n_chunks, chunk_stride, rows, cols, ld_in, ld_out):
    one launch for all of them (include/gfv.h, gfv_reduce_multi)."""
    lib = L.load()
    arr = (L.ReducePiece * len(pieces))()
    for a, p in zip(arr, pieces):
        a.partial = p["partial"].data_ptr() if torch.is_tensor(p["partial"]) else p["partial"]
        a.out = p["out"].data_ptr() if torch.is_tensor(p["out"]) else p["out"]
        a.n_chunks, a.chunk_stride, a.rows, a.cols = p["n_chunks"], p["chunk_stride"], p["rows"], p["cols"]
        a.ld_in = p.get("ld_in", p["cols"])
        a.ld_out = p.get("ld_out", p["cols"])
    L.check(lib.gfv_reduce_multi(arr, len(pieces), L.stream_ptr()), "gfv_reduce_multi")


def csr_prologue_enabled():
    """GFV_CSR_FUSE=0 keeps the neighbour sums as launches of their own."""
    import os
    return os.environ.get("GFV_CSR_FUSE", "1") != "0"


def rowtile_tiles(M):
    return (M + 63) // 64


def ln_rows(M):
    """Rows of an `ln_partial` buffer to provide for a chain launch over M rows (one per 32 rows; csrc/cbwd.hip)."""
    return (M + 31) // 32


def last_ln_rows():
    """How many `ln_partial` rows the last chain launch filled (include/gfv.h: gfv_rowtile_last_ln_rows)."""
    return L.load().gfv_rowtile_last_ln_rows()


def gscale_ld(M):
    """Row length of a `gscale` buffer for M rows: one float per group of 16 rows, whole 128-row workgroups."""
    return (M + 127) // 128 * 8


class Seg:
    """Input segment: rows of `t` (optionally gathered by int32 `idx`), `width` valid columns, row stride `ld`."""

    def __init__(self, t, idx=None, width=None, ld=None, offset=0, csr=None, scale=None, save=None):
        """csr = (rowptr, col): the segment row is scale[m] * the sum of the rows col[rowptr[m] : rowptr[m+1]] of `t` (a
        segmented sum formed in the launch's prologue, include/gfv.h gfv_seg_t); save: [M,128] buffer for the assembled rows."""
        self.t, self.idx = t, idx
        self.width = t.shape[-1] if width is None else width
        self.ld = t.stride(0) if ld is None else ld
        self.offset = offset  # column offset (floats) into the row
        self.csr, self.scale, self.save = csr, scale, save
        assert csr is None or idx is None

    def fill(self, cs):
        cs.ptr = self.t.data_ptr() + 4 * self.offset
        cs.idx = _p(self.idx) if self.csr is None else self.csr[1].data_ptr()
        cs.width = self.width
        cs.ld = self.ld
        cs.csr_rowptr = None if self.csr is None else self.csr[0].data_ptr()
        cs.csr_scale = _p(self.scale)
        cs.save = _p(self.save)


class LayerSpec:
    def __init__(self, W, bias=None, op=L.OP_NONE, save=None, aux=None, stack=None, bias2=None, stack_cols=None):
        """stack (+ bias2): a second [128, K] weight block whose rows follow W's 128 rows in a virtual [256, K] last layer
        (two Linear layers applied to the same input in one launch, outputs = the two 128-wide chunks).
        stack_cols: a second [128, 128] block whose columns follow W's in a virtual [128, 256] layer (the sum of two
        Linear layers applied to two 128-wide input segments).  Either exists only as a split-fp16 image - the blocks'
        images back to back; `stack_ready` tells whether it can be used, a row stack falls back to one launch per block."""
        self.W, self.bias, self.op, self.save, self.aux = W, bias, op, save, aux
        self.stack, self.bias2, self.stack_cols = stack, bias2, stack_cols


class WeightImages:
    """Split-fp16 images (include/gfv.h, gfv_weight_images) of the weight blocks a set of chain launches uses.

    The set builds itself: the first launch that brings a weight block (pointer, row stride, shape) gets its image made
    on the spot; from then on `build()` refreshes every known image in one launch per step (the weights change with
    every optimizer step).  Only blocks inside `static` address ranges (parameters, the engine's persistent transposed
    copies) qualify - a temporary would be gone by the next step; anything else runs on the fp32 MFMA.  `wmax` is the
    device scalar max|W| all images of an engine are scaled with (the caller keeps it current)."""

    def __init__(self, device, wmax):
        self.device, self.wmax = device, wmax
        self.images = {}                            # key -> uint8 tensor
        self.valid = set()                          # keys whose image holds this step's values
        self.static = []                            # sorted [(start, end)]
        self._desc, self._desc_keys, self._max_frags = None, (), 0
        self._retired = []
        self._bf = None                             # the valid images hold bf16 high parts (built under gfv_set_f16split(3))

    def add_static(self, tensors):
        for t in tensors:
            st = t.untyped_storage()
            self.static.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
        self.static.sort()

    def _is_static(self, ptr):
        import bisect
        i = bisect.bisect_right(self.static, (ptr, 1 << 62)) - 1
        return i >= 0 and self.static[i][0] <= ptr < self.static[i][1]

    @staticmethod
    def _parts(key, img):
        """(block key, destination pointer) per described weight block: a stacked image is its blocks' images back to
        back (the image layout is [128-row pass][k group], include/gfv.h)."""
        if key[0] != "stack":
            return [(key, img.data_ptr())]
        per = img.numel() // (len(key) - 1)
        return [(k, img.data_ptr() + i * per) for i, k in enumerate(key[1:])]

    @staticmethod
    def _upload(items, device):
        parts = [pt for key, img in items for pt in WeightImages._parts(key, img)]
        descs = (L.WimgDesc * len(parts))()
        for d, (key, dst) in zip(descs, parts):
            d.W, d.img, d.ldw, d.N, d.K = key[0], dst, key[1], key[2], key[3]
        return torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device), len(parts)

    def lookup(self, W, stack=None):
        """-> image pointer (0: none; the launch then takes the fp32 path)."""
        key = (W.data_ptr(), W.stride(0), W.shape[0], W.shape[1])
        if stack is not None:
            assert W.shape == stack.shape and W.shape[0] == 128
            key = ("stack", key, (stack.data_ptr(), stack.stride(0), stack.shape[0], stack.shape[1]))
        bf = L.load().gfv_f16split_enabled() == 3
        if bf != self._bf:
            # images are built in the product form they are used in (bf16 high parts / fp16 hi + lo, include/gfv.h): the form
            # changed since these were made - rebuild them all now (outside a capture; inside one they simply are not valid)
            self._bf = bf
            self.invalidate()
            if self.images and not torch.cuda.is_current_stream_capturing():
                self.build()
        img = self.images.get(key)
        if img is not None:
            return img.data_ptr() if key in self.valid else 0
        blocks = key[1:] if key[0] == "stack" else (key,)
        if not all(self._is_static(k[0]) for k in blocks) or torch.cuda.is_current_stream_capturing():
            return 0   # (allocations and uploads stay out of a hipGraph capture)
        lib = L.load()
        nbytes = sum(lib.gfv_weight_image_bytes(k[2], k[3]) for k in blocks)
        img = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)
        desc, nd = self._upload([(key, img)], self.device)
        L.check(lib.gfv_weight_images(desc.data_ptr(), nd, img.numel() // 32, self.wmax.data_ptr(), L.stream_ptr()),
                "gfv_weight_images")
        self.images[key] = img
        self.valid.add(key)
        return img.data_ptr()

    def build(self):
        """Refresh every known image from the current weight values: one launch."""
        if not self.images:
            return
        if len(self._desc_keys) != len(self.images) and not torch.cuda.is_current_stream_capturing():
            self._desc_keys = tuple(self.images)
            self._retired.append(self._desc)   # a captured graph may still read the old table
            self._desc, self._ndesc = self._upload(list(self.images.items()), self.device)
            self._max_frags = max(img.numel() for img in self.images.values()) // 32
        if self._desc is None:
            return
        L.check(L.load().gfv_weight_images(self._desc.data_ptr(), self._ndesc, self._max_frags,
                                           self.wmax.data_ptr(), L.stream_ptr()), "gfv_weight_images")
        self._bf = L.load().gfv_f16split_enabled() == 3
        self.valid = set(self._desc_keys)

    def invalidate(self):
        self.valid = set()


_WI = None   # the WeightImages the chain launches currently consult (set by the engine around forward / backward)


def stack_ready(Wa, Wb, rows=False):
    """True when a launch may use the virtual layer stacked from the blocks Wa, Wb (split-fp16 form on, images available)."""
    return (_WI is not None and L.load().gfv_f16split_enabled() and Wa.shape == Wb.shape and Wa.shape[0] == 128
            and _WI.lookup(Wa, Wb) != 0)


def set_weight_images(wi):
    global _WI
    prev, _WI = _WI, wi
    return prev


def rowtile_chain(M, segs, layers, outs, *, in_add=None, in_op=L.IN_NONE, in_gamma=None, in_beta=None, in_aux=None,
                  gadd=None, gadd_s=None, gadd_r=None, in_save=None, ln_partial=None, fin_op=L.FIN_PLAIN,
                  fin_gamma=None, fin_beta=None, fin_aux=None, fin_presave=None, res=None, out_nores=None,
                  padd=None, padd_s=None, padd_r=None, wimg=None, gscale=None, family=0, fin_stats=None, in_stats=None,
                  dw_partial=None, query_fused=False, rc=None):
    """Launch the fused row-tile GEMM chain.  outs / res: list (per 128-wide chunk of the last layer) of
    (tensor, ld) or tensors; see include/gfv.h for the semantics of every field.  gscale: [3, ld] buffer for the
    per-16-row scales of the gradient rows the launch leaves behind; returns True when the launch wrote it (split-fp16
    form), so the weight-gradient launch may take its slots instead of a pass over the rows.  family: 0 = the library picks
    the kernel family, lib.CHAIN_ROW_OWNER / lib.CHAIN_COLUMN_OWNER pin it (include/gfv.h, gfv_rowtile_args_t.flags).
    fin_stats / in_stats: [M, 2] LayerNorm row statistics out of a forward / into a backward launch; dw_partial:
    [gfv_rowtile_dw_partials(), DW_FUSED_FLOATS] workspace of a backward chain with fused weight gradients; query_fused:
    do not launch, return whether the library would run this launch with fused weight gradients (gfv_rowtile_fuses_dw).
    rc = (W2, b2, W3, b3): the FORWARD's second and third Linear of the MLP whose backward this launch is - it then recomputes
    z2 and the LayerNorm input from z1 instead of reading them (include/gfv.h, rc_Wh; needs dw_partial and weight images)."""
    lib = L.load()
    wi = wimg if wimg is not None else _WI
    if layers[-1].stack is not None:
        # two [128, K] blocks applied to the same input: one launch over a virtual 256-row layer when its stacked image
        # exists, otherwise one launch per block (outs[0], outs[1])
        ly = layers[-1]
        h = wi.lookup(ly.W, ly.stack) if (wi is not None and len(layers) == 1 and lib.gfv_f16split_enabled()) else 0
        if not h:
            assert len(layers) == 1 and len(outs) == 2 and res is None and out_nores is None
            kw = dict(in_add=in_add, in_op=in_op, in_gamma=in_gamma, in_beta=in_beta, wimg=wimg)
            rowtile_chain(M, segs, [LayerSpec(ly.W, ly.bias, ly.op)], [outs[0]], in_save=in_save, **kw)
            rowtile_chain(M, segs, [LayerSpec(ly.stack, ly.bias2, ly.op)], [outs[1]], **kw)
            return
    a = L.RowtileArgs()
    a.M = M
    a.nseg = len(segs)
    for i, s in enumerate(segs):
        s.fill(a.seg[i])
    a.in_add = _p(in_add)
    a.in_op = in_op
    a.nlayers = len(layers)
    a.in_gamma, a.in_beta, a.in_aux = _p(in_gamma), _p(in_beta), _p(in_aux)
    a.gadd, a.gadd_s, a.gadd_r = _p(gadd), _p(gadd_s), _p(gadd_r)
    a.in_save, a.ln_partial = _p(in_save), _p(ln_partial)
    for i, ly in enumerate(layers):
        cl = a.layer[i]
        stacked = ly.stack is not None or ly.stack_cols is not None
        cl.W, cl.bias, cl.bias2 = (None if stacked else _p(ly.W)), _p(ly.bias), _p(ly.bias2)   # stacked: image only
        cl.N = ly.W.shape[0] * (2 if ly.stack is not None else 1)
        cl.K = ly.W.shape[1] * (2 if ly.stack_cols is not None else 1)
        assert ly.W.stride(1) == 1
        cl.ldw = 0 if (ly.stack is not None or ly.stack_cols is not None) else ly.W.stride(0)  # column block: parent's stride
        cl.op = ly.op
        cl.save, cl.aux = _p(ly.save), _p(ly.aux)
    if wi is not None:
        hs = [wi.lookup(ly.W, ly.stack if ly.stack is not None else ly.stack_cols) for ly in layers]
        if all(hs):
            for i, h in enumerate(hs):
                a.layer[i].Wh = h
            a.wmax = wi.wmax.data_ptr()
    a.fin_op = fin_op
    a.fin_gamma, a.fin_beta, a.fin_aux, a.fin_presave = _p(fin_gamma), _p(fin_beta), _p(fin_aux), _p(fin_presave)
    for i, o in enumerate(outs):
        t, ld = o if isinstance(o, tuple) else (o, o.stride(0))
        a.out[i] = t.data_ptr() if torch.is_tensor(t) else t
        a.out_ld[i] = ld
    if res is not None:
        for i, o in enumerate(res):
            if o is None:
                continue
            t, ld = o if isinstance(o, tuple) else (o, o.stride(0))
            a.res[i] = t.data_ptr() if torch.is_tensor(t) else t
            a.res_ld[i] = ld
    a.out_nores = _p(out_nores)
    a.flags = family
    a.fin_stats, a.in_stats = _p(fin_stats), _p(in_stats)
    if dw_partial is not None:
        a.dw_partial, a.dw_partial_stride = dw_partial.data_ptr(), dw_partial.stride(0)
    if rc is not None:
        W2, b2, W3, b3 = rc
        h2, h3 = (wi.lookup(W2), wi.lookup(W3)) if wi is not None else (0, 0)
        if not (h2 and h3):
            if query_fused:
                return False
            raise RuntimeError("recompute form: no split-fp16 image of the forward's second / third Linear")
        a.rc_Wh[0], a.rc_Wh[1] = h2, h3
        a.rc_bias[0], a.rc_bias[1] = _p(b2), _p(b3)
    if padd is not None:
        a.padd, a.padd_s, a.padd_r, a.padd_ld = _p(padd), _p(padd_s), _p(padd_r), padd.stride(0)
    if gscale is not None:
        assert gscale.dim() == 2 and gscale.shape[0] >= 3 and gscale.shape[1] >= gscale_ld(M)
        a.gscale, a.gscale_ld = gscale.data_ptr(), gscale.stride(0)
    if query_fused:
        return bool(lib.gfv_rowtile_fuses_dw(C.byref(a)))
    rc = lib.gfv_rowtile_chain(C.byref(a), L.stream_ptr())
    L.check(rc, "gfv_rowtile_chain")
    return gscale is not None and lib.gfv_rowtile_last_path() >= 5


def trans_mlp_fwd(x, res, Wout, bout, gamma, beta, Wpre, bpre, Wpost, bpost, fx1, z, out):
    """The row-local chain of a Transolver block's forward in one launch (include/gfv.h gfv_trans_mlp_fwd).  False: not available
    (fp32-MFMA form, or no split-fp16 image of one of the three weights) - the caller issues the three single-layer launches."""
    lib = L.load()
    if _WI is None or not lib.gfv_f16split_enabled():
        return False
    hs = (_WI.lookup(Wout), _WI.lookup(Wpre), _WI.lookup(Wpost))
    if not all(hs):
        return False
    a = L.TransMlp()
    a.x, a.res, a.img_out, a.img_pre, a.img_post = x.data_ptr(), res.data_ptr(), hs[0], hs[1], hs[2]
    a.b_out, a.b_pre, a.b_post = _p(bout), _p(bpre), _p(bpost)
    a.gamma, a.beta, a.wmax = gamma.data_ptr(), beta.data_ptr(), _WI.wmax.data_ptr()
    a.fx1, a.z, a.out, a.M = fx1.data_ptr(), z.data_ptr(), out.data_ptr(), x.shape[0]
    L.check(lib.gfv_trans_mlp_fwd(C.byref(a), L.stream_ptr()), "gfv_trans_mlp_fwd")
    return True


def trans_mlp_bwd(g, g_add, g_sum, z, fx1, Wpost_t, Wpre_t, Wout_t, gamma, g_z, g_fx1, g_out_x, ln_partial, gscale):
    """... and of its backward (gfv_trans_mlp_bwd); the W*_t are the TRANSPOSED weights ([256,128], [128,256], [128,128])."""
    lib = L.load()
    if _WI is None or not lib.gfv_f16split_enabled():
        return False
    hs = (_WI.lookup(Wpost_t), _WI.lookup(Wpre_t), _WI.lookup(Wout_t))
    if not all(hs):
        return False
    a = L.TransMlpBwd()
    a.g, a.g_add, a.g_sum, a.z, a.fx1 = g.data_ptr(), _p(g_add), _p(g_sum), z.data_ptr(), fx1.data_ptr()
    a.img_post_t, a.img_pre_t, a.img_out_t = hs
    a.gamma, a.wmax = gamma.data_ptr(), _WI.wmax.data_ptr()
    a.g_z, a.g_fx1, a.g_out_x = g_z.data_ptr(), g_fx1.data_ptr(), g_out_x.data_ptr()
    a.ln_partial, a.gscale, a.M = _p(ln_partial), _p(gscale), g.shape[0]
    L.check(lib.gfv_trans_mlp_bwd(C.byref(a), L.stream_ptr()), "gfv_trans_mlp_bwd")
    return True


def linear_dw(G, n_out, segs, M, *, ldg=None, in_add=None, a_op=0, a_gamma=None, a_beta=None, dW=None, db=None,
              want_db=True, accumulate=False, workspace=None, g_offset=0, gscale=None, col_scale=False):
    """dW[n,k] = sum_m G[m,n] A[m,k]; db[n] = sum_m G[m,n].  Returns (dW [n_out,K], db [n_out] or None).
    col_scale (a_op 0 only): per-column power-of-two scales of the activations in the split-fp16 form (include/gfv.h)."""
    if col_scale:
        assert a_op == 0
        a_op = L.DW_COLSCALE
    lib = L.load()
    K = sum(s.width for s in segs)
    dev = G.device
    if dW is None:
        dW = torch.empty((n_out, K), dtype=torch.float32, device=dev)
    if db is None and want_db:
        db = torch.empty((n_out,), dtype=torch.float32, device=dev)
    need = lib.gfv_linear_dw_workspace_floats(M, n_out, K)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty((max(need, 1),), dtype=torch.float32, device=dev)
    cs = (L.Seg * 3)()
    for i, s in enumerate(segs):
        s.fill(cs[i])
    ldg = G.stride(0) if ldg is None else ldg
    rc = lib.gfv_linear_dw_gs(G.data_ptr() + 4 * g_offset, ldg, n_out, cs, len(segs), _p(in_add), a_op, _p(a_gamma),
                              _p(a_beta), M, _p(dW), _p(gscale), _p(db), _p(workspace), 1 if accumulate else 0,
                              L.stream_ptr())
    L.check(rc, "gfv_linear_dw_gs")
    return dW, db
