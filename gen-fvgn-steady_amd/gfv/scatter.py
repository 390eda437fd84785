"""The scatter primitives the reference's tree imports from torch_scatter / torch_geometric, on the atomics-free CSR
segmented reduce of libgfv (`gfv_seg_gather_sum`):

    scatter_add / scatter_sum / scatter_mean / scatter(src, index, dim=0, out=None, dim_size=None[, reduce])
    global_add_pool(x, batch, size=None)

Call sites in the reference: FVMmodel/Models/FVGN/blocks.py:3-4,35,44,92; FVdiscretization/FVscheme.py:17,21,159,175,185,233,244;
FVgrad.py:2,320; FVorder.py:2,81; FVInterpolation.py:19,193,261; utils/utilities.py:5,33,55; importer.py:4,86,88.

A scatter with index `idx` over rows is `out[r] = sum_{m: idx[m] = r} src[m]` = the segmented reduce over the CSR of `idx`
(stable sort: the reference's summation order inside a segment is kept, so results do not depend on launch order and
there are no floating-point atomics).  The CSR of an index tensor is cached on the identity of its STORAGE window (data
pointer, length, stride, in-place version - a fresh view or slice of the same index hits the cache; the cached entry
holds the tensor, so the pointer cannot be recycled under the key).  Backward of a scatter is a gather (`gout[idx]`),
the same kernel over the identity CSR.
Only what the reference's call sites use is built: reduction over dim 0 (or the first dim of a 1-D index), a 1-D index,
CUDA tensors, reduce in {"sum", "add", "mean"}.  The kernels compute in float32: a float64 / float16 / bfloat16 `src` is
converted on the way in and the RESULT IS CAST BACK to src.dtype (torch_scatter preserves the dtype; the arithmetic
here is fp32 whatever the input - a float64 caller gets fp32 accuracy in a float64 tensor); integer `src` is refused.
An index entry outside [0, dim_size) raises a ValueError that says so.
"""
from __future__ import annotations

import torch

from . import ops

_CSR_CACHE = {}
_CSR_CACHE_MAX = 64


def _csr_of(index, n_rows):
    key = (index.data_ptr(), int(index.numel()), tuple(index.stride()), index.dtype, index._version, int(n_rows),
           str(index.device))
    hit = _CSR_CACHE.get(key)
    if hit is not None:
        return hit[1]
    idx = index.reshape(-1).to(torch.int64)
    if idx.numel():
        lo, hi = int(idx.min()), int(idx.max())       # (one host sync per NEW index tensor; cached afterwards)
        if lo < 0 or hi >= n_rows:
            raise ValueError(f"gfv.scatter: index holds entries in [{lo}, {hi}] but the output has {n_rows} rows "
                             "(dim_size / out.shape[0])")
    order = torch.argsort(idx, stable=True)
    counts = torch.bincount(idx, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=index.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    plan = dict(rowptr=rowptr.to(torch.int32), col=order.to(torch.int32).contiguous(),
                inv_count=(1.0 / counts.clamp(min=1).to(torch.float32)).contiguous(),
                ident=torch.arange(index.numel() + 1, dtype=torch.int32, device=index.device),
                idx32=idx.to(torch.int32).contiguous())
    if len(_CSR_CACHE) >= _CSR_CACHE_MAX:
        _CSR_CACHE.pop(next(iter(_CSR_CACHE)))
    _CSR_CACHE[key] = (index, plan)    # (holding the index keeps its data pointer from being reused under the key)
    return plan


class _ScatterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src2d, plan, n_rows, mean):
        out = ops.seg_gather_sum(src2d, plan["rowptr"], plan["col"], n_rows, scale=plan["inv_count"] if mean else None)
        ctx.plan, ctx.mean, ctx.m = plan, mean, src2d.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        # grad_src[m] = g[idx[m]] (/ count[idx[m]]): a gather = the segmented reduce over one-entry rows
        gs = ops.seg_gather_sum(g.contiguous(), plan["ident"], plan["idx32"], ctx.m,
                                src_scale=plan["inv_count"] if ctx.mean else None)
        return gs, None, None, None


def _scatter(src, index, dim, out, dim_size, mean):
    if not src.is_cuda:
        raise RuntimeError("gfv.scatter: tensors must live on the GPU (HIP kernels only, no CPU fallback)")
    if index.dim() != 1:
        raise NotImplementedError("gfv.scatter: 1-D index (the reference's call sites)")
    nd = src.dim()
    if dim < 0:
        dim += nd
    if dim != 0:
        raise NotImplementedError("gfv.scatter: reduction over dim 0 (the reference's call sites)")
    if src.shape[0] != index.shape[0]:
        raise ValueError("src and index disagree along dim 0")
    if out is not None:
        n_rows = out.shape[0]
    elif dim_size is not None:
        n_rows = int(dim_size)
    else:
        n_rows = int(index.max()) + 1 if index.numel() else 0      # (host sync, as in torch_scatter)
    if not src.is_floating_point():
        raise TypeError(f"gfv.scatter: floating-point src only (got {src.dtype}); the kernels sum in float32")
    tail = tuple(src.shape[1:])
    src2d = src.reshape(src.shape[0], -1).to(torch.float32).contiguous()
    if src2d.shape[1] == 0 or n_rows == 0:
        if n_rows == 0 and index.numel():
            raise ValueError("gfv.scatter: a non-empty index with an output of 0 rows")
        res = torch.zeros((n_rows,) + tail, dtype=torch.float32, device=src.device)
    else:
        res = _ScatterFn.apply(src2d, _csr_of(index, n_rows), n_rows, mean).reshape((n_rows,) + tail)
    res = res.to(src.dtype)
    if out is not None:
        if mean:
            raise NotImplementedError("scatter_mean into `out`")
        out += res
        return out
    return res


def _dim0(src, dim):
    # torch_scatter's default dim=-1 on a 1-D src is dim 0
    return 0 if (src.dim() == 1 and dim in (-1, 0)) else dim


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return _scatter(src, index, _dim0(src, dim), out, dim_size, False)


scatter_sum = scatter_add


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    """Sum, then divide by the count clamped at 1 (torch_scatter semantics: empty rows give 0)."""
    return _scatter(src, index, _dim0(src, dim), out, dim_size, True)


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_add(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    raise NotImplementedError(f"gfv.scatter: reduce={reduce!r} is not used on the hot path (SURVEY.md 8b)")


def global_add_pool(x, batch, size=None):
    """torch_geometric.nn.global_add_pool (FVscheme.py:159,185,244): per-graph sums of the rows of x."""
    return scatter_add(x, batch, dim=0, dim_size=size)
