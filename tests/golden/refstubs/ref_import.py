"""Import harness for running the *reference* (``/root/reference/src``) inside the build container.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden.py`` to generate golden vectors and to pin
``oracle/`` against the reference.  It never travels to the GPU box in a usable form (``/root/reference`` does not
exist there) and nothing in the product path imports it.

The reference needs third-party wheels that are not installed here (torch_scatter, torch_geometric, timm, h5py,
pyvista, vtk, ...).  This module fabricates

* *functional* stand-ins for the arithmetic that the hot path really calls (``torch_scatter.scatter*``,
  ``torch_geometric.nn.global_add_pool``, ``torch_geometric.utils.to_torch_coo_tensor``, ``timm trunc_normal_``),
  restated from the published semantics of those libraries with core torch ops, and
* *inert* modules for everything that is only imported, never executed, on the path.

No reference source text is copied here.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import torch

REFERENCE_SRC = "/root/reference/src"


# ----------------------------------------------------------------------------------------------------------------
# functional stand-ins: torch_scatter
# ----------------------------------------------------------------------------------------------------------------
def _broadcast_index(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(src.dim() - index.dim()):
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def _out_size(src, index, dim, out, dim_size):
    if out is not None:
        return None
    size = list(src.size())
    if dim_size is not None:
        size[dim] = int(dim_size)
    elif index.numel() == 0:
        size[dim] = 0
    else:
        size[dim] = int(index.max()) + 1
    return size


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    idx = _broadcast_index(index, src, dim)
    if out is None:
        out = torch.zeros(_out_size(src, index, dim, out, dim_size), dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, idx, src)


scatter_sum = scatter_add


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_add(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_add(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count = _broadcast_index(count, out, dim) if count.dim() == 1 else count
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode="floor")
    return out


def _scatter_minmax(src, index, dim, out, dim_size, reduce):
    idx = _broadcast_index(index, src, dim)
    if out is None:
        out = torch.zeros(_out_size(src, index, dim, out, dim_size), dtype=src.dtype, device=src.device)
    out = out.scatter_reduce(dim, idx, src, reduce="amin" if reduce == "min" else "amax", include_self=False)
    return out, None


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    return _scatter_minmax(src, index, dim, out, dim_size, "min")


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return _scatter_minmax(src, index, dim, out, dim_size, "max")


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    if reduce in ("sum", "add"):
        return scatter_add(src, index, dim, out, dim_size)
    if reduce == "mean":
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == "min":
        return scatter_min(src, index, dim, out, dim_size)[0]
    if reduce == "max":
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)


def _unused(*a, **k):  # imported by name in the reference, never called on the path
    raise NotImplementedError("stub: not on the hot path")


def _make_torch_scatter():
    m = types.ModuleType("torch_scatter")
    m.scatter_add = scatter_add
    m.scatter_sum = scatter_sum
    m.scatter_mean = scatter_mean
    m.scatter_min = scatter_min
    m.scatter_max = scatter_max
    m.scatter = scatter
    m.scatter_softmax = _unused
    m.scatter_mul = _unused
    m.scatter_std = _unused
    return m


# ----------------------------------------------------------------------------------------------------------------
# functional stand-ins: torch_geometric
# ----------------------------------------------------------------------------------------------------------------
class Data:
    """Attribute bag with the small part of the PyG ``Data`` surface the reference touches."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__.keys() if not k.startswith("_")]

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __contains__(self, k):
        return k in self.__dict__

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def clone(self):
        return Data(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


def global_add_pool(x, batch, size=None):
    if batch is None:
        return x.sum(dim=0, keepdim=True)
    if size is None:
        size = int(batch.max()) + 1
    size = int(size)
    out = torch.zeros((size,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    return out.index_add_(0, batch, x)


def global_mean_pool(x, batch, size=None):
    if size is None:
        size = int(batch.max()) + 1
    s = global_add_pool(x, batch, size)
    cnt = torch.bincount(batch, minlength=int(size)).clamp(min=1).to(x.dtype)
    return s / cnt.view(-1, *([1] * (x.dim() - 1)))


def to_torch_coo_tensor(edge_index, edge_attr=None, size=None, is_coalesced=False):
    if size is None:
        n = int(edge_index.max()) + 1
        size = (n, n)
    elif isinstance(size, int):
        size = (size, size)
    if edge_attr is None:
        edge_attr = torch.ones(edge_index.size(1), device=edge_index.device)
    return torch.sparse_coo_tensor(edge_index, edge_attr, size=tuple(size)).coalesce()


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros((n,), dtype=dtype or torch.float32, device=index.device)
    return out.scatter_add_(0, index, torch.ones_like(index, dtype=out.dtype))


class _InertClass:
    def __init__(self, *a, **k):
        pass

    def __init_subclass__(cls, **k):
        pass


def _make_torch_geometric():
    tg = types.ModuleType("torch_geometric")
    tg.__path__ = []
    data = types.ModuleType("torch_geometric.data")
    data.__path__ = []
    data.Data = Data
    data.InMemoryDataset = _InertClass
    data.Dataset = _InertClass
    batch = types.ModuleType("torch_geometric.data.batch")
    batch.Batch = _InertClass
    data.batch = batch
    data.Batch = _InertClass
    nn = types.ModuleType("torch_geometric.nn")
    nn.global_add_pool = global_add_pool
    nn.global_mean_pool = global_mean_pool
    for name in ("knn_graph", "knn", "radius", "radius_graph", "knn_interpolate"):
        setattr(nn, name, _unused)
    nn.GCNConv = _InertClass
    utils = types.ModuleType("torch_geometric.utils")
    utils.to_torch_coo_tensor = to_torch_coo_tensor
    utils.degree = degree
    loader = types.ModuleType("torch_geometric.loader")
    loader.DataLoader = _InertClass
    tg.data, tg.nn, tg.utils, tg.loader = data, nn, utils, loader
    return {
        "torch_geometric": tg,
        "torch_geometric.data": data,
        "torch_geometric.data.batch": batch,
        "torch_geometric.nn": nn,
        "torch_geometric.utils": utils,
        "torch_geometric.loader": loader,
    }


def _make_timm():
    timm = types.ModuleType("timm")
    timm.__path__ = []
    layers = types.ModuleType("timm.layers")
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    models = types.ModuleType("timm.models")
    models.__path__ = []
    mlayers = types.ModuleType("timm.models.layers")
    mlayers.trunc_normal_ = torch.nn.init.trunc_normal_
    models.layers = mlayers
    timm.layers, timm.models = layers, models
    return {"timm": timm, "timm.layers": layers, "timm.models": models, "timm.models.layers": mlayers}


# ----------------------------------------------------------------------------------------------------------------
# inert modules (imported, never executed on the path)
# ----------------------------------------------------------------------------------------------------------------
class _Inert:
    """Object that absorbs any attribute access / call."""

    def __init__(self, name="inert"):
        self._name = name

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Inert(f"{self._name}.{item}")

    def __call__(self, *a, **k):
        return _Inert(self._name + "()")

    def __mro_entries__(self, bases):
        return (_InertClass,)


class _InertModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        return _Inert(f"{self.__name__}.{item}")


_INERT_ROOTS = ("pyvista", "vtk", "h5py", "natsort", "circle_fit", "trimesh", "tensorboard", "statsmodels",
                "torch_sparse", "torch_cluster", "pyg_lib", "vtkmodules")


class _InertFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _INERT_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _InertModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Make ``import FVMmodel...`` (the reference) work in this container.  Idempotent."""
    global _installed
    if _installed:
        return
    if not os.path.isdir(REFERENCE_SRC):
        raise RuntimeError("reference tree not present; golden generation only runs in the build container")
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")  # neutralise @torch.compile on Simulator.forward
    sys.dont_write_bytecode = True
    sys.modules["torch_scatter"] = _make_torch_scatter()
    sys.modules.update(_make_torch_geometric())
    sys.modules.update(_make_timm())
    sys.meta_path.insert(0, _InertFinder())
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    # the reference was developed on a case-insensitive FS: it imports ``Utils.*`` but the directory is ``utils``
    utils = importlib.import_module("utils")
    sys.modules["Utils"] = utils
    for sub in ("utilities", "normalization", "get_param"):
        sys.modules[f"Utils.{sub}"] = importlib.import_module(f"utils.{sub}")
    _installed = True


def reference_modules():
    """Return the handful of reference entry points the golden generator drives."""
    install()
    out = types.SimpleNamespace()
    out.importer = importlib.import_module("FVMmodel.importer")
    out.get_param = importlib.import_module("utils.get_param")
    out.utilities = importlib.import_module("utils.utilities")
    out.FVgrad = importlib.import_module("FVMmodel.FVdiscretization.FVgrad")
    out.FVscheme = importlib.import_module("FVMmodel.FVdiscretization.FVscheme")
    out.EPD = importlib.import_module("FVMmodel.Models.FVGN.EPD")
    out.blocks = importlib.import_module("FVMmodel.Models.FVGN.blocks")
    out.Transolver = importlib.import_module("FVMmodel.Models.GraphTransolver.GraphTransolver")
    return out


def reference_mesh_modules():
    install()
    out = types.SimpleNamespace()
    out.parse_to_h5 = importlib.import_module("Extract_mesh.parse_to_h5")
    # the writers are visualisation only; make them no-ops
    out.parse_to_h5.write_point_cloud_to_vtk = lambda *a, **k: None
    out.Load_mesh = importlib.import_module("Load_mesh.Load_mesh")
    return out
