"""`torch_scatter` names used by the reference's hot path, on libgfv's segmented reduce (see gfv/scatter.py)."""
from gfv.scatter import scatter, scatter_add, scatter_mean, scatter_sum  # noqa: F401

__all__ = ["scatter", "scatter_add", "scatter_mean", "scatter_sum"]
