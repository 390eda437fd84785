"""`torch_geometric` names used by the reference's hot path: nn.global_add_pool, data.Data (see gfv/scatter.py, gfv/graph.py)."""
from . import data, nn  # noqa: F401
