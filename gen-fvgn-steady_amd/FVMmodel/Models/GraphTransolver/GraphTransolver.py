"""Transolver block with the reference's parameter tree (FVMmodel/Models/GraphTransolver/GraphTransolver.py:25-169).
Fixed geometry of the reference's use: 8 heads x 16 dims, 32 slices, mlp_ratio 2 (TransFVGN_v2.py:28-35)."""
import torch
from FVMmodel.padding import require_native
import torch.nn as nn

from gfv import functions as GF
from gfv.plan import build_batch_plan


class Graph_Physics_Attention_1D(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, slice_num=64):
        super().__init__()
        inner_dim = dim_head * heads
        if heads != 8 or dim_head * heads != dim or slice_num != 32 or dropout != 0 or dim > 128 or dim % 16:
            raise NotImplementedError("HIP slice-attention kernels: 8 heads, 32 slices, dim a multiple of 16 up to 128 "
                                      "(below 128 the model runs zero-padded through NNmodel: FVMmodel/padding.py)")
        self.dim_head, self.heads, self.scale = dim_head, heads, dim_head ** -0.5
        self.temperature = nn.Parameter(torch.ones([1, heads, 1, 1]) * 0.5)      # unused by graph_forward (:35)
        self.graph_temperature = nn.Parameter(torch.ones([1, heads, 1]) * 0.5)
        self.in_project_x = nn.Linear(dim, inner_dim)
        self.in_project_fx = nn.Linear(dim, inner_dim)
        self.in_project_slice = nn.Linear(dim_head, slice_num)
        self.to_q = nn.Linear(dim_head, dim_head, bias=False)
        self.to_k = nn.Linear(dim_head, dim_head, bias=False)
        self.to_v = nn.Linear(dim_head, dim_head, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))


class MLP(nn.Module):
    def __init__(self, n_input, hidden_size, n_output, n_layers=1, act="gelu", res=True):
        super().__init__()
        if n_layers != 0 or act != "gelu":
            raise NotImplementedError("only the n_layers=0 GELU MLP of Transolver_block (GraphTransolver.py:154-161)")
        self.linear_pre = nn.Sequential(nn.Linear(n_input, hidden_size), nn.GELU())
        self.linear_post = nn.Linear(hidden_size, n_output)
        self.linears = nn.ModuleList([])


class Transolver_block(nn.Module):
    def __init__(self, num_heads, hidden_dim, dropout, act="gelu", mlp_ratio=4, slice_num=32):
        super().__init__()
        if mlp_ratio != 2:
            raise NotImplementedError("mlp_ratio=2 (TransFVGN_v2.py:33)")
        self.ln_1 = nn.LayerNorm(hidden_dim)  # unused with in_layernorm=False (GraphTransolver.py:166-167)
        self.Attn = Graph_Physics_Attention_1D(hidden_dim, heads=num_heads, dim_head=hidden_dim // num_heads,
                                               dropout=dropout, slice_num=slice_num)
        self.ln_2 = nn.LayerNorm(hidden_dim)
        self.mlp = MLP(hidden_dim, hidden_dim * mlp_ratio, hidden_dim, n_layers=0, res=False, act=act)

    def forward(self, fx, batch, in_layernorm=False):
        if in_layernorm:
            raise NotImplementedError("in_layernorm=True is never used by TransFVGN (TransFVGN_v2.py:46-49)")
        require_native(self.ln_2.normalized_shape[0])
        names, tensors = [], []
        for n, p in self.named_parameters():
            names.append(f"tb.{n}")
            tensors.append(p)
        plan = build_batch_plan(batch)
        return GF.TransolverFn.apply(GF.Engine(), plan, names, fx, *tensors)
