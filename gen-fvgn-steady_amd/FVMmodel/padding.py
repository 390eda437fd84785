"""`hidden_size` below 128 (the reference's `--hidden_size`, utils/get_param.py:69): the model runs zero-padded to 128 columns.

Every HIP kernel of the path works on 128-column latent rows (8 accumulator tiles of 16 columns per lane, 8 heads x 16 dims in
the slice attention).  A model of hidden size h in {16, 32, ..., 112} keeps parameters of its TRUE shapes (state_dict and
checkpoints are the reference's), and `NNmodel.forward` places them into zero tensors of the 128-column shapes with the
differentiable index assignment below - autograd slices the gradients back.  Padded columns stay exactly zero through Linear
(zero weight rows / columns, zero bias), GELU, residuals and the LayerNorm affine (gamma = beta = 0 there); LayerNorm takes its
statistics over the h real columns and the attention scales by (h / 8) ** -0.5 (`gfv_set_hidden_size`, include/gfv.h).

Layouts (what the kernels' fixed column blocks mean, gfv/engine.py):
  node latent  [N, h]      -> columns 0 .. h-1
  edge latent  [E, h]      -> its two halves (blocks.py:35-51 scatters them to the two end nodes) at 0 .. h/2-1 and 64 .. 64+h/2-1
  EdgeBlock input [x_s | x_r | e] (3h)  -> 128-column blocks at 0 / 128 / 256 (node, node, edge layout)
  NodeBlock input [agg | x] (h/2 + h)   -> agg at 0 .. h/2-1, x at 64 .. 64+h-1
  attention inner dim (8 heads x h/8)   -> head k at 16 k .. 16 k + h/8 - 1
  Transolver MLP hidden (2h)            -> 0 .. 2h-1 of 256
"""
import re

import torch


def check_hidden(h):
    if h == 128:
        return
    if h < 16 or h > 128 or h % 16:
        raise NotImplementedError(f"hidden_size={h}: multiples of 16 up to 128 (8 heads, two edge halves, 128-column kernels)")


def require_native(h):
    """The stand-alone operator modules (Encoder, GnBlock, Decoder, ... called on their own) run at the kernels' width."""
    if h != 128:
        raise NotImplementedError(f"hidden_size={h}: a stand-alone block runs at hidden 128; narrower models go through "
                                  "NNmodel, which pads the parameters (FVMmodel/padding.py)")


_MAP_CACHE = {}


def _maps(h, device):
    key = (h, str(device))
    if key not in _MAP_CACHE:
        _MAP_CACHE[key] = _build_maps(h, device)
    return _MAP_CACHE[key]


def _build_maps(h, device):
    ar = lambda n, o=0: torch.arange(n, device=device) + o
    half = h // 2
    dh = h // 8
    return dict(
        plain=ar(h),
        edge=torch.cat((ar(half), ar(half, 64))),
        head=(torch.arange(8, device=device)[:, None] * 16 + torch.arange(dh, device=device)[None, :]).reshape(-1),
        eb_in=torch.cat((ar(h), ar(h, 128), ar(half, 256), ar(half, 256 + 64))),
        nb_in=torch.cat((ar(half), ar(h, 64))),
        mlp2=ar(2 * h),
        dh=ar(dh),
    )


_RULES = [  # (name pattern, (padded rows, row map), (padded cols, col map)); map None = identity at the true size
    (r"encoder\.eb_encoder\.0\.0\.weight$", (128, "plain"), None),
    (r"encoder\.nb_encoder\.0\.0\.weight$", (128, "plain"), None),
    (r"eb_module\.net\.0\.0\.weight$", (128, "plain"), (384, "eb_in")),
    (r"nb_module\.net\.0\.0\.weight$", (128, "plain"), (192, "nb_in")),
    (r"(eb_encoder|eb_module\.net)\.0\.4\.weight$", (128, "edge"), (128, "plain")),
    (r"(eb_encoder|eb_module\.net)\.0\.4\.bias$", (128, "edge"), None),
    (r"(eb_encoder|eb_module\.net)\.1\.(weight|bias)$", (128, "edge"), None),
    (r"Attn\.in_project_(x|fx)\.weight$", (128, "head"), (128, "plain")),
    (r"Attn\.in_project_(x|fx)\.bias$", (128, "head"), None),
    (r"Attn\.in_project_slice\.weight$", None, (16, "dh")),
    (r"Attn\.in_project_slice\.bias$", None, None),
    (r"Attn\.to_(q|k|v)\.weight$", (16, "dh"), (16, "dh")),
    (r"Attn\.to_out\.0\.weight$", (128, "plain"), (128, "head")),
    (r"Attn\.(temperature|graph_temperature)$", None, None),
    (r"mlp\.linear_pre\.0\.weight$", (256, "mlp2"), (128, "plain")),
    (r"mlp\.linear_pre\.0\.bias$", (256, "mlp2"), None),
    (r"mlp\.linear_post\.weight$", (128, "plain"), (256, "mlp2")),
    (r"decoder\.node_decode_module\.4\.weight$", None, (128, "plain")),
    (r"decoder\.node_decode_module\.4\.bias$", None, None),
]


def pad_parameters(names, tensors, h):
    """The parameters of a hidden-`h` model as the 128-column tensors the kernels take (differentiable)."""
    if h == 128:
        return list(tensors)
    maps = _maps(h, tensors[0].device)
    out = []
    for name, p in zip(names, tensors):
        rule = None
        for pat, rows, cols in _RULES:
            if re.search(pat, name):
                rule = (rows, cols)
                break
        if rule is None:   # everything else: h -> 128 at the front, in every dimension of size h
            rule = ((128, "plain") if p.shape[0] == h else None,
                    ((128, "plain") if p.dim() > 1 and p.shape[1] == h else None))
        rows, cols = rule
        if rows is None and (cols is None or p.dim() == 1):
            out.append(p)
            continue
        if p.dim() == 1:
            z = p.new_zeros(rows[0])
            out.append(z.index_put((maps[rows[1]],), p))
            continue
        R = rows[0] if rows is not None else p.shape[0]
        C = cols[0] if cols is not None else p.shape[1]
        ri = maps[rows[1]] if rows is not None else torch.arange(p.shape[0], device=p.device)
        ci = maps[cols[1]] if cols is not None else torch.arange(p.shape[1], device=p.device)
        assert ri.numel() == p.shape[0] and ci.numel() == p.shape[1], (name, tuple(p.shape), rows, cols)
        z = p.new_zeros((R, C))
        out.append(z.index_put((ri[:, None].expand(-1, ci.numel()), ci[None, :].expand(ri.numel(), -1)), p))
    return out
