"""Encode-Process-Decode pieces with the reference's names and signatures (FVMmodel/Models/FVGN/EPD.py:10-270).

The nn.Linear / nn.LayerNorm objects are parameter containers laid out exactly like the reference's
(`build_mlp` = Sequential(Sequential(Linear, GELU, Linear, GELU, Linear), LayerNorm)), so checkpoints are
interchangeable; the arithmetic runs in libgfv's fused row-tile MFMA kernels."""
from FVMmodel.padding import check_hidden, require_native
import torch
import torch.nn as nn

from gfv import functions as GF
from gfv.graph import Data
from gfv.plan import build_gnn_plan
from FVMmodel.Models.FVGN.blocks import EdgeBlock, NodeBlock


def build_mlp(in_size, hidden_size, out_size, drop_out=True, lay_norm=True, dropout_prob=0.2):
    if drop_out:
        raise NotImplementedError("dropout is never enabled on the reference's hot path (EPD.py:101-104,166,173)")
    check_hidden(hidden_size)   # 128, or a multiple of 16 below it (run zero-padded: FVMmodel/padding.py)
    module = nn.Sequential(nn.Linear(in_size, hidden_size), nn.GELU(), nn.Linear(hidden_size, hidden_size), nn.GELU(),
                           nn.Linear(hidden_size, out_size))
    if lay_norm:
        return nn.Sequential(module, nn.LayerNorm(normalized_shape=out_size))
    return module


def build_mlp_from_num_layer(in_size, hidden_size, out_size, drop_out=False, lay_norm=True, dropout_prob=0.2,
                             num_layer=2):
    if drop_out or num_layer != 2:
        raise NotImplementedError("only the configuration used by Decoder (EPD.py:206-213) is built")
    check_hidden(hidden_size)
    layers = [nn.Linear(in_size, hidden_size), nn.GELU(), nn.Linear(hidden_size, hidden_size), nn.GELU(),
              nn.Linear(hidden_size, out_size)]
    if lay_norm:
        layers.append(nn.LayerNorm(normalized_shape=out_size))
    return nn.Sequential(*layers)


def _named(module, prefix):
    names, tensors = [], []
    for n, p in module.named_parameters():
        names.append(f"{prefix}.{n}")
        tensors.append(p)
    return names, tensors


class Encoder(nn.Module):
    def __init__(self, node_input_size=128, edge_input_size=128, hidden_size=128):
        super().__init__()
        self.eb_encoder = build_mlp(edge_input_size, hidden_size, int(hidden_size), drop_out=False)
        self.nb_encoder = build_mlp(node_input_size, hidden_size, int(hidden_size), drop_out=False)
        self.hidden_size = hidden_size

    def forward(self, graph_node, graph_cell=None):
        require_native(self.hidden_size)
        eng = GF.Engine()
        nn_, nt = _named(self.nb_encoder, "mlp")
        en_, et = _named(self.eb_encoder, "mlp")
        x, ea = graph_node.x, graph_node.edge_attr
        node_ = GF.Mlp3Fn.apply(eng, nn_, True, x.shape[1], x, *nt)
        edge_ = GF.Mlp3Fn.apply(eng, en_, True, ea.shape[1], ea, *et)
        return (Data(x=node_, edge_attr=edge_, edge_index=graph_node.edge_index, face=graph_node.face,
                     num_graphs=graph_node.num_graphs, batch=graph_node.batch), node_)


class GnBlock(nn.Module):
    def __init__(self, hidden_size=128, drop_out=False):
        super().__init__()
        eb_input_dim = int(3 * hidden_size)
        nb_input_dim = int(hidden_size + (hidden_size // 2.0))
        self.nb_module = NodeBlock(hidden_size, custom_func=build_mlp(nb_input_dim, hidden_size, int(hidden_size),
                                                                      drop_out=drop_out))
        self.eb_module = EdgeBlock(input_size=hidden_size,
                                   custom_func=build_mlp(eb_input_dim, hidden_size, int(hidden_size), drop_out=drop_out))

    def forward(self, graph_node):
        require_native(self.eb_module.net[0][0].out_features)
        names, tensors = _named(self, "blk")
        plan = build_gnn_plan(graph_node)
        x, e = GF.GnBlockFn.apply(GF.Engine(), plan, names, graph_node.x, graph_node.edge_attr, *tensors)
        return Data(x=x, edge_attr=e, edge_index=graph_node.edge_index, face=graph_node.face,
                    num_graphs=graph_node.num_graphs, batch=graph_node.batch)


class Decoder(nn.Module):
    def __init__(self, hidden_sze=128, node_output_size=3):
        super().__init__()
        self.node_decode_module = build_mlp_from_num_layer(hidden_sze, hidden_sze, node_output_size, drop_out=False,
                                                           lay_norm=False, num_layer=2)

    def forward(self, latent_graph_node=None):
        require_native(self.node_decode_module[0].out_features)
        names, tensors = _named(self.node_decode_module, "mlp")
        return GF.Mlp3Fn.apply(GF.Engine(), names, False, 128, latent_graph_node.x, *tensors)


class EncoderProcesserDecoder(nn.Module):
    def __init__(self, message_passing_num, edge_input_size, node_input_size, node_output_size, drop_out=False,
                 hidden_size=128, params=None):
        super().__init__()
        self.encoder = Encoder(node_input_size=node_input_size, edge_input_size=edge_input_size, hidden_size=hidden_size)
        self.GN_block_list = nn.ModuleList([GnBlock(hidden_size=hidden_size, drop_out=drop_out)
                                            for _ in range(message_passing_num)])
        self.decoder = Decoder(hidden_sze=hidden_size, node_output_size=node_output_size)

    def forward(self, graph_node=None, graph_cell=None):
        latent, _ = self.encoder(graph_node)
        for model in self.GN_block_list:
            latent = model(latent)
        return self.decoder(latent)
