"""TransFVGN_v2 simulator (FVMmodel/Models/TransFVGN/TransFVGN_v2.py:11-105): Encoder, 2 x AttnProcessor
(message_passing_num GnBlocks + Transolver block), Decoder.  `forward` is one autograd node over the HIP engine
(the reference fuses with @torch.compile, :89)."""
import torch
from FVMmodel.padding import require_native
import torch.nn as nn

from gfv import functions as GF
from gfv.plan import build_gnn_plan, build_batch_plan
from FVMmodel.Models.FVGN.EPD import Encoder, Decoder, GnBlock
from FVMmodel.Models.GraphTransolver.GraphTransolver import Transolver_block


class AttnProcessor(nn.Module):
    def __init__(self, message_passing_num=0, hidden_size=128, drop_out=False):
        super().__init__()
        if message_passing_num < 1:
            raise ValueError("message_passing_num must be greater than 0")
        self.GN_block_list = nn.ModuleList([GnBlock(hidden_size=hidden_size, drop_out=drop_out)
                                            for _ in range(message_passing_num)])
        self.TransBlock = Transolver_block(num_heads=8, hidden_dim=hidden_size, dropout=0, act="gelu", mlp_ratio=2,
                                           slice_num=32)

    def forward(self, latent_graph_node, graph_edge):
        node_embedding = latent_graph_node.x
        g = self.GN_block_list[0](latent_graph_node)
        for model in self.GN_block_list[1:]:
            g = model(g)
        g.x = self.TransBlock(g.x + node_embedding, g.batch)
        return g


class Simulator(nn.Module):
    def __init__(self, message_passing_num, edge_input_size, node_input_size, node_output_size, drop_out=False,
                 hidden_size=128, params=None):
        super().__init__()
        self.message_passing_num = message_passing_num
        self.encoder = Encoder(node_input_size=node_input_size, edge_input_size=edge_input_size, hidden_size=hidden_size)
        self.processpr_list = nn.ModuleList([AttnProcessor(message_passing_num=message_passing_num,
                                                           hidden_size=hidden_size, drop_out=False) for _ in range(2)])
        self.decoder = Decoder(hidden_sze=hidden_size, node_output_size=node_output_size)

    def forward(self, graph_node=None, graph_edge=None, graph_cell=None):
        require_native(self.decoder.node_decode_module[0].out_features)
        names, tensors = [], []
        for n, p in self.named_parameters():
            names.append(f"simulator.{n}")
            tensors.append(p)
        plan = build_gnn_plan(graph_node)
        build_batch_plan(graph_node.batch, plan)
        eng = GF.Engine(message_passing_num=self.message_passing_num)
        ea = graph_node.edge_attr
        if ea.shape[1] == 15:  # the kernels read a 16-float padded row
            ea = torch.nn.functional.pad(ea, (0, 1))
        return GF.SimulatorFn.apply(eng, plan, names, graph_node.x, ea, *tensors)
