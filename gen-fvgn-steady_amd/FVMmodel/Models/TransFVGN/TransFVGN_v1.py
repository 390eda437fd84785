"""TransFVGN_v1 simulator (FVMmodel/Models/TransFVGN/TransFVGN_v1.py:10-74): Encoder, message_passing_num GnBlocks, ONE
Transolver block applied to (x + node embedding), Decoder - i.e. one processor of TransFVGN_v2 whose modules hang directly
off the simulator.  `forward` is one autograd node over the HIP engine (the reference fuses with @torch.compile, :53)."""
import torch
from FVMmodel.padding import require_native
import torch.nn as nn

from gfv import functions as GF
from gfv.plan import build_gnn_plan, build_batch_plan
from FVMmodel.Models.FVGN.EPD import Encoder, Decoder, GnBlock
from FVMmodel.Models.GraphTransolver.GraphTransolver import Transolver_block


class Simulator(nn.Module):
    def __init__(self, message_passing_num, edge_input_size, node_input_size, node_output_size, drop_out=False,
                 hidden_size=128, params=None):
        super().__init__()
        self.message_passing_num = message_passing_num
        self.encoder = Encoder(node_input_size=node_input_size, edge_input_size=edge_input_size, hidden_size=hidden_size)
        self.GN_block_list = nn.ModuleList([GnBlock(hidden_size=hidden_size, drop_out=drop_out)
                                            for _ in range(message_passing_num)])
        self.TransBlock = Transolver_block(num_heads=8, hidden_dim=hidden_size, dropout=0, act="gelu", mlp_ratio=2,
                                           slice_num=32)
        self.decoder = Decoder(hidden_sze=hidden_size, node_output_size=node_output_size)

    def forward(self, graph_node=None, graph_edge=None, graph_cell=None):
        require_native(self.decoder.node_decode_module[0].out_features)
        names, tensors = [], []
        for n, p in self.named_parameters():
            names.append(f"simulator.{n}")
            tensors.append(p)
        plan = build_gnn_plan(graph_node)
        build_batch_plan(graph_node.batch, plan)
        eng = GF.Engine(message_passing_num=self.message_passing_num, net="TransFVGN_v1")
        ea = graph_node.edge_attr
        if ea.shape[1] == 15:  # the kernels read a 16-float padded row
            ea = torch.nn.functional.pad(ea, (0, 1))
        return GF.SimulatorFn.apply(eng, plan, names, graph_node.x, ea, *tensors)
