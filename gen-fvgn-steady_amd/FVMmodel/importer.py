"""`NNmodel` - the drop-in model wrapper of the reference (FVMmodel/importer.py:10-313), MI355X-native.

Same constructor (`params` namespace), same `forward(graph_node, graph_node_x, graph_edge, graph_cell, graph_Index,
is_training=True)` 6-tuple, same in-place side effects on `graph_node` (normalised `x`, `edge_attr`, the two norm
flags), same `state_dict` keys / shapes and checkpoint format.  The arithmetic is a fixed sequence of hand-written
HIP kernels (libgfv) wrapped in ONE autograd node; there is no PyTorch/CPU fallback - tensors must be on the GPU.

Deviations (DESIGN.md "Boundary"): `uvp_node` / `uvp_cell` are returned detached (every reference driver detaches
them: pre_train_Adam.py:193-197, solve_with_grad_GPU.py:186-200); `is_training=False` raises, as it does in the
reference (importer.py:243-245 passes a keyword `update_x_attr` does not accept).
"""
import torch
import torch.nn as nn

from FVMmodel.padding import check_hidden, pad_parameters
from gfv import functions as GF
from gfv import lib as L
from gfv.plan import get_plan
from utils.normalization import Normalizer
from utils.utilities import NodeType  # noqa: F401  (re-exported like the reference)


class NNmodel(nn.Module):
    def __init__(self, params) -> None:
        super().__init__()
        self.params = params
        if params.net in ("TransFVGN_v2", "TransFVGN"):
            from FVMmodel.Models.TransFVGN.TransFVGN_v2 import Simulator
        elif params.net == "TransFVGN_v1":
            from FVMmodel.Models.TransFVGN.TransFVGN_v1 import Simulator
        elif params.net == "FVGN":
            raise ImportError("net='FVGN' does not import in the reference either (GenFVGN.py:6); SURVEY.md row #6")
        else:
            raise NotImplementedError(f"net={params.net}: SURVEY.md row f4 (next)")
        if params.node_input_size != 12 or params.node_phi_size != 3 or params.node_output_size != 3:
            raise NotImplementedError("HIP kernels are specialised for 12 node inputs, 3 outputs")
        check_hidden(int(params.hidden_size))   # 128, or a multiple of 16 below it (zero-padded: FVMmodel/padding.py)
        self.hidden_size = int(params.hidden_size)
        self.simulator = Simulator(
            message_passing_num=params.message_passing_num, node_input_size=params.node_input_size,
            edge_input_size=params.node_input_size + 3, node_output_size=params.node_output_size, drop_out=False,
            hidden_size=params.hidden_size, params=params)
        self.node_norm = Normalizer(size=params.node_input_size - params.node_phi_size,
                                    max_accumulations=params.dataset_size)
        self.node_phi_size = params.node_phi_size
        self.initialize_weights()
        self._engine = None
        self._names = None
        self._replay = GF.ReplayCache()   # recorded forward / backward launch lists per (batch, parameter storage): gfv/functions.py

    # importer.py:42-52
    def initialize_weights(self):
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def engine(self):
        if self._engine is None:
            p = self.params
            self._engine = GF.Engine(message_passing_num=p.message_passing_num, integrator=p.integrator,
                                     ncn_smooth=p.ncn_smooth,
                                     net="TransFVGN_v1" if p.net == "TransFVGN_v1" else "TransFVGN_v2",
                                     conserved_form=bool(getattr(p, "conserved_form", True)),
                                     order=getattr(p, "order", "2nd"), hidden=self.hidden_size)
        return self._engine

    def param_names_tensors(self):
        """(names, Parameter objects) in `named_parameters()` order; walked once and kept (159 entries: the walk costs ~0.1 ms per
        call) until the module tree may have changed - `_apply` (.to / .cuda / .float), `load_state_dict`, `register_parameter`."""
        if self._names is None:
            names, tensors = [], []
            for n, p in self.named_parameters():
                names.append(n)
                tensors.append(p)
            self._names = (names, tensors)
        return self._names

    def _apply(self, fn, *args, **kwargs):
        self._names = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._names = None
        return super().load_state_dict(*args, **kwargs)

    def register_parameter(self, name, param):
        self.__dict__["_names"] = None
        return super().register_parameter(name, param)

    def forward(self, graph_node, graph_node_x, graph_edge, graph_cell, graph_Index, is_training=True):
        if not is_training:
            raise TypeError("update_x_attr() got an unexpected keyword argument 'graph'")  # importer.py:243-245
        if not graph_node.norm_uvp:
            raise ValueError(" src/FVMmodel/importer.py The graph node features have already been normalized, "
                             "please check the graph.norm_uvp")                            # importer.py:123-124
        x = graph_node.x
        GF.require_gpu(x)
        # what the kernels of EARLIER calls raised in the device status word (a hidden activation outside the fixed-scale fp16 window,
        # a weight-gradient operand beyond fp16): read from the pinned mirror the backward publishes into - no synchronisation
        L.raise_on_status("NNmodel.forward")
        if not (x.is_contiguous() and x.dtype == torch.float32):
            x = x.contiguous().float()
            graph_node.x = x
        plan = get_plan((graph_node, graph_node_x, graph_edge, graph_cell, graph_Index))
        norm_global = bool(graph_node.norm_global)
        accumulate = norm_global and self.node_norm.should_accumulate()
        names, tensors = self.param_names_tensors()
        tensors = pad_parameters(names, tensors, self.hidden_size)   # (hidden_size 128: as they are)
        losses, uvp_node, uvp_cell, ea15 = GF.ModelFn.apply(
            self.engine(), plan, names, self.node_norm.buffers_dict(), x,
            dict(norm_global=norm_global, accumulate=accumulate), self._replay if self.hidden_size == 128 else None, *tensors)
        if accumulate:
            self.node_norm.note_accumulated()
        graph_node.norm_uvp = False
        if norm_global:
            graph_node.norm_global = False
        graph_node.edge_attr = ea15
        return (losses[:, 0:1], losses[:, 1:2], losses[:, 2:3], losses[:, 3:4], uvp_node, uvp_cell)

    # importer.py:259-313
    def load_checkpoint(self, optimizer=None, scheduler=None, ckpdir=None, device=None):
        if ckpdir is None:
            ckpdir = self.model_dir
        dicts = torch.load(ckpdir, map_location=device)
        self.load_state_dict(dicts["model"])
        if optimizer is not None and "optimizer0" in dicts:
            for i, o in enumerate(optimizer if isinstance(optimizer, list) else [optimizer]):
                if f"optimizer{i}" in dicts:
                    o.load_state_dict(dicts[f"optimizer{i}"])
        if scheduler is not None and "scheduler0" in dicts:
            for i, s in enumerate(scheduler if isinstance(scheduler, list) else [scheduler]):
                if f"scheduler{i}" in dicts:
                    s.load_state_dict(dicts[f"scheduler{i}"])
        print(f"Simulator model and optimizer/scheduler loaded checkpoint {ckpdir}")

    def save_checkpoint(self, path=None, optimizer=None, scheduler=None):
        if path is None:
            path = self.model_dir
        to_save = {"model": self.state_dict()}
        if optimizer is not None:
            for i, o in enumerate(optimizer if isinstance(optimizer, list) else [optimizer]):
                to_save[f"optimizer{i}"] = o.state_dict()
        if scheduler is not None:
            for i, s in enumerate(scheduler if isinstance(scheduler, list) else [scheduler]):
                to_save[f"scheduler{i}"] = s.state_dict()
        torch.save(to_save, path)
        print(f"Simulator model saved at {path}")
