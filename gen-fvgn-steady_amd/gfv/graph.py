"""Minimal graph container + batching of the five graph objects the hot path consumes.

`Data` is an attribute bag with the part of the ``torch_geometric.data.Data`` surface that the reference's model
code touches (EPD.py:144-151, blocks.py:56-63).  `build_batch` assembles ``graph_node, graph_node_x, graph_edge,
graph_cell, graph_Index`` from mesh dicts with the offset / concatenation rules of the reference's
``CustomGraphData.__inc__/__cat_dim__`` (Load_mesh/Graph_loader.py:405-480), the per-object attributes of its five
dataset views (Graph_loader.py:503-784) and ``datapreprocessing`` (Graph_loader.py:131-152).
"""
from __future__ import annotations

import numpy as np
import torch


class Data:
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

    def to(self, device, non_blocking=False):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=non_blocking))
        return self

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def clone(self):
        return Data(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dtype) if dtype is not None else t


def build_batch(meshes, fields=None, device="cpu"):
    """Batch B meshes into the five graph objects.

    fields: list of [N_i,3] float32 dimensional (u,v,p) node states (default: mesh['init_uvp'])."""
    B = len(meshes)
    if fields is None:
        fields = [m["init_uvp"] for m in meshes]
    n_off = e_off = c_off = 0
    node = {k: [] for k in ("x", "edge_index", "face", "pos", "node_type", "y", "batch", "global_idx")}
    nodex = {k: [] for k in ("face_node_x", "support_edge", "A", "B1", "Bx")}
    edge = {k: [] for k in ("face_type", "face_area", "face", "pos", "batch")}
    cell = {k: [] for k in ("edge_index", "cells_face_unv", "cells_area", "pos", "face", "batch")}
    idx = {k: [] for k in ("theta_PDE", "sigma", "uvp_dim", "dt_graph")}
    for b, (m, f) in enumerate(zip(meshes, fields)):
        N, E, C = m["node|pos"].shape[0], m["face|face_node"].shape[1], m["cell|centroid"].shape[0]
        node["x"].append(_t(f, torch.float32))
        node["edge_index"].append(_t(m["face|face_node"]) + n_off)
        node["face"].append(_t(m["cells_node"]) + n_off)
        node["pos"].append(_t(m["node|pos"], torch.float32))
        node["node_type"].append(_t(m["node|node_type"]))
        node["y"].append(_t(m["target|uvp"], torch.float32))
        node["batch"].append(torch.full((N,), b, dtype=torch.int64))
        node["global_idx"].append(torch.arange(n_off, n_off + N))
        nodex["face_node_x"].append(_t(m["face_node_x"]) + n_off)
        nodex["support_edge"].append(_t(m["support_edge"]) + n_off)
        nodex["A"].append(_t(m["A_node_to_node"], torch.float32))
        nodex["B1"].append(_t(m["single_B_node_to_node"], torch.float32))
        nodex["Bx"].append(_t(m["extra_B_node_to_node"], torch.float32))
        edge["face_type"].append(_t(m["face|face_type"]))
        edge["face_area"].append(_t(m["face|face_area"], torch.float32))
        edge["face"].append(_t(m["cells_face"]) + e_off)
        edge["pos"].append(_t(m["face|face_center_pos"], torch.float32))
        edge["batch"].append(torch.full((E,), b, dtype=torch.int64))
        cell["edge_index"].append(_t(m["face|neighbour_cell"]) + c_off)
        cell["cells_face_unv"].append(_t(m["unit_norm_v"], torch.float32))
        cell["cells_area"].append(_t(m["cell|cells_area"], torch.float32).reshape(-1))
        cell["pos"].append(_t(m["cell|centroid"], torch.float32))
        cell["face"].append(_t(m["cells_index"]) + c_off)
        cell["batch"].append(torch.full((C,), b, dtype=torch.int64))
        for k in idx:
            idx[k].append(_t(m[k], torch.float32).reshape(1, -1))
        n_off, e_off, c_off = n_off + N, e_off + E, c_off + C

    theta = torch.cat(idx["theta_PDE"], 0)
    nb = torch.cat(node["batch"], 0)
    graph_node = Data(
        x=torch.cat((torch.cat(node["x"], 0)[:, 0:3], theta[nb]), dim=1),  # datapreprocessing, Graph_loader.py:148-150
        edge_index=torch.cat(node["edge_index"], 1), face=torch.cat(node["face"], 0),
        pos=torch.cat(node["pos"], 0), node_type=torch.cat(node["node_type"], 0), y=torch.cat(node["y"], 0),
        batch=nb, global_idx=torch.cat(node["global_idx"], 0), num_graphs=B,
    )
    graph_node_x = Data(
        face_node_x=torch.cat(nodex["face_node_x"], 1), support_edge=torch.cat(nodex["support_edge"], 1),
        A_node_to_node=torch.cat(nodex["A"], 0), single_B_node_to_node=torch.cat(nodex["B1"], 0),
        extra_B_node_to_node=torch.cat(nodex["Bx"], 0), num_nodes=n_off, num_graphs=B,
    )
    graph_edge = Data(
        face_type=torch.cat(edge["face_type"], 0), face_area=torch.cat(edge["face_area"], 0),
        face=torch.cat(edge["face"], 0), pos=torch.cat(edge["pos"], 0), batch=torch.cat(edge["batch"], 0),
        num_graphs=B,
    )
    graph_cell = Data(
        x=torch.zeros((c_off, 3), dtype=torch.float32), edge_index=torch.cat(cell["edge_index"], 1),
        cells_face_unv=torch.cat(cell["cells_face_unv"], 0), cells_area=torch.cat(cell["cells_area"], 0),
        pos=torch.cat(cell["pos"], 0), face=torch.cat(cell["face"], 0), batch=torch.cat(cell["batch"], 0),
        num_graphs=B,
    )
    graph_Index = Data(
        theta_PDE=theta, sigma=torch.cat(idx["sigma"], 0), uvp_dim=torch.cat(idx["uvp_dim"], 0),
        dt_graph=torch.cat(idx["dt_graph"], 0), num_graphs=B,
    )
    graphs = (graph_node, graph_node_x, graph_edge, graph_cell, graph_Index)
    if str(device) != "cpu":
        graphs = tuple(g.to(device) for g in graphs)
    return graphs


def clone_graphs(graphs):
    return tuple(g.clone() for g in graphs)
