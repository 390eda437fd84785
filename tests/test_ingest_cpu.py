"""COMSOL .mphtxt + BC.json ingest (gfv.ingest) against what the reference's own reader
(Extract_mesh/parse_comsol.py) returns for the same file (tests/golden/make_ingest_golden.py; the same comparison on
the reference's real example meshes is logged in make_ingest_golden.log), then through the rest of the pipeline."""
import json
import os

import numpy as np
import torch

import cases  # noqa: F401  (path setup)
from gfv import ingest, meshgen
from gfv.graph import build_batch
from oracle import fvgn_oracle as O


def test_mphtxt_reader_matches_reference_reader(golden_dir):
    fx = np.load(os.path.join(golden_dir, "ingest_small.npz"))
    raw = ingest.load_comsol_mesh(os.path.join(golden_dir, "ingest_small.mphtxt"),
                                  os.path.join(golden_dir, "ingest_small_BC.json"))
    for k in ("node|pos", "node|node_type", "face|face_node", "cells_node", "cells_index", "cells_face"):
        a, b = np.asarray(fx[k]).reshape(-1), np.asarray(raw[k]).reshape(-1)
        assert a.shape == b.shape and np.array_equal(a.astype(np.float64), b.astype(np.float64)), k
    nt = raw["node|node_type"]
    assert (nt == ingest.INFLOW).sum() > 0 and (nt == ingest.OUTFLOW).sum() > 0 and (nt == ingest.IN_WALL).sum() == 2


def test_mphtxt_write_read_round_trip(tmp_path, golden_dir):
    mf = ingest.read_mphtxt(os.path.join(golden_dir, "ingest_small.mphtxt"))
    p = tmp_path / "again.mphtxt"
    ingest.write_mphtxt(str(p), mf["vertices"], {t: (mf[t]["Elements"], mf[t]["Geometric entity indices"] - 1)
                                                  for t in ("vtx", "edg", "tri", "quad")})
    again = ingest.read_mphtxt(str(p))
    assert np.array_equal(again["vertices"], mf["vertices"])
    for t in ("vtx", "edg", "tri", "quad"):
        assert np.array_equal(again[t]["Elements"], mf[t]["Elements"])
        assert np.array_equal(again[t]["Geometric entity indices"], mf[t]["Geometric entity indices"])


def test_ingested_mesh_runs_through_the_pipeline(golden_dir):
    """file -> raw arrays -> geometry / stencil / moments (gfv.meshgen.finish_mesh) -> batch -> model forward (oracle)."""
    raw = ingest.load_comsol_mesh(os.path.join(golden_dir, "ingest_small.mphtxt"),
                                  os.path.join(golden_dir, "ingest_small_BC.json"), U=0.25)
    mesh = meshgen.finish_mesh(raw)
    assert mesh["cell|centroid"].shape[0] == 225 and abs(float(mesh["uvp_dim"][0, 0]) - 0.25) < 1e-7
    graphs = build_batch([mesh], [meshgen.random_fields(mesh, seed=3)])
    out = O.model_forward(O.init_parameters(cases.WEIGHT_SEED), O.new_normalizer_buffers(), graphs)
    assert all(bool(torch.isfinite(t).all()) for t in out) and float(out[0]) > 0
    bc = json.load(open(os.path.join(golden_dir, "ingest_small_BC.json")))
    assert ingest.expand_bc(bc)["wall"] == [2, 3, 5, 6, 7, 8]
