"""Reference-generated fingerprints of the per-mesh preprocessing: the k-hop reconstruction stencil
`face_node_x` and the WLSQ moment arrays A / B as the REFERENCE's own pipeline (Cosmol_manager.extract_mesh ->
CFDdatasetBase.transform_mesh: parse_to_h5.py:228-254, Load_mesh.py:247-272,421-521) produces them on four of its example
meshes.  tests/test_oracle_golden.py holds gfv.meshgen's host code to them, tests/test_configs_gpu.py the HIP kernels of
gfv.device_prep - directly against the reference, not against each other.  Run: python tests/golden/make_prep_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refstubs"))
sys.path.insert(0, HERE)
import cases  # noqa: E402
import make_real_mesh_golden as MR  # noqa: E402


def main():
    out = {}
    for name in MR.MESHES:
        _R, md, _cap, _mesh, _bc, _th, _params, _init = MR.reference_arrays(name)
        fx = md["face_node_x"].astype(np.int64)
        out[name + ".stencil"] = cases.int_fingerprint(fx)
        out[name + ".stencil_head"] = fx[:, :64].astype(np.int32)
        for k in ("A_node_to_node", "single_B_node_to_node", "extra_B_node_to_node"):
            out[name + "." + k] = cases.fingerprint(np.asarray(md[k], dtype=np.float64))
        print(name, "pairs", fx.shape[1], out[name + ".stencil"])
    np.savez_compressed(os.path.join(HERE, "real_prep_fp.npz"), **out)


if __name__ == "__main__":
    main()
