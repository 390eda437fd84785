"""ctypes binding of libgfv.so (the C ABI declared in include/gfv.h).

Plumbing only: converts torch CUDA tensors to raw device pointers and launches on torch's current HIP stream.
There is NO fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GFV_LIB", os.path.join(HERE, "libgfv.so"))

c_float_p = C.c_void_p
c_int_p = C.c_void_p

OP_NONE, OP_BIAS_GELU, OP_MUL_DGELU = 0, 1, 2
IN_NONE, IN_GELU, IN_LN, IN_LNBWD = 0, 1, 2, 3
FIN_PLAIN, FIN_LN, FIN_LNBWD = 0, 1, 2
CHAIN_ROW_OWNER, CHAIN_COLUMN_OWNER = 1, 2   # gfv_rowtile_args_t.flags
DW_COLSCALE = 8   # gfv_dw_tile_t.a_op flag: per-column power-of-two scales of raw-input activations


class Seg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("idx", C.c_void_p), ("width", C.c_int32), ("ld", C.c_int32),
                ("csr_rowptr", C.c_void_p), ("csr_scale", C.c_void_p), ("save", C.c_void_p)]


class Layer(C.Structure):
    _fields_ = [("W", C.c_void_p), ("bias", C.c_void_p), ("K", C.c_int32), ("N", C.c_int32), ("op", C.c_int32),
                ("ldw", C.c_int32), ("save", C.c_void_p), ("aux", C.c_void_p), ("Wh", C.c_void_p),
                ("bias2", C.c_void_p)]


class RowtileArgs(C.Structure):
    _fields_ = [
        ("M", C.c_int32), ("nseg", C.c_int32), ("seg", Seg * 3), ("in_add", C.c_void_p), ("in_op", C.c_int32),
        ("nlayers", C.c_int32), ("in_gamma", C.c_void_p), ("in_beta", C.c_void_p), ("in_aux", C.c_void_p),
        ("gadd", C.c_void_p), ("gadd_s", C.c_void_p), ("gadd_r", C.c_void_p), ("in_save", C.c_void_p),
        ("ln_partial", C.c_void_p), ("layer", Layer * 3), ("fin_op", C.c_int32), ("hidden", C.c_int32),
        ("fin_gamma", C.c_void_p), ("fin_beta", C.c_void_p), ("fin_aux", C.c_void_p), ("fin_presave", C.c_void_p),
        ("res", C.c_void_p * 3), ("res_ld", C.c_int32 * 3), ("out_ld", C.c_int32 * 3), ("out", C.c_void_p * 3),
        ("out_nores", C.c_void_p), ("padd", C.c_void_p), ("padd_s", C.c_void_p), ("padd_r", C.c_void_p),
        ("padd_ld", C.c_int32), ("flags", C.c_int32), ("wmax", C.c_void_p),
        ("gscale", C.c_void_p), ("gscale_ld", C.c_int32), ("product_form", C.c_int32),
        ("fin_stats", C.c_void_p), ("in_stats", C.c_void_p), ("dw_partial", C.c_void_p), ("dw_partial_stride", C.c_int64),
        ("dw_in", C.c_void_p), ("dw_in_ld", C.c_int32), ("reserved2_", C.c_int32),
        ("rc_Wh", C.c_void_p * 2), ("rc_bias", C.c_void_p * 2),
    ]


ABI_VERSION = 3   # GFV_ABI_VERSION of include/gfv.h this binding is written against

DW_FUSED_FLOATS = 2 * 128 * 128 + 4 * 128   # floats per workgroup block of a fused weight-gradient launch (include/gfv.h)


class WimgDesc(C.Structure):
    _fields_ = [("W", C.c_void_p), ("img", C.c_void_p), ("ldw", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("reserved", C.c_int32)]


class DwTile(C.Structure):
    _fields_ = [("G", C.c_void_p), ("A", C.c_void_p), ("idx", C.c_void_p), ("in_add", C.c_void_p),
                ("a_gamma", C.c_void_p), ("a_beta", C.c_void_p), ("ldg", C.c_int32), ("n_out", C.c_int32),
                ("width", C.c_int32), ("ld", C.c_int32), ("a_op", C.c_int32), ("ld_out", C.c_int32),
                ("out_off", C.c_int64), ("db_off", C.c_int64), ("gscale", C.c_void_p)]


class ReducePiece(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("out", C.c_void_p), ("chunk_stride", C.c_int64), ("n_chunks", C.c_int32),
                ("rows", C.c_int32), ("cols", C.c_int32), ("ld_in", C.c_int32), ("ld_out", C.c_int32), ("reserved", C.c_int32)]


class TransMlp(C.Structure):
    _fields_ = [("x", C.c_void_p), ("res", C.c_void_p), ("img_out", C.c_void_p), ("img_pre", C.c_void_p), ("img_post", C.c_void_p),
                ("b_out", C.c_void_p), ("b_pre", C.c_void_p), ("b_post", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("wmax", C.c_void_p), ("fx1", C.c_void_p), ("z", C.c_void_p), ("out", C.c_void_p), ("M", C.c_int32),
                ("reserved", C.c_int32)]


class TransMlpBwd(C.Structure):
    _fields_ = [("g", C.c_void_p), ("g_add", C.c_void_p), ("g_sum", C.c_void_p), ("z", C.c_void_p), ("fx1", C.c_void_p),
                ("img_post_t", C.c_void_p), ("img_pre_t", C.c_void_p), ("img_out_t", C.c_void_p), ("gamma", C.c_void_p),
                ("wmax", C.c_void_p), ("g_z", C.c_void_p), ("g_fx1", C.c_void_p), ("g_out_x", C.c_void_p),
                ("ln_partial", C.c_void_p), ("gscale", C.c_void_p), ("M", C.c_int32), ("reserved", C.c_int32)]


class FvmMesh(C.Structure):
    """gfv_fvm_mesh_t (include/gfv.h): the mesh tables of a batch for the fused finite-volume launches."""
    _fields_ = ([("N", C.c_int32), ("E", C.c_int32), ("C", C.c_int32), ("B", C.c_int32), ("terms", C.c_int32), ("mode", C.c_int32),
                 ("smooth", C.c_int32), ("reserved", C.c_int32)]
                + [(n, C.c_void_p) for n in (
                    "node_type", "batch", "ftype", "cbatch", "gcell_ptr", "pos", "fpos", "y", "centroid", "area", "theta", "sigma",
                    "uvp_dim", "dt", "crow", "kcell", "kS", "frow", "fk", "nfrow", "nfcol2", "nrow", "ncell", "An", "rn",
                    "xo_rowptr", "xo_in", "xo_B", "sumB")])


_lib = None

_SIGNATURES = {
    "gfv_abi_version": (C.c_int, []),
    "gfv_struct_size": (C.c_int, [C.c_int32]),
    "gfv_seg_gather_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_seg_gather_sum_nnz": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "gfv_seg_gather_sum_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "gfv_seg_gather_sum_ln": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_int64, C.c_int64, C.c_void_p]),
    "gfv_transpose_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_status_flags": (C.c_int, [C.POINTER(C.c_int32)]),
    "gfv_status_mirror": (C.c_int, [C.POINTER(C.POINTER(C.c_int32))]),
    "gfv_status_publish": (C.c_int, [C.c_void_p]),
    "gfv_profile_enable": (C.c_int, [C.c_int]),
    "gfv_profile_collect": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "gfv_profile_reset": (C.c_int, []),
    "gfv_profile_set_sizes": (C.c_int, [C.c_double, C.c_double]),
    "gfv_gather_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                  C.c_void_p]),
    "gfv_get_limit": (C.c_int, [C.c_int32]),
    "gfv_set_limit": (C.c_int, [C.c_int32, C.c_int32]),
    "gfv_limit_name": (C.c_char_p, [C.c_int32]),
    "gfv_rowtile_tiles": (C.c_int, [C.c_int32]),
    "gfv_rowtile_chain": (C.c_int, [C.POINTER(RowtileArgs), C.c_void_p]),
    "gfv_rowtile_last_path": (C.c_int, []),
    "gfv_rowtile_ln_rows": (C.c_int, [C.c_int32]),
    "gfv_rowtile_last_ln_rows": (C.c_int, []),
    "gfv_rowtile_dw_partials": (C.c_int, []),
    "gfv_rowtile_dw_partials_m": (C.c_int, [C.c_int32]),
    "gfv_rowtile_fuses_dw": (C.c_int, [C.POINTER(RowtileArgs)]),
    "gfv_dw_chunks": (C.c_int, [C.c_int32]),
    "gfv_linear_dw_workspace_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "gfv_linear_dw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Seg), C.c_int32, C.c_void_p, C.c_int32,
                                C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_linear_dw_ex": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Seg), C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_void_p]),
    "gfv_linear_dw_gs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(Seg), C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int32, C.c_void_p]),
    "gfv_dw_slabs": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_dw_multi_workspace_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int64]),
    "gfv_dw_multi": (C.c_int, [C.POINTER(DwTile), C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                               C.c_void_p]),
    "gfv_reduce_partials": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_concat_offsets": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_f16split_enabled": (C.c_int, []),
    "gfv_set_f16split": (C.c_int, [C.c_int32]),
    "gfv_set_f16split_thread": (C.c_int, [C.c_int32]),
    "gfv_hidden_size": (C.c_int, []),
    "gfv_set_hidden_size": (C.c_int, [C.c_int32]),
    "gfv_weight_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gfv_weight_absmax": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gfv_weight_images": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "gfv_weight_images_form": (C.c_int, [C.c_void_p]),
    "gfv_reduce_partials_2d": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gfv_reduce_multi": (C.c_int, [C.POINTER(ReducePiece), C.c_int32, C.c_void_p]),
    "gfv_reduce_partials_seg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gfv_transpose": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_slice_softmax_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_slice_softmax_bwd_blocks": (C.c_int, [C.c_int32]),
    "gfv_slice_softmax_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_slice_token_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gfv_slice_attention_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_slice_attention_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_deslice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_slice_gw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_slice_softmax_token": (C.c_int, [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_slice_post_bwd": (C.c_int, [C.c_void_p] * 14 + [C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_phi_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_phi_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_wlsq_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_wlsq_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_wlsq_fwd_ex": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_wlsq_bwd_ex": (C.c_int, [C.c_void_p] * 10 + [C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_wlsq_fwd_full": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_wlsq_bwd_full": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_face_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_cell_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_cell_fwd_ex": (C.c_int, [C.c_void_p] * 19 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gfv_graph_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_cell_to_node": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_fvm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_fvm_bwd_ex": (C.c_int, [C.c_void_p] * 26 + [C.c_int32] * 4 + [C.c_void_p] * 3),
    "gfv_khop_workspace_ints": (C.c_size_t, [C.c_int32, C.c_int32]),
    "gfv_khop_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "gfv_khop_fill": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_wlsq_moments": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_interp2_fwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_interp2_bwd": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "gfv_graph_norm_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "gfv_graph_norm_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gfv_graph_norm_stats_ws": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_normalizer_blocks": (C.c_int, [C.c_int32]),
    "gfv_normalizer_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_node_prep": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_edge_attr": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_adam_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_prep_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "gfv_prep_stats": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_prep_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "gfv_fvm_fwd_tail": (C.c_int, [C.POINTER(FvmMesh)] + [C.c_void_p] * 11),
    "gfv_fvm_bwd_fused": (C.c_int, [C.POINTER(FvmMesh)] + [C.c_void_p] * 10),
    "gfv_adam_state_init": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_float, C.c_void_p]),
    "gfv_train_loss_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_train_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gfv_plan_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "gfv_plan_destroy": (C.c_int, [C.c_void_p]),
    "gfv_plan_table": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "gfv_plan_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "gfv_trans_mlp_fwd": (C.c_int, [C.POINTER(TransMlp), C.c_void_p]),
    "gfv_trans_mlp_bwd": (C.c_int, [C.POINTER(TransMlpBwd), C.c_void_p]),
    "gfv_trans_mlp_ln_rows": (C.c_int, [C.c_int32]),
    "gfv_record_begin": (C.c_int, []),
    "gfv_record_count": (C.c_int, []),
    "gfv_record_end": (C.c_int64, []),
    "gfv_record_length": (C.c_int, [C.c_int64]),
    "gfv_record_replay": (C.c_int, [C.c_int64, C.c_int32, C.c_int32]),
    "gfv_record_free": (C.c_int, [C.c_int64]),
    "gfv_stream_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gfv_record_delay_side": (C.c_int, [C.c_int64, C.c_void_p, C.c_int32]),
}


class PlanDesc(C.Structure):
    """gfv_plan_desc_t (include/gfv.h)."""
    _fields_ = [("n_nodes", C.c_int64), ("n_faces", C.c_int64), ("n_cells", C.c_int64), ("n_incidences", C.c_int64),
                ("n_stencil_pairs", C.c_int64), ("n_support_pairs", C.c_int64),
                ("edge_index", C.c_void_p), ("cells_node", C.c_void_p), ("cells_face", C.c_void_p),
                ("cells_index", C.c_void_p), ("face_node_x", C.c_void_p), ("support_edge", C.c_void_p)]


# table ids of gfv_plan_table, in the order of the enum in include/gfv.h
PLAN_TABLES = ("ES", "ER", "N_ROWPTR", "N_COL_NODE", "N_COL_EDGE2", "INV_DEG", "S_ROWPTR", "S_COL", "R_ROWPTR", "R_COL",
               "X_ROWPTR", "X_OUT", "X_ORDER", "XO_ROWPTR", "XO_IN", "XO_ORDER", "CROW", "K_ORDER", "KFACE", "KNODE", "KCELL",
               "FROW", "FK", "NROW", "NCELL")


def declared_symbols():
    return list(_SIGNATURES)


def register(name, restype, argtypes):
    _SIGNATURES[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype, fn.argtypes = restype, argtypes


_recording = None   # gfv.cmdlist: while a step is being recorded, launches go through a noting proxy


def load(raw=False):
    """Load libgfv.so or raise.  No CPU / PyTorch fallback exists for the product path."""
    global _lib
    if _lib is not None:
        return _lib if (_recording is None or raw) else _recording
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libgfv.so not found at {LIB_PATH}: build it with `python gen-fvgn-steady_amd/gfv/build.py` "
            "(hipcc --offload-arch=gfx950). The HIP extension is mandatory; there is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # raises AttributeError if the symbol is missing
        fn.restype, fn.argtypes = restype, argtypes
    # the binding and the library must agree on the ABI: version and the layout of every argument struct
    if lib.gfv_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libgfv.so has ABI {lib.gfv_abi_version()}, this binding is written for {ABI_VERSION}: rebuild it "
                           "(python gen-fvgn-steady_amd/gfv/build.py)")
    for which, st in enumerate((Seg, Layer, RowtileArgs, WimgDesc, DwTile, ReducePiece, PlanDesc, TransMlp, TransMlpBwd, FvmMesh)):
        if lib.gfv_struct_size(which) != C.sizeof(st):
            raise RuntimeError(f"libgfv.so: struct {st.__name__} is {lib.gfv_struct_size(which)} bytes in the library, "
                               f"{C.sizeof(st)} in the binding")
    _lib = lib
    return lib if (_recording is None or raw) else _recording


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def stream_wait(waiter, waited):
    """`waiter.wait_stream(waited)` through the library (include/gfv.h gfv_stream_wait): the same event record + stream wait,
    and part of a natively recorded step (gfv/cmdlist.py)."""
    check(load().gfv_stream_wait(C.c_void_p(waiter.cuda_stream), C.c_void_p(waited.cuda_stream)), "gfv_stream_wait")


_limit_ids = None


def limit_id(name):
    """Index of a dispatch limit (csrc/gfv_limits.h) by the name of its environment variable, e.g. "GFV_CBWD_MAX_M"."""
    global _limit_ids
    if _limit_ids is None:
        lib, ids, i = load(raw=True), {}, 0
        while True:
            n = lib.gfv_limit_name(i)
            if not n:
                break
            ids[n.decode()] = i
            i += 1
        _limit_ids = ids
    return _limit_ids[name]


def get_limit(name):
    return load(raw=True).gfv_get_limit(limit_id(name))


class limits:
    """`with L.limits(GFV_CBWD_MAX_M=100000): ...` - move dispatch limits for the duration of a block (tests, A/B tools)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        lib = load(raw=True)
        self.old = {k: lib.gfv_get_limit(limit_id(k)) for k in self.kw}
        for k, v in self.kw.items():
            check(lib.gfv_set_limit(limit_id(k), int(v)), "gfv_set_limit")
        return self

    def __exit__(self, *exc):
        lib = load(raw=True)
        for k, v in self.old.items():
            lib.gfv_set_limit(limit_id(k), v)
        return False


FLAG_NAMES = {1: "GFV_FLAG_DW_RANGE (a weight-gradient operand left the fp16 range)",
              2: "GFV_FLAG_CHAIN_RANGE (a hidden activation left the fixed-scale fp16 window of the column-owner chain kernels)"}
_status_word = None


def status_mirror():
    """The library's pinned, device-mapped copy of the status word (include/gfv.h gfv_status_mirror) as a ctypes int32
    pointer; kernels that end a step publish into it, the host reads it without a synchronisation."""
    global _status_word
    if _status_word is None:
        p = C.POINTER(C.c_int32)()
        check(load(raw=True).gfv_status_mirror(C.byref(p)), "gfv_status_mirror")
        _status_word = p
    return _status_word


def status_publish():
    """Enqueue the publication of the status word on the current stream (callers without the fused Adam at their end)."""
    status_mirror()
    check(load().gfv_status_publish(stream_ptr()), "gfv_status_publish")


def raise_on_status(where):
    """Non-blocking check of what the kernels of EARLIER steps raised: a non-zero mirror is cleared (together with the device
    word: that one read synchronises, on the error path only) and turned into a FloatingPointError naming the flags."""
    w = status_mirror()
    v = int(w[0])
    if not v:
        return
    w[0] = 0
    flags = C.c_int32(0)
    load(raw=True).gfv_status_flags(C.byref(flags))   # clears the device word
    v |= int(flags.value)
    names = [n for bit, n in FLAG_NAMES.items() if v & bit] or [f"flags {v:#x}"]
    raise FloatingPointError(f"libgfv status word raised before {where}: " + "; ".join(names) + ".  The split-fp16 products of that "
                             "step lost fp32 accuracy (or overflowed): run with GFV_F16SPLIT=0 (fp32 MFMA) or rescale the model")


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "libgfv works on device memory only"
    return t.data_ptr()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"libgfv: {what} failed with code {rc}")


def f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def i32c(t):
    assert t.dtype == torch.int32 and t.is_contiguous()
    return t
