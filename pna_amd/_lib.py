"""ctypes binding of libpna_amd.so (the C ABI declared in include/pna_amd.h).

PyTorch is used for device memory and streams only: every call below passes raw device pointers
and the current HIP stream through the C ABI.  There is NO CPU or eager-PyTorch fallback -- if the
shared library is missing or a tensor is not on a GPU the call raises.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported first: the library binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PNA_AMD_LIB_PATH: another build of the same ABI -- tools/build_variant.sh, the -DPNA_AMD_EXPERIMENTS build -- for same-box A/B runs)
LIB_PATH = os.environ.get("PNA_AMD_LIB_PATH") or os.path.join(_HERE, "lib", "libpna_amd.so")

PNA_ABI_VERSION = 21
PNA_MAX_AGGR = 8
PNA_MAX_SCALER = 8

AGG_CODES = {"mean": 0, "sum": 1, "max": 2, "min": 3, "std": 4, "var": 5, "var_raw": 6}


class _Args(ctypes.Structure):
    """Base of the *_args mirrors: `struct_size` (the first field of every args struct since ABI 19) is stamped on construction."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = ctypes.sizeof(self)


class PnaTuning(ctypes.Structure):
    _fields_ = [("lanes_per_row", ctypes.c_int32), ("unroll", ctypes.c_int32), ("rows_per_group", ctypes.c_int32),
                ("vec", ctypes.c_int32), ("nt_store", ctypes.c_int32), ("prefetch", ctypes.c_int32),
                ("debug", ctypes.c_int32), ("generic", ctypes.c_int32)]


class PnaSegreduceArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("V", ctypes.c_int32), ("F", ctypes.c_int32),
        ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("x_rows", ctypes.c_int64),
        ("dst_term", ctypes.c_void_p), ("ld_dst", ctypes.c_int64),
        ("edge_term", ctypes.c_void_p), ("ld_edge", ctypes.c_int64),
        ("edge_weight", ctypes.c_void_p),
        ("n_tower", ctypes.c_int32), ("_pad_t", ctypes.c_int32),
        ("tower_stride_in", ctypes.c_int64), ("tower_stride_out", ctypes.c_int64),
        ("n_aggr", ctypes.c_int32), ("aggr", ctypes.c_int32 * PNA_MAX_AGGR),
        ("n_scaler", ctypes.c_int32), ("_pad0", ctypes.c_int32),
        ("row_scale", ctypes.c_void_p * PNA_MAX_SCALER),
        ("out", ctypes.c_void_p), ("ldo", ctypes.c_int64), ("block_stride", ctypes.c_int32), ("_pad1", ctypes.c_int32),
        ("argmax", ctypes.c_void_p), ("argmin", ctypes.c_void_p), ("ld_arg", ctypes.c_int64),
        ("heavy_threshold", ctypes.c_int32), ("seg_len", ctypes.c_int32), ("n_heavy", ctypes.c_int32),
        ("n_seg", ctypes.c_int32),
        ("heavy_rows", ctypes.c_void_p), ("heavy_segptr", ctypes.c_void_p), ("seg_heavy", ctypes.c_void_p),
        ("partials", ctypes.c_void_p),
        ("work_items", ctypes.c_void_p), ("n_work_items", ctypes.c_int32), ("_pad2", ctypes.c_int32),
        ("n_edges", ctypes.c_int64),
        ("tune", PnaTuning),
        ("heavy_out_rows", ctypes.c_void_p), ("out_row_of", ctypes.c_void_p),
        ("edge_type", ctypes.c_void_p), ("n_edge_types", ctypes.c_int32), ("_pad3", ctypes.c_int32),
    ]


class PnaSegreduceBwdArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("V", ctypes.c_int32), ("F", ctypes.c_int32),
        ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
        ("dst_term", ctypes.c_void_p), ("ld_dst", ctypes.c_int64),
        ("edge_term", ctypes.c_void_p), ("ld_edge", ctypes.c_int64),
        ("n_tower", ctypes.c_int32), ("n_aggr", ctypes.c_int32), ("tower_stride_in", ctypes.c_int64),
        ("aggr", ctypes.c_int32 * PNA_MAX_AGGR),
        ("gagg", ctypes.c_void_p), ("ld_g", ctypes.c_int64), ("tower_stride_g", ctypes.c_int64),
        ("mean", ctypes.c_void_p), ("stdv", ctypes.c_void_p), ("var", ctypes.c_void_p),
        ("ld_stat", ctypes.c_int64), ("tower_stride_stat", ctypes.c_int64),
        ("argmax", ctypes.c_void_p), ("argmin", ctypes.c_void_p), ("ld_arg", ctypes.c_int64),
        ("grad_x", ctypes.c_void_p), ("ld_gx", ctypes.c_int64),
        ("grad_dst", ctypes.c_void_p), ("ld_gd", ctypes.c_int64),
        ("grad_edge", ctypes.c_void_p), ("ld_ge", ctypes.c_int64),
        ("heavy_threshold", ctypes.c_int32), ("seg_len", ctypes.c_int32), ("n_heavy", ctypes.c_int32),
        ("n_seg", ctypes.c_int32),
        ("heavy_rows", ctypes.c_void_p), ("heavy_segptr", ctypes.c_void_p), ("seg_heavy", ctypes.c_void_p),
        ("stat_row_of", ctypes.c_void_p), ("stat_node_of", ctypes.c_void_p), ("stat_rows", ctypes.c_int64),
    ]


class PnaPosttransArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("a", ctypes.c_void_p), ("lda", ctypes.c_int64), ("M", ctypes.c_int32), ("K", ctypes.c_int32),
        ("N", ctypes.c_int32), ("n_scaler", ctypes.c_int32),
        ("row_scale", ctypes.c_void_p * PNA_MAX_SCALER),
        ("w_img", ctypes.c_void_p),
        ("h", ctypes.c_void_p), ("ldh", ctypes.c_int64), ("Kh", ctypes.c_int32), ("relu", ctypes.c_int32),
        ("wh_img", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("row_post", ctypes.c_void_p), ("col_scale", ctypes.c_void_p), ("col_shift", ctypes.c_void_p),
        ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64),
        ("y", ctypes.c_void_p), ("ldy", ctypes.c_int64),
        ("pipeline", ctypes.c_int32), ("act_slope", ctypes.c_float),
        ("n_tower", ctypes.c_int32), ("_pad_t", ctypes.c_int32),
        ("tower_stride_a", ctypes.c_int64), ("tower_stride_h", ctypes.c_int64), ("tower_stride_w", ctypes.c_int64),
        ("tower_stride_wh", ctypes.c_int64), ("tower_stride_y", ctypes.c_int64),
        ("row_perm", ctypes.c_void_p), ("tile_image", ctypes.c_void_p), ("image_stride", ctypes.c_int64),
    ]


class PnaFusedSimpleArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
        ("V", ctypes.c_int32), ("F", ctypes.c_int32), ("N", ctypes.c_int32), ("n_scaler", ctypes.c_int32),
        ("row_scale", ctypes.c_void_p * PNA_MAX_SCALER),
        ("w_img", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("col_scale", ctypes.c_void_p),
        ("col_shift", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64),
        ("y", ctypes.c_void_p), ("ldy", ctypes.c_int64), ("relu", ctypes.c_int32), ("heavy_threshold", ctypes.c_int32),
    ]


class PnaSegreduceBwdPullArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("base", ctypes.c_void_p), ("table", ctypes.c_void_p), ("ld_table", ctypes.c_int64), ("col_t", ctypes.c_void_p), ("rank_t", ctypes.c_void_p),
        ("items_t", ctypes.c_void_p), ("n_items_t", ctypes.c_int32), ("run_rowprep", ctypes.c_int32), ("ranks", ctypes.c_void_p), ("ld_rank", ctypes.c_int64),
        ("edge_rows", ctypes.c_void_p), ("ld_edge", ctypes.c_int64), ("pos_t", ctypes.c_void_p), ("items", ctypes.c_void_p),
        ("n_items", ctypes.c_int32), ("_pad_e", ctypes.c_int32),
    ]


class PnaFusedDegreeArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("tile_desc", ctypes.c_void_p), ("tile_ids", ctypes.c_void_p), ("n_records", ctypes.c_int64),
        ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("x_rows", ctypes.c_int64), ("F", ctypes.c_int32), ("N", ctypes.c_int32),
        ("row_perm", ctypes.c_void_p), ("M", ctypes.c_int64), ("n_nodes", ctypes.c_int64),
        ("w_img", ctypes.c_void_p), ("image_stride", ctypes.c_int64), ("bias", ctypes.c_void_p),
        ("col_scale", ctypes.c_void_p), ("col_shift", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64),
        ("y", ctypes.c_void_p), ("ldy", ctypes.c_int64), ("relu", ctypes.c_int32), ("act_slope", ctypes.c_float),
        ("agg_out", ctypes.c_void_p), ("ld_agg", ctypes.c_int64),
        ("x_dst", ctypes.c_void_p), ("ld_xdst", ctypes.c_int64), ("h_self", ctypes.c_void_p), ("ld_h", ctypes.c_int64), ("row_post", ctypes.c_void_p),
        ("spare_workgroups", ctypes.c_int32), ("_pad4", ctypes.c_int32), ("tile_counter", ctypes.c_void_p),
        ("w_img_x3", ctypes.c_void_p), ("image_stride_x3", ctypes.c_int64), ("guard_ws", ctypes.c_void_p), ("guard_ws_bytes", ctypes.c_int64),
        ("arith", ctypes.c_int32), ("_pad5", ctypes.c_int32), ("pre_add", ctypes.c_void_p), ("ld_pre_add", ctypes.c_int64),
        ("y_cols_writable", ctypes.c_int32), ("_pad6", ctypes.c_int32),
    ]


FD_ARITH_GUARDED, FD_ARITH_X3, FD_ARITH_H2 = 0, 1, 2      # include/pna_amd.h PNA_FD_ARITH_*
FD_ARITH_NAMES = {FD_ARITH_GUARDED: "fp16x2_guarded", FD_ARITH_X3: "bf16x3", FD_ARITH_H2: "fp16x2"}


class PnaBnTailArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("y", ctypes.c_void_p), ("ldy", ctypes.c_int64), ("M", ctypes.c_int64), ("N", ctypes.c_int32), ("relu", ctypes.c_int32),
        ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("eps", ctypes.c_float), ("momentum", ctypes.c_float),
        ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p), ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64),
        ("out", ctypes.c_void_p), ("ld_out", ctypes.c_int64), ("save_mean", ctypes.c_void_p), ("save_invstd", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
        ("grad_out", ctypes.c_void_p), ("ld_go", ctypes.c_int64), ("grad_y", ctypes.c_void_p), ("ld_gy", ctypes.c_int64),
        ("grad_gamma", ctypes.c_void_p), ("grad_beta", ctypes.c_void_p),
    ]


class PnaSmallLinearArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("M", ctypes.c_int32), ("K", ctypes.c_int32), ("N", ctypes.c_int32),
        ("act", ctypes.c_int32), ("img", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("act_slope", ctypes.c_float),
        ("_pad", ctypes.c_int32), ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64), ("y", ctypes.c_void_p),
        ("ldy", ctypes.c_int64),
    ]


class PnaPosttransDwArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("gy", ctypes.c_void_p), ("ldg", ctypes.c_int64), ("M", ctypes.c_int64), ("N", ctypes.c_int32), ("n_scaler", ctypes.c_int32),
        ("a", ctypes.c_void_p), ("lda", ctypes.c_int64), ("K", ctypes.c_int32), ("Kh", ctypes.c_int32),
        ("h", ctypes.c_void_p), ("ldh", ctypes.c_int64), ("row_scale", ctypes.c_void_p * PNA_MAX_SCALER),
        ("grad_w", ctypes.c_void_p), ("ldw", ctypes.c_int64), ("grad_b", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64),
    ]


class PnaPosttransDwGroupedArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("gy", ctypes.c_void_p), ("ldg", ctypes.c_int64), ("N", ctypes.c_int32), ("n_scaler", ctypes.c_int32),
        ("a", ctypes.c_void_p), ("lda", ctypes.c_int64), ("K", ctypes.c_int32), ("Kh", ctypes.c_int32),
        ("h", ctypes.c_void_p), ("ldh", ctypes.c_int64),
        ("row_perm", ctypes.c_void_p), ("tile_group", ctypes.c_void_p), ("wg_range", ctypes.c_void_p), ("wg_entry", ctypes.c_void_p),
        ("n_workgroups", ctypes.c_int32), ("n_entries", ctypes.c_int32), ("entry_group", ctypes.c_void_p), ("group_scale", ctypes.c_void_p),
        ("grad_w", ctypes.c_void_p), ("ldw", ctypes.c_int64), ("grad_b", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64), ("a_plan_order", ctypes.c_int32), ("_pad", ctypes.c_int32),
    ]


class PnaTowerLayerArgs(_Args):
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("_abi_reserved", ctypes.c_uint32),
        ("rowptr", ctypes.c_void_p), ("col", ctypes.c_void_p), ("V", ctypes.c_int32), ("n_tower", ctypes.c_int32),
        ("Fi", ctypes.c_int32), ("Fo", ctypes.c_int32), ("divide_input", ctypes.c_int32), ("n_scaler", ctypes.c_int32),
        ("h", ctypes.c_void_p), ("ldh", ctypes.c_int64), ("x_cat", ctypes.c_void_p), ("ldx", ctypes.c_int64),
        ("proj_img", ctypes.c_void_p), ("proj_bias", ctypes.c_void_p),
        ("row_scale", ctypes.c_void_p * PNA_MAX_SCALER),
        ("post_img", ctypes.c_void_p), ("post_bias", ctypes.c_void_p), ("row_post", ctypes.c_void_p),
        ("col_scale", ctypes.c_void_p), ("col_shift", ctypes.c_void_p),
        ("mix_img", ctypes.c_void_p), ("mix_bias", ctypes.c_void_p), ("No", ctypes.c_int32), ("mix_act", ctypes.c_int32),
        ("mix_slope", ctypes.c_float), ("_pad", ctypes.c_int32),
        ("residual", ctypes.c_void_p), ("ld_res", ctypes.c_int64), ("y", ctypes.c_void_p), ("ldy", ctypes.c_int64),
        ("edge_type", ctypes.c_void_p), ("edge_table", ctypes.c_void_p), ("ld_edge_table", ctypes.c_int64),
        ("n_edge_types", ctypes.c_int32), ("no_self_panel", ctypes.c_int32),
    ]


_lib = None


def lib():
    """The loaded shared library; raises (never falls back) when it is absent or has the wrong ABI."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m pna_amd.build` (hipcc, gfx950). "
                "pna_amd has no CPU / eager fallback.")
        L = ctypes.CDLL(LIB_PATH)
        L.pna_abi_version.restype = ctypes.c_int
        L.pna_last_error.restype = ctypes.c_char_p
        L.pna_segreduce_fwd_f32.argtypes = [ctypes.POINTER(PnaSegreduceArgs), ctypes.c_void_p]
        L.pna_segreduce_fwd_f32.restype = ctypes.c_int
        L.pna_segreduce_bwd_f32.argtypes = [ctypes.POINTER(PnaSegreduceBwdArgs), ctypes.c_void_p]
        L.pna_segreduce_bwd_f32.restype = ctypes.c_int
        L.pna_segreduce_bwd_rowprep_f32.argtypes = [ctypes.POINTER(PnaSegreduceBwdArgs), ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.pna_segreduce_bwd_rowprep_f32.restype = ctypes.c_int
        L.pna_segreduce_bwd_argscatter_f32.argtypes = [ctypes.POINTER(PnaSegreduceBwdArgs), ctypes.c_void_p]
        L.pna_segreduce_bwd_argscatter_f32.restype = ctypes.c_int
        L.pna_segreduce_partials_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.pna_segreduce_partials_bytes.restype = ctypes.c_int64
        L.pna_degree_scalers_f32.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p]
        L.pna_degree_scalers_f32.restype = ctypes.c_int
        L.pna_posttrans_f32.argtypes = [ctypes.POINTER(PnaPosttransArgs), ctypes.c_void_p]
        L.pna_posttrans_f32.restype = ctypes.c_int
        L.pna_posttrans_x3_f32.argtypes = [ctypes.POINTER(PnaPosttransArgs), ctypes.c_void_p]
        L.pna_posttrans_x3_f32.restype = ctypes.c_int
        L.pna_posttrans_x3_packed_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                    ctypes.POINTER(ctypes.c_int64)]
        L.pna_posttrans_x3_packed_bytes.restype = ctypes.c_int64
        L.pna_posttrans_x3_pack_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_void_p]
        L.pna_posttrans_x3_pack_f32.restype = ctypes.c_int
        L.pna_collate_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
        L.pna_collate_workspace_bytes.restype = ctypes.c_int64
        L.pna_collate_csr_i32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.pna_collate_csr_i32.restype = ctypes.c_int
        L.pna_pack_rows_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.pna_pack_rows_f32.restype = ctypes.c_int
        L.pna_project_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.pna_project_f32.restype = ctypes.c_int
        L.pna_project_scaled_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                             ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.pna_project_scaled_f32.restype = ctypes.c_int
        L.pna_project_grouped_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        L.pna_project_grouped_f32.restype = ctypes.c_int
        L.pna_fused_simple_f32.argtypes = [ctypes.POINTER(PnaFusedSimpleArgs), ctypes.c_void_p]
        L.pna_fused_simple_f32.restype = ctypes.c_int
        L.pna_fused_degree_image_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
        L.pna_fused_degree_image_bytes.restype = ctypes.c_int64
        L.pna_fused_degree_tile_rows.argtypes = [ctypes.c_int32, ctypes.c_int32]
        L.pna_fused_degree_tile_rows.restype = ctypes.c_int32
        L.pna_fused_degree_pack_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.pna_fused_degree_pack_f32.restype = ctypes.c_int
        L.pna_segreduce_bwd_pull_f32.argtypes = [ctypes.POINTER(PnaSegreduceBwdPullArgs), ctypes.c_void_p]
        L.pna_segreduce_bwd_pull_f32.restype = ctypes.c_int
        L.pna_posttrans_dw_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.pna_posttrans_dw_workspace_bytes.restype = ctypes.c_int64
        L.pna_posttrans_dw_f32.argtypes = [ctypes.POINTER(PnaPosttransDwArgs), ctypes.c_void_p]
        L.pna_posttrans_dw_f32.restype = ctypes.c_int
        L.pna_posttrans_dw_grouped_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.pna_posttrans_dw_grouped_workspace_bytes.restype = ctypes.c_int64
        L.pna_posttrans_dw_grouped_f32.argtypes = [ctypes.POINTER(PnaPosttransDwGroupedArgs), ctypes.c_void_p]
        L.pna_posttrans_dw_grouped_f32.restype = ctypes.c_int
        L.pna_bn_tail_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
        L.pna_bn_tail_workspace_bytes.restype = ctypes.c_int64
        for fn in (L.pna_bn_tail_fwd_f32, L.pna_bn_tail_bwd_f32):
            fn.argtypes = [ctypes.POINTER(PnaBnTailArgs), ctypes.c_void_p]
            fn.restype = ctypes.c_int
        L.pna_fused_tower_image_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
        L.pna_fused_tower_image_bytes.restype = ctypes.c_int64
        L.pna_fused_tower_pack_f32.argtypes = L.pna_fused_degree_pack_f32.argtypes
        L.pna_fused_tower_pack_f32.restype = ctypes.c_int
        L.pna_fused_image_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.pna_fused_image_bytes.restype = ctypes.c_int64
        L.pna_fused_pack_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                         ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        L.pna_fused_pack_f32.restype = ctypes.c_int
        L.pna_fused_degree_guard_bytes.argtypes = [ctypes.c_int64]
        L.pna_fused_degree_guard_bytes.restype = ctypes.c_int64
        L.pna_fused_degree_f32.argtypes = [ctypes.POINTER(PnaFusedDegreeArgs), ctypes.c_void_p]
        L.pna_fused_degree_f32.restype = ctypes.c_int
        L.pna_posttrans_packed_floats.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.POINTER(ctypes.c_int64)]
        L.pna_posttrans_packed_floats.restype = ctypes.c_int64
        L.pna_posttrans_pack_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.pna_posttrans_pack_f32.restype = ctypes.c_int
        L.pna_small_packed_floats.argtypes = [ctypes.c_int32, ctypes.c_int32]
        L.pna_small_packed_floats.restype = ctypes.c_int64
        L.pna_small_pack_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        L.pna_small_pack_f32.restype = ctypes.c_int
        L.pna_small_linear_f32.argtypes = [ctypes.POINTER(PnaSmallLinearArgs), ctypes.c_void_p]
        L.pna_small_linear_f32.restype = ctypes.c_int
        L.pna_tower_post_packed_floats.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.pna_tower_post_packed_floats.restype = ctypes.c_int64
        L.pna_tower_post_pack_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_void_p, ctypes.c_void_p]
        L.pna_tower_post_pack_f32.restype = ctypes.c_int
        L.pna_tower_layer_f32.argtypes = [ctypes.POINTER(PnaTowerLayerArgs), ctypes.c_void_p]
        L.pna_tower_layer_f32.restype = ctypes.c_int
        if L.pna_abi_version() != PNA_ABI_VERSION:
            raise RuntimeError(f"libpna_amd.so ABI {L.pna_abi_version()} != binding {PNA_ABI_VERSION}: rebuild")
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib().pna_last_error().decode()}")


def dev_ptr(t, dtype, what):
    """Device pointer of a tensor the kernels may touch; enforces GPU residency, dtype and unit inner stride."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"pna_amd: `{what}` must live on a GPU (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"pna_amd: `{what}` must be {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.numel() > 0 and t.stride(-1) != 1:
        raise ValueError(f"pna_amd: `{what}` must have unit stride in its last dimension")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
