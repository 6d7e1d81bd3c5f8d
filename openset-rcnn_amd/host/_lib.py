"""ctypes binding of libosr_hip.so (include/osr.h). The product path has no fallback: if the library is
missing or an entry point fails, an exception is raised -- never a silent eager/CPU substitute."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: libosr_hip.so then binds to torch's libamdhip64.so.7)

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "libosr_hip.so")

OSR_MAX_LEVELS = 8
OSR_F32, OSR_F16, OSR_BF16 = 0, 1, 2
ABI_VERSION = 1


class OsrError(RuntimeError):
    pass


ERR_UNSUPPORTED = -2  # include/osr.h OSR_ERR_UNSUPPORTED: nothing was launched, the caller takes its other path


class ConvParams(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("hi", C.c_int32), ("wi", C.c_int32), ("cin", C.c_int32),
        ("ho", C.c_int32), ("wo", C.c_int32), ("cout", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride_h", C.c_int32), ("stride_w", C.c_int32),
        ("pad_h", C.c_int32), ("pad_w", C.c_int32),
        ("in_stride_n", C.c_int64), ("in_stride_h", C.c_int64), ("in_stride_w", C.c_int64),
        ("out_stride_n", C.c_int64), ("out_stride_h", C.c_int64), ("out_stride_w", C.c_int64),
        ("res_stride_n", C.c_int64), ("res_stride_h", C.c_int64), ("res_stride_w", C.c_int64),
        ("relu", C.c_int32), ("res_mode", C.c_int32), ("pad_mode", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("concurrency", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("row_seg_counts", C.c_void_p), ("row_seg_rows", C.c_int32),
    ]


class ConvLevel(C.Structure):
    """include/osr.h osr_conv_level: one pyramid level of a multi-level launch."""
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("deltas", C.c_void_p), ("ctr", C.c_void_p), ("weight", C.c_void_p), ("bias", C.c_void_p),
                ("n", C.c_int32), ("hi", C.c_int32), ("wi", C.c_int32), ("reserved", C.c_int32)]


MAX_CONV_LEVELS = 6


class BottleneckParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("cmid", C.c_int32), ("cout", C.c_int32),
                ("dtype", C.c_int32), ("has_proj", C.c_int32)]


class LossOptions(C.Structure):
    _fields_ = [("box_loss_type", C.c_int32), ("box_smooth_l1_beta", C.c_float), ("aux_smooth_l1_beta", C.c_float)]


BOX_LOSS_TYPES = {"iou": 0, "smooth_l1": 1, "giou": 2, "diou": 3, "ciou": 4}
PLN_DISTANCES = {"COS": 0, "L1": 1, "L2": 2}


class RpnLevels(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int32), ("num_anchors", C.c_int32),
        ("h", C.c_int32 * OSR_MAX_LEVELS), ("w", C.c_int32 * OSR_MAX_LEVELS),
        ("stride", C.c_int32 * OSR_MAX_LEVELS), ("offset", C.c_int64 * OSR_MAX_LEVELS),
    ]


class SgdTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum", C.c_void_p), ("row_scale", C.c_void_p), ("lowp", C.c_void_p),
                ("n", C.c_int64), ("row_elems", C.c_int64), ("lowp_dtype", C.c_int32), ("reserved", C.c_int32)]


class PackTensor(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("cin", C.c_int32),
                ("elem_bytes", C.c_int32), ("reserved", C.c_int32)]


class Pyramid(C.Structure):
    _fields_ = [
        ("num_levels", C.c_int32), ("c", C.c_int32),
        ("h", C.c_int32 * OSR_MAX_LEVELS), ("w", C.c_int32 * OSR_MAX_LEVELS),
        ("scale", C.c_float * OSR_MAX_LEVELS), ("data", C.c_void_p * OSR_MAX_LEVELS),
    ]


P, I32, I64, F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
F32x3, F32x4 = C.c_float * 3, C.c_float * 4

# name -> (restype, argtypes); mirrors include/osr.h one to one
PROTOTYPES = {
    "osr_abi_version": (I32, []),
    "osr_last_error": (C.c_char_p, []),
    "osr_stem_padded_width": (I32, [I32]),
    "osr_preprocess": (I32, [P, I32, I32, I32, I32, I32, I32, C.POINTER(C.c_float), C.POINTER(C.c_float), P, I32, P]),
    "osr_conv2d_fwd": (I32, [C.POINTER(ConvParams), P, P, P, P, P, P]),
    "osr_conv2d_fwd_workspace_bytes": (I64, [C.POINTER(ConvParams)]),
    "osr_conv2d_fwd_describe": (I32, [C.POINTER(ConvParams), I32, P, I32]),
    "osr_conv2d_fwd_masked": (I32, [C.POINTER(ConvParams), P, P, P, P, P, P, P]),
    "osr_conv2d_chain_fwd": (I32, [C.POINTER(ConvParams), P, P, P, P, P, I32, P, P, P]),
    "osr_conv2d_chain_fwd_ex": (I32, [C.POINTER(ConvParams), P, P, P, P, P, I32, P, P, P, P]),
    "osr_bottleneck_fwd": (I32, [C.POINTER(BottleneckParams), P, P, P, P, P, P, P, P, P, P, P]),
    "osr_resize_tmp_bytes": (I64, [I32, I32]),
    "osr_resize_bilinear_u8": (I32, [P, I32, I32, I64, P, P, I32, P, P, I32, I32, I32, I32, I32, P, I64, P, P]),
    "osr_stem_maxpool_fwd": (I32, [P, I32, I32, I32, P, I32, P, P, I32, P]),
    "osr_stem_maxpool_fwd_raw": (I32, [P, I32, I32, I32, I32, I32, I32, C.POINTER(C.c_float), C.POINTER(C.c_float), P, I32, P, P, I32, P]),
    "osr_maxpool3x3s2": (I32, [P, I32, I32, I32, I32, P, I32, P]),
    "osr_subsample2": (I32, [P, I32, I32, I32, I32, P, I32, P]),
    "osr_gemm_f32": (I32, [P, I64, P, P, P, I64, I32, I32, I32, I32, P]),
    "osr_gemm_f32_tn_workspace_bytes": (I64, [I32, I32, I32]),
    "osr_gemm_f32_tn": (I32, [P, I64, P, I64, P, I64, I32, I32, I32, P, I64, P]),
    "osr_cfrpn_head_tail": (I32, [P, I32, I64, I32, P, P, P, P, P, P, P]),
    "osr_cfrpn_head_fwd": (I32, [C.POINTER(ConvParams), P, P, P, P, P, P, P, P]),
    "osr_cfrpn_head_fwd_ex": (I32, [C.POINTER(ConvParams), P, P, P, P, P, P, P, P, P]),
    "osr_conv2d_fwd_pair": (I32, [C.POINTER(ConvParams), P, P, P, I32, I32, P, P, P, I32, I32, P, P]),
    "osr_conv2d_fwd_levels": (I32, [C.POINTER(ConvParams), I32, C.POINTER(ConvLevel), P, P, P]),
    "osr_cfrpn_head_fwd_levels": (I32, [C.POINTER(ConvParams), I32, C.POINTER(ConvLevel), P, P, P, P, P]),
    "osr_rpn_select_capacity": (I32, [C.POINTER(RpnLevels), I32]),
    "osr_rpn_select_workspace_bytes": (I64, [C.POINTER(RpnLevels), I32, I32]),
    "osr_rpn_select": (I32, [C.POINTER(RpnLevels), P, P, P, I32, P, I32, F32, P, P, P, P, P, P, P, I64, P]),
    "osr_rpn_select_ex": (I32, [C.POINTER(RpnLevels), P, P, P, I32, P, I32, F32, I32, P, P, P, P, P, P, P, P, P, I64, P]),
    "osr_fastrcnn_candidates": (I32, [P, P, I32, I32, P, P, I32, I32, P, P, F32, P, P, P, P, P, P]),
    "osr_roi_align_fwd": (I32, [C.POINTER(Pyramid), I32, I32, P, P, I64, I32, I32, I32, I32, P, I32, P]),
    "osr_roi_align_fwd_ordered": (I32, [C.POINTER(Pyramid), I32, I32, P, P, I64, I32, I32, I32, I32, P, P, P, I32, P]),
    "osr_roi_align_fwd_ordered_ex": (I32, [C.POINTER(Pyramid), I32, I32, P, P, I64, I32, I32, I32, I32, P, P, I32, P, I32, P]),
    "osr_roi_locality_order_workspace_bytes": (I64, [I32, I64]),
    "osr_roi_locality_order": (I32, [C.POINTER(Pyramid), I32, P, P, I64, I32, I32, I32, P, P, P, I64, P]),
    "osr_box_predictor_tail": (I32, [P, I64, I32, P, P, P, P, P, P, C.POINTER(C.c_float), I32, F32, P, P, P, P, P, P]),
    "osr_nms_topk_workspace_bytes": (I64, [I32, I64]),
    "osr_nms_topk": (I32, [P, P, P, P, I32, I64, P, F32, I32, P, P, P, I64, P]),
    "osr_gather_rows": (I32, [P, I64, I32, P, P, I32, I32, P, P]),
    "osr_l2_normalize_rows": (I32, [P, I32, I32, P, P]),
    "osr_pln_tail": (I32, [P, I64, I32, P, I32, I32, F32, I64, P, P, I32, P, P, P]),
    "osr_pln_tail_ex": (I32, [P, I64, I32, P, I32, I32, I32, F32, I64, P, P, I32, P, P, P]),
    "osr_softmax_candidates": (I32, [P, I32, P, P, P, P, I32, I32, I64, F32, F32, P, P, P, P, P, P, P, P, P, P]),
    "osr_assemble_detections": (I32, [P, P, P, P, P, I64, I32, P, P, P, P, I64, I32, I32, I64, P, P, P, P, P, P]),
    "osr_detector_postprocess": (I32, [P, P, P, P, I32, I32, P, P, P, P, P, P, P]),
    # training step, forward half
    "osr_rpn_match_anchors": (I32, [P, P, I32, P, P, I32, F32, F32, F32, F32, P, P, P, P, P, I64, P]),
    "osr_subsample_labels": (I32, [P, P, I32, I64, I32, F32, P, P, P]),
    "osr_rpn_anchor_targets": (I32, [P, P, I32, P, P, I32, P, P, P, P, P]),
    "osr_rpn_losses_fwd": (I32, [P, P, I32, P, P, P, P, P, P, F32, F32, I32, P, P, I64, P]),
    "osr_rpn_losses_fwd_ex": (I32, [P, P, I32, P, P, P, P, P, P, F32, F32, I32, P, P, P, I64, P]),
    "osr_roi_match_sample_workspace_bytes": (I64, [I32, I64, I32]),
    "osr_roi_match_and_sample": (I32, [P, P, P, I64, P, P, P, I32, I32, P, I32, I32, F32, F32, P, P, P, P, P, P, P, P, P, I64, P]),
    "osr_roi_box_losses_fwd": (I32, [P, I32, P, I32, I32, P, P, P, P, I64, I32, P, F32, F32, P, P, I64, P]),
    "osr_roi_box_losses_fwd_ex": (I32, [P, I32, P, I32, I32, P, P, P, P, I64, I32, P, F32, F32, P, P, P, I64, P]),
    "osr_pln_loss_fwd": (I32, [P, I64, I32, P, I32, P, P, F32, F32, F32, F32, P, P, I64, P]),
    "osr_pln_loss_fwd_ex": (I32, [P, I64, I32, P, I32, I32, I32, P, P, F32, F32, F32, F32, P, P, I64, P]),
    "osr_softmax_ce_loss_fwd": (I32, [P, I64, I32, P, I32, F32, P, P, I64, P]),
    # training step, backward half
    "osr_conv2d_wgrad_workspace_bytes": (I64, [P]),
    "osr_conv2d_wgrad": (I32, [P, P, P, P, I32, P, I64, P]),
    "osr_bias_grad": (I32, [P, I32, I64, I32, P, I32, P, I64, P]),
    "osr_rpn_losses_bwd": (I32, [P, P, I32, P, P, P, P, P, P, F32, F32, I32, F32, P, P]),
    "osr_rpn_losses_bwd_ex": (I32, [P, P, I32, P, P, P, P, P, P, F32, F32, I32, F32, P, P, P]),
    "osr_cfrpn_tail_bwd_workspace_bytes": (I64, []),
    "osr_cfrpn_tail_bwd": (I32, [P, I32, I64, P, P, P, P, P, I32, P, I64, P]),
    "osr_rpn_sparse_rows_workspace_bytes": (I64, []),
    "osr_rpn_sparse_rows": (I32, [P, I64, I32, P, P, P, P, I64, P]),
    "osr_rpn_gather_cols": (I32, [P, P, I32, I32, P, I32, P, P, P, P]),
    "osr_rpn_scatter_cols_add": (I32, [P, I32, P, P, P, I32, P]),
    "osr_roi_box_losses_bwd": (I32, [P, P, P, P, P, I64, I32, P, F32, F32, F32, P, P, I64, P]),
    "osr_roi_box_losses_bwd_ex": (I32, [P, P, P, P, P, I64, I32, P, F32, F32, F32, P, P, P, I64, P]),
    "osr_softmax_ce_loss_bwd": (I32, [P, I64, I32, P, I32, F32, F32, P, P, I64, P]),
    "osr_pln_loss_bwd_workspace_bytes": (I64, [I64]),
    "osr_pln_loss_bwd": (I32, [P, I64, I32, P, I32, P, P, F32, F32, F32, F32, F32, P, P, I32, P, I64, P]),
    "osr_pln_loss_bwd_ex": (I32, [P, I64, I32, P, I32, I32, I32, P, P, F32, F32, F32, F32, F32, P, P, I32, P, I64, P]),
    "osr_roi_align_bwd": (I32, [P, I32, P, P, I64, I32, I32, I32, I32, P, I32, P]),
    "osr_roi_align_bwd_dense": (I32, [P, I32, P, P, I64, I32, I32, I32, I32, I32, P, I32, I32, P]),
    "osr_relu_mask": (I32, [P, I32, P, I32, I64, P]),
    "osr_add_cast": (I32, [P, P, P, I32, I64, P]),
    "osr_pool_bwd": (I32, [P, I32, I32, P, P, I32, I32, I32, I32, I32, I32, P]),
    "osr_sgd_step": (I32, [P, P, P, I64, F32, F32, F32, F32, P, I64, P, I32, P, P]),
    "osr_check_finite": (I32, [P, I64, P, P]),
    "osr_pack_dgrad_weight": (I32, [P, P, I32, I32, I32, I32, I32, P]),
    "osr_sgd_step_multi": (I32, [P, P, I32, I32, F32, F32, F32, F32, P, P]),
    "osr_pack_dgrad_weight_multi": (I32, [P, P, I32, P]),
}

_lib = None


def load() -> C.CDLL:
    """Load libosr_hip.so; raises OsrError when it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OsrError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError => header/library mismatch
        fn.restype = res
        fn.argtypes = args
    ver = lib.osr_abi_version()
    if ver != ABI_VERSION:
        raise OsrError(f"libosr_hip.so ABI version {ver} != binding version {ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, name: str) -> None:
    if status != 0:
        msg = load().osr_last_error()
        raise OsrError(f"{name} failed (status {status}): {msg.decode() if msg else ''}")
