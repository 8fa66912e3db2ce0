"""ctypes binding of liblivingscenes_hip.so (the C ABI declared in include/livingscenes_hip.h).

There is NO CPU fallback: if the HIP library is missing, or a tensor handed to an operator is not a HIP
("cuda") tensor, the call raises.  PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LS_LIB_PATH") or os.path.join(_HERE, "lib", "liblivingscenes_hip.so")   # LS_LIB_PATH: A/B builds

LS_MAX_LAYERS = 8
FLAG_CONTRACT_FMA = 1
FLAG_KNN_MFMA_FILTER = 2
FLAG_KNN_VALU_ONLY = 4
FLAG_KABSCH_RAW_WEIGHTS = 8
OPT_SDF_TRAIN_SPLITK, OPT_SDF_BF16X2, OPT_ENCODE_GRAPH, OPT_EDGE_STAGED, OPT_EDGE_FUSE_Q, OPT_EDGE_FUSE_T, OPT_GLOB_FUSE, OPT_DEBUG_EDGE, OPT_GEMM_OVERLAP = 1, 2, 3, 4, 5, 6, 7, 8, 9
ABI_VERSION = 103   # == LS_ABI_VERSION in include/livingscenes_hip.h: a library of another version is refused (argument layouts differ)
KABSCH_OK, KABSCH_RANK1, KABSCH_RANK0, KABSCH_NONFINITE = 0, 1, 2, 3


class LsError(RuntimeError):
    pass


class ModelDesc(ctypes.Structure):
    _fields_ = [
        ("num_layers", ctypes.c_int32),
        ("feat_dim", ctypes.c_int32 * LS_MAX_LAYERS),
        ("down_factor", ctypes.c_int32 * LS_MAX_LAYERS),
        ("atten_start_layer", ctypes.c_int32),
        ("atten_head_c", ctypes.c_int32),
        ("res_global_start_layer", ctypes.c_int32),
        ("num_knn", ctypes.c_int32),
        ("c_dim", ctypes.c_int32),
        ("center_pred", ctypes.c_int32),
        ("center_pred_scale", ctypes.c_int32),
        ("scale_factor", ctypes.c_float),
        ("neg_slope", ctypes.c_float),
        ("dec_num_linear", ctypes.c_int32),
        ("dec_width", ctypes.c_int32),
        ("dec_latent_in", ctypes.c_int32),
        ("off_l0", ctypes.c_int64),
        ("off_edge", ctypes.c_int64 * LS_MAX_LAYERS),
        ("off_glob", ctypes.c_int64 * LS_MAX_LAYERS),
        ("off_convc", ctypes.c_int64),
        ("off_inv_t", ctypes.c_int64),
        ("off_c_fc0_t", ctypes.c_int64),
        ("off_c_misc", ctypes.c_int64),
        ("off_dec_w", ctypes.c_int64 * 12),
        ("off_dec_b", ctypes.c_int64 * 12),
        ("off_dec_inv_t", ctypes.c_int64 * 12),
        ("off_dec_so3_t", ctypes.c_int64 * 12),
        ("off_dec_len", ctypes.c_int64 * 12),
        ("blob_floats", ctypes.c_int64),
    ]


class ProfileEntry(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("layer", ctypes.c_int32), ("launches", ctypes.c_int32), ("total_ms", ctypes.c_float)]


KERNEL_KINDS = ["prologue", "fps", "knn", "gemm_edge", "edge_l0", "edge_pool", "edge_attn", "mean", "gemm_glob", "vn_act",
                "gemm_tail", "tail", "sdf_prep", "sdf_affine", "gemm_sdf", "sdf_out"]

_P = ctypes.c_void_p
_I = ctypes.c_int
_U = ctypes.c_uint
_F = ctypes.c_float
_D = ctypes.c_double
_SZ = ctypes.c_size_t

# name -> (restype, argtypes); every symbol include/livingscenes_hip.h declares
class SoftminProblem(ctypes.Structure):
    """ls_softmin_problem (include/livingscenes_hip.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("pot_y", ctypes.c_void_p), ("prev", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("logw", ctypes.c_float), ("N", ctypes.c_int), ("M", ctypes.c_int)]


class AdamGroup(ctypes.Structure):
    """ls_adam_group (include/livingscenes_hip.h)."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("n", ctypes.c_longlong),
                ("lr", ctypes.c_double)]


SIGNATURES = {
    "ls_version": (_I, []),
    "ls_last_error": (ctypes.c_char_p, []),
    "ls_device_count": (_I, []),
    "ls_knn_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I, _U]),
    "ls_knn_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _U, _P, _P, _P, _SZ, _P]),
    "ls_fps_workspace_bytes": (_SZ, [_I, _I, _I]),
    "ls_fps_f32": (_I, [_P, _P, _I, _I, _I, _U, _P, _P, _P, _SZ, _P]),
    "ls_gemm_workspace_bytes": (_SZ, [_I, _I, _I]),
    "ls_gemm_f32": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _SZ, _P]),
    "ls_gemm_rowmax_parts": (_I, [_I]),
    "ls_gemm_f32_ex": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "ls_rowmax_f32": (_I, [_P, _I, _I, _I, _P, _P]),
    "ls_gemm_w_planes_bytes": (_SZ, [_I, _I]),
    "ls_gemm_presplit_w_f32": (_I, [_P, _I, _I, _I, _P, _P, _SZ, _P]),
    "ls_gemm_f32_planes": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P]),
    "ls_encode_prologue_f32": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "ls_cosine_scores_workspace_bytes": (_SZ, [_I, _I]),
    "ls_cosine_scores_f32": (_I, [_P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "ls_greedy_match_f32": (_I, [_P, _I, _I, _P, _P, _P]),
    "ls_nn_match_f32": (_I, [_P, _I, _I, _P, _P, _P]),
    "ls_sinkhorn_match_f32": (_I, [_P, _I, _I, _F, _F, _I, _F, _P, _P, _P]),
    "ls_kabsch_batched_f32": (_I, [_P, _P, _P, _I, _I, _U, _P, _P, _P, _P, _P]),
    "ls_kabsch_codes_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "ls_kabsch_residual_matrix_f32": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "ls_icp_workspace_bytes": (_SZ, [_I, _I]),
    "ls_icp_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _U, _P, _P, _P, _P, _P, _SZ, _P]),
    "ls_model_create": (_I, [ctypes.POINTER(ModelDesc), _P, ctypes.POINTER(_P)]),
    "ls_model_destroy": (None, [_P]),
    "ls_model_set_option": (_I, [_P, _I, _I]),
    "ls_profile_knn_stats": (_I, [_P, ctypes.POINTER(ctypes.c_ulonglong), _I]),
    "ls_model_get_option": (_I, [_P, _I, ctypes.POINTER(ctypes.c_int)]),
    "ls_se3_transform_f32": (_I, [_P, _P, _I, _I, _P, _P]),
    "ls_smooth_l1_f32": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "ls_sinkhorn_softmin_batched_f32": (_I, [_P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "ls_sinkhorn_softmin_multi_f32": (_I, [_P, _I, _P, _I, _I, _P]),
    "ls_mse_f32": (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    "ls_adam_step_f32": (_I, [ctypes.POINTER(AdamGroup), _I, _D, _D, _D, _I, _P]),
    "ls_se3_adam_step_f32": (_I, [_P, _P, _P, _I, _I, _D, _D, _D, _D, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ls_encoder_workspace_bytes": (_SZ, [_P, _I, _I]),
    "ls_encode": (_I, [_P, _P, _I, _I, _I, _U, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ls_vn_edgeconv_workspace_bytes": (_SZ, [_P, _I, _I, _I, _I, _I]),
    "ls_vn_edgeconv_pool_f32": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "ls_vn_edgeconv_attn_f32": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "ls_vn_lna_workspace_bytes": (_SZ, [_P, _I, _I, _I]),
    "ls_vn_lna_f32": (_I, [_P, _I, _P, _I, _I, _P, _P, _SZ, _P]),
    "ls_encoder_tail_workspace_bytes": (_SZ, [_P, _I, _I]),
    "ls_encoder_tail_f32": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "ls_sdf_workspace_bytes": (_SZ, [_P, _I, _I]),
    "ls_sdf_decode": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _SZ, _P]),
    "ls_sdf_rows_workspace_bytes": (_SZ, [_P, _I, ctypes.c_longlong]),
    "ls_sdf_decode_rows": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, ctypes.c_longlong, _P, _P, _SZ, _P]),
    "ls_sdf_train_workspace_bytes": (_SZ, [_P, _I, _I]),
    "ls_sdf_decode_train": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _SZ, _P]),
    "ls_sdf_backward": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _SZ, _P, _P, _P, _P, _P, _P]),
    "ls_sinkhorn_softmin_f32": (_I, [_P, _P, _P, _I, _I, _F, _P, _P, _P]),
    "ls_mise_state_bytes": (_SZ, [_I, _I]),
    "ls_mise_lattice_points": (ctypes.c_longlong, [_I, _I]),
    "ls_mise_init": (_I, [_P, _SZ, _I, _I, _P]),
    "ls_mise_query": (_I, [_P, _I, _I, _F, _P, _P, _I, _P, _P]),
    "ls_mise_update": (_I, [_P, _I, _I, ctypes.c_double, _P, _P, _I, _P]),
    "ls_mise_to_dense": (_I, [_P, _I, _I, _P, _P]),
    "ls_mcubes_workspace_bytes": (_SZ, [_I, _I, _I]),
    "ls_marching_cubes_f64": (_I, [_P, _I, _I, _I, ctypes.c_double, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _SZ, _P]),
    "ls_simplify_mesh_f64_host": (_I, [_P, ctypes.c_longlong, _P, ctypes.c_longlong, _I, ctypes.c_double, _I, _P, _P, _P]),
    "ls_profile_begin": (_I, [_P]),
    "ls_profile_end": (_I, [_P, ctypes.POINTER(ProfileEntry), _I, ctypes.POINTER(ctypes.c_int)]),
}

_lib = None


def load():
    """Load the shared library (raises LsError if it has not been built -- no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LsError(f"{LIB_PATH} is missing: run `python -m livingscenes_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback for the HIP operators.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        got = int(lib.ls_version())
        if got != ABI_VERSION:
            raise LsError(f"{LIB_PATH} reports C-ABI version {got}, this binding was written for {ABI_VERSION}: rebuild the library "
                          "(`python -m livingscenes_amd.build --force`) -- a stale library would reinterpret arguments silently")
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().ls_last_error().decode(errors="replace")
        raise LsError(f"{what} failed (status {rc}): {msg}")


def call(device, name, *args):
    """Invoke operator `name` with `device` current: the library launches on the passed stream, and a kernel launched on the NULL
    (default) stream goes to whatever device is current -- tensors on a non-current GPU would otherwise fault or compute garbage."""
    with torch.cuda.device(device):
        check(getattr(load(), name)(*args), name)


def ptr(t):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not torch.is_tensor(t):
        raise LsError("expected a torch tensor")
    if t.device.type != "cuda":
        raise LsError(f"HIP operator called with a {t.device.type} tensor: the MI355X path has no CPU fallback")
    if not t.is_contiguous():
        raise LsError("HIP operators need contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
