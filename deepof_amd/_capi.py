"""ctypes signatures of the libdeepof_hip C ABI (include/deepof_hip.h).

``bind(cdll)`` attaches argtypes/restypes; ``check(lib, rc)`` raises with the library's error text.
The product loads ``deepof_amd/csrc/libdeepof_hip.so`` through ``deepof_amd._lib`` (and fails
loudly if it is missing); this module itself has no knowledge of where a library comes from.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 17

# hyper[] indices (enum in deepof_hip.h)
H_KLW, H_LAMBDA_DISTILL, H_KM_LATENT, H_KM_LOSS, H_REPEL_W, H_REPEL_LS = 0, 1, 2, 3, 4, 5
H_NONEMPTY_W, H_NONEMPTY_FLOOR, H_NONEMPTY_P, H_L1_ACT, H_DISTILL_T, H_CONF_W = 6, 7, 8, 9, 10, 11
H_CONF_THR, H_HAS_TEACHER, H_LOGVAR_LO, H_LOGVAR_HI = 12, 13, 14, 15
H_CLIP, H_WD, H_LR0, H_BC0, H_ACTIVE0, H_VQ_BETA = 16, 17, 18, 22, 30, 34
H_TF_W, H_CAT_W, H_TEMPORAL_W, H_SCATTER_W, H_SCATTER_BETA, H_COUNT = 36, 37, 38, 39, 40, 42
SEG_ENCODER, SEG_DECODER, SEG_GMM, SEG_HEADS, SEG_COUNT = 0, 1, 2, 3, 4
LOG_KEYS = ("total_loss", "reconstruct_loss", "kl_div", "cat_clust_loss", "kmeans_loss", "activity_l1",
            "prior_loss", "distill_loss", "tf_clust_loss", "nonempty_loss", "temporal_loss", "scatter_loss",
            "repel_loss", "kl_weight")
LOG_ENC_REC, LOG_VQ, LOG_POPULATED, LOG_POS_SIM, LOG_NEG_SIM = 14, 15, 16, 17, 18
MAX_ROT = 8
SIMILARITIES = {"cosine": 0, "dot": 1, "euclidean": 2, "edit": 2}
CONTRASTIVE_LOSSES = {"nce": 0, "dcl": 1, "dlc": 1, "hard_dcl": 2, "fc": 3}
LOG_COUNT = 20


class VadeDims(C.Structure):
    _fields_ = [("batch", C.c_int32), ("window", C.c_int32), ("n_nodes", C.c_int32), ("n_edges", C.c_int32),
                ("latent", C.c_int32), ("n_clusters", C.c_int32), ("mc_samples", C.c_int32)]


class Augment(C.Structure):
    _fields_ = [("start", C.c_void_p), ("n_rot", C.c_int32), ("rot_pivot", C.c_int32 * 8),
                ("rot_nodes", C.c_uint64 * 8), ("theta", C.c_void_p), ("interp_t0", C.c_void_p),
                ("interp_len", C.c_void_p), ("noise", C.c_void_p)]


class SchedItem(C.Structure):
    _fields_ = [("table", C.c_void_p), ("cursor", C.c_void_p), ("len", C.c_int32), ("hyper_index", C.c_int32),
                ("advance", C.c_int32), ("scale", C.c_float)]


SCHED_MAX_ITEMS = 4


class NoiseBuf(C.Structure):
    _fields_ = [("out", C.c_void_p), ("n", C.c_int64)]


NOISE_MAX_BUFS = 2


class TurtleDims(C.Structure):
    _fields_ = [("batch", C.c_int32), ("n_views", C.c_int32), ("n_clusters", C.c_int32), ("view_dim", C.c_int32 * 8)]


class TurtleHyper(C.Structure):
    _fields_ = [("gamma", C.c_float), ("alpha", C.c_float), ("delta", C.c_float), ("head_temp", C.c_float),
                ("task_temp", C.c_float), ("inner_lr", C.c_float), ("head_wd", C.c_float), ("lr_theta", C.c_float),
                ("rho", C.c_float), ("inner_steps", C.c_int32), ("normalize_feats", C.c_int32)]


class PreprocDims(C.Structure):
    _fields_ = [("n_frames", C.c_int64), ("n_videos", C.c_int32), ("n_cols", C.c_int32), ("n_animals", C.c_int32),
                ("n_node_cols", C.c_int32), ("n_edge_cols", C.c_int32), ("n_angle_cols", C.c_int32),
                ("speed_mode", C.c_int32), ("dist_mode", C.c_int32), ("coord_mode", C.c_int32),
                ("log_distances", C.c_int32), ("inter_scale", C.c_int32), ("fit_global", C.c_int32), ("clip", C.c_double),
                ("scale_kind", C.c_int32), ("reserved", C.c_int32), ("col_keep", C.c_void_p), ("video_scaler_in", C.c_void_p)]


PP_KINDS = {"other": 0, "coord": 1, "speed": 2, "dist_inner": 3, "dist_intra": 4, "angle": 5}
PP_MODES = {None: 0, "per_column": 1, "groupwise": 2}
PP_INTER_SCALE = {"mean": 0, "geom": 1, "global": 2}
PP_SCALE_KINDS = {"standard": 0, "minmax": 1, "robust": 2}
PP_ORDER_DOUBLES = 7  # n + six order statistics (median pair, 25th-percentile pair, 75th-percentile pair)
PP_STAT_DOUBLES = 5   # (n, mean, M2, min, max)
PP_MAX_COLS, PP_MAX_ANIMALS = 512, 8

_P = C.c_void_p
_I64 = C.c_int64
_I32 = C.c_int32
COMM_ID_BYTES = 128

SIGNATURES = {
    "dof_last_error_string": (C.c_char_p, []),
    "dof_abi_version": (C.c_int, []),
    "dof_window_gather": (C.c_int, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "dof_window_gather_range": (C.c_int, [_P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "dof_window_gather_bf16": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "dof_widen_bf16": (C.c_int, [_P, _P, _I64, _P]),
    "dof_comm_unique_id": (C.c_int, [_P]),
    "dof_comm_create": (C.c_int, [_P, _I32, _I32, C.POINTER(_P)]),
    "dof_comm_destroy": (C.c_int, [_P]),
    "dof_comm_abort": (C.c_int, [_P]),
    "dof_flat_allreduce": (C.c_int, [_P, _P, _I64, _P]),
    "dof_comm_broadcast": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "dof_vade_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_vade_tcn_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_vqvae_tcn_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_vade_tfm_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_vqvae_tfm_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_contrastive_tfm_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_tfm_dropout_site_count": (_I32, [_P]),
    "dof_tfm_dropout_site_name": (C.c_char_p, [_P, _I32]),
    "dof_tfm_dropout_site_offset": (_I64, [_P, _I32]),
    "dof_tfm_dropout_site_numel": (_I64, [_P, _I32]),
    "dof_tfm_dropout_site_p": (C.c_float, [_P, _I32]),
    "dof_tfm_set_dropout": (C.c_int, [_P, _P, C.c_uint32]),
    "dof_tfm_set_dropout_counter": (C.c_int, [_P, _P]),
    "dof_vade_plan_destroy": (None, [_P]),
    "dof_vade_param_count": (_I32, [_P]),
    "dof_vade_param_name": (C.c_char_p, [_P, _I32]),
    "dof_vade_param_offset": (_I64, [_P, _I32]),
    "dof_vade_param_numel": (_I64, [_P, _I32]),
    "dof_vade_param_total": (_I64, [_P]),
    "dof_vade_param_shape": (_I32, [_P, _I32, C.POINTER(_I64)]),
    "dof_vade_set_trainable": (C.c_int, [_P, _I32, _I32, _P]),
    "dof_vade_set_batchnorm_training": (C.c_int, [_P, _I32]),
    "dof_vade_workspace_bytes": (_I64, [_P]),
    "dof_vade_bind": (C.c_int, [_P, _P, _P]),
    "dof_vade_ws_tensor": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dof_vade_forward": (C.c_int, [_P] * 13),
    "dof_vade_loss_grads": (C.c_int, [_P] * 10 + [_I32, _P, _P, _P]),
    "dof_vqvae_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_vqvae_forward": (C.c_int, [_P] * 11),
    "dof_vqvae_loss_grads": (C.c_int, [_P] * 9),
    "dof_optimizer_step": (C.c_int, [_P] * 7 + [C.c_float, _P]),
    "dof_schedule_apply": (C.c_int, [_P, C.POINTER(SchedItem), _I32, _P]),
    "dof_step_begin": (C.c_int, [_P, C.POINTER(SchedItem), _I32, C.c_uint64, _P, C.POINTER(NoiseBuf), _I32, _P]),
    "dof_vade_set_log_accumulator": (C.c_int, [_P, _P]),
    "dof_turtle_param_total": (_I64, [C.POINTER(TurtleDims)]),
    "dof_turtle_param_offset": (_I64, [C.POINTER(TurtleDims), _I32, _I32, _I32]),
    "dof_turtle_workspace_bytes": (_I64, [C.POINTER(TurtleDims)]),
    "dof_turtle_fit_step": (C.c_int, [C.POINTER(TurtleDims), C.POINTER(TurtleHyper), C.POINTER(_P), _P, _P, _P, _I32,
                                      _I32, _P, _P, _P]),
    "dof_turtle_predict": (C.c_int, [C.POINTER(TurtleDims), C.c_float, C.POINTER(_P), _P, _I64, _P, _P]),
    "dof_contrastive_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_contrastive_tcn_plan_create": (C.c_int, [C.POINTER(VadeDims), _P, _P, _P, C.POINTER(_P)]),
    "dof_contrastive_views": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, C.POINTER(Augment), _P, _P, _P]),
    "dof_contrastive_encode": (C.c_int, [_P, _P, _P, _P, _I32, _P, _P]),
    "dof_contrastive_loss": (C.c_int, [_P, _P, _P, _I32, _I32, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "dof_contrastive_backward": (C.c_int, [_P, _P, _P, _P, _I32, _P]),
    "dof_preprocess_workspace_bytes": (_I64, [C.POINTER(PreprocDims)]),
    "dof_preprocess_tables": (C.c_int, [C.POINTER(PreprocDims)] + [_P] * 16),
    "dof_preprocess_video_stats": (C.c_int, [C.POINTER(PreprocDims)] + [_P] * 10),
    "dof_preprocess_fit_global": (C.c_int, [C.POINTER(PreprocDims), _I32, _P, _P, _P, _P]),
    "dof_preprocess_raw_moments": (C.c_int, [C.POINTER(PreprocDims)] + [_P] * 6),
    "dof_preprocess_order_stats": (C.c_int, [C.POINTER(PreprocDims)] + [_P] * 11),
}


def bind(cdll):
    missing = [n for n in SIGNATURES if not hasattr(cdll, n)]
    if missing:
        raise RuntimeError(f"libdeepof_hip is missing exported symbols: {missing}")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    if cdll.dof_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libdeepof_hip ABI {cdll.dof_abi_version()} != expected {ABI_VERSION}")
    return cdll


class DofError(RuntimeError):
    pass


def check(lib, rc, what=""):
    if rc != 0:
        msg = lib.dof_last_error_string()
        raise DofError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
