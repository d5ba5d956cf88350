"""Trainer configuration records (fields and defaults of the reference's CommonFitCfg / TurtleTeacherCfg / VaDECfg /
ContrastiveCfg, /root/reference/deepof/clustering/model_utils_new.py:37-189 -- interface data a caller may rely on)
and input validation (check_model_inputs, model_utils_new.py:377-449).

The records are generated from compact field tables: one row = name, type, default.
"""
from __future__ import annotations

from dataclasses import asdict, field, make_dataclass
from typing import Optional

_TYPES = {"float": float, "int": int, "bool": bool, "str": str, "Optional[str]": Optional[str],
          "Optional[int]": Optional[int], "Optional[float]": Optional[float]}


def _record(name: str, table: str):
    """Dataclass from a whitespace table 'field type default' (default parsed as a Python literal)."""
    fields = []
    for row in table.strip().splitlines():
        fname, ftype, default = row.split(None, 2)
        fields.append((fname, _TYPES[ftype], field(default=eval(default, {}))))  # noqa: S307 (literals from this file)
    cls = make_dataclass(name, fields)
    cls.__module__ = __name__
    return cls


CommonFitCfg = _record("CommonFitCfg", """
    learning_rate               float          0.0003
    model_name                  str            'VaDE'
    encoder_type                str            'recurrent'
    batch_size                  int            1024
    latent_dim                  int            6
    epochs                      int            10
    n_components                int            10
    output_path                 str            '.'
    data_path                   str            '.'
    log_history                 bool           True
    pretrained                  Optional[str]  None
    save_weights                bool           True
    run                         int            0
    num_workers                 int            0
    prefetch_factor             int            0
    use_amp                     bool           False
    interaction_regularization  float          0.0
    kmeans_loss                 float          0.0
    diag_max_batches            int            4
    seed                        Optional[int]  None
    limit_train_batches         Optional[int]  1000
    limit_val_batches           Optional[int]  1000
""")

TurtleTeacherCfg = _record("TurtleTeacherCfg", """
    use_turtle_teacher             bool             False
    teacher_gamma                  float            8.0
    teacher_outer_steps            int              500
    teacher_inner_steps            int              100
    teacher_normalize_feats        bool             True
    teacher_head_temp              float            0.35
    teacher_task_temp              float            0.35
    teacher_alpha_sample_entropy   float            2.0
    lambda_distill                 float            4.0
    lambda_decay_start             int              10
    lambda_end_weight              float            0.2
    lambda_cooldown                int              10
    distill_sharpen_T              float            0.5
    distill_conf_weight            bool             False
    distill_conf_thresh            float            0.3
    generic_lambda_distill         float            2.0
    generic_distill_sharpen_T      float            0.5
    generic_distill_conf_weight    bool             True
    generic_distill_conf_thresh    float            0.6
    generic_distill_warmup_epochs  int              1
    distill_class_reweight_beta    float            1.0
    distill_class_reweight_cap     Optional[float]  3.0
    include_latent_view            bool             True
    include_edges_view             bool             False
    include_nodes_view             bool             True
    include_angles_view            bool             False
    pca_nodes_dim                  int              32
    pca_edges_dim                  int              32
    pca_angles_dim                 int              32
    batch_size_nodes               int              4096
    batch_size_edges               int              8192
    batch_size_angles              int              8192
    teacher_refresh_every          Optional[int]    None
    teacher_freeze_at              Optional[int]    10
    reinit_gmm_on_refresh          bool             False
    teacher_batch_size             int              2048
    pca_backend                    str              'device'
""")

VaDECfg = _record("VaDECfg", """
    learning_rate_pretrain           float  0.001
    gmm_learning_rate                float  0.001
    pretrain_epochs                  int    10
    reg_cat_clusters                 float  0.0
    recluster                        bool   False
    freeze_gmm_epochs                int    0
    freeze_decoder_epochs            int    0
    prior_loss_weight                float  0.0
    reg_scatter_weight               float  0.0
    temporal_cohesion_weight         float  0.0
    reg_scatter_beta                 float  1.0
    repel_weight                     float  0.0
    repel_length_scale               float  1.0
    tf_cluster_weight                float  0.0
    nonempty_weight                  float  0.02
    nonempty_p                       float  2.0
    nonempty_floor_percent           float  0.05
    kmeans_loss_pretrain             float  1.0
    repel_weight_pretrain            float  0.5
    repel_length_scale_pretrain      float  0.5
    nonempty_weight_pretrain         float  0.02
    nonempty_p_pretrain              float  2.0
    nonempty_floor_percent_pretrain  float  0.05
    kl_annealing_mode                str    'tf_sigmoid'
    kl_max_weight                    float  1.0
    kl_warmup                        int    5
    kl_end_weight                    float  0.2
    kl_cooldown                      int    5
    kl_annealing_mode_pretrain       str    'tf_sigmoid'
    kl_max_weight_pretrain           float  0.2
    kl_warmup_pretrain               int    15
    kl_end_weight_pretrain           float  0.2
    kl_cooldown_pretrain             int    10
""")

ContrastiveCfg = _record("ContrastiveCfg", """
    temperature                      float  0.1
    contrastive_similarity_function  str    'cosine'
    contrastive_loss_function        str    'nce'
    beta                             float  0.1
    tau                              float  0.1
    aug_min_shift                    int    1
    aug_max_shift                    int    6
    aug_p_shift                      float  0.8
    aug_max_rot                      int    30
    aug_n_rot                        int    4
    aug_p_rot                        float  0.0
    aug_max_interp                   int    8
    aug_min_interp                   int    3
    aug_p_interp                     float  0.3
    aug_noise_sigma                  float  0.03
    aug_p_noise                      float  0.0
""")


def cfg_lines(title: str, cfg) -> list:
    if cfg is None:
        return []
    return [f"[{title}]"] + [f"{k}: {v}" for k, v in asdict(cfg).items()] + [""]


_MODELS = ("vade", "vqvae", "contrastive")
_ENCODERS = ("recurrent", "tcn", "transformer")
_ANNEAL = ("linear", "sigmoid", "tf_sigmoid")
_SIMS = ("cosine", "dot", "euclidean", "edit")
_LOSSES = ("nce", "dcl", "dlc", "fc", "hard_dcl")  # the reference accepts the "dlc" typo; both spellings work here (Q13)


SUPPORTED_LATENT_DIMS = (4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 32)
# the TCN / transformer families stop at 16 (row-per-window latent kernels); 7, 9, 14 (round 6) are built behind the recurrent
# blocks only
RECURRENT_ONLY_LATENT_DIMS = (7, 9, 14, 20, 24, 32)
MAX_CONTRASTIVE_NODES = 64
TRANSFORMER_KEY_DIMS = tuple(range(4, 68, 4))   # every value of min(64, 3 N) // 4 * 4 (models_new.py:1013-1019)


def check_model_inputs(preprocessed_object, adjacency_matrix, meta_info, encoder_type, batch_size, latent_dim, epochs,
                       output_path, model_name, kl_annealing_mode, contrastive_similarity_function,
                       contrastive_loss_function, pretrained):
    """AssertionError on invalid trainer inputs (model_utils_new.py:377-449)."""
    if not pretrained:
        assert isinstance(preprocessed_object, (tuple, list)) and len(preprocessed_object) == 2, \
            "preprocessed_object must be a (train_dict, val_dict) tuple"
        assert adjacency_matrix is not None and getattr(adjacency_matrix, "ndim", 0) == 2 and \
            adjacency_matrix.shape[0] == adjacency_matrix.shape[1], "adjacency_matrix must be a square 2-D array"
    assert str(model_name).lower() in _MODELS, f"model_name must be one of {_MODELS}"
    assert str(encoder_type).lower() in _ENCODERS, f"encoder_type must be one of {_ENCODERS}"
    assert str(kl_annealing_mode).lower() in _ANNEAL, f"kl_annealing_mode must be one of {_ANNEAL}"
    assert str(contrastive_similarity_function).lower() in _SIMS, f"similarity must be one of {_SIMS}"
    assert str(contrastive_loss_function).lower() in _LOSSES, f"contrastive loss must be one of {_LOSSES}"
    for name, v in (("batch_size", batch_size), ("latent_dim", latent_dim), ("epochs", epochs)):
        assert pretrained or (isinstance(v, (int,)) and v > 0), f"{name} must be a positive integer"
    assert pretrained or isinstance(output_path, str), "output_path must be a string"
    if not pretrained:
        # limits of this build's kernels, reported before any data is uploaded (the reference accepts any value)
        if int(latent_dim) not in SUPPORTED_LATENT_DIMS:
            raise NotImplementedError(f"latent_dim={latent_dim}: libdeepof_hip has plans for latent_dim in "
                                      f"{SUPPORTED_LATENT_DIMS}")
        if int(latent_dim) in RECURRENT_ONLY_LATENT_DIMS and str(encoder_type).lower() != "recurrent":
            raise NotImplementedError(f"latent_dim={latent_dim} with encoder_type={encoder_type!r}: this size is built for "
                                      f"the recurrent encoder only")
        if str(encoder_type).lower() == "transformer" and str(model_name).lower() != "contrastive":
            # the reference's transformer decoder: 8 heads over d_model = 4 * latent_dim (models_new.py:1272-1277, 1494)
            if (4 * int(latent_dim)) % 8 != 0:   # (an explicit raise: a bare assert disappears under python -O)
                raise AssertionError("d_model must be divisible by num_heads")
        if str(encoder_type).lower() == "transformer":
            # TFMEncoderPT's key_dim = min(64, 3 N) rounded down to a multiple of its 4 heads (models_new.py:1013-1019)
            kd = max(4, min(64, 3 * int(adjacency_matrix.shape[0])) // 4 * 4)
            if kd not in TRANSFORMER_KEY_DIMS:
                raise NotImplementedError(f"transformer encoder: {adjacency_matrix.shape[0]} nodes give key_dim {kd}; "
                                          f"libdeepof_hip has kernels for key_dim in {TRANSFORMER_KEY_DIMS}")
        if str(model_name).lower() == "contrastive" and adjacency_matrix.shape[0] > MAX_CONTRASTIVE_NODES:
            raise NotImplementedError(f"contrastive model: {adjacency_matrix.shape[0]} nodes, the view kernel "
                                      f"handles at most {MAX_CONTRASTIVE_NODES}")
