"""The project-level entry above the trainer: ``Coordinates.deep_unsupervised_embedding``
(/root/reference/deepof/data.py:3247-3404) as a plain function.

The reference method belongs to the pandas-side ``Coordinates`` object; what it does for the hot path is a keyword
mapping onto ``train_deepof_model`` -- reproduced here, quirks included:

* ``save_weights`` of the trainer is fed from **``save_checkpoints``** (default False), not from the caller's
  ``save_weights`` (data.py:3386): by default nothing is written and both returned models hold the last
  weights;
* ``output_path`` becomes ``<project>/<output_path>/Trained_models``, ``data_path`` ``<project>/Tables``, a relative
  ``pretrained`` name is looked up under ``<project>/Trained_models/models``;
* ``embedding_model`` -> ``model_name``; the binning arguments only feed ``_preprocess_time_bins``, whose result the
  trainer never sees (data.py:3350-3352) -- accepted and ignored here;
* extra keyword arguments go straight through to ``train_deepof_model`` (meta_info, teacher options, ...).

INTEGRATION.md shows the two-line change that makes the reference's method call this function.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np

from .training import train_deepof_model


def deep_unsupervised_embedding(
    preprocessed_object: Tuple,
    adjacency_matrix: np.ndarray = None,
    # binning info (vestigial for the trainer)
    bin_size=None, bin_index=None, precomputed_bins=None, samples_max=None,
    # model info
    embedding_model: str = "VaDE", encoder_type: str = "recurrent", batch_size: int = 64, latent_dim: int = 4,
    epochs: int = 150, log_history: bool = True, log_hparams: bool = False, n_clusters: int = 10,
    kmeans_loss: float = 0.0, temperature: float = 0.1, contrastive_similarity_function: str = "cosine",
    contrastive_loss_function: str = "nce", beta: float = 0.1, tau: float = 0.1, output_path: str = "",
    pretrained: str = False, save_checkpoints: bool = False, save_weights: bool = True, input_type: str = False,
    run: int = 0, kl_annealing_mode: str = "linear", kl_warmup: int = 15, reg_cat_clusters: float = 0.0,
    recluster: bool = False, interaction_regularization: float = 0.0, bootstrap_training: bool = False,
    bootstrap_block_len: int = 250, random_seed: int = 0,
    project_dir: str = ".",
    **kwargs,
):
    """-> (model_val, model_score, model_teacher_init_or_None, log_summary).  ``project_dir`` stands for the
    reference's ``os.path.join(self._project_path, self._project_name)``; every other argument is the reference's."""
    del bin_size, bin_index, precomputed_bins, samples_max, log_hparams, save_weights, input_type  # not trainer inputs
    if pretrained:
        pretrained = os.path.join(project_dir, "Trained_models", "models", pretrained)
    try:
        return train_deepof_model(
            preprocessed_object=preprocessed_object, adjacency_matrix=adjacency_matrix, model_name=embedding_model,
            encoder_type=encoder_type, batch_size=batch_size, latent_dim=latent_dim, epochs=epochs,
            log_history=log_history, n_clusters=n_clusters, kmeans_loss=kmeans_loss, temperature=temperature,
            contrastive_similarity_function=contrastive_similarity_function,
            contrastive_loss_function=contrastive_loss_function, beta=beta, tau=tau,
            output_path=os.path.join(project_dir, output_path, "Trained_models"),
            data_path=os.path.join(project_dir, "Tables"), pretrained=pretrained or None,
            save_weights=save_checkpoints, run=run, kl_annealing_mode=kl_annealing_mode, kl_warmup=kl_warmup,
            reg_cat_clusters=reg_cat_clusters, recluster=recluster,
            interaction_regularization=interaction_regularization, bootstrap_training=bootstrap_training,
            bootstrap_block_len=bootstrap_block_len, random_seed=random_seed, **kwargs)
    except IndexError:
        raise ValueError("No pretrained model found for the given parameters. Please train a model first.")
