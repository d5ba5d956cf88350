"""``python -m deepof_amd.cli`` -- the argparse surface of the reference's ``deepof_train_embeddings``
(/root/reference/deepof/deepof_train_embeddings.py:26-223: same 26 flags, short forms and defaults) in front of
``deepof_amd``'s trainer.

The reference script is stale at v0.9.0 (it imports a module that no longer exists and references undefined names,
SURVEY F3 / Q21); only its flags and its call sequence (:360-377, 405-412) are a contract:

    tables -> graph dataset (window_size / window_step / val_num held-out videos, scale="standard")
           -> deep_unsupervised_embedding(batch_size, latent_dim=encoding_size, embedding_model, encoder_type,
              n_clusters=n_components, output_path, save_checkpoints=False, save_weights=True, input_type,
              kmeans_loss, reg_cat_clusters=cat_kl_loss, epochs=max_epochs, run)
           -> embedding_per_video -> embeddings / soft counts written next to the models.

What this entry does NOT do is the reference's ETL (``Project.create``: video / DLC ingestion, arena detection,
smoothing -- out of scope): ``--train-path`` points at the merged feature tables that ETL produces,
as a pickle ``{"tables": {video: DataFrame | (frames, C) array}, "columns": [...]}`` (or just ``{video: DataFrame}``).
The ETL-side flags (--arena-dims, --smooth-alpha, --exclude-bodyparts, --automatic-changepoints, --load-project,
--exp-condition-path, --animal-to-preprocess) are accepted for command-line compatibility and reported as unused;
hyper-parameter tuning (--hyperparameter-tuning, optuna) is not part of this build and raises.
"""
from __future__ import annotations

import argparse
import os
import pickle
import sys

import numpy as np


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Autoencoder training for DeepOF animal pose recognition")
    p.add_argument("--animal-ids", "-ids", type=str, default="",
                   help="Id of the animals in the loaded dataset to use. Empty string by default")
    p.add_argument("--animal-to-preprocess", "-idprep", type=str, default=None,
                   help="Id of the animal to preprocess if multiple animals are being tracked. None by default, which "
                        "results in all animals being processed.")
    p.add_argument("--arena-dims", "-adim", type=int, default=380,
                   help="diameter in mm of the utilised arena. Used for scaling purposes")
    p.add_argument("--automatic-changepoints", "-ruptures", choices=["False", "linear", "rbf"], nargs="?", default="False",
                   help="Algorithm to use to rupture the time series; False = sliding windows")
    p.add_argument("--batch-size", "-bs", type=int, default=128, help="set training batch size")
    p.add_argument("--n-components", "-k", type=int, default=15,
                   help="set the number of components for the unsupervised model")
    p.add_argument("--encoding-size", "-es", type=int, default=8, help="set the number of dimensions of the latent space")
    p.add_argument("--embedding-model", "-embedding", type=str, choices=["VQVAE", "VaDE", "Contrastive"], default="VQVAE",
                   help="Algorithm to use to embed and cluster the time series")
    p.add_argument("--encoder-type", "-encoder", type=str, choices=["recurrent", "TCN", "transformer"], default="recurrent",
                   help="Encoder architecture to use when embedding the time series")
    p.add_argument("--exclude-bodyparts", "-exc", type=str, default="", help="Excludes the indicated bodyparts from all analyses")
    p.add_argument("--hpt-trials", "-n", type=int, default=25, help="number of hyperparameter tuning iterations")
    p.add_argument("--hyperparameter-tuning", "-tune", choices=[False, "bayopt", "hyperband"], default=False,
                   help="hyperparameter tuning mode (not part of this build)")
    p.add_argument("--hyperparameters", "-hp", type=str, default=None, help="Path to a pickled dictionary of network hyperparameters")
    p.add_argument("--input-type", "-d", type=str, default="graph", help="Select an input type: coords or graph")
    p.add_argument("--output-path", "-o", type=str, default=".", help="Sets the base directory where to output results")
    p.add_argument("--kmeans-loss", "-kmeans", type=float, default=0.0,
                   help="If > 0, adds a regularization term controlling for correlation between latent dimensions")
    p.add_argument("--cat-kl-loss", "-catkl", type=float, default=0.0,
                   help="If > 0, adds a KL term between cluster assignment frequencies and a uniform distribution")
    p.add_argument("--smooth-alpha", "-sa", type=float, default=2, help="exponential smoothing factor of the input data (ETL)")
    p.add_argument("--train-path", "-tp", type=str, help="set training set path")
    p.add_argument("--val-num", "-vn", type=int, default=5, help="number of videos of the training set to use for validation")
    p.add_argument("--window-size", "-ws", type=int, default=25, help="sliding window size")
    p.add_argument("--window-step", "-wt", type=int, default=1, help="sliding window step")
    p.add_argument("--max-epochs", "-epochs", type=int, default=150, help="maximum number of epochs to train")
    p.add_argument("--load-project", "-load", type=str, default=None, help="(ETL) load an existing project")
    p.add_argument("--run", "-rid", type=int, default=0, help="run ID of the experiment (for naming output files only)")
    p.add_argument("--exp-condition-path", "-ec", type=str, default=None, help="(ETL) experimental condition file")
    return p


def load_tables(path: str):
    """-> ({video: (frames, C) float64 array}, columns)."""
    with open(path, "rb") as f:
        blob = pickle.load(f)
    tables = blob["tables"] if isinstance(blob, dict) and "tables" in blob else blob
    columns = blob.get("columns") if isinstance(blob, dict) and "tables" in blob else None
    out = {}
    for key, t in tables.items():
        if hasattr(t, "columns"):
            if columns is None:
                columns = list(t.columns)
            t = t.to_numpy(dtype=float)
        out[key] = np.asarray(t, dtype=np.float64)
    if columns is None:
        raise ValueError("the table file carries no column labels (DataFrames, or a 'columns' entry)")
    return out, [tuple(c) if isinstance(c, (list, tuple)) else c for c in columns]


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.train_path:
        raise ValueError("Set a valid data path for the training to run")
    assert args.input_type in ["coords", "graph"], "Invalid input type. Type python model_training.py -h for help."
    if args.input_type != "graph":
        raise NotImplementedError("input_type='coords' (use_gnn=False) is not built: the trainer always runs the graph models")
    if args.hyperparameter_tuning:
        # the reference's own branch calls deepof.model_utils.tune_search, a module its tree no longer has
        # (deepof_train_embeddings.py:429); the per-epoch trial hooks it would use are in training.fit_* (trial=...)
        raise NotImplementedError("the tuning driver is not part of this build (the reference's calls a module it no longer ships); "
                                  "pass an optuna trial to deepof_amd.training.fit_VADE / fit_VQVAE / fit_contrastive instead")
    unused = {k: getattr(args, k) for k in ("arena_dims", "smooth_alpha", "exclude_bodyparts", "automatic_changepoints",
                                             "load_project", "exp_condition_path", "animal_to_preprocess", "hyperparameters")}
    print("ETL-side options accepted but not used here:", unused)
    from .api import deep_unsupervised_embedding
    from .inference import embedding_per_video
    from .preprocess import graph_dataset_from_tables

    tables, columns = load_tables(os.path.abspath(args.train_path))
    animal_ids = [a for a in args.animal_ids.split(",")] if args.animal_ids else [""]
    keys = sorted(tables)
    test_keys = keys[len(keys) - min(args.val_num, max(len(keys) - 1, 0)):] if len(keys) > 1 else []
    (train, val), meta, adjacency, pre = graph_dataset_from_tables(
        tables, columns, animal_ids, window_size=args.window_size, window_step=args.window_step, test_keys=test_keys)
    print("Training windows:", len(train), train.x_shape, train.a_shape, "| validation windows:", len(val))
    trained = deep_unsupervised_embedding(
        (train, val), adjacency_matrix=adjacency, batch_size=args.batch_size, latent_dim=args.encoding_size,
        embedding_model=args.embedding_model, encoder_type=args.encoder_type, n_clusters=args.n_components,
        output_path=args.output_path, save_checkpoints=False, save_weights=True, input_type=args.input_type,
        kmeans_loss=float(args.kmeans_loss), reg_cat_clusters=float(args.cat_kl_loss), epochs=args.max_epochs, run=args.run,
        meta_info=meta, project_dir=".")
    model = trained[0]
    embeddings, soft_counts = embedding_per_video(pre, model)
    out_dir = os.path.join(args.output_path, "Trained_models")
    os.makedirs(out_dir, exist_ok=True)
    tag = f"{args.embedding_model}_{args.encoder_type}_encoding={args.encoding_size}_k={args.n_components}_run={args.run}"
    with open(os.path.join(out_dir, f"deepof_unsupervised_{tag}_embeddings.pkl"), "wb") as f:
        pickle.dump(embeddings, f)
    with open(os.path.join(out_dir, f"deepof_unsupervised_{tag}_soft_counts.pkl"), "wb") as f:
        pickle.dump(soft_counts, f)
    print("Done!")
    return trained, embeddings, soft_counts


if __name__ == "__main__":
    main(sys.argv[1:])
