"""Per-video inference: embeddings and soft cluster assignments of every window of every video.

Mirrors ``embedding_per_video`` (/root/reference/deepof/clustering/model_utils_new.py:452-748): for each video the
reference re-runs the table preprocessing with the training run's global scaler (``get_graph_dataset(...,
window_step=1, pretrained_scaler=global_scaler)`` :563-575), materialises all stride-1 windows, pushes them through
the model 256 at a time (:597-619) and wraps the stitched outputs into table dicts; for a contrastive model (encoder
only) the soft counts come from a post-hoc decoder over the embeddings (:677-733).

Here the frame tables of ALL videos are preprocessed in one device call (the per-video standardisation does not
couple videos and the global scaler is given, so this equals the per-video calls), stay resident, and every chunk of
windows is gathered straight from them into the model's static batch buffers; the encoder forward of a chunk is
replayed as a hipGraph.  Videos are independent units: under ``torch.distributed`` rank r takes videos r, r+world, ...
and the per-video results are all-gathered as plain arrays.

The returned dicts ``{video key: ndarray}`` hold what the reference's ``TableDict`` values hold in memory:
``embeddings[key]`` (frames - W + 1, latent_dim) and ``soft_counts[key]`` (frames - W + 1, K) (tests/test_data.py:1014).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi
from .models import Contrastive, VaDE, VQVAE
from .preprocess import PreprocessedTables, preprocess_tables
from .stepping import StepGraphs


def _video_shares(n_videos: int):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


class VideoEncoder:
    """Chunked window encoder over resident frame tables: gather -> eval forward (no decoder) -> (embedding, soft counts).

    One instance per (model, chunk size); the forward of a full chunk is captured once and replayed."""

    def __init__(self, model, chunk: int = 4096, use_graphs: Optional[bool] = None):
        if not isinstance(model, (VaDE,)):  # VQVAE and Contrastive derive from VaDE
            raise TypeError("model must be a deepof_amd VaDE / VQVAE / Contrastive")
        self.model, self.chunk = model, int(chunk)
        self.kind = "contrastive" if isinstance(model, Contrastive) else "vqvae" if isinstance(model, VQVAE) else "vade"
        self.graphs = StepGraphs(model.device, use_graphs)
        self._buf: Dict[int, dict] = {}

    def _static(self, n: int) -> dict:
        b = self._buf.get(n)
        if b is None:
            eng = self.model.engine(n)
            f32 = dict(dtype=torch.float32, device=eng.device)
            b = dict(eng=eng, x=torch.empty(n, eng.T, eng.N, 3, **f32), a=torch.empty(n, eng.T, eng.E, 1, **f32),
                     z=torch.empty(n, eng.L, **f32), q=torch.empty(n, max(eng.K, 1), **f32))
            self._buf[n] = b
        return b

    def _forward(self, b: dict):
        eng, lib = b["eng"], b["eng"].lib
        st = eng._stream()
        if self.kind == "vade":     # VaDEPT.forward in eval mode: z = z_mean, q = p(c | z)   (model_utils_new.py:611)
            rc = lib.dof_vade_forward(eng.plan, eng.params.data_ptr(), eng.prior.data_ptr(), b["x"].data_ptr(),
                                      b["a"].data_ptr(), None, b["z"].data_ptr(), b["q"].data_ptr(), None, None, None, None, st)
            _capi.check(lib, rc, "dof_vade_forward")
        elif self.kind == "vqvae":  # outputs [4] (encoder output) and [3] (soft counts) of the 6-tuple (:614)
            rc = lib.dof_vqvae_forward(eng.plan, eng.params.data_ptr(), b["x"].data_ptr(), b["a"].data_ptr(),
                                       b["z"].data_ptr(), None, b["q"].data_ptr(), None, None, None, st)
            _capi.check(lib, rc, "dof_vqvae_forward")
        else:                       # ContrastivePT.forward (:617), eval mode
            eng.contrastive_encode(b["x"], b["a"], train=False, out=b["z"], count=False)

    @torch.no_grad()
    def encode_rows(self, node_table: torch.Tensor, edge_table: torch.Tensor, first_row: int, n_windows: int,
                    row_step: int = 1) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """Windows first_row + i * row_step, i < n_windows, of the frame tables -> (embeddings (n, L), soft counts (n, K) | None)."""
        model = self.model
        was_training = model.training
        model.eval()
        L, K = model.latent_dim, model.n_components
        emb = torch.empty(n_windows, L, dtype=torch.float32, device=model.device)
        soft = None if self.kind == "contrastive" else torch.empty(n_windows, K, dtype=torch.float32, device=model.device)
        lib = model._base.lib
        for s in range(0, n_windows, self.chunk):
            n = min(self.chunk, n_windows - s)
            b = self._static(n)
            eng = b["eng"]
            rc = lib.dof_window_gather_range(node_table.data_ptr(), edge_table.data_ptr(), int(first_row + s * row_step),
                                             int(row_step), n, eng.T, eng.N, eng.E, b["x"].data_ptr(), b["a"].data_ptr(),
                                             eng._stream())
            _capi.check(lib, rc, "dof_window_gather_range")
            self.graphs.run((self.kind, n), lambda: self._forward(b))
            emb[s:s + n].copy_(b["z"])
            if soft is not None:
                soft[s:s + n].copy_(b["q"][:, :K])
        model.train(was_training)
        return emb, soft


def embedding_per_video(tables, model, meta_info: Optional[dict] = None, *, columns: Optional[Sequence] = None,
                        animal_ids=("",), global_scaler: Optional[dict] = None, scale: str = "standard",
                        samples_max: int = 227272, keys: Optional[Sequence[str]] = None, chunk: int = 4096,
                        softcounts_extraction_method: Optional[str] = None, states_per_gate: int = 8, M_gates: int = 3,
                        gating_series: Optional[dict] = None, shard_videos: bool = True, lib=None,
                        use_graphs: Optional[bool] = None) -> Tuple[Dict[str, np.ndarray], Dict[str, np.ndarray]]:
    """Embeddings and soft counts per video with a trained model (model_utils_new.py:452-748).

    ``tables``: either a ``PreprocessedTables`` (frame tables already on the device, e.g. the ``pre`` returned by
    ``graph_dataset_from_tables``) or ``{video key: (frames, C) raw merged table}`` together with ``columns``,
    ``meta_info`` (node / edge columns + the three ``*_standardize`` modes the model was trained with) and the training
    run's ``global_scaler`` -- the reference's ``pretrained_scaler`` path.  Windows: ``model.window_size`` frames,
    stride 1, never across videos.  Contrastive models: ``softcounts_extraction_method`` "gmm" (default here; the
    reference's default "msm" needs ``deeptime``, see ``deepof_amd.soft_counts``) decodes soft counts from the embeddings.
    """
    if isinstance(tables, PreprocessedTables):
        pre = tables
    else:
        if meta_info is None or columns is None:
            raise ValueError("raw tables need `columns` and `meta_info` (node_columns / edge_columns / *_standardize)")
        if global_scaler is None:
            raise ValueError("raw tables need the training run's global_scaler (pretrained_scaler path of the reference)")
        pre = preprocess_tables(tables, columns, animal_ids, meta_info["node_columns"], meta_info["edge_columns"], (),
                                scale=scale, samples_max=samples_max, dist_standardize=meta_info.get("dist_standardize", "groupwise"),
                                speed_standardize=meta_info.get("speed_standardize", "groupwise"),
                                coord_standardize=meta_info.get("coord_standardize", "groupwise"),
                                pretrained_scaler=global_scaler, device=model.device, lib=lib)
    W = int(model.window_size)
    wanted = list(pre.keys) if keys is None else [k for k in pre.keys if k in set(keys)]
    dist, rank, world = _video_shares(len(wanted)) if shard_videos else (None, 0, 1)
    enc = VideoEncoder(model, chunk, use_graphs)
    emb_out: Dict[str, np.ndarray] = {}
    soft_out: Dict[str, np.ndarray] = {}
    for vi, key in enumerate(wanted):
        if vi % world != rank:
            continue
        i = pre.keys.index(key)
        lo, hi = int(pre.video_off[i]), int(pre.video_off[i + 1])
        nw = hi - lo - W + 1
        if nw <= 0:
            continue   # shorter than one window: the reference's rolling_window yields no rows either
        emb, soft = enc.encode_rows(pre.node_table, pre.edge_table, lo, nw)
        emb_out[key] = emb.cpu().numpy()
        if soft is not None:
            soft_out[key] = soft.cpu().numpy()
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, (emb_out, soft_out))
        emb_out = {k: v for part in gathered for k, v in part[0].items()}
        soft_out = {k: v for part in gathered for k, v in part[1].items()}
        emb_out = {k: emb_out[k] for k in wanted if k in emb_out}
        soft_out = {k: soft_out[k] for k in wanted if k in soft_out}
    if isinstance(model, Contrastive):
        from .soft_counts import contrastive_soft_counts
        method = softcounts_extraction_method or "gmm"
        soft_out = contrastive_soft_counts(emb_out, method=method, n_clusters_per_gate=states_per_gate, M_gates=M_gates,
                                           gating_series=gating_series)
    elif softcounts_extraction_method is not None:
        from .soft_counts import contrastive_soft_counts
        soft_out = contrastive_soft_counts(emb_out, method=softcounts_extraction_method, n_clusters_per_gate=states_per_gate,
                                           M_gates=M_gates, gating_series=gating_series)
    return emb_out, soft_out
