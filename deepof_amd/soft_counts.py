"""Post-hoc soft counts for models without a cluster head.

A contrastive model is an encoder only; the reference turns its embeddings into soft cluster assignments after the
fact (``embedding_per_video`` model_utils_new.py:677-733 -> post_hoc.py):

* ``get_contrastive_soft_counts_gmm`` (post_hoc.py:1028-1172): per gate and per gate bin, a full-covariance Gaussian
  mixture (``N_clusters_per_gate`` components, ``reg_covar`` 1e-5, k-means init, seed ``random_state + 17 b + 3 g``) is
  fitted on up to ``sample_size`` reservoir-sampled embeddings of that bin; every window's responsibilities fill its
  bin's block of a ``(n, M_gates * N_clusters_per_gate)`` matrix initialised to 1e-4, which is smoothed over time
  (moving average, re-normalised) and row-normalised.  A "gate" is a per-window scalar series (e.g. the distance between
  two animals, binned by quantile edges) or a categorical behaviour series (bin = value).
* ``get_contrastive_soft_counts_msm_pcca`` (:1474-1594): k-means microstates -> Markov state model -> PCCA+ macrostate
  memberships.  It is built on the ``deeptime`` package, which this image does not have; ``method="msm"`` /
  ``"combined"`` raise and name that.

Without gating information (single animal: the reference then keeps one gate with one bin, post_hoc.py:995-996) the
decoder is a plain mixture over all windows.  This is host-side statistics on the trainer's outputs (scikit-learn, like the
reference); it sits behind the hot path, not on it.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np


def reservoir_rows(segments: List[np.ndarray], n: int, seed: int = 0) -> np.ndarray:
    """At most n rows of the stacked segments; beyond n the classic reservoir replacement with
    numpy.random.default_rng(seed).integers(0, seen + 1) per row, rows visited in order (post_hoc.py:757-781)."""
    total = sum(s.shape[0] for s in segments)
    if total <= n:
        return np.concatenate(segments, axis=0)
    rng = np.random.default_rng(seed)
    buf = np.empty((n, segments[0].shape[1]), dtype=np.float32)
    stacked = np.concatenate(segments, axis=0)
    buf[:] = stacked[:n]
    # the draw for row `seen` (seen >= n) is integers(0, seen + 1): one generator call per row, same sequence
    for seen in range(n, total):
        j = int(rng.integers(0, seen + 1))
        if j < n:
            buf[j] = stacked[seen]
    return buf


def temporal_smooth(P: np.ndarray, win: Optional[int]) -> np.ndarray:
    """Uniform moving average over time (edge-replicated), rows re-normalised (post_hoc.py:610-618)."""
    from scipy.ndimage import uniform_filter1d
    if win is None or win <= 1 or P.shape[0] < win:
        return P
    out = uniform_filter1d(P.astype(np.float32), size=win, axis=0, mode="nearest")
    return out / np.maximum(out.sum(axis=1, keepdims=True), 1e-12)


def gate_edges_from_series(keys: Sequence[str], gating_series: Dict[str, Dict[Any, np.ndarray]], gates: Sequence,
                           M_gates: int) -> Dict[Any, np.ndarray]:
    """Quantile bin edges of every gate over all videos, open at both ends (compute_gate_edges, post_hoc.py:646-704)."""
    edges = {}
    qs = np.linspace(0.0, 1.0, int(M_gates) + 1)
    for gate in gates:
        full = np.concatenate([np.asarray(gating_series[k][gate], dtype=np.float64) for k in keys])
        e = np.nanquantile(full, qs).astype(np.float64)
        e[0], e[-1] = -np.inf, np.inf
        edges[gate] = e
    return edges


def gate_masks(keys: Sequence[str], lengths: Dict[str, int], gating_series: Dict[str, Dict[Any, np.ndarray]], gates: Sequence,
               M_gates: int, categorical: bool, gate_edges: Optional[Dict[Any, np.ndarray]]) -> Dict[Any, Dict[int, Dict[str, np.ndarray]]]:
    """masks[gate][bin][key]: windows of `key` whose gate value lies in the bin -- `value == b` for categorical
    (behaviour) gates, `edges[b] < value <= edges[b+1]` otherwise (post_hoc.py:707-754)."""
    out: Dict[Any, Dict[int, Dict[str, np.ndarray]]] = {}
    for gate in gates:
        out[gate] = {}
        for b in range(M_gates):
            out[gate][b] = {}
            for key in keys:
                g = np.asarray(gating_series[key][gate])[: lengths[key]]
                if categorical:
                    out[gate][b][key] = g == b
                else:
                    e = np.asarray(gate_edges[gate], dtype=np.float64)
                    if len(e) != M_gates + 1:
                        raise ValueError(f"gate_edges[{gate!r}] must have length {M_gates + 1}, got {len(e)}")
                    out[gate][b][key] = (g > e[b]) & (g <= e[b + 1])
    return out


def contrastive_soft_counts_gmm(embeddings: Dict[str, np.ndarray], *, gating_series: Optional[Dict[str, Dict[Any, np.ndarray]]] = None,
                                categorical_gates: bool = False, n_clusters_per_gate: int = 8, M_gates: int = 3,
                                gate_edges: Optional[Dict[Any, np.ndarray]] = None, reg_covar: float = 1e-5,
                                sample_size: int = 200000, random_state: int = 0,
                                temporal_smooth_win: Optional[int] = 3) -> Dict[Any, Dict[str, np.ndarray]]:
    """Gated Gaussian-mixture decoder (post_hoc.py:1028-1172) -> {gate: {video key: (n, M_gates * n_clusters) float32}}."""
    from sklearn.mixture import GaussianMixture
    keys = list(embeddings.keys())
    Z = {k: np.asarray(embeddings[k], dtype=np.float32) for k in keys}
    lengths = {k: Z[k].shape[0] for k in keys}
    C, M = int(n_clusters_per_gate), int(M_gates)
    if gating_series is None:  # a single animal: the reference keeps one gate with ONE bin (post_hoc.py:995-996)
        gating_series, M = {k: {"": np.zeros(lengths[k])} for k in keys}, 1
    gates = list(gating_series[keys[0]].keys())
    if not categorical_gates and gate_edges is None:
        gate_edges = gate_edges_from_series(keys, gating_series, gates, M)
    masks = gate_masks(keys, lengths, gating_series, gates, M, categorical_gates, gate_edges)
    models: Dict[Any, List] = {}
    for gi, gate in enumerate(gates):
        models[gate] = []
        for b in range(M):
            seed = int(random_state + 17 * b + 3 * gi)
            segs = [Z[k][np.flatnonzero(masks[gate][b][k])] for k in keys]
            segs = [s for s in segs if s.shape[0] > 0]
            if sum(s.shape[0] for s in segs) < max(10, C):
                models[gate].append(None)
                continue
            fit_on = reservoir_rows(segs, int(sample_size), seed=seed)
            models[gate].append(GaussianMixture(n_components=C, covariance_type="full", reg_covar=float(reg_covar),
                                                random_state=seed, init_params="kmeans", max_iter=200, tol=1e-3).fit(fit_on))
    out: Dict[Any, Dict[str, np.ndarray]] = {gate: {} for gate in gates}
    for key in keys:
        for gate in gates:
            P = np.full((lengths[key], M * C), 1e-4, dtype=np.float32)
            for b in range(M):
                mask = masks[gate][b][key]
                block = slice(b * C, (b + 1) * C)
                if models[gate][b] is None:
                    if np.any(mask):
                        P[mask, block] = 1.0 / C
                    continue
                idx = np.flatnonzero(mask)
                if idx.size:
                    P[idx, block] = models[gate][b].predict_proba(Z[key][idx]).astype(np.float32, copy=False)
            if temporal_smooth_win and temporal_smooth_win > 1:
                P = temporal_smooth(P, temporal_smooth_win)
            out[gate][key] = P / np.maximum(P.sum(axis=1, keepdims=True), 1e-12)
    return out


def contrastive_soft_counts(embeddings: Dict[str, np.ndarray], method: str = "gmm", n_clusters_per_gate: int = 8, M_gates: int = 3,
                            gating_series: Optional[dict] = None, gate: Any = None, **kw) -> Dict[str, np.ndarray]:
    """The soft-count table dict ``embedding_per_video`` returns for `softcounts_extraction_method` (model_utils_new.py:677-733):
    the decoder's output for one gate (default: the first; a single animal has exactly one)."""
    if method in ("msm", "combined"):
        raise NotImplementedError(
            f"softcounts_extraction_method={method!r} is the reference's MSM-PCCA decoder (post_hoc.py:1474-1594), which is "
            "built on the `deeptime` package; it is not installed here and is not re-implemented. Use method='gmm'.")
    if method != "gmm":
        raise ValueError('For "softcounts_extraction_method" only "gmm", "msm" or "combined" are supported!')
    # embedding_per_video calls the GMM decoder with temporal_smooth_win=3 (model_utils_new.py:689)
    by_gate = contrastive_soft_counts_gmm(embeddings, gating_series=gating_series, n_clusters_per_gate=n_clusters_per_gate,
                                          M_gates=M_gates, temporal_smooth_win=kw.pop("temporal_smooth_win", 3), **kw)
    gates = list(by_gate.keys())
    return by_gate[gates[0] if gate is None else gate]
