"""Post-hoc soft counts for models without a cluster head.

A contrastive model is an encoder only; the reference turns its embeddings into soft cluster assignments after the
fact (``embedding_per_video`` model_utils_new.py:677-733 -> post_hoc.py):

* ``get_contrastive_soft_counts_gmm`` (post_hoc.py:1028-1172): per gate and per gate bin, a full-covariance Gaussian
  mixture (``N_clusters_per_gate`` components, ``reg_covar`` 1e-5, k-means init, seed ``random_state + 17 b + 3 g``) is
  fitted on up to ``sample_size`` reservoir-sampled embeddings of that bin; every window's responsibilities fill its
  bin's block of a ``(n, M_gates * N_clusters_per_gate)`` matrix initialised to 1e-4, which is smoothed over time
  (moving average, re-normalised) and row-normalised.  A "gate" is a per-window scalar series (e.g. the distance between
  two animals, binned by quantile edges) or a categorical behaviour series (bin = value).
* ``get_contrastive_soft_counts_msm_pcca`` (:1474-1594): per gate bin, MiniBatchKMeans microstates on the standardised
  embeddings of the bin's runs, a Markov state model at ``lagtime`` over the runs' microstate trajectories, PCCA+
  macrostate memberships, and every window takes the membership row of its microstate.  The reference's orchestration
  (:1175-1365) is restated here and pinned by ``posthoc_msm.npz``; the three ``deeptime`` calls inside it are restated
  from their published algorithms in ``deepof_amd.msm_pcca`` (deeptime is absent from this image: that core is
  parity-unpinned and says so).
* ``method="combined"`` (model_utils_new.py:709-729): the MSM decoder plus a behaviour-gated GMM on quality-based "chaos"
  labels (``get_supervised_chaos`` :375-443, ``add_chaos_gates`` :446-555).

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


def mask_to_runs(mask: np.ndarray, min_len: int = 2) -> List[tuple]:
    """[(start, end)) of the runs of True of at least ``min_len`` (post_hoc.py:582-598)."""
    idx = np.flatnonzero(np.asarray(mask, dtype=bool))
    if idx.size == 0:
        return []
    cut = np.flatnonzero(np.diff(idx) > 1)
    starts = np.concatenate(([idx[0]], idx[cut + 1]))
    ends = np.concatenate((idx[cut] + 1, [idx[-1] + 1]))
    return [(int(s), int(e)) for s, e in zip(starts, ends) if e - s >= min_len]


def contrastive_soft_counts_msm_pcca(embeddings: Dict[str, np.ndarray], *, gating_series: Optional[Dict[str, Dict[Any, np.ndarray]]] = None,
                                     categorical_gates: bool = False, n_clusters_per_gate: int = 10, M_gates: int = 3,
                                     gate_edges: Optional[Dict[Any, np.ndarray]] = None, sample_size: int = 200000,
                                     random_state: int = 0, temporal_smooth_win: Optional[int] = 3, n_micro: int = 400,
                                     min_micro_per_macro: int = 3, lagtime: int = 3) -> Dict[Any, Dict[str, np.ndarray]]:
    """Gated MSM + PCCA+ decoder (post_hoc.py:1365-1594) -> {gate: {video key: (n, M_gates * n_clusters) float32}}."""
    from sklearn.cluster import MiniBatchKMeans
    from sklearn.preprocessing import StandardScaler

    from .msm_pcca import fit_pcca_memberships
    keys = list(embeddings.keys())
    Z = {k: np.asarray(embeddings[k], dtype=np.float32) for k in keys}
    lengths = {k: Z[k].shape[0] for k in keys}
    C, M = int(n_clusters_per_gate), int(M_gates)
    if gating_series is None:
        gating_series, M = {k: {"": np.zeros(lengths[k])} for k in keys}, 1
    gates = list(gating_series[keys[0]].keys())
    if not categorical_gates and gate_edges is None:
        gate_edges = gate_edges_from_series(keys, gating_series, gates, M)
    masks = gate_masks(keys, lengths, gating_series, gates, M, categorical_gates, gate_edges)
    models: Dict[Any, List] = {}
    for gi, gate in enumerate(gates):
        models[gate] = []
        for b in range(M):
            seed = int(random_state + 1000 * gi + 17 * b)
            spatial, temporal, n_windows = [], [], 0
            for key in keys:                                      # _collect_segments_for_gate_bin
                mask = masks[gate][b][key]
                n_windows += int(mask.sum())
                for s0, e0 in mask_to_runs(mask, min_len=2):
                    seg = Z[key][s0:e0]
                    spatial.append(seg)
                    if seg.shape[0] >= lagtime + 2:
                        temporal.append(seg)
            if not spatial or n_windows < max(50, 5 * C):
                models[gate].append(None)
                continue
            scaler = StandardScaler()                             # _fit_microstates_kmeans
            X_fit = scaler.fit_transform(reservoir_rows(spatial, int(sample_size), seed=seed))
            n_micro_eff = max(int(min(n_micro, max(min_micro_per_macro * C, X_fit.shape[0] // 50))), 2)
            kmeans = MiniBatchKMeans(n_clusters=n_micro_eff, batch_size=4096, max_iter=200, random_state=seed,
                                     init="k-means++", n_init="auto").fit(X_fit)
            if not temporal:
                models[gate].append(None)
                continue
            dtrajs = [np.asarray(kmeans.predict(scaler.transform(seg)), dtype=np.int32) for seg in temporal]
            try:
                active, chi = fit_pcca_memberships(dtrajs, lagtime, C)
            except Exception:  # noqa: BLE001  (the reference swallows estimator failures the same way, :1444-1451)
                active, chi = None, None
            if active is None or chi is None:
                models[gate].append(None)
                continue
            m2m = np.full((n_micro_eff, C), 1.0 / C, dtype=np.float32)   # inactive microstates: uniform (_build_micro2macro)
            for i in range(active.shape[0]):
                if 0 <= int(active[i]) < n_micro_eff:
                    m2m[int(active[i])] = chi[i]
            models[gate].append({"scaler": scaler, "kmeans": kmeans, "micro2macro": m2m})
    out: Dict[Any, Dict[str, np.ndarray]] = {gate: {} for gate in gates}
    for key in keys:
        for gate in gates:
            P = np.full((lengths[key], M * C), 1e-4, dtype=np.float32)
            for b in range(M):
                model, mask = models[gate][b], masks[gate][b][key]
                block = slice(b * C, (b + 1) * C)
                if model is None:
                    if np.any(mask):
                        P[mask, block] = 1.0 / C
                    continue
                for s0, e0 in mask_to_runs(mask, min_len=1):
                    d = np.asarray(model["kmeans"].predict(model["scaler"].transform(Z[key][s0:e0])), dtype=np.int32)
                    P[s0:e0, block] = model["micro2macro"][d]
            if temporal_smooth_win and temporal_smooth_win > 1:
                P = temporal_smooth(P, temporal_smooth_win)
            out[gate][key] = P / np.maximum(P.sum(axis=1, keepdims=True), 1e-12)
    return out


def supervised_chaos(quality: Dict[str, np.ndarray], quality_columns: Sequence, animal_ids: Sequence[str] = ("",),
                     quality_threshold: float = 0.75, frac_bps_below: float = 0.5) -> Dict[str, Dict[str, np.ndarray]]:
    """get_supervised_chaos (post_hoc.py:375-443): per video {"<animal>_chaos" ...: (frames,) 0/1, "anychaos"}: a frame is
    chaotic for an animal when at least ``frac_bps_below`` of its body parts have a tracking quality below
    ``quality_threshold`` (or not finite).  ``quality``: {video: (frames, len(quality_columns)) likelihoods}."""
    ids = [""] if (animal_ids is None or list(animal_ids) in ([], [""])) else \
        ([a + "_" for a in animal_ids] if len(animal_ids) > 1 else list(animal_ids))
    out = {}
    for key, q in quality.items():
        q = np.asarray(q, dtype=np.float32)
        tab, per_animal = {}, []
        for mid in ids:
            cols = [i for i, c in enumerate(quality_columns) if str(c).startswith(f"{mid}")]
            if not cols:
                raise ValueError("Provided animal_id is not in quality table!")
            arr = q[:, cols]
            bad = (~np.isfinite(arr)) | (arr < float(quality_threshold))
            chaos = (bad.mean(axis=1) >= float(frac_bps_below)).astype(np.float32)
            tab[f"{mid}chaos"] = chaos
            per_animal.append(chaos.astype(bool))
        tab["anychaos"] = np.logical_or.reduce(per_animal).astype(np.float32)
        out[key] = tab
    return out


def add_chaos_gates(soft_counts: Dict[Any, Dict[str, np.ndarray]], soft_counts_chaos: Dict[str, np.ndarray],
                    chaos: Dict[str, Dict[str, np.ndarray]], window_size: int) -> Dict[Any, Dict[str, np.ndarray]]:
    """add_chaos_gates (post_hoc.py:446-555): a window is chaotic when any of its ``window_size`` frames is; the regular
    states are zeroed on chaotic windows, the chaos decoder's states on the others, and the chaotic half of the chaos
    decoder's columns is appended."""
    out = {}
    for gate, by_key in soft_counts.items():
        out[gate] = {}
        for key, sc1 in by_key.items():
            sc1 = np.array(sc1, copy=True)
            sc2 = np.array(soft_counts_chaos[key], copy=True)
            n = sc1.shape[0]
            raw = np.asarray(chaos[key]["anychaos"], dtype=np.float32)[: n + window_size - 1]
            if sc2.shape[0] != n or raw.shape[0] < n:
                raise ValueError(f"Soft_counts and soft_counts_chaos must have same length, annotations must have same "
                                 f"lenght or longer (Error at key{key!r}): {sc1.shape[0]} vs {sc2.shape[0]} vs {raw.shape[0]}")
            win = np.convolve(raw, np.ones(window_size, dtype=np.float32), mode="valid") > 0
            if win.shape[0] != n:
                raise ValueError(f"Convolved length mismatch for key {key!r}: {win.shape[0]} vs expected {n}")
            sc1[win] = 0
            sc2[~win] = 0
            if sc2.shape[1] % 2 != 0:
                raise ValueError(f"Chaos soft counts for key {key!r} have an odd number of columns ({sc2.shape[1]})")
            out[gate][key] = np.concatenate([sc1, sc2[:, sc2.shape[1] // 2:]], axis=1)
    return out


def contrastive_soft_counts(embeddings: Dict[str, np.ndarray], method: str = "gmm", n_clusters_per_gate: int = 8, M_gates: int = 3,
                            gating_series: Optional[dict] = None, gate: Any = None, **kw) -> Dict[str, np.ndarray]:
    """The soft-count table dict ``embedding_per_video`` returns for `softcounts_extraction_method` (model_utils_new.py:677-733):
    the decoder's output for one gate (default: the first; a single animal has exactly one)."""
    if method in ("msm", "combined"):
        # embedding_per_video calls the MSM decoder with temporal_smooth_win=1, n_micro=400, lagtime=3
        # (model_utils_new.py:694-708)
        quality = kw.pop("quality", None)
        quality_columns = kw.pop("quality_columns", None)
        animal_ids = kw.pop("animal_ids", ("",))
        window_size = kw.pop("window_size", None)
        q_thr, q_frac = kw.pop("quality_threshold", 0.75), kw.pop("frac_bps_below", 0.5)
        by_gate = contrastive_soft_counts_msm_pcca(embeddings, gating_series=gating_series,
                                                   n_clusters_per_gate=n_clusters_per_gate, M_gates=M_gates,
                                                   temporal_smooth_win=kw.pop("temporal_smooth_win", 1), **kw)
        if method == "combined":
            if quality is None or quality_columns is None or window_size is None:
                raise ValueError('method="combined" needs the tracking-quality tables: quality={video: (frames, parts)}, '
                                 "quality_columns=[...], window_size=<window length>")
            chaos = supervised_chaos(quality, quality_columns, animal_ids, q_thr, q_frac)
            lengths = {k: np.asarray(v).shape[0] for k, v in embeddings.items()}
            # window-level behaviour gate: the reference's gating front end turns the frame series into one value per
            # window start (post_hoc.py:_preprocess_gates with supervised annotations); here: the label of the window's
            # first frame, as for every other behaviour gate of this module
            series = {k: {"behavior_combinations": np.asarray(chaos[k]["anychaos"][: lengths[k]], dtype=np.int64)} for k in lengths}
            chaos_counts = contrastive_soft_counts_gmm(embeddings, gating_series=series, categorical_gates=True,
                                                       n_clusters_per_gate=n_clusters_per_gate, M_gates=2,
                                                       temporal_smooth_win=1)["behavior_combinations"]
            by_gate = add_chaos_gates(by_gate, chaos_counts, chaos, int(window_size))
        gates = list(by_gate.keys())
        return by_gate[gates[0] if gate is None else gate]
    if method != "gmm":
        raise ValueError('For "softcounts_extraction_method" only "gmm", "msm" or "combined" are supported!')
    # embedding_per_video calls the GMM decoder with temporal_smooth_win=3 (model_utils_new.py:689)
    by_gate = contrastive_soft_counts_gmm(embeddings, gating_series=gating_series, n_clusters_per_gate=n_clusters_per_gate,
                                          M_gates=M_gates, temporal_smooth_win=kw.pop("temporal_smooth_win", 3), **kw)
    gates = list(by_gate.keys())
    return by_gate[gates[0] if gate is None else gate]
