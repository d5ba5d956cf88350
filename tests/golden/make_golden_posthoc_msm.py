"""Golden for the MSM-PCCA soft-count decoder and the chaos gates (the part round 2 left out).

The REFERENCE's own functions -- get_contrastive_soft_counts_msm_pcca, _fit_msmpcca_models, _collect_segments_for_gate_bin,
_fit_microstates_kmeans, _segments_to_dtrajs, _fit_pcca_memberships, _get_active_state_symbols, _pcca_memberships,
_build_micro2macro, _mask_to_runs (+ the helpers of make_golden_posthoc.py), get_supervised_chaos, add_chaos_gates --
are compiled by name from /root/reference/deepof/post_hoc.py and executed on synthetic embeddings / quality tables.

What the fixture pins and what it cannot: ``deeptime`` (pyproject.toml:58, ^0.4.5) is not installed, so the three deeptime
objects the reference instantiates (TransitionCountEstimator, MaximumLikelihoodMSM, the MSM's .pcca()) are thin adaptors
over deepof_amd.msm_pcca here.  The fixture therefore pins the reference's ORCHESTRATION around them (runs, microstate
k-means with its seeds and sizes, active-set mapping, padding, uniform rows for inactive microstates, decode, smoothing,
chaos windows) bit for bit, and is silent about deeptime's numerics -- stated as "parity unpinned" in deepof_amd/msm_pcca.py.
Output: posthoc_msm.npz (inputs + expected soft counts; data only)."""
import ast
import os
import sys
import types

import numpy as np
import pandas as pd
from scipy.ndimage import uniform_filter1d
from sklearn.cluster import MiniBatchKMeans
from sklearn.preprocessing import StandardScaler

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden_posthoc import REF, _Bar  # noqa: E402
from deepof_amd import msm_pcca as MP  # noqa: E402

NAMES = ["get_contrastive_soft_counts_msm_pcca", "_fit_msmpcca_models", "_collect_segments_for_gate_bin",
         "_fit_microstates_kmeans", "_segments_to_dtrajs", "_fit_pcca_memberships", "_get_active_state_symbols",
         "_pcca_memberships", "_build_micro2macro", "_mask_to_runs", "_reservoir_sample", "_temporal_smooth",
         "_build_gate_masks", "_get_Z", "_gate_to_tag", "get_supervised_chaos", "add_chaos_gates"]


class TransitionCountEstimator:          # adaptor: deeptime.markov.TransitionCountEstimator
    def __init__(self, lagtime, count_mode):
        assert count_mode == "sliding"
        self.lagtime = lagtime

    def fit(self, dtrajs):
        self.C = MP.sliding_count_matrix(dtrajs, self.lagtime)
        return self

    def fetch_model(self):
        return types.SimpleNamespace(count_matrix=self.C)


class MaximumLikelihoodMSM:              # adaptor: deeptime.markov.msm.MaximumLikelihoodMSM(reversible=True)
    def __init__(self, reversible):
        assert reversible

    def fit(self, count_model):
        C = count_model.count_matrix
        self.active = MP.largest_connected_set(C)
        self.T, self.pi = MP.reversible_mle(C[np.ix_(self.active, self.active)])
        return self

    def fetch_model(self):
        T, pi = self.T, self.pi
        return types.SimpleNamespace(n_states=T.shape[0], state_symbols=self.active,
                                     pcca=lambda m: types.SimpleNamespace(memberships=MP.pcca_memberships(T, pi, m)))


def load():
    from typing import Any, Dict, List, Optional, Tuple
    tree = ast.parse(open(REF).read(), filename=REF)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in NAMES]
    assert len(keep) == len(NAMES), sorted(set(NAMES) - {n.name for n in keep})
    deepof = types.SimpleNamespace(utils=types.SimpleNamespace(save_dt=lambda arr, path, big: arr),
                                   data=types.SimpleNamespace(TableDict=lambda d, **k: d))
    ns = dict(np=np, pd=pd, os=os, Any=Any, Dict=Dict, List=List, Optional=Optional, Tuple=Tuple, MiniBatchKMeans=MiniBatchKMeans,
              StandardScaler=StandardScaler, uniform_filter1d=uniform_filter1d, tqdm=types.SimpleNamespace(tqdm=_Bar),
              PROGRESS_BAR_FIXED_WIDTH=30, deepof=deepof, get_dt=lambda d, k, **kw: d[k],
              TransitionCountEstimator=TransitionCountEstimator, MaximumLikelihoodMSM=MaximumLikelihoodMSM)
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns


def metastable_embeddings(rng, n, L, n_states=4, stay=0.97):
    """A slow switching process between well separated centres: what an MSM is for."""
    centres = rng.standard_normal((n_states, L)) * 3.0
    s = np.empty(n, dtype=np.int64)
    s[0] = rng.integers(0, n_states)
    for t in range(1, n):
        s[t] = s[t - 1] if rng.random() < stay else rng.integers(0, n_states)
    return (centres[s] + 0.6 * rng.standard_normal((n, L))).astype(np.float32)


def main():
    ns = load()
    coords = types.SimpleNamespace(_project_path="/tmp", _project_name="p", _very_large_project=False, get_exp_conditions={})
    rng = np.random.default_rng(7)
    out = {}
    cases = [("single", [700, 520, 610], 6, 3, 1, [""], False, 3, 40, 3),
             ("dist", [900, 800], 8, 3, 2, [("A", "B")], False, 1, 60, 2)]
    for tag, lens, L, C, M, gates, categorical, smooth, n_micro, lag in cases:
        keys = [f"v{i}" for i in range(len(lens))]
        emb = {k: metastable_embeddings(rng, n, L) for k, n in zip(keys, lens)}
        if M == 1:
            series = {k: {g: np.zeros(n) for g in gates} for k, n in zip(keys, lens)}
            edges = {g: np.array([-np.inf, np.inf]) for g in gates}
        else:   # a slowly varying distance-like gate (long runs inside each bin)
            series = {k: {g: np.abs(np.cumsum(rng.standard_normal(n)) * 0.2) for g in gates} for k, n in zip(keys, lens)}
            edges = {}
            for g in gates:
                e = np.nanquantile(np.concatenate([series[k][g] for k in keys]), np.linspace(0, 1, M + 1)).astype(np.float64)
                e[0], e[-1] = -np.inf, np.inf
                edges[g] = e
        emb_len = {k: n for k, n in zip(keys, lens)}

        def prep(coordinates, embeddings, animal_ids, window_size, supervised_annotations, M_gates, embedding_gates, gate_edges):
            masks = ns["_build_gate_masks"](keys=keys, emb_len=emb_len, dist_series_dict=series, gates=gates, M_gates=M,
                                            supervised_annotations=supervised_annotations, gate_edges=gate_edges)
            return keys, gates, masks, dict(emb), M

        ns["_preprocess_gates"] = prep
        res = ns["get_contrastive_soft_counts_msm_pcca"](coords, emb, ["A"], window_size=12, supervised_annotations=None,
                                                        N_clusters_per_gate=C, M_gates=M, gate_edges=edges, sample_size=200000,
                                                        random_state=0, temporal_smooth_win=smooth, n_micro=n_micro,
                                                        min_micro_per_macro=3, lagtime=lag)
        p = f"{tag}::"
        out[p + "cfg"] = np.array([L, C, M, smooth, n_micro, lag], dtype=np.int64)
        out[p + "keys"] = np.array(keys)
        for k in keys:
            out[p + f"emb::{k}"] = emb[k]
            for gi, g in enumerate(gates):
                out[p + f"series::{gi}::{k}"] = np.asarray(series[k][g], dtype=np.float64)
                out[p + f"soft::{gi}::{k}"] = np.asarray(res[g][k], dtype=np.float32)
        for gi, g in enumerate(gates):
            out[p + f"edges::{gi}"] = edges[g]
    # ---- chaos labels and chaos gates
    cols = ["B_Nose", "B_Tail_base", "B_Center", "W_Nose", "W_Tail_base", "W_Center"]
    W = 5
    q = {k: rng.random((n + W - 1, len(cols))) for k, n in (("v0", 60), ("v1", 45))}
    q["v0"][10:14, :3] = 0.1
    q["v0"][30, 3:] = np.nan
    q["v1"][5:9] = 0.2
    coords2 = types.SimpleNamespace(_project_path="/tmp", _project_name="p", _very_large_project=False, get_exp_conditions={},
                                    _animal_ids=["B", "W"], _tables={k: None for k in q},
                                    get_quality=lambda: {k: pd.DataFrame(v, columns=cols) for k, v in q.items()})
    chaos = ns["get_supervised_chaos"](coords2, 0.75, 0.5)
    sc = {("B", "W"): {k: rng.random((v.shape[0] - W + 1, 6)).astype(np.float32) for k, v in q.items()}}
    sc_chaos = {"behavior_combinations": {k: rng.random((v.shape[0] - W + 1, 4)).astype(np.float32) for k, v in q.items()}}
    comb = ns["add_chaos_gates"](coords2, sc, sc_chaos, chaos, W)
    out["chaos::cols"] = np.array(cols)
    out["chaos::W"] = np.int64(W)
    for k in q:
        out[f"chaos::quality::{k}"] = q[k]
        for c in chaos[k].columns:
            out[f"chaos::label::{c}::{k}"] = chaos[k][c].to_numpy(np.float32)
        out[f"chaos::sc::{k}"] = sc[("B", "W")][k]
        out[f"chaos::sc_chaos::{k}"] = sc_chaos["behavior_combinations"][k]
        out[f"chaos::combined::{k}"] = np.asarray(comb[("B", "W")][k], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "posthoc_msm.npz"), **out)
    print("posthoc_msm.npz", os.path.getsize(os.path.join(HERE, "posthoc_msm.npz")))


if __name__ == "__main__":
    main()
