"""Golden for the post-hoc soft-count decoder: the REFERENCE's get_contrastive_soft_counts_gmm
(/root/reference/deepof/post_hoc.py:1028-1172) with its helpers _reservoir_sample, _temporal_smooth, _build_gate_masks,
_get_Z, _gate_to_tag compiled by name from the file where it lies (post_hoc.py as a whole needs the full DeepOF
dependency set) and executed on synthetic embeddings / gating series.  Only the Coordinates-bound front end
(_preprocess_gates: gating series from the project's distance tables) is replaced: it hands the prepared series to the
reference's own _build_gate_masks.  Output: posthoc.npz (inputs + expected soft counts; data only)."""
import ast
import os
import sys
import types

import numpy as np
from scipy.ndimage import uniform_filter1d
from sklearn.mixture import GaussianMixture

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/deepof/post_hoc.py"
NAMES = ["get_contrastive_soft_counts_gmm", "_reservoir_sample", "_temporal_smooth", "_build_gate_masks", "_get_Z", "_gate_to_tag"]


class _Bar:
    def __init__(self, it=None, **k):
        self.it = it

    def __iter__(self):
        return iter(self.it)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, *a):
        pass


def load():
    from typing import Any, Dict, List, Optional, Tuple
    tree = ast.parse(open(REF).read(), filename=REF)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in NAMES]
    assert len(keep) == len(NAMES), [n.name for n in keep]
    deepof = types.SimpleNamespace(utils=types.SimpleNamespace(save_dt=lambda arr, path, big: arr),
                                   data=types.SimpleNamespace(TableDict=lambda d, **k: d))
    ns = dict(np=np, os=os, Any=Any, Dict=Dict, List=List, Optional=Optional, Tuple=Tuple, GaussianMixture=GaussianMixture,
              uniform_filter1d=uniform_filter1d, tqdm=types.SimpleNamespace(tqdm=_Bar), PROGRESS_BAR_FIXED_WIDTH=30, deepof=deepof,
              get_dt=lambda d, k, **kw: d[k])
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns


def main():
    ns = load()
    coords = types.SimpleNamespace(_project_path="/tmp", _project_name="p", _very_large_project=False, get_exp_conditions={})
    rng = np.random.default_rng(3)
    out = {}
    cases = [
        # tag, lengths, L, C, M, gates, categorical, sample_size, smooth
        ("single", [180, 97, 140], 6, 4, 1, [""], False, 150, 3),
        ("dist", [160, 120], 8, 3, 3, [("A", "B"), ("A", "C")], False, 200000, 3),
        ("behav", [90, 75, 60], 4, 3, 2, [""], True, 100, 1),
    ]
    for tag, lens, L, C, M, gates, categorical, sample_size, smooth in cases:
        keys = [f"v{i}" for i in range(len(lens))]
        centers = rng.standard_normal((5, L)) * 2.0
        emb = {k: (centers[rng.integers(0, 5, n)] + 0.5 * rng.standard_normal((n, L))).astype(np.float32) for k, n in zip(keys, lens)}
        if categorical:
            series = {k: {g: rng.integers(0, M, n) for g in gates} for k, n in zip(keys, lens)}
            series[keys[0]][gates[0]][:] = 0      # ... and bin 1 stays small overall
            for k in keys[1:]:
                series[k][gates[0]][:] = np.where(rng.random(len(series[k][gates[0]])) < 0.03, 1, 0)
            edges = None
        else:
            series = {k: {g: np.abs(rng.standard_normal(n)) * (1 + gi) for gi, g in enumerate(gates)} for k, n in zip(keys, lens)}
            qs = np.linspace(0, 1, M + 1)     # compute_gate_edges, post_hoc.py:695-702
            edges = {}
            for g in gates:
                e = np.nanquantile(np.concatenate([series[k][g] for k in keys]), qs).astype(np.float64)
                e[0], e[-1] = -np.inf, np.inf
                edges[g] = e
        emb_len = {k: n for k, n in zip(keys, lens)}

        def prep(coordinates, embeddings, animal_ids, window_size, supervised_annotations, M_gates, embedding_gates, gate_edges):
            masks = ns["_build_gate_masks"](keys=keys, emb_len=emb_len, dist_series_dict=series, gates=gates, M_gates=M,
                                            supervised_annotations=supervised_annotations, gate_edges=gate_edges)
            return keys, gates, masks, dict(emb), M

        ns["_preprocess_gates"] = prep
        res = ns["get_contrastive_soft_counts_gmm"](coords, emb, ["A"], window_size=12,
                                                   supervised_annotations=(object() if categorical else None),
                                                   N_clusters_per_gate=C, M_gates=M, gate_edges=edges, sample_size=sample_size,
                                                   random_state=0, temporal_smooth_win=smooth)
        p = f"{tag}::"
        out[p + "cfg"] = np.array([L, C, M, int(categorical), sample_size, smooth], dtype=np.int64)
        out[p + "keys"] = np.array(keys)
        out[p + "n_gates"] = np.int64(len(gates))
        for k in keys:
            out[p + f"emb::{k}"] = emb[k]
            for gi, g in enumerate(gates):
                out[p + f"series::{gi}::{k}"] = np.asarray(series[k][g], dtype=np.float64)
                out[p + f"soft::{gi}::{k}"] = np.asarray(res[g][k], dtype=np.float32)
        if edges is not None:
            for gi, g in enumerate(gates):
                out[p + f"edges::{gi}"] = edges[g]
    # the reservoir sampler on its own, beyond the buffer size
    segs = [rng.standard_normal((n, 3)).astype(np.float32) for n in (40, 25, 60)]
    out["reservoir::segs"] = np.concatenate(segs)
    out["reservoir::lens"] = np.array([40, 25, 60])
    out["reservoir::out"] = ns["_reservoir_sample"](segs, 50, seed=11)
    np.savez_compressed(os.path.join(HERE, "posthoc.npz"), **out)
    print("posthoc.npz", os.path.getsize(os.path.join(HERE, "posthoc.npz")))


if __name__ == "__main__":
    main()
