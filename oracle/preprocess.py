"""CPU restatement of the reference's pose-table preprocessing (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

SURVEY.md section 8(f) row N2.  Follows, under /root/reference/deepof:
  infer_column_types        utils.py:2395-2422   coords / speeds / distances (inner, intra) / angles from column labels
  scale_table               utils.py:2425-2566   per-animal size factor (nan-median nose--tail-base length), size
                                                 normalisation (with the .loc quirk noted in scale_table below),
                                                 log1p of distances, per-video standardisation
  _pp_pass1_collect_samples utils.py:2665-2792   rows sampled per video (RandomState(2).choice, <= samples_max)
  _pp_fit_global_scaler     utils.py:2795-2863   global scalers on the per-video-standardised samples
  _pp_apply_global          utils.py:2866-2921
  _pp_pass2_scale_and_save  utils.py:2924-3027   |z| > interpolate_normalized -> NaN -> linear interpolation in time,
                                                 angle interpolation, _pp_sanitize_numeric (:2577-2583)
  TableDict.preprocess      data.py:3773-3916    (time bins and the train/test split are control plane: inputs here)
sklearn.preprocessing.StandardScaler 1.7 (fit = NaN-ignoring two-pass mean/variance with float64 accumulators,
near-constant features get scale 1) and pandas ``interpolate(limit_direction="both")`` (= numpy.interp over row
positions, flat beyond the first/last valid row) are restated with numpy; only ``scale="standard"`` is covered.
A table is a float64 array (frames, C) plus the list of column labels: ``(bodypart, "x"|"y")`` coordinates,
``bodypart`` speeds, ``(bp1, bp2)`` distances, 3-tuples angles.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

MODES = ("per_column", "groupwise", None)


# ---------------------------------------------------------------------------------------------------------
def column_types(columns: Sequence) -> Dict[str, List[int]]:
    """Column indices by kind (utils.py:2395-2422)."""
    coords = [i for i, c in enumerate(columns) if isinstance(c, tuple) and len(c) == 2 and c[1] in ("x", "y")]
    bodyparts = {columns[i][0] for i in coords}
    speeds = [i for i, c in enumerate(columns) if isinstance(c, str) and c in bodyparts]
    dists = [i for i, c in enumerate(columns)
             if isinstance(c, tuple) and len(c) == 2 and c[0] in bodyparts and c[1] in bodyparts]
    angles = [i for i, c in enumerate(columns) if isinstance(c, tuple) and len(c) == 3]

    def prefix(bp):
        return bp.split("_", 1)[0] if "_" in bp else None

    inner = [i for i in dists if prefix(columns[i][0]) == prefix(columns[i][1])]
    intra = [i for i in dists if prefix(columns[i][0]) != prefix(columns[i][1])]
    return dict(coords=coords, speeds=speeds, dists=dists, inner=inner, intra=intra, angles=angles,
                bodyparts=sorted(bodyparts))


def standard_fit(x: np.ndarray):
    """StandardScaler.fit on a 2-D float64 array -> (mean, scale), NaNs ignored (sklearn extmath._incremental_mean_and_var
    first call + _is_constant_feature + _handle_zeros_in_scale)."""
    x = np.asarray(x, dtype=np.float64)
    n = (x.shape[0] - np.isnan(x).sum(axis=0)).astype(np.float64)
    with np.errstate(all="ignore"):
        s = np.nansum(x, axis=0)
        mean = s / n
        t = x - mean
        corr = np.nansum(t, axis=0)
        var = (np.nansum(t * t, axis=0) - corr ** 2 / n) / n
        eps = np.finfo(np.float64).eps
        constant = var <= n * eps * var + (n * mean * eps) ** 2
        scale = np.sqrt(var)
    scale[constant] = 1.0
    return mean, scale


def _standardize(out: np.ndarray, cols: List[int], mode: Optional[str]):
    if not cols or mode is None:
        return
    if mode == "per_column":
        m, s = standard_fit(out[:, cols])
        out[:, cols] = (out[:, cols] - m) / s
    else:
        m, s = standard_fit(out[:, cols].reshape(-1, 1))
        out[:, cols] = (out[:, cols] - m[0]) / s[0]


def size_factors(tab: np.ndarray, columns: Sequence, animal_ids, size_ref=("Nose", "Tail_base")):
    """{animal id: size factor}, default factor (utils.py:2477-2494)."""
    pos = {c: i for i, c in enumerate(columns)}
    s_by = {}
    for aid in animal_ids:
        a = size_ref[0] if aid is None else f"{aid}_{size_ref[0]}"
        b = size_ref[1] if aid is None else f"{aid}_{size_ref[1]}"
        need = [(a, "x"), (a, "y"), (b, "x"), (b, "y")]
        if all(c in pos for c in need):
            dx = tab[:, pos[need[0]]] - tab[:, pos[need[2]]]
            dy = tab[:, pos[need[1]]] - tab[:, pos[need[3]]]
            with np.errstate(all="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    s_by[aid] = np.nanmedian(np.hypot(dx, dy))
        else:
            s_by[aid] = np.nan
    valid = [v for v in s_by.values() if np.isfinite(v) and v > 0]
    s_default = float(np.nanmedian(valid)) if valid else 1.0
    return {k: (v if np.isfinite(v) and v > 0 else s_default) for k, v in s_by.items()}, s_default


def scale_table(tab: np.ndarray, columns: Sequence, animal_ids=None, inter_scale: str = "mean", standardize: bool = True,
                dist_standardize="per_column", speed_standardize="per_column", coord_standardize="per_column",
                log_distances: bool = True) -> np.ndarray:
    """scale_table(scale="standard") (utils.py:2425-2566)."""
    out = np.array(tab, dtype=np.float64, copy=True)
    ct = column_types(columns)
    bodyparts = ct["bodyparts"]

    def split(bp):
        return bp.split("_", 1) if "_" in bp else (None, bp)

    if animal_ids is None:
        pre = {split(bp)[0] for bp in bodyparts if split(bp)[0] is not None}
        animal_ids = sorted(pre) or [None]
    animal_ids = list(animal_ids)
    bp_aid = {bp: split(bp)[0] for bp in bodyparts}
    s_by, s_default = size_factors(out, columns, animal_ids)

    def comb(s1, s2):
        if inter_scale == "mean":
            return 0.5 * (s1 + s2)
        if inter_scale == "geom":
            return float(np.sqrt(s1 * s2))
        return s_default

    pos = {c: i for i, c in enumerate(columns)}
    for aid in animal_ids:
        bps = [bp for bp in bodyparts if bp_aid.get(bp) == aid] if aid is not None else \
              [bp for bp in bodyparts if bp_aid.get(bp) is None]
        if not bps:
            continue
        cols = [pos[(bp, ax)] for bp in bps for ax in ("x", "y") if (bp, ax) in pos] + [pos[bp] for bp in bps if bp in pos]
        out[:, cols] = out[:, cols] / s_by[aid]
    for i in ct["dists"]:
        a1, a2 = bp_aid.get(columns[i][0]), bp_aid.get(columns[i][1])
        s = s_by.get(a1, s_default) if a1 == a2 else comb(s_by.get(a1, s_default), s_by.get(a2, s_default))
        # Reference quirk (utils.py:2523-2529): the merged table's columns are a FLAT index of mixed labels, so
        # ``out.loc[:, (bp1, bp2)]`` is a list-like of the two labels bp1, bp2 -- the two SPEED columns -- and not
        # the distance column.  Distances are therefore never size-normalised; every speed column is divided once
        # more for each distance column its body part takes part in (a missing speed column is a KeyError there).
        pair = [pos[columns[i][0]], pos[columns[i][1]]]
        out[:, pair] = out[:, pair] / s
    if log_distances and ct["dists"]:
        arr = out[:, ct["dists"]]
        arr[arr < 0] = 0.0
        out[:, ct["dists"]] = np.log1p(arr)
    if not standardize:
        return out
    _standardize(out, ct["speeds"], speed_standardize)
    if dist_standardize == "per_column":
        _standardize(out, ct["dists"], "per_column")
    elif dist_standardize == "groupwise":
        _standardize(out, ct["inner"], "groupwise")
        _standardize(out, ct["intra"], "groupwise")
    _standardize(out, ct["coords"], coord_standardize)
    return out


def interpolate_both(col: np.ndarray) -> np.ndarray:
    """pandas Series.interpolate(limit_direction="both") on one float64 column: numpy.interp over row positions."""
    col = np.array(col, dtype=np.float64, copy=True)
    bad = np.isnan(col)
    if bad.all() or not bad.any():
        return col
    idx = np.arange(col.shape[0], dtype=np.float64)
    col[bad] = np.interp(idx[bad], idx[~bad], col[~bad])
    return col


def sample_rows(lengths: Sequence[int], samples_max: int) -> List[np.ndarray]:
    """The row subsets pass 1 draws, one RandomState(2) shared by all videos in key order (utils.py:2679, :2718)."""
    rng = np.random.RandomState(2)
    out = []
    for n in lengths:
        take = min(samples_max, n)
        out.append(rng.choice(n, size=take, replace=False) if take > 0 else np.zeros(0, dtype=np.int64))
    return out


def preprocess(tables: Dict[str, np.ndarray], columns: Sequence, animal_ids, samples_max: int = 227272,
               dist_standardize="groupwise", speed_standardize="groupwise", coord_standardize="groupwise",
               log_distances: bool = True, interpolate_normalized: float = 10, pretrained_scaler: Optional[dict] = None):
    """TableDict.preprocess(scale="standard") up to (not including) window extraction.
    Returns ({key: (frames, C) float64}, global scaler as {"speed"|"dist"|"dist_inner"|"dist_intra"|"coord": (mean, scale)})."""
    ct = column_types(columns)
    keys = [k for k in sorted(tables) if not np.isnan(tables[k]).all()]
    local = {}
    for k in keys:
        non_angle = [i for i in range(len(columns)) if i not in ct["angles"]]
        sub_cols = [columns[i] for i in non_angle]
        loc = np.array(tables[k], dtype=np.float64, copy=True)
        loc[:, non_angle] = scale_table(tables[k][:, non_angle], sub_cols, animal_ids, standardize=True,
                                        dist_standardize=dist_standardize, speed_standardize=speed_standardize,
                                        coord_standardize=None, log_distances=log_distances)
        local[k] = loc
    if pretrained_scaler is not None:
        gs = pretrained_scaler
    else:
        idx = sample_rows([local[k].shape[0] for k in keys], samples_max)
        gs = {}

        def fit(name, cols, mode):
            if not cols or mode is None:
                return
            parts = [local[k][i][:, cols] for k, i in zip(keys, idx) if len(i)]
            if not parts:
                return
            if mode == "per_column":
                gs[name] = standard_fit(np.vstack(parts))
            else:
                m, s = standard_fit(np.concatenate([p.reshape(-1) for p in parts]).reshape(-1, 1))
                gs[name] = (m, s)

        fit("speed", ct["speeds"], speed_standardize)
        if dist_standardize == "per_column":
            fit("dist", ct["dists"], "per_column")
        elif dist_standardize == "groupwise":
            fit("dist_inner", ct["inner"], "groupwise")
            fit("dist_intra", ct["intra"], "groupwise")
        fit("coord", ct["coords"], coord_standardize)
    out = {}
    for k in keys:
        tab = local[k].copy()

        def apply(name, cols):
            if cols and gs.get(name) is not None:
                m, s = gs[name]
                tab[:, cols] = (tab[:, cols] - m) / s

        if speed_standardize is not None:
            apply("speed", ct["speeds"])
        if dist_standardize == "per_column":
            apply("dist", ct["dists"])
        elif dist_standardize == "groupwise":
            apply("dist_inner", ct["inner"])
            apply("dist_intra", ct["intra"])
        if coord_standardize is not None:
            apply("coord", ct["coords"])
        clip_cols = list(dict.fromkeys(ct["speeds"] + ct["dists"] + ct["coords"]))
        if interpolate_normalized and clip_cols:
            arr = tab[:, clip_cols]
            with np.errstate(invalid="ignore"):
                arr[np.abs(arr) > interpolate_normalized] = np.nan
            tab[:, clip_cols] = arr
        for c in range(tab.shape[1]):
            tab[:, c] = interpolate_both(tab[:, c])
        tab[np.isnan(tab)] = 0.0
        out[k] = tab
    return out, gs
