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
near-constant features get scale 1), MinMaxScaler (``X * scale_ + min_`` with ``scale_ = 1 / handle_zeros(max - min)``,
``min_ = -data_min * scale_``, NaNs ignored) and pandas ``interpolate(limit_direction="both")`` (= numpy.interp over
row positions, flat beyond the first/last valid row) are restated with numpy; ``scale`` "standard", "minmax" and "robust" are
covered (``_pp_make_scaler`` utils.py:2570), and ``_pp_filter_low_variance`` (utils.py:2604-2620).
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


def minmax_fit(x: np.ndarray):
    """MinMaxScaler.fit on a 2-D float64 array -> (scale_, min_), NaNs ignored (sklearn _data.py partial_fit:
    data_range below 10 eps counts as constant -> divisor 1)."""
    x = np.asarray(x, dtype=np.float64)
    import warnings
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lo, hi = np.nanmin(x, axis=0), np.nanmax(x, axis=0)
    rng = hi - lo
    div = rng.copy()
    div[div < 10 * np.finfo(np.float64).eps] = 1.0
    scale = 1.0 / div
    return scale, 0.0 - lo * scale


def robust_fit(x: np.ndarray):
    """RobustScaler.fit on a 2-D float64 array -> (center_, scale_): np.nanmedian, np.nanpercentile(25, 75) (linear
    interpolation), inter-quartile ranges below 10 eps -> 1 (sklearn _data.py RobustScaler.fit)."""
    import warnings
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        center = np.nanmedian(x, axis=0)
        q = np.nanpercentile(x, (25.0, 75.0), axis=0)
    scale = q[1] - q[0]
    scale[scale < 10 * np.finfo(np.float64).eps] = 1.0
    return center, scale


def fit_scaler(x: np.ndarray, kind: str):
    return standard_fit(x) if kind == "standard" else minmax_fit(x) if kind == "minmax" else robust_fit(x)


def apply_scaler(x: np.ndarray, pair, kind: str) -> np.ndarray:
    """transform(): StandardScaler / RobustScaler (x - a) / b, MinMaxScaler x * scale_ + min_ (each operation rounded on
    its own)."""
    a, b = pair
    return x * a + b if kind == "minmax" else (x - a) / b


def _standardize(out: np.ndarray, cols: List[int], mode: Optional[str], kind: str = "standard"):
    if not cols or mode is None:
        return
    if mode == "per_column":
        out[:, cols] = apply_scaler(out[:, cols], fit_scaler(out[:, cols], kind), kind)
    else:
        a, b = fit_scaler(out[:, cols].reshape(-1, 1), kind)
        out[:, cols] = apply_scaler(out[:, cols], (a[0], b[0]), kind)


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
                log_distances: bool = True, scale: str = "standard") -> np.ndarray:
    """scale_table (utils.py:2425-2566), scale "standard" or "minmax"."""
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
    _standardize(out, ct["speeds"], speed_standardize, scale)
    if dist_standardize == "per_column":
        _standardize(out, ct["dists"], "per_column", scale)
    elif dist_standardize == "groupwise":
        _standardize(out, ct["inner"], "groupwise", scale)
        _standardize(out, ct["intra"], "groupwise", scale)
    _standardize(out, ct["coords"], coord_standardize, scale)
    return out


def low_variance_keep(tab: np.ndarray, columns: Sequence, threshold) -> List[int]:
    """Column indices _pp_filter_low_variance keeps (utils.py:2604-2620): pandas var (ddof 1, NaNs skipped) > threshold
    first, then the "pheno" columns (again, if they also passed: the reference concatenates the two index lists)."""
    import warnings
    n = (~np.isnan(tab)).sum(axis=0).astype(np.float64)
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mean = np.nansum(tab, axis=0) / n
        var = np.where(n > 1, np.nansum((tab - mean) ** 2, axis=0) / (n - 1), np.nan)
        keep = list(np.where(var > threshold)[0])
    return keep + [i for i, c in enumerate(columns) if "pheno" in str(c)]


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
               log_distances: bool = True, interpolate_normalized: float = 10, pretrained_scaler: Optional[dict] = None,
               scale: str = "standard", filter_low_variance=False):
    """TableDict.preprocess up to (not including) window extraction, scale "standard" | "minmax" | "robust".
    Returns ({key: (frames, C) float64}, global scaler as {"speed"|"dist"|"dist_inner"|"dist_intra"|"coord": pair}) with
    pair = (mean_, scale_) for "standard" and (scale_, min_) for "minmax".  ``filter_low_variance``: every video must keep
    the same columns (the case the product covers); the returned tables then hold the kept columns only, in order."""
    keys = [k for k in sorted(tables) if not np.isnan(tables[k]).all()]
    if filter_low_variance:
        return _preprocess_filtered(tables, keys, columns, animal_ids, samples_max, dist_standardize, speed_standardize,
                                    coord_standardize, log_distances, interpolate_normalized, pretrained_scaler, scale,
                                    filter_low_variance)
    ct = column_types(columns)
    local = {}
    for k in keys:
        non_angle = [i for i in range(len(columns)) if i not in ct["angles"]]
        sub_cols = [columns[i] for i in non_angle]
        loc = np.array(tables[k], dtype=np.float64, copy=True)
        loc[:, non_angle] = scale_table(tables[k][:, non_angle], sub_cols, animal_ids, standardize=True,
                                        dist_standardize=dist_standardize, speed_standardize=speed_standardize,
                                        coord_standardize=None, log_distances=log_distances, scale=scale)
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
                gs[name] = fit_scaler(np.vstack(parts), scale)
            else:
                gs[name] = fit_scaler(np.concatenate([p.reshape(-1) for p in parts]).reshape(-1, 1), scale)

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
                tab[:, cols] = apply_scaler(tab[:, cols], gs[name], scale)

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
        if scale == "standard" and interpolate_normalized and clip_cols:   # utils.py:2993
            arr = tab[:, clip_cols]
            with np.errstate(invalid="ignore"):
                arr[np.abs(arr) > interpolate_normalized] = np.nan
            tab[:, clip_cols] = arr
        for c in range(tab.shape[1]):
            tab[:, c] = interpolate_both(tab[:, c])
        tab[np.isnan(tab)] = 0.0
        out[k] = tab
    return out, gs


def _preprocess_filtered(tables, keys, columns, animal_ids, samples_max, dist_standardize, speed_standardize, coord_standardize,
                         log_distances, interpolate_normalized, pretrained_scaler, scale, threshold):
    """preprocess() with _pp_filter_low_variance: every video is scaled on its own kept columns (pass 1 filters the whole
    table, utils.py:2701; pass 2 sets the angles aside first, :2962-2966), the global scalers see the kept columns (per-column
    sections: the first video's columns, :2651-2656) and a dropped column comes back as zeros (:3011-3015)."""
    n_cols = len(columns)
    all_angles = set(column_types(columns)["angles"])
    kept = {k: sorted(set(low_variance_keep(tables[k], columns, threshold)) - all_angles) for k in keys}
    local, local_cols = {}, {}
    for k in keys:
        idx = kept[k]
        assert idx, "the entire table was filtered out"
        sub_cols = [columns[i] for i in idx]
        local[k] = scale_table(tables[k][:, idx], sub_cols, animal_ids, standardize=True, dist_standardize=dist_standardize,
                               speed_standardize=speed_standardize, coord_standardize=None, log_distances=log_distances, scale=scale)
        local_cols[k] = sub_cols
    if pretrained_scaler is not None:
        gs = pretrained_scaler
    else:
        rows = sample_rows([local[k].shape[0] for k in keys], samples_max)
        gs = {}

        def fit(name, which, mode):
            if mode is None:
                return
            parts, ref = [], None
            for k, r in zip(keys, rows):
                ct = column_types(local_cols[k])
                cols = [local_cols[k][i] for i in ct[which]]
                if not cols or not len(r):
                    continue
                if mode == "per_column":
                    ref = ref or cols
                    assert cols == ref, "per-column sections need the same columns in every video"
                    parts.append(local[k][r][:, ct[which]])
                else:
                    parts.append(local[k][r][:, ct[which]].reshape(-1))
            if parts:
                gs[name] = fit_scaler(np.vstack(parts) if mode == "per_column" else np.concatenate(parts).reshape(-1, 1), scale)

        fit("speed", "speeds", speed_standardize)
        if dist_standardize == "per_column":
            fit("dist", "dists", "per_column")
        elif dist_standardize == "groupwise":
            fit("dist_inner", "inner", "groupwise")
            fit("dist_intra", "intra", "groupwise")
        fit("coord", "coords", coord_standardize)
    out = {}
    for k in keys:
        tab, ct = local[k].copy(), column_types(local_cols[k])

        def apply(name, cols):
            if cols and gs.get(name) is not None:
                tab[:, cols] = apply_scaler(tab[:, cols], gs[name], scale)

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
        if scale == "standard" and interpolate_normalized and clip_cols:
            arr = tab[:, clip_cols]
            with np.errstate(invalid="ignore"):
                arr[np.abs(arr) > interpolate_normalized] = np.nan
            tab[:, clip_cols] = arr
        full = np.full((tab.shape[0], n_cols), np.nan)
        full[:, kept[k]] = tab
        for a in all_angles:
            full[:, a] = tables[k][:, a]
        for c in range(n_cols):
            full[:, c] = interpolate_both(full[:, c])
        full[np.isnan(full)] = 0.0
        out[k] = full
    return out, gs
