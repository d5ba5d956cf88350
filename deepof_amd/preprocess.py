"""Pose-table preprocessing on the device: the host side of ``dof_preprocess_tables``.

Mirrors ``TableDict.preprocess`` (/root/reference/deepof/data.py:3773-3916; scale "standard" or "minmax",
``filter_low_variance`` where every video drops the same columns) up to window
extraction, and the column bookkeeping of ``get_graph_dataset`` (data.py:2797-2880): raw merged tables
(coordinates + speeds + distances [+ angles]) of every video go to the device ONCE as float64; size
normalisation, log1p, per-video and global standardisation, clipping, interpolation and the fp32 cast run there and
leave the resident frame tables ``dof_window_gather`` builds batches from.  No CPU path: without the HIP library
this module raises.

Reference behaviour kept on purpose (utils.py:2523-2529): the merged table has a flat column index, so
``out.loc[:, (bp1, bp2)]`` addresses the two *speed* columns bp1, bp2.  Distances are never size-normalised and a
speed column is divided once more for every distance column its body part appears in.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi

SIZE_REF = ("Nose", "Tail_base")


def _is_pair(c) -> bool:
    return isinstance(c, tuple) and len(c) == 2


def classify_columns(columns: Sequence) -> np.ndarray:
    """int32 kind per column, the rules of infer_column_types (utils.py:2395-2422)."""
    bodyparts = {c[0] for c in columns if _is_pair(c) and c[1] in ("x", "y")}
    kinds = np.zeros(len(columns), dtype=np.int32)
    for i, c in enumerate(columns):
        if _is_pair(c) and c[1] in ("x", "y"):
            kinds[i] = _capi.PP_KINDS["coord"]
        elif isinstance(c, str) and c in bodyparts:
            kinds[i] = _capi.PP_KINDS["speed"]
        elif _is_pair(c) and c[0] in bodyparts and c[1] in bodyparts:
            first = [bp.split("_", 1)[0] if "_" in bp else None for bp in c]
            kinds[i] = _capi.PP_KINDS["dist_inner" if first[0] == first[1] else "dist_intra"]
        elif isinstance(c, tuple) and len(c) == 3:
            kinds[i] = _capi.PP_KINDS["angle"]
    return kinds


@dataclass
class ColumnPlan:
    kinds: np.ndarray        # (C,) int32
    size_ref: np.ndarray     # (A, 4) int32
    chain_off: np.ndarray    # (C+1,) int32
    chain: np.ndarray        # (k, 4) int32: (animal 1, animal 2, same animal, source column)
    animal_ids: List


def column_plan(columns: Sequence, animal_ids) -> ColumnPlan:
    """Who divides which column by what (scale_table stage 1, utils.py:2458-2529)."""
    kinds = classify_columns(columns)
    where = {c: i for i, c in enumerate(columns)}
    bodyparts = sorted({c[0] for i, c in enumerate(columns) if kinds[i] == _capi.PP_KINDS["coord"]})
    owner = {bp: (bp.split("_", 1)[0] if "_" in bp else None) for bp in bodyparts}
    if animal_ids is None:
        found = sorted({o for o in owner.values() if o is not None})
        animal_ids = found or [None]
    animal_ids = list(animal_ids)
    if len(animal_ids) > _capi.PP_MAX_ANIMALS:
        raise ValueError(f"at most {_capi.PP_MAX_ANIMALS} animals")
    code = {}
    for i, aid in enumerate(animal_ids):
        code.setdefault(aid, i)   # s_by_aid is a dict: a repeated id keeps one entry
    size_ref = np.full((len(animal_ids), 4), -1, dtype=np.int32)
    for i, aid in enumerate(animal_ids):
        a, b = (SIZE_REF if aid is None else (f"{aid}_{SIZE_REF[0]}", f"{aid}_{SIZE_REF[1]}"))
        need = [(a, "x"), (a, "y"), (b, "x"), (b, "y")]
        if all(n in where for n in need):
            size_ref[i] = [where[n] for n in need]
    chains: List[List[Tuple[int, int, int, int]]] = [[] for _ in columns]
    for aid in animal_ids:
        mine = [bp for bp in bodyparts if owner[bp] == aid]
        for bp in mine:
            for lab in ((bp, "x"), (bp, "y"), bp):
                if lab in where:
                    chains[where[lab]].append((code[aid], code[aid], 1, where[lab]))
    for i, c in enumerate(columns):
        if kinds[i] in (_capi.PP_KINDS["dist_inner"], _capi.PP_KINDS["dist_intra"]):
            o1, o2 = owner.get(c[0]), owner.get(c[1])
            entry = (code.get(o1, -1), code.get(o2, -1), int(o1 == o2), i)
            for bp in c:   # the reference's .loc[:, (bp1, bp2)] hits the two speed columns (module docstring)
                if bp not in where:
                    raise KeyError(f"distance column {c} needs the speed column {bp!r} (as in the reference)")
                chains[where[bp]].append(entry)
    off = np.zeros(len(columns) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(ch) for ch in chains])
    flat = np.array([e for ch in chains for e in ch], dtype=np.int32).reshape(-1, 4)
    return ColumnPlan(kinds, size_ref, off, flat, animal_ids)


def sample_mask(lengths: Sequence[int], samples_max: int) -> Optional[np.ndarray]:
    """Rows entering the global fit: per video ``RandomState(2).choice(len, min(samples_max, len), replace=False)``
    with ONE generator advanced through the videos in key order (utils.py:2679, :2718).  None = every row."""
    rng = np.random.RandomState(2)
    picks, partial = [], False
    for n in lengths:
        take = min(int(samples_max), int(n))
        idx = rng.choice(int(n), size=take, replace=False) if take > 0 else np.zeros(0, dtype=np.int64)
        partial |= take < n
        picks.append(idx)
    if not partial:
        return None
    mask = np.zeros(int(np.sum(lengths)), dtype=np.uint8)
    off = 0
    for n, idx in zip(lengths, picks):
        mask[off + idx] = 1
        off += int(n)
    return mask


@dataclass
class PreprocessedTables:
    """Resident fp32 frame tables of all videos (rows of video i: video_off[i] .. video_off[i+1]-1)."""
    node_table: torch.Tensor
    edge_table: torch.Tensor
    angle_table: Optional[torch.Tensor]
    video_off: np.ndarray
    keys: List[str]
    global_scaler: Optional[dict]
    size_factors: torch.Tensor          # (videos, animals + 1) float64, last column = default factor
    video_scaler: torch.Tensor          # (videos, C, 2) float64 per-video (mean, scale)
    columns: List = field(default_factory=list)


_SECTIONS = (("speed", ("speed",)), ("dist", ("dist_inner", "dist_intra")), ("dist_inner", ("dist_inner",)),
             ("dist_intra", ("dist_intra",)), ("coord", ("coord",)))


def _scaler_to_dict(per_col: np.ndarray, kinds: np.ndarray, modes: Dict[str, Optional[str]], log_distances: bool,
                    scale: str = "standard", present: Optional[np.ndarray] = None) -> Optional[dict]:
    """(C,2) -> the legacy dict layout of GlobalScalerSpec.to_legacy_dict (utils.py:2362-2374): (mean, scale) pairs for
    "standard", (data_min, data_range with the near-zero ranges already replaced by 1) pairs for "minmax", (center,
    inter-quartile range) pairs for "robust"."""
    out = {"kind": scale, "speed": None, "dist": None, "dist_inner": None, "dist_intra": None, "coord": None,
           "speed_mode": modes["speed"], "dist_mode": modes["dist"], "coord_mode": modes["coord"], "log_distances": log_distances}

    def cols(names):   # columns the low-variance filter removed from every video are not features of the scalers
        return [i for i, k in enumerate(kinds) if k in [_capi.PP_KINDS[n] for n in names] and (present is None or present[i])]

    def put(name, names, mode):
        idx = cols(names)
        if not idx or mode is None:
            return
        sub = per_col[idx]
        out[name] = (sub[:, 0].copy(), sub[:, 1].copy()) if mode == "per_column" else (sub[:1, 0].copy(), sub[:1, 1].copy())

    put("speed", ("speed",), modes["speed"])
    if modes["dist"] == "per_column":
        put("dist", ("dist_inner", "dist_intra"), "per_column")
    elif modes["dist"] == "groupwise":
        put("dist_inner", ("dist_inner",), "groupwise")
        put("dist_intra", ("dist_intra",), "groupwise")
    put("coord", ("coord",), modes["coord"])
    return None if all(out[k] is None for k in ("speed", "dist", "dist_inner", "dist_intra", "coord")) else out


def _scaler_from_dict(gs: dict, kinds: np.ndarray, modes: Dict[str, Optional[str]], scale: str = None,
                      present: Optional[np.ndarray] = None) -> np.ndarray:
    """Legacy dict (pairs as written by _scaler_to_dict, or fitted sklearn StandardScalers / MinMaxScalers) -> (C,2)
    (offset, divisor), identity where nothing applies.  ``present`` (C,) marks the columns the low-variance filter kept
    in at least one video: a per-column section fitted on filtered tables has one entry per PRESENT column (what
    _scaler_to_dict writes, and what a sklearn scaler fitted on the reference's filtered tables holds), which is mapped
    back onto those columns; a section with an entry for every column of its kind is accepted as well."""
    if scale is not None and gs.get("kind") is not None and gs["kind"] != scale:
        raise ValueError(f"pretrained scaler is of kind {gs['kind']!r}, scale={scale!r} was requested")
    per_col = np.tile(np.array([0.0, 1.0]), (len(kinds), 1))

    def pair(v):
        if hasattr(v, "center_"):       # RobustScaler
            return np.atleast_1d(np.asarray(v.center_, dtype=np.float64)), np.atleast_1d(np.asarray(v.scale_, dtype=np.float64))
        if hasattr(v, "data_min_"):     # MinMaxScaler: X * scale_ + min_ == (X - data_min_) / handle_zeros(data_range_)
            rng = np.atleast_1d(np.asarray(v.data_range_, dtype=np.float64)).copy()
            rng[rng < 10 * np.finfo(np.float64).eps] = 1.0
            return np.atleast_1d(np.asarray(v.data_min_, dtype=np.float64)), rng
        if hasattr(v, "mean_"):
            return np.atleast_1d(np.asarray(v.mean_, dtype=np.float64)), np.atleast_1d(np.asarray(v.scale_, dtype=np.float64))
        return np.atleast_1d(np.asarray(v[0], dtype=np.float64)), np.atleast_1d(np.asarray(v[1], dtype=np.float64))

    def put(name, names, mode):
        idx = [i for i, k in enumerate(kinds) if k in [_capi.PP_KINDS[n] for n in names]]
        if not idx or mode is None or gs.get(name) is None:
            return
        m, s = pair(gs[name])
        if mode == "per_column" and len(m) != len(idx):
            kept = [i for i in idx if present is None or present[i]]
            if len(m) != len(kept):
                raise ValueError(f"pretrained scaler section {name!r} has {len(m)} columns, the tables have {len(idx)}"
                                 + (f" ({len(kept)} after the low-variance filter)" if len(kept) != len(idx) else ""))
            idx = kept
        per_col[idx, 0] = m if mode == "per_column" else m[0]
        per_col[idx, 1] = s if mode == "per_column" else s[0]

    put("speed", ("speed",), modes["speed"])
    if modes["dist"] == "per_column":
        put("dist", ("dist_inner", "dist_intra"), "per_column")
    elif modes["dist"] == "groupwise":
        put("dist_inner", ("dist_inner",), "groupwise")
        put("dist_intra", ("dist_intra",), "groupwise")
    put("coord", ("coord",), modes["coord"])
    return per_col


class _Call:
    """Descriptor tensors + one invocation of the C ABI for a set of (local) videos."""

    def __init__(self, lib, device, arrays, plan, out_cols, n_node, n_edge, n_ang, modes, log_distances, inter_scale, clip,
                 raw_device=None, scale="standard", keep=None):
        self.lib, self.device, self.plan = lib, torch.device(device), plan
        self.lengths = [a.shape[0] for a in arrays]
        self.video_off = np.zeros(len(arrays) + 1, dtype=np.int64)
        self.video_off[1:] = np.cumsum(self.lengths)
        self.n_frames, self.n_cols = int(self.video_off[-1]), len(plan.kinds)
        if raw_device is None:   # one host-to-device copy per table, straight into its rows (no host-side concatenation)
            raw_device = torch.empty(self.n_frames, self.n_cols, dtype=torch.float64, device=self.device)
            for a, lo, hi in zip(arrays, self.video_off[:-1], self.video_off[1:]):
                raw_device[int(lo):int(hi)].copy_(torch.from_numpy(np.ascontiguousarray(a)), non_blocking=True)
        if raw_device.dtype != torch.float64 or tuple(raw_device.shape) != (self.n_frames, self.n_cols):
            raise ValueError("raw_device must be the concatenated float64 tables")
        self.raw = raw_device
        dev = self._dev
        self.d_off, self.d_kind = dev(self.video_off), dev(plan.kinds)
        self.d_ref = dev(plan.size_ref.reshape(-1) if plan.size_ref.size else np.zeros(4, np.int32))
        self.d_coff = dev(plan.chain_off)
        self.d_chain = dev(plan.chain.reshape(-1) if plan.chain.size else np.zeros(4, np.int32))
        self.d_out = dev(out_cols)
        self.counts = (n_node, n_edge, n_ang)
        self.dims = _capi.PreprocDims(n_frames=self.n_frames, n_videos=len(arrays), n_cols=self.n_cols, n_animals=len(plan.animal_ids),
                                      n_node_cols=n_node, n_edge_cols=n_edge, n_angle_cols=n_ang,
                                      speed_mode=_capi.PP_MODES[modes["speed"]], dist_mode=_capi.PP_MODES[modes["dist"]],
                                      coord_mode=_capi.PP_MODES[modes["coord"]], log_distances=int(bool(log_distances)),
                                      inter_scale=_capi.PP_INTER_SCALE[inter_scale], fit_global=1, clip=float(clip or 0),
                                      scale_kind=_capi.PP_SCALE_KINDS[scale])
        self.d_keep = dev(np.asarray(keep, dtype=np.uint8)) if keep is not None else None   # (videos, C), 0 = filtered out
        self.dims.col_keep = self.d_keep.data_ptr() if self.d_keep is not None else None
        ws_bytes = lib.dof_preprocess_workspace_bytes(ctypes.byref(self.dims))
        if ws_bytes < 0:
            _capi.check(lib, -1, "dof_preprocess_workspace_bytes")
        self.ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        self.stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def _dev(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    @staticmethod
    def _ptr(t):
        return t.data_ptr() if t is not None else None

    def video_stats(self, mask) -> torch.Tensor:
        """(videos, C, 5) float64 (n, mean, M2, min, max) of the sampled, per-video-scaled rows."""
        d_mask = self._dev(mask) if mask is not None else None
        ystat = torch.empty(len(self.lengths), self.n_cols, _capi.PP_STAT_DOUBLES, dtype=torch.float64, device=self.device)
        p = self._ptr
        _capi.check(self.lib, self.lib.dof_preprocess_video_stats(ctypes.byref(self.dims), p(self.raw), p(self.d_off), p(self.d_kind),
                                                                  p(self.d_ref), p(self.d_coff), p(self.d_chain), p(d_mask), p(ystat),
                                                                  p(self.ws), self.stream), "dof_preprocess_video_stats")
        return ystat

    def raw_moments(self) -> torch.Tensor:
        """(videos, C, 5) float64 (n, mean, M2, min, max) of the raw values of every column."""
        mom = torch.empty(len(self.lengths), self.n_cols, _capi.PP_STAT_DOUBLES, dtype=torch.float64, device=self.device)
        p = self._ptr
        _capi.check(self.lib, self.lib.dof_preprocess_raw_moments(ctypes.byref(self.dims), p(self.raw), p(self.d_off), p(self.d_kind),
                                                                  p(mom), p(self.ws), self.stream), "dof_preprocess_raw_moments")
        return mom

    def order_stats(self, video_scaler: Optional[torch.Tensor], mask) -> torch.Tensor:
        """scale="robust": (videos, C, 7) per-video rows, or with the per-video scalers given the (C, 7) rows of the pooled
        sampled values -- n and the six order statistics around the median and the 25th / 75th percentile."""
        shape = (len(self.lengths), self.n_cols) if video_scaler is None else (self.n_cols,)
        out = torch.empty(*shape, _capi.PP_ORDER_DOUBLES, dtype=torch.float64, device=self.device)
        d_mask = self._dev(mask) if (mask is not None and video_scaler is not None) else None
        p = self._ptr
        _capi.check(self.lib, self.lib.dof_preprocess_order_stats(ctypes.byref(self.dims), p(self.raw), p(self.d_off), p(self.d_kind),
                                                                  p(self.d_ref), p(self.d_coff), p(self.d_chain), p(video_scaler),
                                                                  p(d_mask), p(out), p(self.ws), self.stream),
                    "dof_preprocess_order_stats")
        return out

    def fit_global(self, ystat_all: torch.Tensor) -> torch.Tensor:
        scaler = torch.empty(self.n_cols, 2, dtype=torch.float64, device=self.device)
        p = self._ptr
        _capi.check(self.lib, self.lib.dof_preprocess_fit_global(ctypes.byref(self.dims), int(ystat_all.shape[0]), p(self.d_kind),
                                                                 p(ystat_all.contiguous()), p(scaler), self.stream),
                    "dof_preprocess_fit_global")
        return scaler

    def tables(self, mask, scaler: Optional[torch.Tensor]):
        """Run the whole pipeline on these videos; ``scaler`` given = apply it instead of fitting."""
        n_node, n_edge, n_ang = self.counts
        self.dims.fit_global = int(scaler is None)
        d_scaler = scaler if scaler is not None else torch.empty(self.n_cols, 2, dtype=torch.float64, device=self.device)
        d_mask = self._dev(mask) if (mask is not None and scaler is None) else None
        node = torch.empty(self.n_frames, n_node, dtype=torch.float32, device=self.device)
        edge = torch.empty(self.n_frames, n_edge, dtype=torch.float32, device=self.device)
        ang = torch.empty(self.n_frames, n_ang, dtype=torch.float32, device=self.device) if n_ang else None
        sizes = torch.empty(len(self.lengths), len(self.plan.animal_ids) + 1, dtype=torch.float64, device=self.device)
        vsc = torch.empty(len(self.lengths), self.n_cols, 2, dtype=torch.float64, device=self.device)
        p = self._ptr
        _capi.check(self.lib, self.lib.dof_preprocess_tables(ctypes.byref(self.dims), p(self.raw), p(self.d_off), p(self.d_kind),
                                                             p(self.d_ref), p(self.d_coff), p(self.d_chain), p(self.d_out), p(d_mask),
                                                             p(d_scaler), p(sizes), p(vsc), p(node), p(edge), p(ang), p(self.ws),
                                                             self.stream), "dof_preprocess_tables")
        return node, edge, ang, sizes, vsc, d_scaler


def preprocess_tables(tables: Dict[str, np.ndarray], columns: Sequence, animal_ids, node_columns: Sequence,
                      edge_columns: Sequence, angle_columns: Sequence = (), *, scale: str = "standard",
                      samples_max: int = 227272, dist_standardize: Optional[str] = "groupwise",
                      speed_standardize: Optional[str] = "groupwise", coord_standardize: Optional[str] = "groupwise",
                      log_distances: bool = True, interpolate_normalized: float = 10, pretrained_scaler: Optional[dict] = None,
                      filter_low_variance=False, inter_scale: str = "mean", device="cuda", lib=None,
                      raw_device: Optional[torch.Tensor] = None, shard_videos: bool = False) -> PreprocessedTables:
    """``TableDict.preprocess`` on the device, ``scale`` "standard", "minmax" or "robust" (utils.py:2570
    ``_pp_make_scaler``; "robust" = exact medians / quartiles by radix selection).  ``filter_low_variance`` (utils.py:2604):
    a column is dropped where its raw variance (pandas ``var``, ddof 1, NaNs skipped) is not above the threshold;
    the device path covers the case in which every video drops the SAME columns (the reference otherwise scales
    tables with differing column sets per video, which its own window extraction cannot stack) and raises otherwise.  ``tables``: {video key: (frames, C) float64
    array or DataFrame}; ``columns``: the C labels; ``node_columns`` / ``edge_columns`` / ``angle_columns``: the labels
    the frame tables keep, in output order (get_graph_dataset's node_sorting / edge_sorting / angle_sorting indices).

    ``shard_videos=True`` under an initialised ``torch.distributed`` group: rank r preprocesses videos r, r+world, ...
    (sorted key order); the ranks exchange the per-video statistics the global scalers are fitted on (one all-gather
    of (videos, C, 3) float64), fit identical scalers, finish their own videos and all-gather the frame tables, so
    every rank ends up with the tables of ALL videos -- bit-identical to the single-process result."""
    if scale not in ("standard", "minmax", "robust"):
        raise ValueError(f"Invalid scaler: {scale}. Choose from {{'standard', 'minmax', 'robust'}}")   # utils.py:2572-2573
    for m in (dist_standardize, speed_standardize, coord_standardize):
        if m not in _capi.PP_MODES:
            raise ValueError("standardisation modes are 'per_column', 'groupwise' or None")
    if lib is None:
        from ._lib import load_hip_library
        lib = load_hip_library()
    device = torch.device(device)
    columns = list(columns)
    if len(columns) > _capi.PP_MAX_COLS:
        raise ValueError(f"at most {_capi.PP_MAX_COLS} table columns")
    arrays, keys = [], []
    for k in sorted(tables.keys()):
        t = tables[k]
        t = np.asarray(t.to_numpy(float) if hasattr(t, "to_numpy") else t, dtype=np.float64)
        if t.ndim != 2 or t.shape[1] != len(columns):
            raise ValueError(f"table {k!r} has shape {t.shape}, expected (frames, {len(columns)})")
        if t.shape[0] == 0 or (np.isnan(t[0]).all() and np.isnan(t).all()):
            continue   # data.py / utils.py:2694-2697: tables without a single value are dropped
        arrays.append(t)
        keys.append(k)
    if not arrays:
        raise ValueError("no table holds any value")
    plan = column_plan(columns, animal_ids)
    where = {c: i for i, c in enumerate(columns)}
    out_cols = np.array([where[c] for c in list(node_columns) + list(edge_columns) + list(angle_columns)], dtype=np.int32)
    lengths = [a.shape[0] for a in arrays]
    video_off = np.zeros(len(arrays) + 1, dtype=np.int64)
    video_off[1:] = np.cumsum(lengths)
    modes = {"speed": speed_standardize, "dist": dist_standardize, "coord": coord_standardize}
    keep = None
    if filter_low_variance:
        keep, raw_device = _filter_low_variance(lib, device, arrays, columns, plan, modes, filter_low_variance, raw_device)
    n_node, n_edge, n_ang = len(node_columns), len(edge_columns), len(angle_columns)
    fit_global = pretrained_scaler is None
    mask = sample_mask(lengths, samples_max) if fit_global else None
    scaler_in = None if fit_global else torch.from_numpy(_scaler_from_dict(
        pretrained_scaler, plan.kinds, modes, scale, None if keep is None else keep.any(axis=0))).to(device)
    common = dict(plan=plan, out_cols=out_cols, n_node=n_node, n_edge=n_edge, n_ang=n_ang, modes=modes, log_distances=log_distances,
                  inter_scale=inter_scale, scale=scale,
                  clip=interpolate_normalized if scale == "standard" else 0)   # utils.py:2993: only "standard" clips
    import torch.distributed as dist
    world = dist.get_world_size() if (shard_videos and dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        call = _Call(lib, device, arrays, raw_device=raw_device, keep=keep, **common)
        if scale == "robust":
            node, edge, ang, sizes, vsc, d_scaler = _robust_tables(call, plan, modes, mask, scaler_in)
        else:
            node, edge, ang, sizes, vsc, d_scaler = call.tables(mask, scaler_in)
    else:
        node, edge, ang, sizes, vsc, d_scaler = _sharded(lib, device, arrays, video_off, mask, scaler_in, world, dist.get_rank(),
                                                         len(plan.animal_ids), common, keep, modes)
    scaler = pretrained_scaler if not fit_global else _scaler_to_dict(d_scaler.cpu().numpy(), plan.kinds, modes, bool(log_distances),
                                                                               scale, None if keep is None else keep.any(axis=0))
    return PreprocessedTables(node, edge, ang, video_off, keys, scaler, sizes, vsc, columns)


def robust_center_scale(rows: np.ndarray, scaled: np.ndarray) -> np.ndarray:
    """(..., C, 7) order-statistics rows -> (..., C, 2) (center_, scale_) of sklearn's RobustScaler: np.nanmedian (mean of the
    two middle values) and np.nanpercentile(25, 75) with numpy's linear rule (a + (b - a) t below t = 1/2, b - (b - a)(1 - t)
    from there on), inter-quartile ranges below 10 eps replaced by 1; identity where ``scaled`` (C,) is False."""
    n = rows[..., 0]
    with np.errstate(all="ignore"):
        center = (rows[..., 1] + rows[..., 2]) / 2.0
        m = np.maximum(n - 1, 0)

        def lerp(a, b, t):
            d = b - a
            return np.where(t >= 0.5, b - d * (1 - t), a + d * t)

        q25 = lerp(rows[..., 3], rows[..., 4], (m % 4) / 4.0)
        q75 = lerp(rows[..., 5], rows[..., 6], ((3 * m) % 4) / 4.0)
        scale = q75 - q25
        scale = np.where(scale < 10 * np.finfo(np.float64).eps, 1.0, scale)
    out = np.stack([center, scale], axis=-1)
    out[..., ~scaled, 0] = 0.0
    out[..., ~scaled, 1] = 1.0
    return out


def _robust_sections(plan, modes):
    """(C,) masks: columns scaled per video, columns scaled by the global scaler."""
    K = _capi.PP_KINDS
    on = {K["speed"]: modes["speed"], K["dist_inner"]: modes["dist"], K["dist_intra"]: modes["dist"], K["coord"]: modes["coord"]}
    per_video = np.array([k != K["coord"] and on.get(int(k)) is not None for k in plan.kinds])
    globally = np.array([on.get(int(k)) is not None for k in plan.kinds])
    return per_video, globally


def _robust_tables(call: "_Call", plan, modes, mask, scaler_in):
    """scale="robust" on one process: per-video order statistics -> per-video (median, IQR); order statistics of the
    per-video-scaled sampled rows of all videos -> global (median, IQR); then the common output pass."""
    per_video, globally = _robust_sections(plan, modes)
    vs = torch.from_numpy(robust_center_scale(call.order_stats(None, None).cpu().numpy(), per_video)).to(call.device)
    if scaler_in is None:
        scaler_in = torch.from_numpy(robust_center_scale(call.order_stats(vs, mask).cpu().numpy(), globally)).to(call.device)
    call.video_scaler_in = vs          # kept alive for the call
    call.dims.video_scaler_in = vs.data_ptr()
    return call.tables(None, scaler_in)


def low_variance_keep(moments: np.ndarray, columns: Sequence, threshold) -> np.ndarray:
    """(videos, C) bool: the columns ``_pp_filter_low_variance`` keeps per video (utils.py:2604-2620): variance with
    ddof = 1 over the non-missing rows strictly above the threshold (NaN variance = dropped), "pheno" columns always."""
    n, m2 = moments[..., 0], moments[..., 2]
    with np.errstate(all="ignore"):
        var = np.where(n > 1, m2 / (n - 1), np.nan)
        keep = var > float(threshold)
    keep[:, [i for i, c in enumerate(columns) if "pheno" in str(c)]] = True
    return keep


def _filter_low_variance(lib, device, arrays, columns, plan, modes, threshold, raw_device):
    """The low-variance filter: raw moments on the device, the (videos x columns) decisions here.  Returns the keep mask
    (videos, C) bool and the uploaded raw tables.  A dropped column is absent from that video's scaling and comes back
    as zeros (utils.py:2966, :3011-3015); angle columns are set aside before the filter runs (:2962-2964)."""
    n_cols = len(columns)
    probe = _Call(lib, device, arrays, plan=ColumnPlan(plan.kinds, np.zeros((0, 4), np.int32), np.zeros(n_cols + 1, np.int32),
                                                       np.zeros((0, 4), np.int32), []),
                  out_cols=np.zeros(0, np.int32), n_node=0, n_edge=0, n_ang=0, modes={"speed": None, "dist": None, "coord": None},
                  log_distances=False, inter_scale="mean", clip=0, raw_device=raw_device)
    keep = low_variance_keep(probe.raw_moments().cpu().numpy(), columns, threshold)
    K = _capi.PP_KINDS
    keep[:, plan.kinds == K["angle"]] = True
    scaled = np.isin(plan.kinds, [K["coord"], K["speed"], K["dist_inner"], K["dist_intra"]])
    if not (keep[:, plan.kinds != K["angle"]]).any(axis=1).all():
        raise AssertionError("Error! During preprocessing the entire table was filtered out due to low variance!")   # utils.py:2615
    where = {c: i for i, c in enumerate(columns)}
    for v in range(keep.shape[0]):
        for i in np.flatnonzero(scaled & keep[v]):
            c = columns[i]
            if plan.kinds[i] in (K["dist_inner"], K["dist_intra"]):
                gone = [bp for bp in c if not keep[v, where[bp]]]
                if gone:   # the reference's .loc[:, (bp1, bp2)] on the filtered table
                    raise KeyError(f"distance column {c} needs the speed column {gone[0]!r}, which filter_low_variance dropped")
            bp = c[0] if _is_pair(c) else c
            if plan.kinds[i] != K["coord"] and not any(keep[v, where[(b, ax)]] for b in ((bp,) if isinstance(c, str) else c)
                                                       for ax in ("x", "y") if (b, ax) in where):
                raise NotImplementedError(f"filter_low_variance dropped both coordinates of a body part of column {c}: the "
                                          "reference then stops treating the column as a speed / distance")
    for name, kinds_of in (("speed", ("speed",)), ("dist", ("dist_inner", "dist_intra")), ("coord", ("coord",))):
        if modes[name] == "per_column":   # one scaler feature per column: sklearn refuses tables with other column sets
            sel = np.isin(plan.kinds, [K[k] for k in kinds_of])
            if (keep[:, sel] != keep[:1, sel]).any():
                raise ValueError(f"filter_low_variance keeps different {name} columns in different videos; the per-column "
                                 "global scaler (fitted on the first video's columns) cannot be applied to them")
    return keep, probe.raw


def _all_gather_rows(local: torch.Tensor, rows_per_rank: List[int]) -> List[torch.Tensor]:
    """all_gather of tensors whose first dimension differs per rank (padded to the longest)."""
    import torch.distributed as dist
    longest = max(rows_per_rank)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in rows_per_rank]
    dist.all_gather(parts, pad)
    return [p[:n] for p, n in zip(parts, rows_per_rank)]


def _sharded(lib, device, arrays, video_off, mask, scaler_in, world, rank, n_animals, common, keep=None, modes=None):
    """Videos i = rank, rank + world, ... on this rank; two collectives (statistics rows, finished tables).

    scale="robust": medians and quartiles do not merge, so the split is by what each selection ranges over.  The
    per-video order statistics (every row of every video -- the bulk of the selection work) stay with the video's
    owner and their (median, IQR) rows are all-gathered; the pooled selection that fits the GLOBAL scaler ranges over
    the sampled rows of all videos, scaled per video with whole-video size factors: every rank runs it over all
    tables (one extra upload of tables the rank already holds on the host; the selection is deterministic, so every
    rank gets the single-process scaler bit for bit and no histogram exchange is needed)."""
    n_videos, n_cols = len(arrays), len(common["plan"].kinds)
    owner = [list(range(r, n_videos, world)) for r in range(world)]
    mine = owner[rank]
    lengths = np.diff(video_off)
    call = _Call(lib, device, [arrays[i] for i in mine], keep=None if keep is None else keep[mine], **common) if mine else None
    local_mask = None
    if mask is not None and mine:
        local_mask = np.concatenate([mask[video_off[i]:video_off[i + 1]] for i in mine])
    order = [i for r in range(world) for i in owner[r]]          # global index of the rows as gathered
    back = torch.from_numpy(np.argsort(np.array(order))).to(device)
    robust = common["scale"] == "robust"
    if robust:
        plan = common["plan"]
        per_video, globally = _robust_sections(plan, modes)
        vs_local = torch.from_numpy(robust_center_scale(call.order_stats(None, None).cpu().numpy(), per_video)).to(device) if mine \
            else torch.zeros(0, n_cols, 2, dtype=torch.float64, device=device)
        if scaler_in is None:
            vs_all = torch.cat(_all_gather_rows(vs_local, [len(o) for o in owner]))[back]   # global video order
            full = _Call(lib, device, arrays, keep=keep, **common)
            scaler_in = torch.from_numpy(robust_center_scale(full.order_stats(vs_all.contiguous(), mask).cpu().numpy(), globally)).to(device)
            del full
        if mine:
            call.video_scaler_in = vs_local   # kept alive for the call
            call.dims.video_scaler_in = vs_local.data_ptr()
    elif scaler_in is None:
        ystat = call.video_stats(local_mask) if mine else torch.zeros(0, n_cols, _capi.PP_STAT_DOUBLES, dtype=torch.float64,
                                                                      device=device)
        ystat_all = torch.cat(_all_gather_rows(ystat, [len(o) for o in owner]))[back]      # global video order
        helper = call if call is not None else _Call(lib, device, [arrays[0][:1]], **common)
        scaler_in = helper.fit_global(ystat_all)
    n_node, n_edge, n_ang = common["n_node"], common["n_edge"], common["n_ang"]
    if mine:
        node, edge, ang, sizes, vsc, _ = call.tables(None, scaler_in)
    else:
        node = torch.zeros(0, n_node, device=device)
        edge = torch.zeros(0, n_edge, device=device)
        ang = torch.zeros(0, n_ang, device=device) if n_ang else None
        sizes = torch.zeros(0, n_animals + 1, dtype=torch.float64, device=device)
        vsc = torch.zeros(0, n_cols, 2, dtype=torch.float64, device=device)
    rows = [int(sum(lengths[i] for i in o)) for o in owner]

    def assemble(local, per_video: bool):
        parts = _all_gather_rows(local, [len(o) for o in owner] if per_video else rows)
        if per_video:
            return torch.cat(parts)[back]
        pieces = {}
        for r, o in enumerate(owner):
            off = 0
            for i in o:
                pieces[i] = parts[r][off:off + int(lengths[i])]
                off += int(lengths[i])
        return torch.cat([pieces[i] for i in range(n_videos)])

    return (assemble(node, False), assemble(edge, False), assemble(ang, False) if n_ang else None, assemble(sizes, True),
            assemble(vsc, True), scaler_in)


def graph_dataset_from_tables(tables: Dict[str, np.ndarray], columns: Sequence, animal_ids=("",), *, graph_preset: str = "deepof_14",
                              window_size: int = 25, window_step: int = 1, test_keys: Sequence[str] = (), device="cuda", lib=None,
                              dist_standardize: Optional[str] = "per_column", speed_standardize: Optional[str] = "per_column",
                              coord_standardize: Optional[str] = "per_column", **preprocess_kw):
    """``Coordinates.get_graph_dataset(preprocess=True)`` (/root/reference/deepof/data.py:2644-2906) from merged raw
    tables: body-part graph -> sorted node / edge feature columns (:2797-2835; nodes absent from the tables are
    dropped from the graph, :2776-2781) -> device preprocessing -> window datasets over the resident frame tables.

    Returns ``((train, val), meta_info, adjacency, pre)``: ``train`` / ``val`` go to ``train_deepof_model`` as
    ``preprocessed_object``; ``pre.global_scaler`` is the fitted scaler.  ``test_keys`` are the held-out videos (the
    reference picks them at random in ``get_training_set``; that choice is control plane and stays with the caller)."""
    from .dataset import WindowDataset
    from .graph import adjacency_from_graph, bodypart_graph
    if lib is None:
        from ._lib import load_hip_library
        lib = load_hip_library()
    columns = list(columns)
    have = set(columns)
    nodes, edges = bodypart_graph(list(animal_ids), graph_preset)
    nodes = [n for n in nodes if (n, "x") in have and (n, "y") in have]
    keep = set(nodes)
    edges = [e for e in edges if e[0] in keep and e[1] in keep]
    missing = [n for n in nodes if n not in have]
    if missing:
        raise KeyError(f"speed columns missing for {missing}")
    by_pair = {frozenset(c): c for c in columns if _is_pair(c) and c[1] not in ("x", "y")}
    edge_cols = []
    for e in edges:
        if frozenset(e) not in by_pair:
            raise KeyError(f"distance column for edge {e} missing")
        edge_cols.append(by_pair[frozenset(e)])
    node_cols = [(n, "x") for n in nodes] + [(n, "y") for n in nodes] + nodes
    pre = preprocess_tables(tables, columns, animal_ids, node_cols, edge_cols, (), dist_standardize=dist_standardize,
                            speed_standardize=speed_standardize, coord_standardize=coord_standardize, device=device, lib=lib,
                            **preprocess_kw)
    test = [k for k in pre.keys if k in set(test_keys)]
    train = WindowDataset.from_device_tables(pre, window_size, window_step, lib, keys=[k for k in pre.keys if k not in set(test)])
    val = WindowDataset.from_device_tables(pre, window_size, window_step, lib, keys=test) if test else train
    meta = {"node_columns": node_cols, "edge_columns": edge_cols, "angle_columns": [],
            "shape_train": [(len(train),) + train.x_shape[:1] + (len(node_cols),), (len(train),) + train.a_shape[:1] + (len(edge_cols),)],
            "dist_standardize": dist_standardize, "speed_standardize": speed_standardize, "coord_standardize": coord_standardize}
    return (train, val), meta, adjacency_from_graph(nodes, edges), pre
