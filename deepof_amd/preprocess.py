"""Pose-table preprocessing on the device: the host side of ``dof_preprocess_tables``.

Mirrors ``TableDict.preprocess`` (/root/reference/deepof/data.py:3773-3916, scale="standard") up to window
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
    chain: np.ndarray        # (k, 3) int32
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
    chains: List[List[Tuple[int, int, int]]] = [[] for _ in columns]
    for aid in animal_ids:
        mine = [bp for bp in bodyparts if owner[bp] == aid]
        for bp in mine:
            for lab in ((bp, "x"), (bp, "y"), bp):
                if lab in where:
                    chains[where[lab]].append((code[aid], code[aid], 1))
    for i, c in enumerate(columns):
        if kinds[i] in (_capi.PP_KINDS["dist_inner"], _capi.PP_KINDS["dist_intra"]):
            o1, o2 = owner.get(c[0]), owner.get(c[1])
            entry = (code.get(o1, -1), code.get(o2, -1), int(o1 == o2))
            for bp in c:   # the reference's .loc[:, (bp1, bp2)] hits the two speed columns (module docstring)
                if bp not in where:
                    raise KeyError(f"distance column {c} needs the speed column {bp!r} (as in the reference)")
                chains[where[bp]].append(entry)
    off = np.zeros(len(columns) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(ch) for ch in chains])
    flat = np.array([e for ch in chains for e in ch], dtype=np.int32).reshape(-1, 3)
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


def _scaler_to_dict(per_col: np.ndarray, kinds: np.ndarray, modes: Dict[str, Optional[str]], log_distances: bool) -> Optional[dict]:
    """(C,2) -> the legacy dict layout of GlobalScalerSpec.to_legacy_dict (utils.py:2362-2374), (mean, scale) pairs."""
    out = {"kind": "standard", "speed": None, "dist": None, "dist_inner": None, "dist_intra": None, "coord": None,
           "speed_mode": modes["speed"], "dist_mode": modes["dist"], "coord_mode": modes["coord"], "log_distances": log_distances}

    def cols(names):
        return [i for i, k in enumerate(kinds) if k in [_capi.PP_KINDS[n] for n in names]]

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


def _scaler_from_dict(gs: dict, kinds: np.ndarray, modes: Dict[str, Optional[str]]) -> np.ndarray:
    """Legacy dict (pairs (mean, scale) or fitted sklearn StandardScalers) -> (C,2), identity where nothing applies."""
    per_col = np.tile(np.array([0.0, 1.0]), (len(kinds), 1))

    def pair(v):
        if hasattr(v, "mean_"):
            return np.atleast_1d(np.asarray(v.mean_, dtype=np.float64)), np.atleast_1d(np.asarray(v.scale_, dtype=np.float64))
        return np.atleast_1d(np.asarray(v[0], dtype=np.float64)), np.atleast_1d(np.asarray(v[1], dtype=np.float64))

    def put(name, names, mode):
        idx = [i for i, k in enumerate(kinds) if k in [_capi.PP_KINDS[n] for n in names]]
        if not idx or mode is None or gs.get(name) is None:
            return
        m, s = pair(gs[name])
        if mode == "per_column" and len(m) != len(idx):
            raise ValueError(f"pretrained scaler section {name!r} has {len(m)} columns, the tables have {len(idx)}")
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


def preprocess_tables(tables: Dict[str, np.ndarray], columns: Sequence, animal_ids, node_columns: Sequence,
                      edge_columns: Sequence, angle_columns: Sequence = (), *, scale: str = "standard",
                      samples_max: int = 227272, dist_standardize: Optional[str] = "groupwise",
                      speed_standardize: Optional[str] = "groupwise", coord_standardize: Optional[str] = "groupwise",
                      log_distances: bool = True, interpolate_normalized: float = 10, pretrained_scaler: Optional[dict] = None,
                      filter_low_variance=False, inter_scale: str = "mean", device="cuda", lib=None,
                      raw_device: Optional[torch.Tensor] = None) -> PreprocessedTables:
    """``TableDict.preprocess`` for ``scale="standard"`` on the device.  ``tables``: {video key: (frames, C) float64
    array or DataFrame}; ``columns``: the C labels; ``node_columns`` / ``edge_columns`` / ``angle_columns``: the labels
    the frame tables keep, in output order (get_graph_dataset's node_sorting / edge_sorting / angle_sorting indices)."""
    if scale != "standard":
        raise NotImplementedError("only scale='standard' (the reference default) runs on the device")
    if filter_low_variance:
        raise NotImplementedError("filter_low_variance is not supported")
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
    plan = column_plan(columns, animal_ids)
    where = {c: i for i, c in enumerate(columns)}
    out_cols = np.array([where[c] for c in list(node_columns) + list(edge_columns) + list(angle_columns)], dtype=np.int32)
    if len(out_cols) > _capi.PP_MAX_OUT:
        raise ValueError(f"at most {_capi.PP_MAX_OUT} output columns")
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
    lengths = [a.shape[0] for a in arrays]
    video_off = np.zeros(len(arrays) + 1, dtype=np.int64)
    video_off[1:] = np.cumsum(lengths)
    n_frames = int(video_off[-1])
    modes = {"speed": speed_standardize, "dist": dist_standardize, "coord": coord_standardize}
    fit_global = pretrained_scaler is None
    mask = sample_mask(lengths, samples_max) if fit_global else None
    if raw_device is None:
        raw_device = torch.from_numpy(np.concatenate(arrays) if len(arrays) > 1 else arrays[0]).to(device)
    if raw_device.dtype != torch.float64 or tuple(raw_device.shape) != (n_frames, len(columns)):
        raise ValueError("raw_device must be the concatenated float64 tables")

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)

    scaler_host = np.tile(np.array([0.0, 1.0]), (len(columns), 1)) if fit_global else _scaler_from_dict(pretrained_scaler, plan.kinds, modes)
    d_off, d_kind, d_ref = dev(video_off), dev(plan.kinds), dev(plan.size_ref.reshape(-1) if plan.size_ref.size else np.zeros(4, np.int32))
    d_coff = dev(plan.chain_off)
    d_chain = dev(plan.chain.reshape(-1) if plan.chain.size else np.zeros(3, np.int32))
    d_out, d_scaler = dev(out_cols), dev(scaler_host)
    d_mask = dev(mask) if mask is not None else None
    n_node, n_edge, n_ang = len(node_columns), len(edge_columns), len(angle_columns)
    dims = _capi.PreprocDims(n_frames=n_frames, n_videos=len(arrays), n_cols=len(columns), n_animals=len(plan.animal_ids),
                             n_node_cols=n_node, n_edge_cols=n_edge, n_angle_cols=n_ang,
                             speed_mode=_capi.PP_MODES[speed_standardize], dist_mode=_capi.PP_MODES[dist_standardize],
                             coord_mode=_capi.PP_MODES[coord_standardize], log_distances=int(bool(log_distances)),
                             inter_scale=_capi.PP_INTER_SCALE[inter_scale], fit_global=int(fit_global),
                             clip=float(interpolate_normalized or 0))
    ws_bytes = lib.dof_preprocess_workspace_bytes(ctypes.byref(dims))
    if ws_bytes < 0:
        _capi.check(lib, -1, "dof_preprocess_workspace_bytes")
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=device)
    node = torch.empty(n_frames, n_node, dtype=torch.float32, device=device)
    edge = torch.empty(n_frames, n_edge, dtype=torch.float32, device=device)
    ang = torch.empty(n_frames, n_ang, dtype=torch.float32, device=device) if n_ang else None
    sizes = torch.empty(len(arrays), len(plan.animal_ids) + 1, dtype=torch.float64, device=device)
    vsc = torch.empty(len(arrays), len(columns), 2, dtype=torch.float64, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    ptr = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    _capi.check(lib, lib.dof_preprocess_tables(ctypes.byref(dims), ptr(raw_device), ptr(d_off), ptr(d_kind), ptr(d_ref),
                                               ptr(d_coff), ptr(d_chain), ptr(d_out), ptr(d_mask), ptr(d_scaler), ptr(sizes),
                                               ptr(vsc), ptr(node), ptr(edge), ptr(ang), ptr(ws), stream),
                "dof_preprocess_tables")
    scaler = pretrained_scaler if not fit_global else _scaler_to_dict(d_scaler.cpu().numpy(), plan.kinds, modes, bool(log_distances))
    return PreprocessedTables(node, edge, ang, video_off, keys, scaler, sizes, vsc, columns)
