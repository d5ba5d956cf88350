"""Host side of the contrastive augmentations: graph pre-computation and the random draws.

The reference draws inside its augmentation functions and then transforms tensors with torch ops
(training.py:2128-2403).  Here the draws are resolved on the host / with torch's device generator in the
same order and with the same distributions, and the transformation itself is one HIP kernel
(``dof_contrastive_views``) that takes the resolved draws (see ``engine.contrastive_views``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .config import ContrastiveCfg


def edge_index_from_meta(meta_info: dict, n_nodes: int) -> Tuple[np.ndarray, np.ndarray]:
    """(global (E,2), within-animal (E_local,2)) int32 node-index pairs from ``meta_info`` (node names are the
    first n_nodes ``(bodypart, 'x')`` columns; an edge is local if both names share the prefix before the first
    underscore).  Reference: training.py:1936-2004 _build_edge_from_metainfo."""
    if "node_columns" not in meta_info or "edge_columns" not in meta_info:
        raise RuntimeError("meta_info must contain 'node_columns' and 'edge_columns'.")
    names: List[str] = []
    for c in meta_info["node_columns"]:
        if isinstance(c, tuple) and len(c) == 2 and c[1] == "x":
            names.append(c[0])
            if len(names) == n_nodes:
                break
    if len(names) != n_nodes:
        raise RuntimeError(f"Failed to infer {n_nodes} node names from meta_info['node_columns']. Got {len(names)}.")
    idx = {n: i for i, n in enumerate(names)}
    pairs = []
    for u, v in meta_info["edge_columns"]:
        if u not in idx or v not in idx:
            raise RuntimeError(f"Edge ({u},{v}) contains node(s) not found in inferred node list.")
        pairs.append((idx[u], idx[v]))
    key = [n.split("_", 1)[0] if "_" in n else "" for n in names]
    glob = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    local = np.asarray([p for p in pairs if key[p[0]] == key[p[1]]], dtype=np.int32).reshape(-1, 2)
    return glob, local


@dataclass
class RotationPrecomp:
    """Joint-like rotation candidates (training.py:2051-2125): triplets (a, b, c) of a centre b with two of its
    neighbours, and the node sets hanging off a / off c when the path through b is cut."""
    triplets: List[Tuple[int, int, int]]
    centers: List[int]
    branches_a: List[List[int]]
    branches_c: List[List[int]]


def build_rotation_precomp(edge_index_local: Sequence[Sequence[int]], n_nodes: int) -> RotationPrecomp:
    nbrs: List[List[int]] = [[] for _ in range(n_nodes)]
    for u, v in edge_index_local:
        nbrs[int(u)].append(int(v))
        nbrs[int(v)].append(int(u))

    def branch(center: int, side: int) -> List[int]:
        seen = {side}
        todo = [side]
        while todo:
            u = todo.pop()
            for v in nbrs[u]:
                if v != center and v not in seen:
                    seen.add(v)
                    todo.append(v)
        return sorted(seen)

    pc = RotationPrecomp([], [], [], [])
    for b in range(n_nodes):
        nb = nbrs[b]
        for i in range(len(nb)):
            for j in range(i + 1, len(nb)):
                pc.triplets.append((nb[i], b, nb[j]))
                pc.centers.append(b)
                pc.branches_a.append(branch(b, nb[i]))
                pc.branches_c.append(branch(b, nb[j]))
    return pc


def draw_augmentation(batch: int, t_full: int, n_nodes: int, cfg: ContrastiveCfg, precomp: RotationPrecomp,
                      device, generator: torch.Generator = None, host_generator: torch.Generator = None) -> dict:
    """One set of draws for ``dof_contrastive_views`` -- the distributions and the order of
    _make_augmented_view (time shift, rotations, interpolation, noise).  Per-sample draws live on ``device``;
    the handful of per-batch choices (which triplets, which side) are drawn on the host."""
    B, half = int(batch), t_full // 2
    kw = dict(device=device, generator=generator)
    out = {}
    # time shift (training.py:2128-2165)
    base = (t_full - half) // 2
    gate = torch.rand(B, **kw) < cfg.aug_p_shift
    mag = torch.randint(cfg.aug_min_shift, cfg.aug_max_shift + 1, (B,), **kw)
    sgn = torch.randint(0, 2, (B,), **kw) * 2 - 1
    out["start"] = (base + mag * sgn * gate.long()).clamp(0, t_full - half).to(torch.int32)
    # rotations (training.py:2167-2250): up to n_rot triplets, every centre at most twice
    M = len(precomp.triplets)
    out["rot_pivot"], out["rot_nodes"] = [], []
    if cfg.aug_n_rot > 0 and cfg.aug_max_rot > 0 and cfg.aug_p_rot > 0 and M > 0:
        gate = (torch.rand(B, **kw) < cfg.aug_p_rot).float()
        used = [0] * n_nodes
        chosen = []
        for k in torch.randperm(M, generator=host_generator).tolist():
            if used[precomp.centers[k]] >= 2:
                continue
            used[precomp.centers[k]] += 1
            chosen.append(k)
            if len(chosen) >= cfg.aug_n_rot:
                break
        thetas = []
        max_rad = float(cfg.aug_max_rot) * math.pi / 180.0
        for k in chosen:
            side_a = bool(torch.rand((), generator=host_generator) < 0.5)
            out["rot_pivot"].append(precomp.centers[k])
            out["rot_nodes"].append(precomp.branches_a[k] if side_a else precomp.branches_c[k])
            thetas.append((torch.rand(B, **kw) * 2.0 - 1.0) * max_rad * gate)
        out["theta"] = torch.stack(thetas) if thetas else torch.zeros(0, B, device=device)
    # one linearly interpolated segment (training.py:2299-2370); needs frames t0-1 and t0+len inside the view
    if cfg.aug_max_interp > 0 and cfg.aug_p_interp > 0 and half >= 3:
        gate = torch.rand(B, **kw) < cfg.aug_p_interp
        ln = torch.randint(cfg.aug_min_interp, cfg.aug_max_interp + 1, (B,), **kw)
        t0 = torch.minimum(torch.randint(1, half - 1, (B,), **kw), (half - ln - 1).clamp_min(1))
        ln = torch.minimum(ln, half - 1 - t0).clamp_min(0)  # the reference indexes past the view here; keep it inside
        out["interp_t0"] = t0.to(torch.int32)
        out["interp_len"] = (ln * gate.long()).to(torch.int32)
    # per-node offsets on x or y, and on speed (training.py:2253-2296)
    if cfg.aug_noise_sigma > 0 and cfg.aug_p_noise > 0:
        gate = (torch.rand(B, **kw) < cfg.aug_p_noise).float().view(B, 1)
        axis = torch.randint(0, 2, (B, n_nodes), **kw)
        off = cfg.aug_noise_sigma * torch.randn((B, n_nodes), **kw) * gate
        ds = cfg.aug_noise_sigma * torch.randn((B, n_nodes), **kw) * gate
        out["noise"] = torch.stack([off * (axis == 0).float(), off * (axis == 1).float(), ds], dim=-1)
    return out
