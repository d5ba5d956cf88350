"""Body-part graph presets and the CensNet graph operators.

Restates (not copies) the connectivity presets of the reference's ``connect_mouse``
(/root/reference/deepof/utils.py:416-508) and the adjacency/edge ordering used by
``Coordinates.get_graph_dataset`` (/root/reference/deepof/data.py:2791-2793): nodes in
sorted-name order, edges as sorted (min,max) name pairs, which is also the column order of
the incidence matrix (upper-triangular non-zeros, row-major;
/root/reference/deepof/clustering/censNetConv_pt.py:296-332).

The graph operators (normalised adjacency with self-loops, incidence, line-graph filter)
follow censNetConv_pt.py:161-370 and are tiny one-off host computations (numpy, float32 out).
"""
from __future__ import annotations

from itertools import combinations
from typing import Dict, List, Sequence, Tuple

import numpy as np

_PRESETS: Dict[str, Dict[str, List[str]]] = {
    "deepof_14": {
        "Nose": ["Left_ear", "Right_ear"],
        "Spine_1": ["Center", "Left_ear", "Right_ear"],
        "Center": ["Left_fhip", "Right_fhip", "Spine_2"],
        "Spine_2": ["Left_bhip", "Right_bhip", "Tail_base"],
        "Tail_base": ["Tail_1"],
        "Tail_1": ["Tail_2"],
        "Tail_2": ["Tail_tip"],
    },
    "deepof_11": {
        "Nose": ["Left_ear", "Right_ear"],
        "Spine_1": ["Center", "Left_ear", "Right_ear"],
        "Center": ["Left_fhip", "Right_fhip", "Spine_2"],
        "Spine_2": ["Left_bhip", "Right_bhip", "Tail_base"],
    },
    "deepof_8": {
        "Nose": ["Left_ear", "Right_ear"],
        "Center": ["Left_fhip", "Right_fhip", "Tail_base", "Left_ear", "Right_ear"],
        "Tail_base": ["Tail_tip"],
    },
}


def bodypart_graph(
    animal_ids: Sequence[str] = ("",), preset: str = "deepof_14"
) -> Tuple[List[str], List[Tuple[str, str]]]:
    """Return (sorted node names, sorted edge name pairs) for one or more animals."""
    if isinstance(animal_ids, str):
        animal_ids = [animal_ids]
    animal_ids = list(animal_ids) or [""]
    nodes, edges = set(), set()

    def _add(a, b):
        nodes.update((a, b))
        edges.add((a, b) if a < b else (b, a))

    for aid in animal_ids:
        pre = f"{aid}_" if aid else ""
        for src, dsts in _PRESETS[preset].items():
            for d in dsts:
                _add(pre + src, pre + d)
    for a, b in combinations(animal_ids, 2):
        _add(f"{a}_Nose", f"{b}_Nose")
        _add(f"{a}_Tail_base", f"{b}_Tail_base")
        _add(f"{a}_Nose", f"{b}_Tail_base")
        _add(f"{b}_Nose", f"{a}_Tail_base")
    return sorted(nodes), sorted(edges)


def adjacency_from_graph(nodes: Sequence[str], edges: Sequence[Tuple[str, str]]) -> np.ndarray:
    idx = {n: i for i, n in enumerate(nodes)}
    adj = np.zeros((len(nodes), len(nodes)), dtype=np.float32)
    for a, b in edges:
        adj[idx[a], idx[b]] = 1.0
        adj[idx[b], idx[a]] = 1.0
    return adj


def make_meta_info(nodes: Sequence[str], edges: Sequence[Tuple[str, str]]) -> dict:
    """``meta_info`` dict in the shape the reference trainer expects (data.py:2795-2800)."""
    node_columns = [(n, "x") for n in nodes] + [(n, "y") for n in nodes] + list(nodes)
    return {"node_columns": node_columns, "edge_columns": list(edges)}


def _gcn_filter(a: np.ndarray) -> np.ndarray:
    a_hat = a.astype(np.float64) + np.eye(a.shape[0])
    deg = a_hat.sum(axis=1)
    deg[deg == 0] = 1.0
    dinv = deg ** -0.5
    return (dinv[:, None] * a_hat) * dinv[None, :]


def incidence_matrix(adjacency: np.ndarray) -> np.ndarray:
    """N x E incidence; edge e = e-th non-zero of triu(adjacency) in row-major order."""
    tri = np.triu(np.asarray(adjacency))
    rows, cols = np.nonzero(tri)
    inc = np.zeros((adjacency.shape[0], rows.shape[0]), dtype=np.float32)
    e = np.arange(rows.shape[0])
    inc[rows, e] = 1.0
    inc[cols, e] = 1.0
    return inc


def censnet_operators(adjacency: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(laplacian N x N, edge_laplacian E x E, incidence N x E), float32."""
    adjacency = np.asarray(adjacency, dtype=np.float64)
    lap = _gcn_filter(adjacency)
    inc = incidence_matrix(adjacency)
    line = inc.T.astype(np.float64) @ inc.astype(np.float64) - 2.0 * np.eye(inc.shape[1])
    edge_lap = _gcn_filter(line)
    return lap.astype(np.float32), edge_lap.astype(np.float32), inc


def edge_index_from_graph(nodes: Sequence[str], edges: Sequence[Tuple[str, str]]):
    """(global (E,2), within-animal (E_local,2)) int32 node-index pairs in ``edges`` order; an edge is local
    when both endpoints carry the same animal prefix (text before the first underscore).
    Mirrors reference training.py:1936-2004 _build_edge_from_metainfo."""
    idx = {n: i for i, n in enumerate(nodes)}
    key = [n.split("_", 1)[0] if "_" in n else "" for n in nodes]
    glob = np.array([(idx[u], idx[v]) for u, v in edges], dtype=np.int32).reshape(-1, 2)
    local = np.array([e for e in glob.tolist() if key[e[0]] == key[e[1]]], dtype=np.int32).reshape(-1, 2)
    return glob, local
