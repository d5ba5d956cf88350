"""Golden fixtures for the window tensor build and the body-part graphs, produced by the
REFERENCE's own functions executed in place (build container only):

* ``rolling_window``      /root/reference/deepof/utils.py:3354-3377   (compiled by name from the file where it lies:
                          the module as a whole needs cv2 / numba / ...; nothing is copied)
* ``reorder_and_reshape`` /root/reference/deepof/clustering/dataset.py:16-26 (imported through the shim)
* ``connect_mouse``       /root/reference/deepof/utils.py:416-508 + the sorted node / edge / adjacency assembly of
                          ``get_graph_dataset`` (data.py:2791-2793), evaluated here with networkx as that code does.

Output: windows_graph.npz (inputs + expected outputs; data only).
"""
import ast
import os
import sys
from itertools import combinations

import networkx as nx
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import REF, load_reference  # noqa: E402

R = load_reference()


def compile_from_utils(names):
    path = REF + "/deepof/utils.py"
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names)
    ns = dict(np=np, nx=nx, combinations=combinations)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def main():
    rolling_window, connect_mouse = compile_from_utils(["rolling_window", "connect_mouse"])
    out = {}
    rng = np.random.default_rng(17)
    # ---- windows: (frames, window, step, nodes, edges)
    for ci, (F, W, step, N, E) in enumerate([(40, 25, 1, 14, 14), (31, 7, 3, 5, 4), (60, 50, 5, 28, 32), (25, 25, 1, 3, 2),
                                             (33, 24, 2, 11, 12)]):
        nodes = rng.standard_normal((F, 3 * N))
        edges = rng.standard_normal((F, E))
        wn = np.ascontiguousarray(rolling_window(nodes, W, step))
        we = np.ascontiguousarray(rolling_window(edges, W, step))
        out[f"w{ci}::cfg"] = np.array([F, W, step, N, E], dtype=np.int64)
        out[f"w{ci}::node_table"], out[f"w{ci}::edge_table"] = nodes, edges
        out[f"w{ci}::node_windows"], out[f"w{ci}::edge_windows"] = wn, we
        out[f"w{ci}::x"] = R.D.reorder_and_reshape(wn).astype(np.float32)          # dataset.py:16-26, then the fp32 cast
        out[f"w{ci}::a"] = np.expand_dims(we, -1).astype(np.float32)               # dataset.py:81 / :204-206
    # ---- graphs
    gi = 0
    for preset in ("deepof_14", "deepof_11", "deepof_8"):
        for ids in ([""], ["B", "W"], ["A", "B", "C"]):
            if preset != "deepof_14" and len(ids) == 3:
                continue
            graph = connect_mouse(animal_ids=list(ids), graph_preset=preset)
            nodes = sorted(graph.nodes())
            edges = sorted(tuple(sorted(e)) for e in graph.edges())
            adj = np.asarray(nx.adjacency_matrix(graph, nodelist=nodes).todense()).astype(np.float32)
            out[f"g{gi}::preset"] = np.array(preset)
            out[f"g{gi}::ids"] = np.array(ids)
            out[f"g{gi}::nodes"] = np.array(nodes)
            out[f"g{gi}::edges"] = np.array(edges)
            out[f"g{gi}::adj"] = adj
            gi += 1
    out["n_window_cases"], out["n_graph_cases"] = np.int64(5), np.int64(gi)
    np.savez_compressed(os.path.join(HERE, "windows_graph.npz"), **out)
    print("windows_graph.npz", os.path.getsize(os.path.join(HERE, "windows_graph.npz")))


if __name__ == "__main__":
    main()
