"""Oracle (test infrastructure): window tensor build, NumPy.

Restates
* ``rolling_window``  /root/reference/deepof/utils.py:3354-3377  (stride-`step` sliding windows),
* ``reorder_and_reshape``  /root/reference/deepof/clustering/dataset.py:16-26
  (column blocks [x_1..x_N | y_1..y_N | s_1..s_N] -> (n, W, N, 3)),
* the edge expand ``(n, W, E) -> (n, W, E, 1)``  dataset.py:81,
* ``RecurrentEncoderPT.tf_style_group_reshape``  models_new.py:120-138  (the index *scramble*
  (B,T,G,F)->(B,G,T,F) that is NOT a transpose).
"""
import numpy as np


def rolling_window(table: np.ndarray, window_size: int, window_step: int = 1) -> np.ndarray:
    """(frames, C) -> (n_windows, W, C); window i starts at frame i*step."""
    n = table.shape[0] - window_size + 1
    starts = np.arange(0, n, window_step)
    idx = starts[:, None] + np.arange(window_size)[None, :]
    return table[idx]


def node_windows_to_x(node_windows: np.ndarray) -> np.ndarray:
    """(n, W, 3N) column-block layout -> (n, W, N, 3) float32."""
    n_cols = node_windows.shape[2]
    assert n_cols % 3 == 0
    nn = n_cols // 3
    return (
        node_windows.reshape(node_windows.shape[0], node_windows.shape[1], 3, nn)
        .transpose(0, 1, 3, 2)
        .astype(np.float32)
    )


def edge_windows_to_a(edge_windows: np.ndarray) -> np.ndarray:
    """(n, W, E) -> (n, W, E, 1) float32."""
    return edge_windows[..., None].astype(np.float32)


def gather_windows(node_table, edge_table, starts, window_size):
    """Frame tables + window start frames -> reference-layout batch (x, a).

    node_table (frames, 3N) [x.. y.. s..], edge_table (frames, E); starts int array (B,).
    Equivalent to rolling_window(step=1)[starts] followed by reorder_and_reshape.
    """
    starts = np.asarray(starts, dtype=np.int64)
    idx = starts[:, None] + np.arange(window_size)[None, :]
    return node_windows_to_x(node_table[idx]), edge_windows_to_a(edge_table[idx])


def group_scramble_index(T: int, G: int, F: int) -> np.ndarray:
    """Flat source index (into a (T,G,F) window) for every element of the (G,T,F) output.

    out[g', t', f'] = window.flat[src[g', t', f']].  Closed form: L = (f'*T + t')*G + g';
    (c, t) = divmod(L, T); (g, f) = divmod(c, F); src = (t*G + g)*F + f.
    """
    gp, tp, fp = np.meshgrid(np.arange(G), np.arange(T), np.arange(F), indexing="ij")
    lin = (fp * T + tp) * G + gp
    c, t = np.divmod(lin, T)
    g, f = np.divmod(c, F)
    return ((t * G + g) * F + f).astype(np.int64)


def group_scramble(x: np.ndarray) -> np.ndarray:
    """(B,T,G,F) -> (B,G,T,F), the reference's 'TF-style' grouping."""
    B, T, G, F = x.shape
    src = group_scramble_index(T, G, F)
    return x.reshape(B, -1)[:, src.reshape(-1)].reshape(B, G, T, F)
