"""Host cost of taking the REFERENCE's input format at C2 size: {video: (node windows (n, W, 3N), edge windows (n, W, E))}
(40 videos x 15,000 frames, W = 25, stride 1: 599,040 windows, 3.35 GB of float32) -> WindowDataset.from_preprocessed,
which checks every element of the overlap and folds the windows back into frame tables (1 / W of the bytes are uploaded).
Runs on any device (the fold is numpy); python tools/bench_ingest.py [cpu|cuda]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepof_amd.dataset import WindowDataset  # noqa: E402


def main():
    dev = sys.argv[1] if len(sys.argv) > 1 else ("cuda" if torch.cuda.is_available() else "cpu")
    V, F, W, N, E = 40, 15_000, 25, 14, 14
    rng = np.random.default_rng(0)
    pre = {}
    for v in range(V):
        tn = np.cumsum(rng.standard_normal((F, 3 * N)).astype(np.float32) * 0.05, axis=0)
        te = np.abs(np.cumsum(rng.standard_normal((F, E)).astype(np.float32) * 0.05, axis=0))
        wn = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(tn, W, axis=0).transpose(0, 2, 1))
        we = np.ascontiguousarray(np.lib.stride_tricks.sliding_window_view(te, W, axis=0).transpose(0, 2, 1))
        pre[f"vid{v:02d}"] = (wn, we)
    gb = sum(a.nbytes + b.nbytes for a, b in pre.values()) / 1e9
    lib = object()  # any non-None value selects the table-rebuilding path; the library itself is only used by fetch()
    if dev == "cuda":
        from deepof_amd._lib import load_hip_library
        lib = load_hip_library()
    t0 = time.perf_counter()
    ds = WindowDataset.from_preprocessed(pre, torch.device(dev), lib)
    if dev == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert ds.node_table is not None and ds.length == V * (F - W + 1), "windows were not recognised as sliding windows"
    print(json.dumps({"videos": V, "windows": ds.length, "input_gb": round(gb, 3), "device": dev, "ingest_s": round(dt, 3),
                      "resident_mb": round((ds.node_table.numel() + ds.edge_table.numel()) * 4 / 1e6, 1),
                      "host_threads": torch.get_num_threads()}))


if __name__ == "__main__":
    main()
