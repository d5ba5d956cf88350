#!/usr/bin/env python
"""Teacher PCA views at BASELINE C2's size: IncrementalPCA of the flattened window positions
(700 features), speeds (350) and edges (350) of every training window.

  python tools/bench_teacher_views.py [--frames 600000] [--sklearn]

Prints seconds per view for the device path (DeviceIncrementalPCA: Gram GEMM + symmetric eigen-solve per batch, windows
gathered on the fly from the resident frame tables) and, with --sklearn, for the reference's host IncrementalPCA.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=600_000)
    ap.add_argument("--sklearn", action="store_true")
    args = ap.parse_args()
    from bench import synth_tables_fast
    from deepof_amd import teacher as TT
    from deepof_amd._lib import load_hip_library
    from deepof_amd.dataset import WindowDataset
    lib = load_hip_library()
    tn, te = synth_tables_fast(args.frames, 14, 14, 0, "cuda")

    class Pre:
        node_table, edge_table, video_off, keys = tn, te, np.array([0, args.frames]), ["v"]

    ds = WindowDataset.from_device_tables(Pre, 25, 1, lib)
    out = {"windows": len(ds)}
    for backend in (["device", "sklearn"] if args.sklearn else ["device"]):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fp, fs = TT.fit_nodes_pca(ds, backend=backend)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fe = TT.extract_pca_edges_view(ds, backend=backend)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[backend] = {"nodes_views_s": t1 - t0, "edges_view_s": t2 - t1, "shapes": [list(fp.shape), list(fs.shape), list(fe.shape)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
