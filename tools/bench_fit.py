#!/usr/bin/env python
"""End-to-end wall clock of train_deepof_model on BASELINE C2's data set (600k frames resident on the device), teacher on.

  python tools/bench_fit.py [--frames 600000] [--epochs 3] [--no-teacher]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=600_000)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--no-teacher", action="store_true")
    ap.add_argument("--model", default="VaDE")
    args = ap.parse_args()
    from bench import synth_tables_fast
    from deepof_amd import training as TR
    from deepof_amd._lib import load_hip_library
    from deepof_amd.dataset import WindowDataset
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph, make_meta_info
    lib = load_hip_library()
    nodes, edges = bodypart_graph([""])
    tn, te = synth_tables_fast(args.frames, 14, 14, 0, "cuda")
    n_val = args.frames // 10

    class Pre:
        node_table, edge_table, video_off, keys = tn, te, np.array([0, args.frames - n_val, args.frames]), ["train", "val"]

    train = WindowDataset.from_device_tables(Pre, 25, 1, lib, keys=["train"])
    val = WindowDataset.from_device_tables(Pre, 25, 1, lib, keys=["val"])
    out = tempfile.mkdtemp()
    # time the training epochs of the fit from the outside (synchronised before and after each)
    acc = {"steps": 0, "seconds": 0.0, "epochs": []}
    orig_epoch = TR.VadeStepper.train_epoch

    def timed_epoch(self, dataset, seed, shuffle=True):
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = orig_epoch(self, dataset, seed, shuffle)
        torch.cuda.synchronize()
        dt_e = time.perf_counter() - t
        acc["steps"] += self.log_steps
        acc["seconds"] += dt_e
        acc["epochs"].append(round(1e3 * dt_e / max(1, self.log_steps), 4))
        return res

    TR.VadeStepper.train_epoch = timed_epoch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, _, _, logs = TR.train_deepof_model(
        preprocessed_object=(train, val), adjacency_matrix=adjacency_from_graph(nodes, edges), meta_info=make_meta_info(nodes, edges),
        encoder_type="recurrent", batch_size=1024, latent_dim=8, epochs=args.epochs, output_path=out, n_clusters=10,
        model_name=args.model, use_turtle_teacher=not args.no_teacher, save_weights=False, pretrain_epochs=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"model": args.model, "train_windows": len(train), "val_windows": len(val), "epochs": args.epochs,
                      "teacher": not args.no_teacher, "wall_s": dt, "final_train_loss": float(logs["train"]["total_loss"][-1]),
                      "train_steps": acc["steps"], "train_ms_per_step": 1e3 * acc["seconds"] / max(1, acc["steps"]),
                      "train_ms_per_step_by_epoch": acc["epochs"]}))


if __name__ == "__main__":
    main()
