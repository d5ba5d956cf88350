#!/usr/bin/env python
"""Pose-table preprocessing on one GPU: BASELINE C2's data set, raw tables resident in HBM.

  python tools/bench_preprocess.py [--videos 40] [--frames 15000] [--iters 20] [--animals 1|2] [--no-cpu]

One JSON line: frames/s of the whole dof_preprocess_tables call (8 launches), its HBM roofline figure
(algorithmic bytes = 8 B x raw columns read once + 4 B x output columns written once, per frame), the upload-
inclusive rate, and the oracle (numpy restatement of the reference's pandas / sklearn path) timed on a sample of
videos on the host.
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
    ap.add_argument("--videos", type=int, default=40)
    ap.add_argument("--frames", type=int, default=15_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--animals", type=int, default=1)
    ap.add_argument("--mode", default="groupwise")
    ap.add_argument("--scale", default="standard", choices=["standard", "minmax", "robust"])
    ap.add_argument("--filter", type=float, default=0.0, help="filter_low_variance threshold (0 = off)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-log", action="store_true", help="log_distances=False")
    args = ap.parse_args()
    import parity_common as PC
    from deepof_amd._lib import load_hip_library
    from deepof_amd.preprocess import preprocess_tables
    lib = load_hip_library()
    one = ["Nose", "Left_ear", "Right_ear", "Spine_1", "Center", "Spine_2", "Left_fhip", "Right_fhip", "Left_bhip", "Right_bhip",
           "Tail_base", "Tail_1", "Tail_2", "Tail_tip"]
    bps, aids = (one, [""]) if args.animals == 1 else ([f"{a}_{b}" for a in ("B", "W") for b in one], ["B", "W"])
    tabs, cols = PC.synth_raw_tables(args.videos, args.frames, bps, seed=3, nan_rate=0.001)
    node_cols, edge_cols, _ = PC.preprocess_output_columns(cols)
    edge_cols = edge_cols[:14 * args.animals + (4 if args.animals == 2 else 0)]
    mode = None if args.mode == "none" else args.mode
    kw = dict(dist_standardize=mode, speed_standardize=mode, coord_standardize=mode, log_distances=not args.no_log)
    plain = args.scale == "standard" and not args.filter
    if not plain:
        kw.update(scale=args.scale, filter_low_variance=args.filter or False)
    keys = sorted(tabs)
    torch.zeros(1, device="cuda")                      # context up before anything is timed
    raw = torch.empty(sum(tabs[k].shape[0] for k in keys), len(cols), dtype=torch.float64, device="cuda")

    def upload():
        off = 0
        for k in keys:
            n = tabs[k].shape[0]
            raw[off:off + n].copy_(torch.from_numpy(tabs[k]), non_blocking=True)
            off += n
        torch.cuda.synchronize()

    upload()
    t0 = time.perf_counter()
    upload()
    upload_s = time.perf_counter() - t0
    n_frames, C = raw.shape
    n_out = len(node_cols) + len(edge_cols)

    def run():
        return preprocess_tables(tabs, cols, aids, node_cols, edge_cols, (), device="cuda", lib=lib, raw_device=raw, **kw)

    for _ in range(3):
        res = run()
    torch.cuda.synchronize()
    wall = []
    for _ in range(args.iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run()
        torch.cuda.synchronize()
        wall.append(time.perf_counter() - t0)
    # device-only time: replay the C call alone
    # the other scalers / the filter are several C calls with host decisions in between: the whole host call is their figure
    dev_ms = device_time(lib, res, raw, tabs, cols, aids, node_cols, edge_cols, kw, args.iters) if plain else float(np.median(wall) * 1e3)
    algo = n_frames * (8 * C + 4 * n_out)
    out = {"metric": "pose-table preprocessing (scale_table + global scaler + clip/interpolate -> fp32 frame tables)",
           "value": n_frames / (dev_ms * 1e-3), "unit": "frames/s", "ms_per_call": dev_ms, "n_frames": int(n_frames),
           "raw_columns": int(C), "output_columns": int(n_out), "videos": args.videos, "dtype": "f64 -> f32",
           "config": {"workload": f"C2 data set: {args.videos} videos x {args.frames} frames, {len(bps)} body parts, modes={args.mode}, "
                                  f"scale={args.scale}, filter_low_variance={args.filter or False}"
                                  + ("" if plain else " (ms_per_call = median wall time of the whole host call incl. its syncs)")},
           "roofline": {"bound": "hbm", "achieved": algo / (dev_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": algo / (dev_ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_call": int(algo),
                        "bytes_per_frame": 8 * C + 4 * n_out},
           "host_call_ms_median": float(np.median(wall) * 1e3),
           "upload_ms": upload_s * 1e3, "frames_per_s_incl_upload": n_frames / (upload_s + dev_ms * 1e-3)}
    if not args.no_cpu:
        from oracle import preprocess as op
        sample = {k: tabs[k] for k in keys[:4]}
        t0 = time.perf_counter()
        op.preprocess(sample, cols, aids, **kw)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": sum(v.shape[0] for v in sample.values()) / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"oracle/preprocess.py (numpy restatement) on the first 4 videos ({4 * args.frames} frames)"}
    print(json.dumps(out))


def device_time(lib, res, raw, tabs, cols, aids, node_cols, edge_cols, kw, iters):
    """ms of one dof_preprocess_tables call, HIP events on the launch stream, descriptors already on the device."""
    import ctypes
    from deepof_amd import _capi
    from deepof_amd import preprocess as PP
    plan = PP.column_plan(cols, aids)
    where = {c: i for i, c in enumerate(cols)}
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()   # noqa: E731
    out_cols = dev(np.array([where[c] for c in node_cols + edge_cols], dtype=np.int32))
    d_off, d_kind = dev(res.video_off), dev(plan.kinds)
    d_ref = dev(plan.size_ref.reshape(-1))
    d_coff, d_chain = dev(plan.chain_off), dev(plan.chain.reshape(-1) if plan.chain.size else np.zeros(4, np.int32))
    d_scaler = torch.zeros(len(cols), 2, dtype=torch.float64, device="cuda")
    dims = _capi.PreprocDims(n_frames=raw.shape[0], n_videos=len(res.keys), n_cols=len(cols), n_animals=len(plan.animal_ids),
                             n_node_cols=len(node_cols), n_edge_cols=len(edge_cols), n_angle_cols=0,
                             speed_mode=_capi.PP_MODES[kw["speed_standardize"]], dist_mode=_capi.PP_MODES[kw["dist_standardize"]],
                             coord_mode=_capi.PP_MODES[kw["coord_standardize"]], log_distances=int(kw["log_distances"]), inter_scale=0, fit_global=1, clip=10.0)
    ws = torch.empty(lib.dof_preprocess_workspace_bytes(ctypes.byref(dims)), dtype=torch.uint8, device="cuda")
    node, edge = torch.empty_like(res.node_table), torch.empty_like(res.edge_table)
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _capi.check(lib, lib.dof_preprocess_tables(ctypes.byref(dims), raw.data_ptr(), d_off.data_ptr(), d_kind.data_ptr(),
                                                   d_ref.data_ptr(), d_coff.data_ptr(), d_chain.data_ptr(), out_cols.data_ptr(), None,
                                                   d_scaler.data_ptr(), None, None, node.data_ptr(), edge.data_ptr(), None,
                                                   ws.data_ptr(), st), "dof_preprocess_tables")

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    assert torch.equal(node, res.node_table) and torch.equal(edge, res.edge_table)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        call()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


if __name__ == "__main__":
    main()
