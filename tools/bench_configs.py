#!/usr/bin/env python
"""Secondary BASELINE configurations on one GPU (the headline C2 line is bench.py's job).

  python tools/bench_configs.py [--steps 50] [--warmup 10] [--only c3,c4,c5]

Prints one JSON line per configuration: whole train steps (window build -> forward -> loss -> backward ->
clip + Adam) on device-resident synthetic data, fp32, eager launches on the current stream.
  c3  VQ-VAE, 14 body parts, window 25, codebook 512, batch 4096
  c4  Contrastive, TCN encoder on half windows (window 50 -> 25), batch 8192, nce / cosine (as BASELINE names it)
  c4r the same step with the recurrent encoder
  c5  VaDE, 2 animals (28 nodes, 32 edges), window 50, k=25, batch 4096, main phase
  infer  N1: gather + eval-mode encoder forward (embeddings + soft counts) of the C2 model, batch 4096
  c2tcn  the headline C2 workload (VaDE, 14 body parts, window 25, k=10, batch 1024) with the TCN encoder/decoder
  c2tfm  the same workload with the transformer encoder/decoder (dropout from the on-device counter hash)
  c5tcn  C5's shape (2 animals, window 50, k=25, batch 4096) with the TCN encoder/decoder
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

from bench import synth_tables_fast  # noqa: E402


def init_params(eng, seed=0):
    """Random weights of a trained-like scale for every tensor of the plan (norm scales around 1)."""
    g = torch.Generator().manual_seed(seed)
    for n in eng.names:
        shape = eng.layout[n][2]
        v = torch.randn(shape, generator=g) * (0.3 if len(shape) > 1 else 0.1)
        if "norm" in n and n.endswith("weight"):
            v = 1.0 + v
        eng.view(n).copy_(v)


def timed(step, steps, warmup):
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def init_tcn_params(eng, seed=0):
    """Reference initialisers of the TCN family (conv ~ N(0, 0.05), BatchNorm identity, zero biases)."""
    g = torch.Generator().manual_seed(seed)
    for n in eng.names:
        if n.endswith("running_var") or ((".bn" in n or ".head.2" in n or ".head.5" in n) and n.endswith("weight")):
            eng.view(n).fill_(1.0)
        elif n.endswith("running_mean") or (n.endswith("bias") and ("tcn." in n or ".bn" in n or ".head." in n or "decoder.fc" in n)):
            eng.view(n).zero_()
        elif "_tcn." in n or ".tcn." in n:
            eng.view(n).copy_(torch.randn(eng.layout[n][2], generator=g) * 0.05)


def run_vade_like(kind, ids, T, K, B, steps, warmup, frames=200_000):
    from deepof_amd import _capi
    from deepof_amd.engine import create_vade_engine
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph
    dev = torch.device("cuda")
    nodes, edges = bodypart_graph(ids)
    N, E, L, S = len(nodes), len(edges), 8, 32
    eng = create_vade_engine(B, T, adjacency_from_graph(nodes, edges), L, K, S, device=dev, kind=kind)
    init_params(eng)
    if kind.endswith("_tcn"):
        init_tcn_params(eng)
    if kind.endswith("_tfm"):  # LayerNorm / BatchNorm identity, zero biases (the family's initial state)
        for n in eng.names:
            if n.endswith("running_var") or ((".norm" in n or ".head.2" in n or ".head.5" in n) and n.endswith("weight")):
                eng.view(n).fill_(1.0)
            elif n.endswith("running_mean") or n.endswith("bias"):
                eng.view(n).zero_()
        eng.set_dropout(None, seed=1234)
    if kind.startswith("vqvae"):
        eng.view("vq_layer.codebook").uniform_(0.0, 1.0)
    tn, te = synth_tables_fast(frames, N, E, 0, dev)
    n_batches = (frames - T + 1) // B
    x, a = torch.empty(B, T, N, 3, device=dev), torch.empty(B, T, E, 1, device=dev)
    eps, eps_mc = torch.empty(B, L, device=dev), torch.empty(S, B, L, device=dev)
    tau = torch.softmax(torch.randn(B, K, device=dev) * 2, dim=-1)
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, 5e-4)
    if kind.startswith("vade"):
        eng.configure_vade_phase(False, 1.0, tau, 4.0)
    else:
        eng.set_hyper(vq_beta=1.0, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
    lib = eng.lib

    def step(i):
        b0 = (i * 7919 % n_batches) * B
        _capi.check(lib, lib.dof_window_gather_range(tn.data_ptr(), te.data_ptr(), b0, 1, B, T, N, E, x.data_ptr(),
                                                     a.data_ptr(), torch.cuda.current_stream().cuda_stream))
        eng.push_hyper()
        if kind.startswith("vade"):
            eps.normal_()
            eps_mc.normal_()
            eng.loss_grads(x, a, eps, eps_mc, tau, pretrain=False)
        else:
            eng.vq_loss_grads(x, a)
        eng.optimizer_step()

    sec = timed(step, steps, warmup)
    logs = eng.read_logs() if kind.startswith("vade") else eng.read_vq_logs()
    assert np.isfinite(logs["total_loss"]), logs
    return sec, logs["total_loss"], N, E


def run_contrastive(B, Tf, steps, warmup, frames=200_000, kind="contrastive_tcn"):
    from deepof_amd import _capi, graph as G
    from deepof_amd.augment import build_rotation_precomp, draw_augmentation
    from deepof_amd.config import ContrastiveCfg
    from deepof_amd.engine import contrastive_views, create_vade_engine
    dev = torch.device("cuda")
    nodes, edges = G.bodypart_graph([""])
    N, E, L = len(nodes), len(edges), 8
    adj = G.adjacency_from_graph(nodes, edges)
    ei, eil = G.edge_index_from_graph(nodes, edges)
    eid = torch.from_numpy(ei).to(dev)
    e1 = create_vade_engine(B, Tf // 2, adj, L, 1, device=dev, kind=kind)
    e2 = create_vade_engine(B, Tf // 2, adj, L, 1, device=dev, kind=kind, shared=e1)
    init_params(e1)
    if kind == "contrastive_tcn":
        g = torch.Generator().manual_seed(0)
        for n in e1.names:  # reference inits: conv ~ N(0, 0.05), BatchNorm identity, zero biases
            if n.endswith("running_var") or ((".bn" in n or ".head.2" in n or ".head.5" in n) and n.endswith("weight")):
                e1.view(n).fill_(1.0)
            elif n.endswith("running_mean") or n.endswith("bias"):
                e1.view(n).zero_()
            elif "_tcn." in n:
                e1.view(n).copy_(torch.randn(e1.layout[n][2], generator=g) * 0.05)
            if ".spatial_gnn_block." in n:
                e1.set_trainable(n, False)  # reference quirk Q11
    for seg in range(_capi.SEG_COUNT):
        e1.set_lr(seg, 1e-3)
    e1.set_hyper(clip=0.75, wd=1e-4)
    tn, te = synth_tables_fast(frames, N, E, 0, dev)
    n_batches = (frames - Tf + 1) // B
    x_full, a_unused = torch.empty(B, Tf, N, 3, device=dev), torch.empty(B, Tf, E, 1, device=dev)
    cfg = ContrastiveCfg(aug_p_rot=0.5, aug_p_noise=0.5)  # reference defaults leave rotations / noise off
    pc = build_rotation_precomp(eil.tolist(), N)
    gen, hgen = torch.Generator(device=dev).manual_seed(0), torch.Generator().manual_seed(0)
    lib = e1.lib

    def step(i):
        st = torch.cuda.current_stream().cuda_stream
        b0 = (i * 7919 % n_batches) * B
        _capi.check(lib, lib.dof_window_gather_range(tn.data_ptr(), te.data_ptr(), b0, 1, B, Tf, N, E,
                                                     x_full.data_ptr(), a_unused.data_ptr(), st))
        draws = draw_augmentation(B, Tf, N, cfg, pc, dev, gen, hgen)
        x, a = contrastive_views(lib, x_full, eid, None, st)
        xa, aa = contrastive_views(lib, x_full, eid, draws, st)
        e1.push_hyper()
        z, za = e1.contrastive_encode(x, a, True), e2.contrastive_encode(xa, aa, True)
        dz, dza = e1.contrastive_loss(z, za, "cosine", "nce", 0.1, 0.1, 0.1)
        e1.contrastive_backward(dz, False)
        e2.contrastive_backward(dza, True)
        e1.optimizer_step()

    sec = timed(step, steps, warmup)
    logs = e1.read_contrastive_logs()
    assert np.isfinite(logs["total_loss"]), logs
    return sec, logs["total_loss"], N, E


def run_inference(B, steps, warmup, frames=200_000):
    """N1: embedding_per_video's inner loop -- gather a batch of windows, eval-mode VaDE forward without the decoder,
    embeddings z (B,L) + soft counts q (B,K) -- for C2's model."""
    from deepof_amd import _capi
    from deepof_amd.engine import create_vade_engine
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph
    dev = torch.device("cuda")
    nodes, edges = bodypart_graph([""])
    N, E, T, L, K = len(nodes), len(edges), 25, 8, 10
    eng = create_vade_engine(B, T, adjacency_from_graph(nodes, edges), L, K, 32, device=dev)
    init_params(eng, seed=0)
    tn, te = synth_tables_fast(frames, N, E, 0, dev)
    x = torch.empty(B, T, N, 3, device=dev)
    a = torch.empty(B, T, E, 1, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    nb = (frames - T + 1) // B
    acc = torch.zeros((), device=dev)

    def step(i):
        _capi.check(eng.lib, eng.lib.dof_window_gather_range(tn.data_ptr(), te.data_ptr(), (i % nb) * B, 1, B, T, N, E,
                                                             x.data_ptr(), a.data_ptr(), st), "gather")
        out = eng.forward(x, a, None, want_loc=False)
        acc.add_(out["q"][0, 0])

    sec = timed(step, steps, warmup)
    return sec, float(acc), N, E


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--only", default="c3,c4,c4r,c5,c2tcn,c2tfm")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs a ROCm GPU")
    for name in args.only.split(","):
        if name == "c3":
            B = 4096
            sec, loss, N, E = run_vade_like("vqvae", [""], 25, 512, B, args.steps, args.warmup)
            desc = "C3: VQ-VAE recurrent, N=14,E=14, window=25, codebook=512, latent=8, batch=4096"
        elif name in ("c4", "c4r"):
            B = 8192
            kind = "contrastive_tcn" if name == "c4" else "contrastive"
            sec, loss, N, E = run_contrastive(B, 50, args.steps, args.warmup, kind=kind)
            desc = (f"C4{'' if name == 'c4' else ' shape with the RECURRENT encoder'}: contrastive "
                    f"{'TCN' if name == 'c4' else 'recurrent'} encoder, nce/cosine, N=14,E=14, window 50 -> half 25, "
                    "latent=8, batch=8192, both views + augmentations")
        elif name == "c2tcn":
            B = 1024
            sec, loss, N, E = run_vade_like("vade_tcn", [""], 25, 10, B, args.steps, args.warmup)
            desc = "C2 with the TCN family: VaDE TCN encoder/decoder, N=14,E=14, window=25, k=10, latent=8, batch=1024, main phase"
        elif name == "c5tcn":
            B = 4096
            sec, loss, N, E = run_vade_like("vade_tcn", ["B", "W"], 50, 25, B, args.steps, args.warmup)
            desc = "C5 with the TCN family: VaDE TCN encoder/decoder, 2 animals N=28,E=32, window=50, k=25, latent=8, batch=4096, main phase"
        elif name == "c2tfm":
            B = 1024
            sec, loss, N, E = run_vade_like("vade_tfm", [""], 25, 10, B, args.steps, args.warmup)
            desc = "C2 with the transformer family: VaDE transformer encoder/decoder, N=14,E=14, window=25, k=10, latent=8, batch=1024, main phase, dropout on"
        elif name == "infer":
            B = 4096
            sec, loss, N, E = run_inference(B, args.steps, args.warmup)
            desc = "N1 inference (embedding_per_video inner loop): C2 model, eval forward without decoder, batch=4096"
            print(json.dumps({"metric": "pose-windows/sec (embedding + soft counts)", "value": B / sec, "unit": "windows/s",
                              "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3, "dtype": "f32",
                              "data": "synthetic", "config": {"workload": desc}}), flush=True)
            continue
        elif name == "c5":
            B = 4096
            sec, loss, N, E = run_vade_like("vade", ["B", "W"], 50, 25, B, args.steps, args.warmup)
            desc = "C5: VaDE recurrent, 2 animals N=28,E=32, window=50, k=25, latent=8, batch=4096, main phase"
        else:
            raise SystemExit(f"unknown config {name}")
        print(json.dumps({"metric": "pose-windows/sec (train step)", "value": B / sec, "unit": "windows/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": sec * 1e3, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": desc, "final_total_loss": loss}}), flush=True)


if __name__ == "__main__":
    main()
