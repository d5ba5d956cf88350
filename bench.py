#!/usr/bin/env python
"""Headline benchmark: pose-windows/sec of the VaDE train step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
  python bench.py --config c4|c5 --gpus N ...        (BASELINE configs[3] / [4], the two that name data parallelism)

One "step" = one pass of the hot path over one batch of synthetic windows, inputs resident in
HBM: window gather from the frame table -> VaDE forward -> VadeLoss -> backward -> [RCCL
all-reduce of the flat gradient] -> clip + Adam -- issued through the SAME stepper the product's
fit loop uses (deepof_amd.training.VadeStepper.step: hipGraph replay per step).  Workload = BASELINE config C2
(VaDE recurrent, 14 body parts, window 25, k=10, latent 8, batch 1024 per GPU, main phase with distillation).
Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the HBM-bound window-gather kernel measured live with HIP events on a full
                 materialisation of the C2 dataset (5,824 algorithmic bytes/window)
  cpu_baseline : the CPU oracle (a port of the reference PyTorch path) timed on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# every source the C2 step's kernels are compiled from (the headers with device code included: dof_rt.h carries the MFMA /
# piece-split helpers): profiles/rNN_step_pmc.json is quoted only while all of them are unchanged
STEP_SOURCES = ["k_gather.hip", "k_rnn.hip", "k_grum16.inc.h", "k_grumx.inc.h", "k_sum_partials.inc.h", "dof_rt.h", "launchers.h",
                "k_reduce.hip", "vade.hip", "k_decoder.inc.h", "k_graph_latent.inc.h"]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling
FP32_PEAK_FLOPS = 157.3e12  # fp32 vector peak (= the f32-input MFMA rate; MI355X_MICROARCH.md)


def synth_tables(n_frames, n_nodes, n_edges, seed):
    """per column an AR(1) walk (rho 0.95) of numpy.random.default_rng(seed) standard normals over ALL
    frames, standardised to mean 0 / std 1, clipped to +-10; column order [x.. | y.. | speed..] + edges.  float32."""
    from scipy.signal import lfilter
    rng = np.random.default_rng(seed)
    cols = 3 * n_nodes + n_edges
    out = np.empty((n_frames, cols), dtype=np.float32)
    for c0 in range(0, cols, 8):  # column blocks bound the float64 temporaries
        e = rng.standard_normal((n_frames, min(8, cols - c0)))
        w = lfilter([1.0], [1.0, -0.95], e, axis=0)
        w = (w - w.mean(0)) / (w.std(0) + 1e-12)
        out[:, c0:c0 + w.shape[1]] = np.clip(w, -10.0, 10.0)
    return out[:, : 3 * n_nodes].copy(), out[:, 3 * n_nodes:].copy()


def synth_tables_fast(n_frames, n_nodes, n_edges, seed, device):
    """The tables of synth_tables on the device (name kept for tools/)."""
    tn, te = synth_tables(n_frames, n_nodes, n_edges, seed)
    return torch.from_numpy(tn).to(device), torch.from_numpy(te).to(device)


def cpu_baseline(P, x, a, L, K, budget_s=100.0):
    """Oracle ('port' of the reference PyTorch-CPU path) timed on a bounded sample of the same workload: same
    architecture and initial weights (state_dict P), one batch of the SAME synthetic windows the device path trains on,
    main phase with distillation.  SURVEY 8(d) asks for the host's cores with the count printed: eager PyTorch on these
    tiny operators does not scale, so the sweep {1, 8, 32, all hardware threads} is timed (a few steps each, bounded by
    ``budget_s`` in total) and every figure is reported; ``value`` / ``cores`` = the best of them."""
    from oracle import vade as OV

    B, T = x.shape[0], x.shape[1]
    g = torch.Generator().manual_seed(0)
    tau = torch.softmax(torch.randn(B, K, generator=g), dim=-1)
    pi = tau.mean(0).clamp_min(1e-8)
    w = pi.pow(-1.0)
    w = (w / w.mean()).clamp_max(3.0)
    cfg = OV.VadeLossCfg(K, False, lambda_distill=4.0, class_weight=w, teacher_marginal=pi)
    all_threads = torch.get_num_threads()
    t_start = time.perf_counter()

    def timed(threads, n_warm, n_steps):
        torch.set_num_threads(threads)
        P_run = {k: v.clone() for k, v in P.items()}
        opt = OV.AdamState()
        times = []
        for i in range(n_warm + n_steps):
            t0 = time.perf_counter()
            eps = torch.randn(B, L)
            eps_mc = torch.randn(32, B, L)
            OV.vade_train_step(P_run, opt, x, a, cfg, 1.0, 5e-4, 2e-4, eps, eps_mc, tau)
            dt = time.perf_counter() - t0
            if i >= n_warm:
                times.append(dt)
            if time.perf_counter() - t_start > budget_s and times:
                break
        return B / float(np.median(times)), len(times)

    sweep = {}
    for threads, n_warm, n_steps in ((8, 2, 6), (1, 1, 2), (32, 1, 2), (all_threads, 1, 1)):
        threads = min(threads, all_threads)
        if threads in sweep:
            continue
        if time.perf_counter() - t_start > budget_s:
            sweep[threads] = None
            continue
        sweep[threads] = timed(threads, n_warm, n_steps)
    torch.set_num_threads(all_threads)
    done = {t: v for t, v in sweep.items() if v is not None}
    cores = max(done, key=lambda t: done[t][0])
    return {"value": done[cores][0], "unit": "windows/s", "cores": cores, "kind": "port",
            "threads_sweep": {str(t): (None if v is None else round(v[0], 1)) for t, v in sweep.items()},
            "host_hardware_threads": all_threads,
            "sample": f"one batch of {B} synthetic windows (the device path's own), oracle/vade.py train steps (torch CPU "
                      f"fp32): median of " + ", ".join(f"{v[1]} step(s) on {t} thread(s)" for t, v in done.items())
                      + "; one warm-up step or two before each"}


# ------------------------------------------------------------------------------------------------
# product paths of the other configurations: the model classes (reference initialisers) and the fit loops' own steppers
# ------------------------------------------------------------------------------------------------
def _device_dataset(ids, T, frames, n_animals, seed, dev, lib):
    from types import SimpleNamespace

    from deepof_amd.dataset import WindowDataset
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph
    nodes, edges = bodypart_graph(ids)
    tn, te = synth_tables_fast(n_animals * frames, len(nodes), len(edges), seed, dev)
    pre = SimpleNamespace(node_table=tn, edge_table=te, keys=[f"animal{i}" for i in range(n_animals)],
                          video_off=np.arange(n_animals + 1, dtype=np.int64) * frames)
    return WindowDataset.from_device_tables(pre, T, 1, lib), adjacency_from_graph(nodes, edges), nodes, edges, tn, te


def _time_steps(one_step, steps, warmup):
    for i in range(warmup):
        one_step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one_step(warmup + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def vade_stepper_setup(ids, T, K, B, encoder_type, frames, dev, rank=0, use_graphs=None, latent=8):
    """VaDE through deepof_amd.training.VadeStepper in the main phase with distillation, as fit_VADE configures it
    (training.py:1643-1755 of the reference): lr after epoch 0 = 5e-4 / 2e-4, KL weight tf_sigmoid warm-up 5 epochs -> 1,
    lambda 4 held 10 epochs.  Returns (stepper, model, dataset, batch starts, tables)."""
    from deepof_amd.config import CommonFitCfg, TurtleTeacherCfg, VaDECfg
    from deepof_amd.dataset import batch_starts
    from deepof_amd.models import VaDE
    from deepof_amd.schedules import WeightSchedule
    from deepof_amd.stepping import DeviceSchedule
    from deepof_amd.training import VadeStepper, _set_lrs
    from deepof_amd._lib import load_hip_library
    lib = load_hip_library()
    ds, adj, nodes, edges, tn, te = _device_dataset(ids, T, frames, 2, rank, dev, lib)
    N, E, L = len(nodes), len(edges), latent
    torch.manual_seed(0)
    model = VaDE((T, N, 3), (T, E, 1), adj, L, K, encoder_type=encoder_type, kmeans_loss=1.0, batch_size=B, device=dev)
    eng = model._base
    common = CommonFitCfg(model_name="vade", encoder_type=encoder_type, batch_size=B, latent_dim=L, epochs=1,
                          n_components=K, output_path=".")
    stepper = VadeStepper(model, common, VaDECfg(), TurtleTeacherCfg(), use_graphs=use_graphs)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    tau_star = torch.softmax(torch.randn(len(ds), K, device=dev, generator=g) * 2.0, dim=-1)
    model.set_pretrain_mode(False)
    model.train()
    if encoder_type != "recurrent":
        model.set_censnet_trainable(True)   # fit_VADE's main-phase optimiser holds the CensNet tensors (Q11)
    stepper.set_mode("main")
    nb = len(ds) // B
    vcfg, tcfg = stepper.vade, stepper.teacher
    stepper.kl_scheduler = DeviceSchedule(WeightSchedule(nb, mode=vcfg.kl_annealing_mode, warmup_epochs=vcfg.kl_warmup,
                                                         max_weight=vcfg.kl_max_weight, cooldown_epochs=vcfg.kl_cooldown,
                                                         end_weight=vcfg.kl_end_weight), dev)
    stepper.set_teacher(tau_star, tcfg.lambda_distill, DeviceSchedule(
        WeightSchedule(nb, mode=vcfg.kl_annealing_mode, warmup_epochs=0, at_max_epochs=tcfg.lambda_decay_start,
                       max_weight=tcfg.lambda_distill, cooldown_epochs=tcfg.lambda_cooldown,
                       end_weight=tcfg.lambda_end_weight), dev))
    eng.reset_optimizer()
    _set_lrs(eng, 5e-4, 2e-4)
    eng.push_hyper()
    stepper.begin_logs()
    # batches in the reference loader's order: seeded block shuffle of the batch starts (dataset.py:589-597), full batches
    starts = [int(v) for v in batch_starts(len(ds), B, 1, 0, True) if v + B <= len(ds)]
    return stepper, model, ds, starts, (tn, te), tau_star


def run_vade_product(ids, T, K, B, encoder_type, steps, warmup, frames=100_000, window_storage="fp32", latent=8):
    dev = torch.device("cuda")
    stepper, model, ds, starts, _tabs, _tau = vade_stepper_setup(ids, T, K, B, encoder_type, frames, dev, latent=latent)
    ds.window_storage = window_storage   # "bf16": batches gathered as bf16 and widened for the fp32 step kernels

    def one_step(i):
        s0 = starts[i % len(starts)]
        stepper.step(ds, s0, s0 + B, True, True)

    sec = _time_steps(one_step, steps, warmup)
    logs = model._base.read_logs()
    assert np.isfinite(logs["total_loss"]), logs
    return sec, logs["total_loss"], "deepof_amd.training.VadeStepper.step"


def run_vqvae_product(B, K, steps, warmup, frames=100_000, T=25):
    """C3 through the VQ-VAE fit loop's stepper (fit_VQVAE's configuration: Adam lr 1e-3, weight decay 1e-4, clip 0.75)."""
    from deepof_amd import _capi
    from deepof_amd._lib import load_hip_library
    from deepof_amd.dataset import batch_starts
    from deepof_amd.models import VQVAE
    from deepof_amd.training import VQVAEStepper
    dev = torch.device("cuda")
    ds, adj, nodes, edges, _tn, _te = _device_dataset([""], T, frames, 2, 0, dev, load_hip_library())
    torch.manual_seed(0)
    model = VQVAE(ds.x_shape, ds.a_shape, adj, 8, K, encoder_type="recurrent", use_gnn=True, kmeans_loss=0.0,
                  batch_size=B, device=dev)
    eng = model._base
    eng.reset_optimizer()
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, 1e-3)
    eng.set_hyper(vq_beta=model.beta, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
    eng.push_hyper()
    model.train()
    stepper = VQVAEStepper(model)
    log_sum = torch.zeros(_capi.LOG_COUNT, dtype=torch.float64, device=dev)
    starts = [int(v) for v in batch_starts(len(ds), B, 1, 0, True) if v + B <= len(ds)]

    def one_step(i):
        s0 = starts[i % len(starts)]
        stepper.step(ds, s0, s0 + B, True, None, None, log_sum)

    sec = _time_steps(one_step, steps, warmup)
    logs = eng.read_vq_logs()
    assert np.isfinite(logs["total_loss"]), logs
    return sec, logs["total_loss"], "deepof_amd.training.VQVAEStepper.step"


def contrastive_stepper_setup(B, Tf, encoder_type, frames, dev, rank=0, use_graphs=None):
    """C4 through fit_contrastive's stepper: both views with the reference's default augmentations, TCN encoder on the
    half windows, NCE / cosine, Adam lr 1e-3 + weight decay 1e-4, clip 0.75 (CensNet outside the optimiser, Q11).
    Returns (one_step(i), stepper, model, dataset, (node table, edge table)); rank: this rank's synthetic animals
    (the model's initial state is the same on every rank -- DDP's broadcast)."""
    from deepof_amd import _capi
    from deepof_amd._lib import load_hip_library
    from deepof_amd.augment import edge_index_from_meta
    from deepof_amd.config import ContrastiveCfg
    from deepof_amd.dataset import batch_starts
    from deepof_amd.graph import make_meta_info
    from deepof_amd.models import Contrastive
    from deepof_amd.training import ContrastiveStepper
    ds, adj, nodes, edges, _tn, _te = _device_dataset([""], Tf, frames, 2, rank, dev, load_hip_library())
    ccfg = ContrastiveCfg()
    torch.manual_seed(0)
    model = Contrastive(ds.x_shape, ds.a_shape, adj, latent_dim=8, encoder_type=encoder_type, use_gnn=True,
                        similarity_function=ccfg.contrastive_similarity_function,
                        loss_function=ccfg.contrastive_loss_function, temperature=ccfg.temperature, beta=ccfg.beta,
                        tau=ccfg.tau, batch_size=B, device=dev)
    eng = model._base
    ei_g, ei_l = edge_index_from_meta(make_meta_info(nodes, edges), ds.x_shape[1])
    stepper = ContrastiveStepper(model, ei_g, ei_l, ccfg, seed=0)
    if use_graphs is not None:
        from deepof_amd.training import StepGraphs
        stepper.graphs = StepGraphs(model.device, use_graphs)
    eng.reset_optimizer()
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, 1e-3)
    eng.set_hyper(clip=0.75, wd=1e-4)
    eng.push_hyper()
    model.train()
    log_sum = torch.zeros(_capi.LOG_COUNT, dtype=torch.float64, device=dev)
    starts = [int(v) for v in batch_starts(len(ds), B, 1, 0, True) if v + B <= len(ds)]

    def one_step(i):
        s0 = starts[i % len(starts)]
        stepper.step(ds, s0, s0 + B, True, None, None, log_sum)

    return one_step, stepper, model, ds, (_tn, _te)


def run_contrastive_product(B, Tf, encoder_type, steps, warmup, frames=100_000):
    one_step, _stepper, model, _ds, _tabs = contrastive_stepper_setup(B, Tf, encoder_type, frames, torch.device("cuda"))
    eng = model._base
    sec = _time_steps(one_step, steps, warmup)
    logs = eng.read_contrastive_logs()
    assert np.isfinite(logs["total_loss"]), logs
    return sec, logs["total_loss"], "deepof_amd.training.ContrastiveStepper.step"


def secondary_configs(steps=12, warmup=6):
    """The other BASELINE configurations (and the two other encoder families at the C2 shape) timed in this same run,
    so that their rates are driver-visible too: whole train steps on device-resident synthetic data through the SAME
    steppers the fit loops use (hipGraph replay), models built by the product classes with the reference's
    initialisers.  The TCN lines warm up for 25 steps: the one-pass BatchNorm statistics engage once the running means
    have caught up with the batch means, which is where a fit spends its time.  Never part of
    `value`; a failing configuration reports its error string."""
    rows = []
    plan = [("C3 VQ-VAE recurrent, codebook 512, batch 4096", lambda: run_vqvae_product(4096, 512, steps, warmup), 4096),
            ("C5 VaDE recurrent, 2 animals (N=28,E=32), window 50, k=25, batch 4096",
             lambda: run_vade_product(["B", "W"], 50, 25, 4096, "recurrent", steps, warmup), 4096),
            ("C4 contrastive TCN encoder, window 50 -> 25, batch 8192",
             lambda: run_contrastive_product(8192, 50, "TCN", max(4, steps // 3), 25), 8192),
            ("C2 shape, VaDE TCN encoder/decoder, batch 1024", lambda: run_vade_product([""], 25, 10, 1024, "TCN", steps, 25), 1024),
            ("C5 shape, VaDE TCN encoder/decoder: 2 animals (N=28,E=32), window 50 (8-sequence time-resident convolutions), k=25, batch 4096",
             lambda: run_vade_product(["B", "W"], 50, 25, 4096, "TCN", max(4, steps // 3), 12), 4096),
            ("C2 shape, VaDE transformer encoder/decoder (dropout on), batch 1024",
             lambda: run_vade_product([""], 25, 10, 1024, "transformer", steps, warmup), 1024),
            ("C2 with bf16 window storage (BASELINE configs[1]: batches gathered as bf16, fp32 arithmetic), batch 1024",
             lambda: run_vade_product([""], 25, 10, 1024, "recurrent", 4 * steps, warmup, window_storage="bf16"), 1024),
            ("C2 at latent 4 (the API's default latent_dim; GRU(8,8) / GRU(16->4) layers on the lane-per-unit kernels k_gru3_*), batch 1024",
             lambda: run_vade_product([""], 25, 10, 1024, "recurrent", steps, warmup, latent=4), 1024),
            ("C2 at latent 6 (the tutorial's latent_dim; GRU(12,12) / GRU(24->6) layers, k_gru3_* on padded lane groups), batch 1024",
             lambda: run_vade_product([""], 25, 10, 1024, "recurrent", steps, warmup, latent=6), 1024),
            ("C2 at latent 16 (GRU(32,32) / GRU(64->16) layers on the GEMM-shaped matrix-pipe kernels k_grum_*), batch 1024",
             lambda: run_vade_product([""], 25, 10, 1024, "recurrent", steps, warmup, latent=16), 1024),
            ("C2 at latent 32 (GRU(64,64) / GRU(128->32) layers, k_grum_*), batch 1024",
             lambda: run_vade_product([""], 25, 10, 1024, "recurrent", max(6, steps // 2), warmup, latent=32), 1024)]
    for name, fn, B in plan:
        try:
            sec, loss, path = fn()
            rows.append({"workload": name, "value": B / sec, "unit": "windows/s", "ms_per_step": sec * 1e3, "dtype": "f32",
                         "path": path, "final_total_loss": loss})
        except Exception as exc:  # noqa: BLE001  (a secondary line must never take the headline down)
            rows.append({"workload": name, "error": f"{type(exc).__name__}: {exc}"[:300]})
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    return rows


def _source_sha(files):
    """sha256 over kernel sources: measured counter files in profiles/ are only quoted while the sources they were
    measured on are unchanged (no stale constants in the line)."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "deepof_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def _latest_profile(suffix, sha):
    """The newest committed profiles/rNN_<suffix> whose `source_sha` equals `sha` -> (record, repo-relative name), else
    (None, None): counter files are quoted only while the sources they were measured on are unchanged."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")), reverse=True):
        try:
            rec = json.load(open(f))
        except ValueError:
            continue
        if rec.get("source_sha") == sha:
            return rec, os.path.relpath(f, ROOT)
    return None, None


def spawn_ranks(n):
    """Re-run this command line as n ranks under torch.distributed.run on this node (rendezvous on 127.0.0.1, a free
    port); rank 0's JSON line is the child's stdout, the exit code is the launcher's."""
    import socket
    import subprocess
    if os.environ.get("DOF_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < n:
        raise SystemExit(f"bench.py --gpus {n}: {torch.cuda.device_count()} GPU(s) visible; one rank per GPU is required")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only mode the host driver supports
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=("c2", "c4", "c5"), default="c2",
                    help="c2 (default, the headline): BASELINE configs[1].  c4 / c5: BASELINE configs[3] / configs[4] -- the two that "
                         "name data parallelism -- through the same steppers, barriers, self-check and data_parallel report")
    ap.add_argument("--batch", type=int, default=None, help="windows per GPU (default: the configuration's 1024 / 8192 / 4096)")
    ap.add_argument("--frames", type=int, default=None, help="frames per synthetic animal (2 animals per rank; default 600,000 / 100,000)")
    ap.add_argument("--latent", type=int, default=8, help="latent size of the headline model (8 = C2; 16 for profiling that path)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (C3, C4, C5, TCN / transformer)")
    ap.add_argument("--gather-iters", type=int, default=20)
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the timed steps keep stepping for about this long (second, longer measurement)")
    ap.add_argument("--log-every", type=int, default=0, help="debug: print the loss terms every N steps (adds syncs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` run bare: start the N ranks ourselves (one process per GPU, RCCL), exactly as the
        # driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` would.
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and os.environ.get("DOF_BENCH_FORCE_PG") != "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    # DOF_BENCH_SHARE_GPU=1 (debug only): all ranks on cuda:0 over gloo, to exercise the N > 1 control flow (all-reduce,
    # barriers, max-over-ranks, rank-0 JSON) on a 1-GPU box; RCCL refuses two ranks on one device.  Never a result.
    share = os.environ.get("DOF_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, {torch.cuda.device_count()} visible "
                         f"(no oversubscription: RCCL refuses two ranks on one device)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1 or os.environ.get("DOF_BENCH_FORCE_PG") == "1":  # FORCE_PG: a 1-rank RCCL group (exercises the DP path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from deepof_amd import _capi

    cfg = args.config
    B = args.batch if args.batch else {"c2": 1024, "c4": 8192, "c5": 4096}[cfg]
    L, S = args.latent, 32
    n_animals, F = 2, args.frames if args.frames else (600_000 if cfg == "c2" else 100_000)
    graphs_arg = False if args.no_graph else None
    # ---- the product objects: model (reference initialisers, identical on every rank = DDP's broadcast), the fit
    # loop's stepper, a device-resident dataset of 2 animals per rank
    tau_star = None
    if cfg == "c4":   # contrastive, TCN encoder on the half windows of 50-step windows
        T, K = 50, 0
        one_step, stepper, model, ds, (tn, te) = contrastive_stepper_setup(B, T, "TCN", F, dev, rank=rank, use_graphs=graphs_arg)
        workload = (f"C4: contrastive TCN encoder + augmented second view, 14 body parts (N=14,E=14), window=50 -> halves of 25, "
                    f"nce / cosine, latent={L}, batch={B}/GPU, fp32")
        metric = "pose-windows/sec (train step) contrastive TCN win=50"
        step_path = "deepof_amd.training.ContrastiveStepper.step (the fit loop's step)"
        read_logs = lambda: model._base.read_contrastive_logs()   # noqa: E731
    else:
        ids, T, K = ([""], 25, 10) if cfg == "c2" else (["B", "W"], 50, 25)
        stepper, model, ds, starts, (tn, te), tau_star = vade_stepper_setup(
            ids, T, K, B, "recurrent", F, dev, rank=rank, use_graphs=graphs_arg, latent=args.latent)

        def one_step(i):
            s0 = starts[i % len(starts)]
            stepper.step(ds, s0, s0 + B, True, True)   # gather + forward + loss + backward [+ all-reduce] + clip/Adam

        if cfg == "c2":
            workload = (f"C2: VaDE GM-VAE recurrent, 14 body parts (N=14,E=14), window=25, k=10, latent={args.latent}, "
                        f"batch={B}/GPU, main phase (MC-KL S=32 + distillation), fp32")
            metric = "pose-windows/sec (train step) VaDE 14-bp win=25"
        else:
            workload = (f"C5: VaDE GM-VAE recurrent, 2 animals (N=28,E=32), window=50, k=25, latent={args.latent}, "
                        f"batch={B}/GPU, main phase (MC-KL S=32 + distillation), fp32")
            metric = "pose-windows/sec (train step) VaDE 2 animals 28-bp win=50"
        step_path = "deepof_amd.training.VadeStepper.step (the fit loop's step)"
        read_logs = lambda: model._base.read_logs()   # noqa: E731
    eng = model._base
    lib = eng.lib
    N, E = eng.N, eng.E
    n_windows = len(ds)
    win_per_animal = F - T + 1
    initial_state = eng.state_dict() if (rank == 0 and cfg == "c2") else None   # (no step has run yet)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def show(tag, i):
        if args.log_every and rank == 0 and i % args.log_every == 0:
            print(tag, i, {k: round(v, 4) for k, v in read_logs().items()}, file=sys.stderr, flush=True)

    for i in range(args.warmup):
        one_step(i)
        show("warmup", i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
        show("step", i)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_own = elapsed   # this rank's clock (the line carries every rank's, so a slow rank is visible from the line alone)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    logs = read_logs()
    if not np.isfinite(logs["total_loss"]):
        raise SystemExit(f"non-finite loss after benchmark steps: {logs}")
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    # ---- a second, longer measurement of the same loop (the contract's K steps last only tens of milliseconds)
    sustained = None
    if args.sustain_seconds > 0:
        n_more = max(args.steps, int(args.sustain_seconds / max(ms_per_step * 1e-3, 1e-6)))
        if world > 1:
            import torch.distributed as dist
            nm = torch.tensor([n_more], device=dev)
            dist.broadcast(nm, src=0)
            n_more = int(nm.item())
        barrier()
        t1 = time.perf_counter()
        for i in range(n_more):
            one_step(args.warmup + args.steps + i)
        barrier()
        el2 = time.perf_counter() - t1
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        sustained = {"steps": n_more, "ms_per_step": 1e3 * el2 / n_more, "value": world * B * n_more / el2}

    dp = None
    if world > 1 or os.environ.get("DOF_BENCH_FORCE_PG") == "1":
        import torch.distributed as dist
        from deepof_amd.training import dp_form, dp_check_verdict, _dp_switches, _native_comm
        devs = [None] * dist.get_world_size()
        dist.all_gather_object(devs, torch.cuda.current_device())
        per_rank_ms = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank_ms, 1e3 * elapsed_own / args.steps)
        # latency of the gradient all-reduce alone, eagerly enqueued on the current stream (HIP events, 100 calls), in the
        # form the step uses and through torch.distributed: a bad scaling point can be read off the line
        scratch = torch.zeros_like(eng.grads)

        def ar_latency_us(fn, calls=100):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(calls):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / calls

        native, _one_graph = _dp_switches(eng, dist)
        lat = {"torch.distributed.all_reduce": ar_latency_us(lambda: dist.all_reduce(scratch, op=dist.ReduceOp.SUM))}
        if native:
            comm = _native_comm(eng, dist)
            lat["dof_flat_allreduce"] = ar_latency_us(lambda: comm.all_reduce_(scratch))
        dp = {"backend": dist.get_backend(), "rccl_world_size": dist.get_world_size() if dist.get_backend() == "nccl" else None,
              "world_size": dist.get_world_size(), "form": dp_form(eng, dist), "self_check": dp_check_verdict(eng, dist),
              "allreduce_bytes_per_step": int(eng.grads.numel()) * 4, "collectives_per_step": 1,
              "allreduce_eager_latency_us": lat, "ms_per_step_per_rank": per_rank_ms,
              "devices": devs}
    # algorithmic work of one step against the fp32 vector peak (C2 7.6 MFLOP forward per window, training =
    # 3 x; C4: 4.25 TFLOP per step of 8192 windows; C5: not tabulated)
    flops_per_step = {"c2": 3.0 * 7.6e6 * B, "c4": 4.25e12 * B / 8192.0, "c5": None}[cfg]
    out = {
        "metric": metric, "value": value, "unit": "windows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": world * B, "window": T, "parallelism": f"dp{world}",
                   "hip_graph": stepper.graphs.enabled, "graph_replays": stepper.graphs.replays,
                   "path": step_path,
                   "final_total_loss": logs["total_loss"]},
        "sustained": sustained, "data_parallel": dp,
        "roofline_step": None if flops_per_step is None else {
            "flop_frac": flops_per_step / (ms_per_step * 1e-3) / FP32_PEAK_FLOPS,
            "achieved_tflops": flops_per_step / (ms_per_step * 1e-3) / 1e12, "peak_tflops": FP32_PEAK_FLOPS / 1e12,
            "algorithmic_flops_per_step": flops_per_step, "hbm_frac": None, "hbm_bytes_per_step": None},
    }
    # HBM bytes of one step from the round's PMC passes (tools/profile_step_hbm.sh): quoted only while the kernel sources
    # are the ones the passes ran on
    if B == 1024 and cfg == "c2":
        rec, name = _latest_profile("step_pmc.json", _source_sha(STEP_SOURCES))
        if rec is not None:
            hb = rec["hbm_bytes_per_step"]
            out["roofline_step"].update(hbm_bytes_per_step=hb, hbm_frac=hb / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9),
                                        hbm_bytes_source=name)

    if rank == 0:
        # ---- roofline of the HBM-bound window-gather kernel: full materialisation of this rank's dataset
        nw = n_windows
        xg = torch.empty(nw, T, N, 3, device=dev)
        ag = torch.empty(nw, T, E, 1, device=dev)

        def stream():
            return torch.cuda.current_stream().cuda_stream

        def gather_all():
            for i in range(n_animals):
                lo = i * win_per_animal
                _capi.check(lib, lib.dof_window_gather_range(
                    tn.data_ptr(), te.data_ptr(), i * F, 1, win_per_animal, T, N, E,
                    xg[lo:].data_ptr(), ag[lo:].data_ptr(), stream()))

        gather_all()
        gather_all()   # two untimed passes: page the output in, settle clocks
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(args.gather_iters):
            gather_all()
        ev1.record()
        torch.cuda.synchronize()
        launches = args.gather_iters * n_animals
        sec_per_launch = ev0.elapsed_time(ev1) * 1e-3 / launches
        bytes_per_window = T * (3 * N + E) * 4 + (3 * N + E) * 4
        alg_bytes = win_per_animal * bytes_per_window
        achieved = alg_bytes / sec_per_launch / 1e9
        # HBM bytes/launch from rocprofv3 PMC passes of the same launch (tools/gather_pmc.sh -> profiles/rNN_gather_pmc.json,
        # the newest one taken on the present kernel source); null when the source has changed since
        traffic, traffic_src = None, None
        if win_per_animal == 599976 and (T, N, E) == (25, 14, 14):
            rec, traffic_src = _latest_profile("gather_pmc.json", _source_sha(["k_gather.hip"]))
            if rec is not None:
                traffic = rec["hbm_bytes_per_launch"]
        out["roofline"] = {"kernel": "k_window_gather", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "bytes_per_window": bytes_per_window, "windows_per_launch": win_per_animal,
                           "avg_launch_ms": sec_per_launch * 1e3}
        del xg, ag
        # same-box denominators ("use the measured copy peak as denominator too"): a pure fill of the launch's
        # 3.49 GB (the gather is 96 % stores) and a device copy moving the same bytes (half read, half written)
        nfl = alg_bytes // 4
        buf = torch.empty(nfl, device=dev)

        def timed_gbs(fn, moved, iters=10):
            fn(); fn()
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(iters):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            return moved / (ev0.elapsed_time(ev1) * 1e-3 / iters) / 1e9
        fill_gbs = timed_gbs(lambda: buf.fill_(1.0), nfl * 4)
        half = nfl // 2
        copy_gbs = timed_gbs(lambda: buf[:half].copy_(buf[half:2 * half]), 2 * half * 4)
        del buf
        measured_peak = max(fill_gbs, copy_gbs)
        out["roofline"].update(measured_peak=measured_peak, frac_of_measured=achieved / measured_peak,
                               measured_peak_kernels={"fill_3.49GB": fill_gbs, "copy_3.49GB_moved": copy_gbs},
                               measured_peak_note="same process, same box, HIP events; torch fill_/copy_ elementwise kernels")
        # the same launch writing bf16 (BASELINE configs[1] names bf16; SURVEY 8(d): 3,024 B per C2 window)
        xg16 = torch.empty(nw, T, N, 3, device=dev, dtype=torch.bfloat16)
        ag16 = torch.empty(nw, T, E, 1, device=dev, dtype=torch.bfloat16)

        def gather_all16():
            for i in range(n_animals):
                lo = i * win_per_animal
                _capi.check(lib, lib.dof_window_gather_bf16(tn.data_ptr(), te.data_ptr(), None, i * F, 1, win_per_animal, T, N, E,
                                                            xg16[lo:].data_ptr(), ag16[lo:].data_ptr(), stream()))

        gather_all16()
        gather_all16()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(args.gather_iters):
            gather_all16()
        ev1.record()
        torch.cuda.synchronize()
        sec16 = ev0.elapsed_time(ev1) * 1e-3 / launches
        bpw16 = T * (3 * N + E) * 2 + (3 * N + E) * 4
        out["roofline_bf16_storage"] = {"kernel": "k_window_gather<.., OUT16>", "bound": "hbm",
                                        "achieved": win_per_animal * bpw16 / sec16 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": win_per_animal * bpw16 / sec16 / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                        "bytes_per_window": bpw16, "windows_per_launch": win_per_animal,
                                        "avg_launch_ms": sec16 * 1e3, "windows_per_s": win_per_animal / sec16,
                                        "measured_peak": measured_peak,
                                        "frac_of_measured": win_per_animal * bpw16 / sec16 / 1e9 / measured_peak}
        del xg16, ag16
        if not args.no_cpu_baseline and world == 1 and cfg == "c2":
            xb, ab = ds.fetch(starts[0], starts[0] + B)   # the first batch the device path trained on
            out["cpu_baseline"] = cpu_baseline(initial_state, xb.cpu(), ab.cpu(), L, K)
        if not args.no_secondary and world == 1 and cfg == "c2":
            del stepper, ds, tau_star, model, eng
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["secondary"] = secondary_configs()
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
