#!/usr/bin/env python
"""Headline benchmark: pose-windows/sec of the VaDE train step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic windows, inputs resident in
HBM: window gather from the frame table -> VaDE forward -> VadeLoss -> backward -> [RCCL
all-reduce of the flat gradient] -> clip + Adam.  Workload = BASELINE config C2 (VaDE recurrent,
14 body parts, window 25, k=10, latent 8, batch 1024 per GPU, main phase with distillation).
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
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling


def synth_tables(n_frames, n_nodes, n_edges, seed):
    """AR(1) (rho 0.95) random walks per column, standardised, clipped to +-10."""
    rng = np.random.default_rng(seed)
    cols = 3 * n_nodes + n_edges
    noise = rng.standard_normal((n_frames, cols)).astype(np.float32)
    out = np.empty_like(noise)
    acc = np.zeros(cols, dtype=np.float32)
    # block-wise AR(1) to keep generation fast for large tables
    rho = np.float32(0.95)
    for i in range(n_frames):
        acc = rho * acc + noise[i]
        out[i] = acc
    out = (out - out.mean(0)) / (out.std(0) + 1e-6)
    np.clip(out, -10, 10, out=out)
    return out[:, : 3 * n_nodes].copy(), out[:, 3 * n_nodes:].copy()


def synth_tables_fast(n_frames, n_nodes, n_edges, seed, device):
    """Same process on device (lfilter via cumulative products in chunks would be overkill): use a
    short CPU seed table tiled with phase shifts + device noise; standardised, clipped."""
    base_n, base_e = synth_tables(20000, n_nodes, n_edges, seed)
    reps = (n_frames + 19999) // 20000
    tn = torch.from_numpy(base_n).to(device).repeat(reps, 1)[:n_frames].contiguous()
    te = torch.from_numpy(base_e).to(device).repeat(reps, 1)[:n_frames].contiguous()
    g = torch.Generator(device=device).manual_seed(seed)
    tn += 0.05 * torch.randn(tn.shape, device=device, generator=g)
    te += 0.05 * torch.randn(te.shape, device=device, generator=g)
    return tn.clamp_(-10, 10), te.clamp_(-10, 10)


def init_params(eng, seed=0):
    """Random-init weights of the reference architecture (PyTorch default inits; models_new.py)."""
    g = torch.Generator().manual_seed(seed)
    L = eng.L
    for n in eng.names:
        shape = eng.layout[n][2]
        if "norm" in n and n.endswith("weight"):
            v = torch.ones(shape)
        elif "norm" in n and n.endswith("bias"):
            v = torch.zeros(shape)
        elif ".gru" in n:
            hid = shape[0] // 3
            v = (torch.rand(shape, generator=g) * 2 - 1) / np.sqrt(hid)
        elif n.startswith("latent_space.gmm"):
            v = torch.randn(shape, generator=g) * np.sqrt(2.0 / (shape[0] + shape[1]))
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = (torch.rand(shape, generator=g) * 2 - 1) / np.sqrt(fan_in)
        else:
            v = (torch.rand(shape, generator=g) * 2 - 1) * 0.1
        eng.view(n).copy_(v)


def cpu_baseline(P, B, T, N, E, L, K, steps=3, warm=1):
    """Oracle ('port' of the reference PyTorch-CPU path) timed on a bounded sample of the same workload:
    same architecture and initial weights (state_dict P), same batch size, main phase with distillation."""
    from oracle import vade as OV

    torch.manual_seed(0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, T, N, 3, generator=g)
    a = torch.randn(B, T, E, 1, generator=g)
    tau = torch.softmax(torch.randn(B, K, generator=g), dim=-1)
    pi = tau.mean(0).clamp_min(1e-8)
    w = pi.pow(-1.0)
    w = (w / w.mean()).clamp_max(3.0)
    cfg = OV.VadeLossCfg(K, False, lambda_distill=4.0, class_weight=w, teacher_marginal=pi)
    all_cores = torch.get_num_threads()
    results = {}
    # tiny-op eager PyTorch does not scale with threads: report the best of a few thread counts
    for threads in sorted({min(8, all_cores), min(32, all_cores), all_cores}):
        torch.set_num_threads(threads)
        P_run = {k: v.clone() for k, v in P.items()}
        opt = OV.AdamState()
        times = []
        for i in range(warm + steps):
            t0 = time.perf_counter()
            eps = torch.randn(B, L)
            eps_mc = torch.randn(32, B, L)
            OV.vade_train_step(P_run, opt, x, a, cfg, 1.0, 5e-4, 2e-4, eps, eps_mc, tau)
            dt = time.perf_counter() - t0
            if i >= warm:
                times.append(dt)
        results[threads] = B / float(np.median(times))
    torch.set_num_threads(all_cores)
    best = max(results, key=results.get)
    others = ", ".join(f"{t} threads: {v:.1f}" for t, v in sorted(results.items()))
    return {"value": results[best], "unit": "windows/s", "cores": best, "kind": "port",
            "sample": f"{steps} train steps of batch {B} after {warm} warm-up per thread count (oracle/vade.py, "
                      f"torch CPU fp32, median; host has {all_cores} threads; windows/s by thread count: {others})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--frames", type=int, default=600_000, help="frames per synthetic animal (2 animals per rank)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-iters", type=int, default=20)
    ap.add_argument("--log-every", type=int, default=0, help="debug: print the loss terms every N steps (adds syncs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    # DOF_BENCH_SHARE_GPU=1 (debug only): all ranks on cuda:0 over gloo, to exercise the N > 1 control flow (all-reduce,
    # barriers, max-over-ranks, rank-0 JSON) on a 1-GPU box; RCCL refuses two ranks on one device.  Never a result.
    share = os.environ.get("DOF_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from deepof_amd import _capi
    from deepof_amd.engine import create_vade_engine
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph
    from parity_common import configure_phase

    nodes, edges = bodypart_graph([""])
    adj = adjacency_from_graph(nodes, edges)
    N, E = len(nodes), len(edges)
    B, T, L, K, S = args.batch, 25, 8, 10, 32
    eng = create_vade_engine(B, T, adj, L, K, S, device=dev)
    lib = eng.lib
    init_params(eng, seed=0)  # identical on every rank (DDP broadcast equivalent)
    initial_state = eng.state_dict() if rank == 0 else None

    # --- device-resident dataset: 2 animals per rank, concatenated frame tables + window start rows
    n_animals, F = 2, args.frames
    tn, te = synth_tables_fast(n_animals * F, N, E, seed=rank, device=dev)
    win_per_animal = F - T + 1
    starts = torch.cat([torch.arange(win_per_animal, device=dev, dtype=torch.int64) + i * F for i in range(n_animals)])
    n_windows = int(starts.numel())
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    tau_star = torch.softmax(torch.randn(n_windows, K, device=dev, generator=g) * 2.0, dim=-1)
    pi = tau_star.mean(0).clamp_min(1e-8)
    cw = pi.pow(-1.0)
    cw = (cw / cw.mean()).clamp_max(3.0)

    # main phase with distillation (reference defaults, training.py:592-719; lr after epoch 0: 5e-4 / 2e-4)
    for seg in (_capi.SEG_ENCODER, _capi.SEG_DECODER, _capi.SEG_HEADS):
        eng.set_lr(seg, 5e-4)
    eng.set_lr(_capi.SEG_GMM, 2e-4)
    configure_phase(eng, K, False, 1.0, None, 4.0)
    eng.set_teacher(cw, pi)
    eng.push_hyper()

    x = torch.empty(B, T, N, 3, device=dev)
    a = torch.empty(B, T, E, 1, device=dev)
    eps = torch.empty(B, L, device=dev)
    eps_mc = torch.empty(S, B, L, device=dev)
    tau = torch.empty(B, K, device=dev)
    batch_rows = torch.empty(B, dtype=torch.int64, device=dev)
    n_batches = n_windows // B
    perm = torch.randperm(n_batches, generator=torch.Generator().manual_seed(0)).tolist()

    def stream():
        return torch.cuda.current_stream().cuda_stream

    def step_body():
        # batch = contiguous block of windows (reference loader: block shuffle of batch starts, dataset.py:589-634)
        _capi.check(lib, lib.dof_window_gather(tn.data_ptr(), te.data_ptr(), batch_rows.data_ptr(), B, T, N, E,
                                               x.data_ptr(), a.data_ptr(), stream()))
        eps.normal_()
        eps_mc.normal_()
        eng.loss_grads(x, a, eps, eps_mc, tau, pretrain=False)

    def opt_body():
        eng.optimizer_step()

    graph_a = graph_b = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        eng.advance_adam()
        eng.push_hyper()
        with torch.cuda.stream(side):
            batch_rows.copy_(starts[:B])
            tau.copy_(tau_star[:B])
            step_body()
            opt_body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        eng.reset_optimizer()
        graph_a, graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_a):
            step_body()
        with torch.cuda.graph(graph_b):
            opt_body()

    def one_step(i):
        b0 = perm[i % n_batches] * B
        batch_rows.copy_(starts[b0:b0 + B])
        tau.copy_(tau_star[b0:b0 + B])
        eng.advance_adam()
        eng.push_hyper()
        if graph_a is not None:
            graph_a.replay()
        else:
            step_body()
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM)
            eng.grads.mul_(1.0 / world)
        if graph_b is not None:
            graph_b.replay()
        else:
            opt_body()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
        if args.log_every and rank == 0 and i % args.log_every == 0:
            print("warmup", i, {k: round(v, 4) for k, v in eng.read_logs().items()}, file=sys.stderr, flush=True)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(args.warmup + i)
        if args.log_every and rank == 0 and i % args.log_every == 0:
            print("step", i, {k: round(v, 4) for k, v in eng.read_logs().items()}, file=sys.stderr, flush=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    logs = eng.read_logs()
    if not np.isfinite(logs["total_loss"]):
        raise SystemExit(f"non-finite loss after benchmark steps: {logs}")
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * B * args.steps / elapsed

    out = {
        "metric": "pose-windows/sec (train step) VaDE 14-bp win=25", "value": value, "unit": "windows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: VaDE GM-VAE recurrent, 14 body parts (N=14,E=14), window=25, k=10, latent=8, "
                               f"batch={B}/GPU, main phase (MC-KL S=32 + distillation), fp32",
                   "global_batch": world * B, "window": T, "parallelism": f"dp{world}",
                   "hip_graph": graph_a is not None, "final_total_loss": logs["total_loss"]},
    }

    if rank == 0:
        # ---- roofline of the HBM-bound window-gather kernel: full materialisation of this rank's dataset
        nw = n_windows
        xg = torch.empty(nw, T, N, 3, device=dev)
        ag = torch.empty(nw, T, E, 1, device=dev)

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
        traffic = None  # HBM bytes/launch from rocprofv3 PMC passes of this same launch (profiles/r01_gather_pmc.json)
        pmc_file = os.path.join(ROOT, "profiles", "r01_gather_pmc.json")
        if os.path.exists(pmc_file) and win_per_animal == 599976 and (T, N, E) == (25, 14, 14):
            traffic = json.load(open(pmc_file))["hbm_bytes_per_launch"]
        out["roofline"] = {"kernel": "k_window_gather", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "bytes_per_window": bytes_per_window, "windows_per_launch": win_per_animal,
                           "avg_launch_ms": sec_per_launch * 1e3}
        del xg, ag
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(initial_state, B, T, N, E, L, K)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
