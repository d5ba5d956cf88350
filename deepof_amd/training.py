"""Public trainer API of the hot path (mirrors /root/reference/deepof/clustering/training.py).

``train_deepof_model(...)`` keeps the reference's signature, defaults, enum handling, return
tuple ``(model_val, model_score, model_teacher_init_or_None, log_summary)`` and on-disk artefacts
(checkpoint bundle + ``_info.txt``), so ``Coordinates.deep_unsupervised_embedding`` can call it
unchanged.  The epoch loop, the step and the optimiser run in libdeepof_hip on a
ROCm device; data parallelism = one process per GPU, one RCCL all-reduce of the flat gradient per
step (torch.distributed "nccl").  Model families: VaDE, VQ-VAE and contrastive, each with the recurrent, TCN or
transformer encoder (``encoder_type``); latent_dim in config.SUPPORTED_LATENT_DIMS (check_model_inputs rejects other values up front).
"""
from __future__ import annotations

import math
import os
import sys
import time
import warnings
from copy import deepcopy
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _capi
from .config import (CommonFitCfg, ContrastiveCfg, TurtleTeacherCfg, VaDECfg, cfg_lines, check_model_inputs)
from .dataset import WindowDataset, n_batches
from .models import Contrastive, VaDE, VQVAE
from .schedules import WeightSchedule
from .stepping import DeviceSchedule, StepGraphs, constant_schedule
from .tb_log import log_epoch_to_tensorboard, open_writer

LOG_SUMMARY_KEYS = ("total_loss", "reconstruction_loss", "kl_divergence", "cat_cluster_loss", "kmeans_loss",
                    "distill_loss", "temporal_loss", "scatter_loss", "nonempty_loss", "repel_loss", "tf_cluster_loss",
                    "prior_loss", "activity_l1", "pos_similarity", "neg_similarity", "conf_norm", "bal_norm",
                    "alignment_score")


# ------------------------------------------------------------------------------------------------
# distributed plumbing (model_utils_new.py:196-226)
# ------------------------------------------------------------------------------------------------
def ddp_init_if_needed(backend: str = "nccl"):
    import torch.distributed as dist

    if "RANK" not in os.environ and "SLURM_PROCID" in os.environ:
        os.environ["RANK"] = os.environ["SLURM_PROCID"]
        os.environ["WORLD_SIZE"] = os.environ["SLURM_NTASKS"]
        os.environ["LOCAL_RANK"] = os.environ.get("SLURM_LOCALID", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    if dist.is_available() and dist.is_initialized():
        return True, dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ or not torch.cuda.is_available():
        return False, 0, 1, 0
    if int(os.environ["WORLD_SIZE"]) <= 1:
        return False, 0, 1, 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, init_method="env://")
    return True, dist.get_rank(), dist.get_world_size(), local_rank


TB_WRITER = None  # the running fit's event-file writer (deepof_amd.tb_log), set by train_deepof_model_base


def _dist_state():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def _dp_step(graphs, key, slot, forward_backward, eng, dist, world: int):
    """The data-parallel step: [gather + loss + gradients] -> ONE all-reduce (SUM) of the flat gradient over RCCL ->
    [clip + Adam with grad_scale 1 / world].

    Default on the RCCL ("nccl") backend: the collective is the C ABI's ``dof_flat_allreduce`` on the step's OWN stream,
    captured with everything else into ONE hipGraph -- a data-parallel step is a single graph replay, no event hop to
    a communication stream and no second graph launch.  Before the first step takes that form, `dp_self_check` runs it
    (eagerly and captured) on a scratch buffer against ``torch.distributed.all_reduce``; on a mismatch, an error or a
    timeout EVERY rank falls back to the safe form -- ``torch.distributed.all_reduce`` between two captured graphs --
    with one warning line, and `dp_form` reports the form actually taken.  ``DOF_DP_NATIVE=0`` asks for the safe form
    directly; ``DOF_DP_ONE_GRAPH=1`` with it captures the torch collective into the step graph (opt-in: never validated
    multi-rank); ``DOF_DP_ONE_GRAPH=0`` keeps two graphs around an eagerly enqueued native collective.  Other backends
    (gloo: the CPU tests) always take the torch.distributed form, two graphs.  Decided once per process group
    (``_dp_switches``).  Reference: DDP's bucketed gradient all-reduce, /root/reference/deepof/clustering/
    model_utils_new.py:196-226, training.py:1567-1576."""
    scale = 1.0 / world
    native, one_graph = _dp_switches(eng, dist)
    if native:
        comm = _native_comm(eng, dist)
        reduce = lambda: comm.all_reduce_(eng.grads)
    else:
        reduce = lambda: dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM)
    if one_graph:
        def whole():
            forward_backward()
            reduce()
            eng.optimizer_step(scale)
        graphs.run(key + ("dp",), whole, None if slot is None else slot + ("dp",))
        return
    graphs.run(key + ("grads",), forward_backward, None if slot is None else slot + ("grads",))
    reduce()
    graphs.run((eng.B, world, "adam"), lambda: eng.optimizer_step(scale))


_DP_SWITCHES = {}
_DP_CHECK = {}   # process-group identity -> the self-check's verdict string ("" when no check was due)


_DP_CHECK_LEAKED = []   # buffers / graphs of a timed-out self-check (deliberately never freed)


def dp_self_check(candidate, dist, like: torch.Tensor, captured: bool = False, timeout_s: float = None, on_timeout=None):
    """Does ``candidate(t)`` -- an in-place SUM over the ranks, enqueued on the current stream -- agree with
    ``torch.distributed.all_reduce`` on a scratch buffer shaped like ``like`` (the flat gradient)?

    Every rank fills the scratch with its own seeded values, reduces one copy through torch.distributed and one through
    the candidate on a side stream, polls an event (never a blocking wait: a collective that hangs must not hang the
    fit) until ``timeout_s`` (DOF_DP_CHECK_TIMEOUT, default 30 s), and compares: bitwise, else 1e-6 of the largest
    magnitude.  ``captured``: the candidate is then also captured into a hipGraph and replayed once -- the form the
    one-graph step uses.  The verdict is agreed across ranks (MIN over an all-reduced flag), so either all ranks take
    the candidate or none does.  Returns (ok, why)."""
    if timeout_s is None:
        timeout_s = float(os.environ.get("DOF_DP_CHECK_TIMEOUT", "30"))
    rank = dist.get_rank()
    dev = like.device
    n = int(like.numel())
    gen = torch.Generator(device="cpu")
    gen.manual_seed(7919 + 104729 * rank)
    probe = torch.randn(max(n, 1), generator=gen, dtype=torch.float32).to(dev)
    ref = probe.clone()
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    ok, why = True, ""
    hung = []

    def agrees(got, what):
        if torch.equal(got, ref):
            return True, ""
        err = float((got - ref).abs().max())
        bar = 1e-6 * max(float(ref.abs().max()), 1.0)
        if err <= bar:
            return True, ""
        return False, f"{what} differs from torch.distributed.all_reduce by {err:.3e} (bar {bar:.1e})"

    def finished(event):
        deadline = time.monotonic() + timeout_s
        while not event.query():
            if time.monotonic() > deadline:
                return False
            time.sleep(0.002)
        return True

    try:
        if dev.type == "cuda":
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                got = probe.clone()
                candidate(got)
                done = torch.cuda.Event()
                done.record(side)
            if not finished(done):
                ok, why = False, f"the collective did not complete within {timeout_s:.0f} s"
                hung += [got, side]
            else:
                ok, why = agrees(got, "the eager collective")
            if ok and captured:
                with torch.cuda.stream(side):
                    got2 = probe.clone()
                    side.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        candidate(got2)
                    g.replay()
                    done2 = torch.cuda.Event()
                    done2.record(side)
                if not finished(done2):
                    ok, why = False, f"the captured collective did not complete within {timeout_s:.0f} s"
                    hung += [got2, g, side]
                else:
                    ok, why = agrees(got2, "the captured collective")
                    del g
        else:
            got = probe.clone()
            candidate(got)
            ok, why = agrees(got, "the collective")
    except Exception as exc:  # a failing candidate must end in the safe form, not in a dead fit
        ok, why = False, f"{type(exc).__name__}: {exc}"
    if hung:
        # a collective that never finished may still be reading its buffers / replaying its graph: keep them alive for the
        # rest of the process, and drop the communicator BEFORE the verdict's all-reduce has to share the device with it
        _DP_CHECK_LEAKED.extend(hung)
        if on_timeout is not None:
            on_timeout()
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if ok and float(flag.item()) != 1.0:
        ok, why = False, "another rank's check failed"
    return ok, why


def _decide_dp_form(rccl: bool, dist, grads: torch.Tensor, native_reduce_factory, on_failure=None, env=None):
    """(native, one_graph, verdict): the form the data-parallel step takes on this process group.  ``rccl``: the group
    runs on RCCL with the gradient on a ROCm device (any other backend takes the torch.distributed form between two
    graphs).  ``native_reduce_factory()`` returns the native in-place SUM (it may raise: no librccl, no communicator)."""
    env = os.environ if env is None else env
    native = rccl and env.get("DOF_DP_NATIVE", "1") != "0"
    # the torch.distributed collective is captured into the step graph only on request: that form was never validated
    # on more than one rank
    one_graph = rccl and env.get("DOF_DP_ONE_GRAPH", "1" if native else "0") != "0"
    verdict = ""
    if native and env.get("DOF_DP_SELF_CHECK", "1") != "0":
        try:
            reduce = native_reduce_factory()
        except Exception as exc:
            reduce = None
            why = f"{type(exc).__name__}: {exc}"
        if reduce is None:
            def reduce(_t, _why=why):
                raise RuntimeError(_why)
        ok, why = dp_self_check(reduce, dist, grads, captured=one_graph, on_timeout=on_failure)
        verdict = "passed" if ok else f"failed ({why})"
        if not ok:
            native, one_graph = False, False
            if on_failure is not None:
                on_failure()
            if dist.get_rank() == 0:
                print(f"deepof_amd: WARNING native data-parallel collective self-check {verdict}; "
                      "falling back to torch.distributed.all_reduce between two graphs", file=sys.stderr, flush=True)
    return native, one_graph, verdict


def _dp_switches(eng, dist):
    """(native collective, one graph) for this process group, decided once (`_decide_dp_form`): the RCCL backend on a
    ROCm device takes the one-graph native form when `dp_self_check` passes, unless DOF_DP_NATIVE=0 /
    DOF_DP_ONE_GRAPH=0 say otherwise; any other backend takes the torch.distributed form between two graphs."""
    ident = _pg_identity(eng, dist)
    sw = _DP_SWITCHES.get(ident)
    if sw is None:
        rccl = eng.params.is_cuda and dist.get_backend() == "nccl"

        def factory():
            comm = _native_comm(eng, dist)
            return lambda t: comm.all_reduce_(t)

        native, one_graph, verdict = _decide_dp_form(rccl, dist, eng.grads, factory, on_failure=abort_native_comm)
        _DP_SWITCHES.clear()
        _DP_CHECK.clear()
        sw = _DP_SWITCHES[ident] = (native, one_graph)
        _DP_CHECK[ident] = verdict
    return sw


def dp_form(eng, dist) -> str:
    """Name of the data-parallel form `_dp_step` takes (bench / log lines)."""
    native, one_graph = _dp_switches(eng, dist)
    return ("dof_flat_allreduce" if native else "torch.distributed.all_reduce") + \
           (" captured in the step graph" if one_graph else " between two graphs")


def dp_check_verdict(eng, dist) -> str:
    """"passed" / "failed (...)" of the native collective's self-check for this process group, "" when none was due."""
    _dp_switches(eng, dist)
    return _DP_CHECK.get(_pg_identity(eng, dist), "")


def _pg_identity(eng, dist):
    """What a cached communicator / switch pair belongs to: the default process group object (a re-initialised group is
    a new object), its shape, and the device."""
    pg = dist.distributed_c10d._get_default_group()
    return (id(pg), dist.get_rank(), dist.get_world_size(), str(eng.params.device))


_NATIVE_COMM = {}


def _native_comm(eng, dist):
    """This process's RCCL communicator behind the C ABI (deepof_amd.comm.NativeComm), created on first use: rank 0's
    unique id travels over the initialised torch.distributed group.  Cached per (process group, rank, world, device);
    a communicator of another group is closed when a new one is made, and `close_native_comm` drops it at the end of
    a fit."""
    ident = _pg_identity(eng, dist)
    comm = _NATIVE_COMM.get(ident)
    if comm is None:
        close_native_comm()
        from .comm import NativeComm
        comm = _NATIVE_COMM[ident] = NativeComm.from_process_group(eng.lib, dist)
    return comm


def close_native_comm():
    for comm in _NATIVE_COMM.values():
        comm.close()
    _NATIVE_COMM.clear()


def abort_native_comm():
    """Drop the communicator without waiting for collectives in flight (``ncclCommAbort``): the self-check's way out of
    a collective that hung or produced wrong sums."""
    for comm in _NATIVE_COMM.values():
        comm.abort()
    _NATIVE_COMM.clear()


def _dp_active(dist, world: int) -> bool:
    """True when steps take the data-parallel route (gradient all-reduce between the two step graphs).
    DOF_FORCE_DP=1 takes it with a 1-rank group too: exercises RCCL + graph replay on a single-GPU box."""
    return world > 1 or (dist is not None and os.environ.get("DOF_FORCE_DP") == "1")


# ------------------------------------------------------------------------------------------------
# checkpoint bundles (model_utils_new.py:263-329, 368-374, 822-938)
# ------------------------------------------------------------------------------------------------
def ckpt_paths(model_name: str, common_cfg: CommonFitCfg):
    ckpt_dir = os.path.join(common_cfg.output_path, "models", model_name.lower(), f"run_{common_cfg.run}")
    os.makedirs(ckpt_dir, exist_ok=True)
    return (ckpt_dir, os.path.join(ckpt_dir, "best_model_val.pth"), os.path.join(ckpt_dir, "best_model_score.pth"),
            os.path.join(ckpt_dir, "model_teacher_init.pth"))


def save_model_info(ckpt_path: str, *, stage: str, epoch=None, train_steps=None, val_total=None, score_value=None,
                    extra=None, common_cfg=None, teacher_cfg=None, vade_cfg=None, contrastive_cfg=None, model=None,
                    log_summary=None, rebuild_spec=None, save_weights: bool = True) -> None:
    info_path = os.path.splitext(ckpt_path)[0] + "_info.txt"
    lines = [f"stage: {stage}"]
    for label, v, cast in (("epoch", epoch, int), ("train_steps", train_steps, int), ("val_total", val_total, float),
                           ("score_value", score_value, float)):
        if v is not None:
            lines.append(f"{label}: {cast(v)}")
    lines.append("")
    if save_weights and model is not None:
        keys = ["state_dict"] + (["rebuild_spec"] if rebuild_spec is not None else []) + \
            (["log_summary"] if log_summary is not None else [])
        lines += ["[checkpoint_format]", "ckpt_contains: bundle", "bundle_keys: " + ", ".join(keys), ""]
    for title, cfg in (("common_cfg", common_cfg), ("teacher_cfg", teacher_cfg), ("vade_cfg", vade_cfg),
                       ("contrastive_cfg", contrastive_cfg)):
        lines += cfg_lines(title, cfg)
    if extra:
        lines += ["[extra]"] + [f"{k}: {extra[k]}" for k in sorted(extra)] + [""]
    os.makedirs(os.path.dirname(ckpt_path), exist_ok=True)
    if save_weights and model is not None:
        payload = {"state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}
        if rebuild_spec is not None:
            payload["rebuild_spec"] = rebuild_spec
        if log_summary is not None:
            payload["log_summary"] = log_summary
        torch.save(payload, ckpt_path)
    with open(info_path, "w", encoding="utf-8") as f:
        f.write("\n".join(lines))


def build_model_from_spec(spec: dict, device=None, batch_size: int = 256, _engine_factory=None) -> VaDE:
    name = str(spec.get("model_name", "vade")).lower()
    if name == "vqvae":
        return VQVAE(tuple(spec["x_shape"]), tuple(spec["a_shape"]), np.asarray(spec["adjacency_matrix"]),
                     int(spec["latent_dim"]), int(spec["n_components"]),
                     encoder_type=spec.get("encoder_type", "recurrent"), use_gnn=bool(spec.get("use_gnn", True)),
                     batch_size=batch_size, device=device, _engine_factory=_engine_factory)
    if name == "contrastive":
        return Contrastive(tuple(spec["x_shape"]), tuple(spec["a_shape"]), np.asarray(spec["adjacency_matrix"]),
                           int(spec["latent_dim"]), encoder_type=spec.get("encoder_type", "TCN"),
                           n_components=int(spec.get("n_components", 1)),
                           use_gnn=bool(spec.get("use_gnn", True)), batch_size=batch_size, device=device,
                           _engine_factory=_engine_factory)
    if name != "vade":
        raise NotImplementedError(f"checkpoint of model {name!r}: unknown model name")
    return VaDE(tuple(spec["x_shape"]), tuple(spec["a_shape"]), np.asarray(spec["adjacency_matrix"]),
                int(spec["latent_dim"]), int(spec["n_components"]), encoder_type=spec.get("encoder_type", "recurrent"),
                use_gnn=bool(spec.get("use_gnn", True)), kmeans_loss=float(spec.get("kmeans_loss", 0.0)),
                batch_size=batch_size, device=device, _engine_factory=_engine_factory)


def load_model_from_ckpt(path: str, device=None, _engine_factory=None):
    """-> (model, log_summary, rebuild_spec, load_report); bundle written by save_model_info (reference-compatible)."""
    bundle = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(bundle, dict) or "state_dict" not in bundle or "rebuild_spec" not in bundle:
        raise RuntimeError(f"{path} is not a model bundle with state_dict + rebuild_spec")
    model = build_model_from_spec(bundle["rebuild_spec"], device=device, _engine_factory=_engine_factory)
    rep = model.load_state_dict(bundle["state_dict"], strict=False)
    model.eval()
    report = {"missing": list(rep.missing_keys), "unexpected": list(rep.unexpected_keys)}   # as the reference reports
    return model, bundle.get("log_summary"), bundle["rebuild_spec"], report


INFERENCE_BATCH = 256  # plan/workspace size of returned models (embedding_per_video encodes 256 windows at a time)


def _clone_model(model: VaDE) -> VaDE:
    """Independent copy (own parameter buffer) -- deepcopy() of the reference.  The copy starts with an
    inference-sized plan (other batch sizes are created on demand), not with a training-sized workspace."""
    if isinstance(model, Contrastive):
        twin = Contrastive(model.input_shape, model.edge_feature_shape, model._adjacency, model.latent_dim,
                           encoder_type=model.encoder_type, n_components=model.n_components, temperature=model.temperature,
                           similarity_function=model.similarity_function,
                           loss_function=model.loss_function, beta=model.beta, tau=model.tau,
                           batch_size=min(model._base.B, INFERENCE_BATCH), _engine_factory=model._factory)
        twin._base.params.copy_(model._base.params)
        for k, t in model._base.num_batches_tracked.items():
            twin._base.num_batches_tracked[k].copy_(t)
        twin.train(model.training)
        return twin
    twin = type(model)((model.window_size, model.input_n_nodes, 3), (model.window_size, model._base.E, 1),
                       model._adjacency, model.latent_dim, model.n_components, encoder_type=model.encoder_type,
                       kmeans_loss=model.kmeans_weight, batch_size=min(model._base.B, INFERENCE_BATCH),
                       _engine_factory=model._factory)
    twin._base.params.copy_(model._base.params)
    twin._base.prior.copy_(model._base.prior)
    for k, t in model._base.num_batches_tracked.items():
        twin._base.num_batches_tracked[k].copy_(t)
    twin.train(model.training)
    return twin


def load_best_checkpoints(model: VaDE, best_path_val: str, best_path_score: str, save_weights: bool):
    """(model_val, model_score): reload the saved bests, else the last weights for both (Q18)."""
    out = []
    for path in (best_path_val, best_path_score):
        if save_weights and os.path.exists(path):
            m = _clone_model(model)
            m.load_state_dict(torch.load(path, map_location="cpu", weights_only=False)["state_dict"], strict=False)
            m.eval()
            out.append(m)
        else:
            out.append(model)
    return out[0], out[1]


# ------------------------------------------------------------------------------------------------
# logging (logging.py:304-349, 427-433)
# ------------------------------------------------------------------------------------------------
def init_log_summary(model_name: str) -> dict:
    summary = {"model_type": model_name}
    for split in ("train", "val"):
        summary[split] = {k: [] for k in LOG_SUMMARY_KEYS}
    return summary


def _update_log_summary(log_summary: dict, train_logs: dict, val_logs: dict) -> dict:
    for split, logs in (("train", train_logs), ("val", val_logs)):
        # (the reference also overwrites every other top-level key -- i.e. "model_type" -- with NaN here,
        #  logging.py:337-342; that side effect is deliberately not reproduced)
        for key in log_summary[split]:
            log_summary[split][key].append(logs.get(key, np.nan))
    return log_summary


def average_logs(logs_list) -> Dict[str, float]:
    acc: Dict[str, Tuple[float, int]] = {}
    for logs in logs_list:
        for k, v in logs.items():
            s, n = acc.get(k, (0.0, 0))
            acc[k] = (s + float(v), n + 1)
    return {k: s / max(n, 1) for k, (s, n) in acc.items()}


def _clip01(v: float) -> float:
    return float(min(1.0, max(0.0, v)))


@torch.no_grad()
def compute_diagnostics(model: VaDE, dataset: WindowDataset, batch_size: int, n_components: int, tau_star=None,
                        distill_sharpen_T: float = 0.5, distill_conf_weight: bool = False,
                        distill_conf_thresh: float = 0.55, max_batches: int = 4) -> Dict[str, float]:
    """conf_norm / bal_norm / alignment_score on <= max_batches validation batches (logging.py:148-301)."""
    total, sum_ent, sum_max, sum_q = 0, 0.0, 0.0, None
    for bi, s in enumerate(range(0, len(dataset), batch_size)):
        if bi >= max_batches:
            break
        x, a = dataset.fetch(s, min(s + batch_size, len(dataset)))
        q = model.group(x, a).float().clamp_min(1e-8)
        q = q / q.sum(dim=-1, keepdim=True).clamp_min(1e-8)
        sum_ent += float(-(q * q.log()).sum())
        sum_max += float(q.max(dim=-1).values.sum())
        total += q.shape[0]
        sum_q = q.sum(dim=0) if sum_q is None else sum_q + q.sum(dim=0)
    out = {k: float("nan") for k in ("diag/q_mean_entropy", "diag/q_marginal_entropy", "diag/q_mean_max_prob",
                                     "diag/teacher_marginal_entropy", "diag/teacher_conf_mean",
                                     "diag/teacher_weight_mean", "diag/kl_marg_q_to_tau", "conf_norm", "bal_norm",
                                     "alignment_score")}
    if total > 0:
        q_marg = (sum_q / total).clamp_min(1e-9)
        mean_ent = sum_ent / total
        q_marg_ent = float(-(q_marg * q_marg.log()).sum())
        log_k = math.log(float(n_components))
        conf = _clip01(1.0 - mean_ent / max(1e-9, log_k))
        out.update({"diag/q_mean_entropy": mean_ent, "diag/q_marginal_entropy": q_marg_ent,
                    "diag/q_mean_max_prob": sum_max / total})
        if tau_star is not None:
            tau = tau_star.detach().to(q_marg.device, q_marg.dtype)
            tau_marg = tau.mean(dim=0).clamp_min(1e-9)
            kl = max(0.0, float((q_marg * (q_marg.log() - tau_marg.log())).sum()))
            bal = _clip01(1.0 - kl / max(1e-9, log_k))
            sharp = torch.softmax(tau.clamp_min(1e-8).log() / distill_sharpen_T, dim=-1) if distill_sharpen_T > 0 else tau
            cmax = sharp.max(dim=1).values
            out.update({"diag/teacher_marginal_entropy": float(-(tau_marg * tau_marg.log()).sum()),
                        "diag/teacher_conf_mean": float(cmax.mean()), "diag/kl_marg_q_to_tau": kl,
                        "diag/teacher_weight_mean": float(((cmax - distill_conf_thresh) / max(1e-6, 1 - distill_conf_thresh)
                                                           ).clamp(0, 1).mean()) if distill_conf_weight else 1.0})
        else:
            bal = _clip01(q_marg_ent / max(1e-9, log_k))
        out.update({"conf_norm": conf, "bal_norm": bal, "alignment_score": conf * bal})
    return out


# ------------------------------------------------------------------------------------------------
# the VaDE step driver
# ------------------------------------------------------------------------------------------------
NOISE_HOOK = None  # see VadeStepper.noise_fn


class VadeStepper:
    """Owns the loss configuration for one phase and issues train / validation steps on the engine.

    A step = [dof_schedule_apply -> noise -> dof_vade_loss_grads -> log accumulation -> dof_optimizer_step] on static
    buffers, replayed as one hipGraph per (batch size, phase, teacher on/off, train/validation) -- see
    ``deepof_amd.stepping``.  Under data parallelism the RCCL all-reduce of the flat gradient runs between two
    graphs (loss/gradients, optimiser); 1 / world is applied inside the optimiser kernel."""

    def __init__(self, model: VaDE, common_cfg: CommonFitCfg, vade_cfg: VaDECfg, teacher_cfg: TurtleTeacherCfg,
                 use_graphs: Optional[bool] = None):
        self.model, self.common, self.vade, self.teacher = model, common_cfg, vade_cfg, teacher_cfg
        self.pretrain = True
        self.kl_scheduler: Optional[DeviceSchedule] = None
        self.lambda_scheduler: Optional[DeviceSchedule] = None
        self.tau_star: Optional[torch.Tensor] = None
        self.graphs = StepGraphs(model.device, use_graphs)
        self._zero_kl = constant_schedule(0.0, model.device)
        self._buffers: Dict[int, SimpleNamespace] = {}
        self.log_sum = torch.zeros(_capi.LOG_COUNT, dtype=torch.float64, device=model.device)
        self.log_steps = 0
        # Parity tests replay reference runs with recorded noise: noise_fn(kind, index, shape) -> tensor supplies the
        # reparameterisation noise ("eps_train") and the Monte-Carlo KL samples ("mc_train" / "mc_val") instead of the
        # device generator (such steps are launched eagerly)
        self.noise_fn = NOISE_HOOK
        self._noise_count = {"eps_train": 0, "mc_train": 0, "mc_val": 0}
        # the device noise stream of this stepper (dof_step_begin): seeded from torch's generator, so torch.manual_seed
        # fixes it; every data-parallel rank draws its own
        _, rank, _world = _dist_state()
        self._noise_seed = (int(torch.randint(0, 2 ** 62, (1,)).item()) + 0x9E3779B97F4A7C15 * rank) & (2 ** 64 - 1)
        self._rng_state = torch.zeros(2, dtype=torch.int32, device=model.device)
        self.set_mode("pretrain")

    def set_mode(self, mode: str):
        self.pretrain = mode == "pretrain"
        K, v = self.common.n_components, self.vade
        eng = self.model._base
        if self.pretrain:
            eng.set_hyper(km_loss=v.kmeans_loss_pretrain, repel_w=v.repel_weight_pretrain,
                          repel_ls=v.repel_length_scale_pretrain, nonempty_w=v.nonempty_weight_pretrain,
                          nonempty_floor=max(1e-4, v.nonempty_floor_percent_pretrain / K),
                          nonempty_p=int(v.nonempty_p_pretrain))
        else:
            eng.set_hyper(km_loss=self.common.kmeans_loss, repel_w=v.repel_weight, repel_ls=v.repel_length_scale,
                          nonempty_w=v.nonempty_weight, nonempty_floor=max(1e-4, v.nonempty_floor_percent / K),
                          nonempty_p=int(v.nonempty_p))
        # main-phase-only regularisers (kernels ignore them while pretraining)
        eng.set_hyper(tf_w=v.tf_cluster_weight, cat_w=v.reg_cat_clusters, temporal_w=v.temporal_cohesion_weight,
                      scatter_w=v.reg_scatter_weight, scatter_beta=v.reg_scatter_beta)
        eng.set_hyper(km_latent=self.model.kmeans_weight, l1_act=0.1, distill_T=self.teacher.distill_sharpen_T,
                      conf_w=1.0 if self.teacher.distill_conf_weight else 0.0, conf_thr=self.teacher.distill_conf_thresh)
        eng.push_hyper()

    def set_teacher(self, tau_star: Optional[torch.Tensor], lambda_distill: float, lambda_scheduler=None):
        """tau* of every training window + the distillation weight: a schedule (WeightSchedule / DeviceSchedule) or,
        without one, the constant ``lambda_distill``."""
        eng = self.model._base
        self.tau_star = tau_star
        if tau_star is None:
            self.lambda_scheduler = None
            eng.set_teacher(None, None)
            eng.set_hyper(lambda_distill=0.0)
            eng.push_hyper()
            return
        if lambda_scheduler is None:
            lambda_scheduler = constant_schedule(float(lambda_distill), eng.device)
        elif not isinstance(lambda_scheduler, DeviceSchedule):
            lambda_scheduler = DeviceSchedule(lambda_scheduler, eng.device)
        self.lambda_scheduler = lambda_scheduler
        pi = tau_star.mean(dim=0).clamp_min(1e-8)
        w = pi.pow(-float(self.teacher.distill_class_reweight_beta))
        w = w / w.mean()
        if self.teacher.distill_class_reweight_cap is not None:
            w = w.clamp_max(float(self.teacher.distill_class_reweight_cap))
        eng.set_teacher(w, pi)
        eng.push_hyper()

    def _static(self, eng) -> SimpleNamespace:
        b = self._buffers.get(eng.B)
        if b is None:
            f32 = dict(dtype=torch.float32, device=eng.device)
            b = SimpleNamespace(x=torch.empty(eng.B, eng.T, eng.N, 3, **f32), a=torch.empty(eng.B, eng.T, eng.E, 1, **f32),
                                eps=torch.empty(eng.B, eng.L, **f32), eps_zero=torch.zeros(eng.B, eng.L, **f32),
                                eps_mc=torch.empty(eng.S, eng.B, eng.L, **f32), tau=torch.empty(eng.B, eng.K, **f32))
            self._buffers[eng.B] = b
        return b

    def step(self, dataset: WindowDataset, s: int, e: int, train: bool, apply_distill: bool):
        """One step on windows [s, e) of ``dataset``; the loss terms are added to ``self.log_sum``."""
        dist, rank, world = _dist_state()
        eng = self.model.engine(e - s)
        st = self._static(eng)
        dataset.fetch(s, e, out=(st.x, st.a))
        lam_now = self.lambda_scheduler.get_weight() if self.lambda_scheduler is not None else 0.0
        use_tau = bool(apply_distill and self.tau_star is not None and lam_now > 0.0)
        if use_tau:
            st.tau.copy_(self.tau_star[s:e])
        kl = self.kl_scheduler if self.kl_scheduler is not None else self._zero_kl
        items = [(kl, _capi.H_KLW, train and self.kl_scheduler is not None, 1.0)]
        if self.lambda_scheduler is not None:
            items.append((self.lambda_scheduler, _capi.H_LAMBDA_DISTILL, train, 1.0 if use_tau else 0.0))
        pretrain = self.pretrain
        inject = self.noise_fn is not None
        if inject:
            def take(kind, shape):
                k = self._noise_count[kind]
                self._noise_count[kind] = k + 1
                return self.noise_fn(kind, k, shape).to(eng.device, torch.float32)
            if train:
                st.eps.copy_(take("eps_train", (eng.B, eng.L)))
            if not pretrain:
                st.eps_mc.copy_(take("mc_train" if train else "mc_val", (eng.S, eng.B, eng.L)))

        if getattr(eng, "_log_accum", None) is not self.log_sum:
            eng.set_log_accumulator(self.log_sum)  # the loss kernels add every step's terms to log_sum themselves

        def forward_backward():
            # (no lambda schedule: hyper[lambda_distill] stays at the 0 set_teacher pushed)
            eps, eps_mc = (st.eps if train else st.eps_zero), (None if pretrain else st.eps_mc)  # eval: z = mean
            if inject:
                eng.schedule_apply(items)
            else:  # schedules + this step's noise in one launch
                eng.step_begin(items, ([st.eps] if train else []) + ([] if pretrain else [st.eps_mc]),
                               self._noise_seed, self._rng_state)
            eng.loss_grads(st.x, st.a, eps, eps_mc, st.tau if use_tau else None, pretrain=pretrain, count=False)

        # slot = which step variant; key = slot + the schedule objects whose tables / cursors the captured body points
        # at.  tau*, the class weights and the teacher marginal are written in place into static buffers, so a teacher
        # refresh needs no new graph; a replaced schedule does, and StepGraphs drops the slot's previous graph then.
        slot = (eng.B, pretrain, train, use_tau, self.model.training, inject)
        key = slot + (kl.uid, getattr(self.lambda_scheduler, "uid", 0))
        if not train:
            self.graphs.run(key + ("val",), forward_backward, slot + ("val",))
        elif _dp_active(dist, world):
            _dp_step(self.graphs, key, slot, forward_backward, eng, dist, world)
        else:
            def whole():
                forward_backward()
                eng.optimizer_step()
            self.graphs.run(key + ("train",), whole, slot + ("train",))
        eng._count_bn("", 1)
        self.log_steps += 1
        if train:
            if self.kl_scheduler is not None:
                self.kl_scheduler.step()
            if self.lambda_scheduler is not None:
                self.lambda_scheduler.step()

    def begin_logs(self):
        self.log_sum.zero_()
        self.log_steps = 0

    def mean_logs(self) -> Dict[str, float]:
        """One host sync per epoch: the per-step loss terms are summed on the device (float64) until here (Q23)."""
        if self.log_steps == 0:
            return {k: float("nan") for k in _capi.LOG_KEYS if k != "kl_weight"}
        m = (self.log_sum / self.log_steps).cpu().tolist()
        out = {k: m[i] for i, k in enumerate(_capi.LOG_KEYS)}
        out.pop("kl_weight", None)
        return out

    def train_epoch(self, dataset: WindowDataset, seed, shuffle=True):
        _, rank, world = _dist_state()
        self.model.train()
        self.begin_logs()
        mean_klw, mean_lam = 0.0, 0.0
        nb = n_batches(len(dataset), self.common.batch_size, world)
        for step, (s, e) in enumerate(dataset.iter_ranges(self.common.batch_size, shuffle, seed, world, rank)):
            self.step(dataset, s, e, True, True)
            if step == int(nb / 2):  # the schedulers have just been stepped, as in training.py:167-181
                mean_klw = self.kl_scheduler.get_weight() if self.kl_scheduler is not None else 0.0
                mean_lam = self.lambda_scheduler.get_weight() if self.lambda_scheduler is not None else 0.0
        return self.mean_logs(), mean_klw, mean_lam

    @torch.no_grad()
    def validate_epoch(self, dataset: WindowDataset):
        self.model.eval()
        self.begin_logs()
        for s, e in dataset.iter_ranges(self.common.batch_size, False, None, 1, 0):
            self.step(dataset, s, e, False, False)
        return self.mean_logs()


class CheckpointSelector:
    """Which epochs become ``best_model_val`` / ``best_model_score`` (training.py:1725-1731 + 1841-1902 for VaDE,
    :1134-1140 + 1199-1250 for VQ-VAE, :1379-1385 + 1459-1505 for the contrastive model).

    * VaDE validation rule (Q19): the validation loss first RISES while the KL weight warms up, so ``best`` starts at
      -inf and follows the loss upwards until one epoch undercuts it by ``tol`` = 0.01; from then on (tolerance 0) an
      epoch is a new best whenever it improves.  The other two models: plain ``val < best`` from +inf.
    * score rule (all models; needs a score source -- VaDE: always, the others: only with the teacher's head): a finite
      alignment score that beats the best, or ties it within 0.01 at a lower validation loss; saved only for
      epochs > max(3, ceil(0.1 * epochs)).  The bests are updated only when an epoch is actually saved."""

    def __init__(self, epochs: int, rising_start: bool):
        self.rising_start = bool(rising_start)
        self.best_val = -float("inf") if rising_start else float("inf")
        self.val_tol = 0.01 if rising_start else 0.0
        self.val_top_reached = False
        self.best_score, self.best_score_val, self.score_tol = -float("inf"), float("inf"), 0.01
        self.score_start_epoch = max(3, math.ceil(0.1 * epochs))

    def update(self, epoch: int, val_total: float, score_value: float, has_score: bool = True) -> Tuple[bool, bool]:
        """-> (save as best_val, save as best_score) for this epoch."""
        if self.rising_start:
            improved_val = (val_total + self.val_tol) < self.best_val
            if not improved_val and not self.val_top_reached:
                self.best_val = val_total
            if improved_val:
                self.val_top_reached, self.best_val, self.val_tol = True, val_total, 0.0
        else:
            improved_val = val_total < self.best_val
            if improved_val:
                self.best_val = val_total
        improved_score = bool(has_score) and math.isfinite(score_value) and (
            score_value > self.best_score
            or (abs(score_value - self.best_score) <= self.score_tol and val_total < self.best_score_val))
        save_score = improved_score and epoch > self.score_start_epoch
        if save_score:
            self.best_score, self.best_score_val = score_value, val_total
        return improved_val, save_score


def _sync_from_rank0(t: torch.Tensor) -> torch.Tensor:
    """Data parallel: replace a replicated-by-construction tensor by rank 0's copy (no-op on one rank)."""
    dist, _rank, world = _dist_state()
    if world > 1:
        t = t.contiguous()
        dist.broadcast(t, src=0)
    return t


def _set_lrs(eng, base_lr: float, gmm_lr: float):
    for seg in (_capi.SEG_ENCODER, _capi.SEG_DECODER, _capi.SEG_HEADS):
        eng.set_lr(seg, base_lr)
    eng.set_lr(_capi.SEG_GMM, gmm_lr)


@torch.no_grad()
def initialize_gmm_from_data(model: VaDE, dataset: WindowDataset, batch_size: int, seed, n_samples: int = 10000,
                             shuffle: bool = True):
    """sklearn diag GMM (reg_covar 1e-4) on <= 10k latent means -> gmm_means / gmm_log_vars (models_new.py:1907-1943)."""
    from sklearn.mixture import GaussianMixture

    _, rank, world = _dist_state()
    model.eval()
    chunks, got = [], 0
    for x, a, _idx, _vid in dataset.iter_batches(batch_size, shuffle, seed, world, rank):
        _, out = model._run(x, a, None, want_loc=False)
        chunks.append(out["z_mean"].cpu())
        got += x.shape[0]
        if got >= n_samples:
            break
    emb = torch.cat(chunks).numpy()[:n_samples]
    gmm = GaussianMixture(n_components=model.n_components, covariance_type="diag", reg_covar=1e-4).fit(emb)
    model.latent_space.gmm_means.copy_(torch.from_numpy(gmm.means_).float())
    model.latent_space.gmm_log_vars.copy_(torch.from_numpy(np.log(gmm.covariances_)).float())


class TrialPruned(Exception):
    """Raised when a tuning trial asks to stop (optuna.TrialPruned when optuna is importable, so that a study catches it)."""


try:   # the hooks are duck-typed (trial.report / trial.should_prune); optuna itself is optional
    import optuna as _optuna
    TrialPruned = _optuna.TrialPruned   # noqa: F811
except Exception:   # noqa: BLE001
    pass


def _report_trial(trial, score_value: float, epoch: int, running_max: float = None) -> float:
    """The reference's tuning hook at the end of an epoch (training.py:1224-1228, 1420-1424, 1853-1857): report the
    epoch's alignment score, stop when the pruner says so.  Returns the reported value (the fit's ``max_score``).
    ``running_max`` (fit_contrastive, training.py:1384, 1420-1422): the reference reports a running maximum that starts
    at -inf and its score is NaN when the hook runs, so -inf is what Optuna sees (a NaN report would be pruned by
    MedianPruner at epoch 0, unlike the reference)."""
    if running_max is not None:
        score_value = score_value if score_value > running_max else running_max
    if trial is not None:
        trial.report(score_value, step=epoch)
        if trial.should_prune():
            raise TrialPruned(f"Pruned at epoch={epoch}, best_score={score_value:.4f}")
    return score_value


def _fit_vade(train_ds: WindowDataset, val_ds: WindowDataset, adjacency_matrix: np.ndarray, common_cfg: CommonFitCfg,
              teacher_cfg: TurtleTeacherCfg, vade_cfg: VaDECfg, device=None, _engine_factory=None, shuffle: bool = True, trial=None):
    """training.py:1522-1918 (pretrain -> GMM init -> main epochs with best-val / best-score selection)."""
    dist, rank, world = _dist_state()
    is_main = rank == 0
    model = VaDE(train_ds.x_shape, train_ds.a_shape, adjacency_matrix, common_cfg.latent_dim, common_cfg.n_components,
                 encoder_type=common_cfg.encoder_type, use_gnn=True, kmeans_loss=vade_cfg.kmeans_loss_pretrain,
                 interaction_regularization=common_cfg.interaction_regularization, batch_size=common_cfg.batch_size,
                 device=device, _engine_factory=_engine_factory)
    eng = model._base
    if world > 1:  # DDP constructor semantics: rank 0's initial weights everywhere
        dist.broadcast(eng.params, src=0)
    rebuild_spec = {"model_name": "vade", "x_shape": train_ds.x_shape, "a_shape": train_ds.a_shape,
                    "adjacency_matrix": np.asarray(adjacency_matrix).astype("float32"),
                    "latent_dim": common_cfg.latent_dim, "n_components": common_cfg.n_components,
                    "encoder_type": common_cfg.encoder_type, "use_gnn": True, "kmeans_loss": common_cfg.kmeans_loss,
                    "interaction_regularization": common_cfg.interaction_regularization, "lens_enabled": False}
    stepper = VadeStepper(model, common_cfg, vade_cfg, teacher_cfg)
    nb = n_batches(len(train_ds), common_cfg.batch_size, world)

    # ---- pretraining: KL vs N(0, I); GMM learning rate 0
    if is_main:
        print("\n--- Pretraining (reconstruction and setting up the latent space) ---")
    model.set_pretrain_mode(True)
    eng.reset_optimizer()
    _set_lrs(eng, vade_cfg.learning_rate_pretrain, 0.0)
    stepper.kl_scheduler = DeviceSchedule(WeightSchedule(nb, mode=vade_cfg.kl_annealing_mode_pretrain,
                                                         warmup_epochs=vade_cfg.kl_warmup_pretrain,
                                                         max_weight=vade_cfg.kl_max_weight_pretrain,
                                                         cooldown_epochs=vade_cfg.kl_cooldown_pretrain,
                                                         end_weight=vade_cfg.kl_end_weight_pretrain), eng.device)
    eng.push_hyper()
    for _ep in range(vade_cfg.pretrain_epochs):
        pre_logs, _klw, _lam = stepper.train_epoch(train_ds, common_cfg.seed, shuffle)
        if TB_WRITER is not None:   # (training.py:1635-1636)
            TB_WRITER.add_scalar("Pretrain/total_loss", pre_logs["total_loss"], _ep)

    # ---- main phase
    model.set_pretrain_mode(False)
    stepper.set_mode("main")
    model.set_censnet_trainable(True)  # the main-phase optimiser is built after the first forward (Q11)
    stepper.kl_scheduler = DeviceSchedule(WeightSchedule(nb, mode=vade_cfg.kl_annealing_mode,
                                                         warmup_epochs=vade_cfg.kl_warmup,
                                                         max_weight=vade_cfg.kl_max_weight,
                                                         cooldown_epochs=vade_cfg.kl_cooldown,
                                                         end_weight=vade_cfg.kl_end_weight), eng.device)
    eng.reset_optimizer()
    _set_lrs(eng, common_cfg.learning_rate, vade_cfg.gmm_learning_rate)
    eng.push_hyper()
    _, best_path_val, best_path_score, _teacher_path = ckpt_paths("vade", common_cfg)
    log_summary = init_log_summary("vade")
    teacher_init_model = None
    teacher_views = {}
    lambda_scheduler = None
    lib = eng.lib
    if teacher_cfg.use_turtle_teacher:
        # training.py:1664-1716: latent view of the pre-trained encoder + PCA views -> TURTLE teacher -> tau*;
        # GMM initialised from tau*-weighted moments; distillation towards tau* during the main phase
        from . import teacher as TT
        if is_main:
            print("\n--- Extracting latents for teacher ---")
        z_all = TT.extract_latents(model, train_ds, batch_size=2048)
        lambda_scheduler = DeviceSchedule(
            WeightSchedule(nb, mode=vade_cfg.kl_annealing_mode, warmup_epochs=0,
                           at_max_epochs=teacher_cfg.lambda_decay_start, max_weight=teacher_cfg.lambda_distill,
                           cooldown_epochs=teacher_cfg.lambda_cooldown, end_weight=teacher_cfg.lambda_end_weight),
            eng.device)
        teacher_cfg.include_latent_view = True  # VaDE has a free and useful latent view thanks to the pre-training
        _teacher, tau_star, teacher_views = TT.maybe_build_turtle_teacher(
            teacher_cfg=teacher_cfg, common_cfg=common_cfg, train_dataset=train_ds, device=eng.device,
            latent_view=z_all, lib=lib)
        if is_main:
            print("\n--- Initializing GMM from teacher tau* ---")
        tau_star = _sync_from_rank0(tau_star.to(eng.device))  # every rank fits its own teacher: keep rank 0's
        TT.initialize_gmm_from_teacher(model, z_all, tau_star, min_var=0.01)
        stepper.set_teacher(tau_star, teacher_cfg.lambda_distill, lambda_scheduler)
        teacher_init_model = _clone_model(model)
        if common_cfg.save_weights and is_main:
            save_model_info(_teacher_path, stage="teacher_init", epoch=vade_cfg.pretrain_epochs - 1,
                            train_steps=vade_cfg.pretrain_epochs * nb,
                            extra={"note": "after pretrain + teacher + GMM init, before main training"},
                            common_cfg=common_cfg, teacher_cfg=teacher_cfg, vade_cfg=vade_cfg, model=model,
                            log_summary=log_summary, rebuild_spec=rebuild_spec, save_weights=common_cfg.save_weights)
    else:
        if is_main:
            print("\n--- Initializing GMM from embeddings (sklearn) ---")
        initialize_gmm_from_data(model, train_ds, common_cfg.batch_size, common_cfg.seed, shuffle=shuffle)
    if world > 1:
        dist.broadcast(eng.params, src=0)
        dist.broadcast(eng.prior, src=0)

    selector = CheckpointSelector(common_cfg.epochs, rising_start=True)
    max_score = 0.0
    for epoch in range(common_cfg.epochs):
        if epoch == 0 and vade_cfg.freeze_gmm_epochs > 0:
            eng.set_active(_capi.SEG_GMM, False)
        if epoch == vade_cfg.freeze_gmm_epochs:  # reference hard-codes these learning rates here (Q22)
            _set_lrs(eng, 5e-4, 2e-4)
            eng.set_active(_capi.SEG_GMM, True)
        if epoch == 0 and vade_cfg.freeze_decoder_epochs > 0:
            eng.set_active(_capi.SEG_DECODER, False)
        if epoch == vade_cfg.freeze_decoder_epochs:
            eng.set_active(_capi.SEG_DECODER, True)
        if (epoch > 0 and teacher_cfg.use_turtle_teacher and teacher_cfg.teacher_refresh_every
                and teacher_cfg.teacher_refresh_every > 0 and epoch % teacher_cfg.teacher_refresh_every == 0
                and (teacher_cfg.teacher_freeze_at is None or epoch <= teacher_cfg.teacher_freeze_at)):
            # training.py:1770-1802: refit the teacher on the CURRENT latents (PCA views are reused)
            from . import teacher as TT
            if is_main:
                print(f"\n--- Refresh TURTLE teacher at epoch {epoch + 1} ---")
            z_curr = TT.extract_latents(model, train_ds, batch_size=2048)
            views = {"z": z_curr}
            for key, flag in (("pca_pos", teacher_cfg.include_nodes_view), ("pca_spd", teacher_cfg.include_nodes_view),
                              ("pca_edges", teacher_cfg.include_edges_view), ("pca_angles", teacher_cfg.include_angles_view)):
                if flag and teacher_views.get(key) is not None:
                    views[key] = teacher_views[key]
            _teacher, tau_star = TT.run_turtle_teacher_on_views(
                views, common_cfg.n_components, gamma=teacher_cfg.teacher_gamma,
                alpha_sample_entropy=teacher_cfg.teacher_alpha_sample_entropy,
                outer_steps=max(200, teacher_cfg.teacher_outer_steps), inner_steps=teacher_cfg.teacher_inner_steps,
                normalize_feats=teacher_cfg.teacher_normalize_feats, verbose=is_main, device=eng.device,
                head_temp=teacher_cfg.teacher_head_temp, task_temp=teacher_cfg.teacher_task_temp,
                batch_size=teacher_cfg.teacher_batch_size, seed=(common_cfg.seed or 0) + epoch, lib=lib)
            tau_star = _sync_from_rank0(tau_star.to(eng.device))
            stepper.set_teacher(tau_star, teacher_cfg.lambda_distill, lambda_scheduler)
            if teacher_cfg.reinit_gmm_on_refresh:
                TT.initialize_gmm_from_teacher(model, z_curr, tau_star, min_var=1e-4)
                if world > 1:  # replicas must not drift apart: rank 0's re-initialised mixture everywhere
                    dist.broadcast(eng.params, src=0)
                    dist.broadcast(eng.prior, src=0)
        eng.push_hyper()  # learning rates / freeze flags of this epoch
        train_logs, klw, lambda_d = stepper.train_epoch(train_ds, common_cfg.seed, shuffle)
        val_logs = stepper.validate_epoch(val_ds)
        diag = compute_diagnostics(model, val_ds, common_cfg.batch_size, common_cfg.n_components, tau_star=stepper.tau_star,
                                   distill_sharpen_T=teacher_cfg.distill_sharpen_T,
                                   distill_conf_weight=teacher_cfg.distill_conf_weight,
                                   distill_conf_thresh=teacher_cfg.distill_conf_thresh,
                                   max_batches=common_cfg.diag_max_batches)
        val_logs.update(diag)
        val_total = float(val_logs.get("total_loss", float("inf")))
        score_value = float(val_logs["alignment_score"])
        log_summary = _update_log_summary(log_summary, train_logs, val_logs)
        log_epoch_to_tensorboard(TB_WRITER, train_logs, val_logs, epoch, score_value)   # (training.py:1838)
        if is_main:
            print(f"Epoch {epoch + 1}/{common_cfg.epochs} | KLw={klw:.3f} | lambda_distill={lambda_d:.3f} | "
                  f"train total={train_logs['total_loss']:.4f} recon={train_logs['reconstruct_loss']:.4f} | "
                  f"val total={val_total:.4f} | align score={score_value:.3f}")
        improved_val, save_score = selector.update(epoch, val_total, score_value)
        max_score = _report_trial(trial, score_value, epoch)
        common_info = dict(common_cfg=common_cfg, teacher_cfg=teacher_cfg, vade_cfg=vade_cfg, model=model,
                           log_summary=log_summary, rebuild_spec=rebuild_spec, save_weights=common_cfg.save_weights)
        if improved_val:
            if common_cfg.save_weights and is_main:
                save_model_info(best_path_val, stage="best_val", epoch=epoch, train_steps=(epoch + 1) * nb,
                                val_total=val_total, **common_info)
        if save_score:
            if common_cfg.save_weights and is_main:
                save_model_info(best_path_score, stage="best_score", epoch=epoch, train_steps=(epoch + 1) * nb,
                                val_total=val_total, score_value=score_value, **common_info)
    model_val, model_score = load_best_checkpoints(model, best_path_val, best_path_score, common_cfg.save_weights)
    if trial is not None:   # tuning mode: the last epoch's score rides along (training.py:1914-1915)
        return model_val, model_score, teacher_init_model, log_summary, max_score
    return model_val, model_score, teacher_init_model, log_summary


class VQVAEStepper:
    """One step of the VQ-VAE fit loop (step_vqvae_distill, training.py:312-389 of the reference, + clip + Adam) on
    windows [s, e) of a device-resident dataset: fetch into static buffers, then [distillation-weight schedule ->
    dof_vqvae_loss_grads -> log sum -> optimiser] replayed as one hipGraph per (batch size, train / val, teacher)."""

    def __init__(self, model, use_graphs: Optional[bool] = None):
        self.model = model
        self.graphs = StepGraphs(model._base.device, use_graphs)
        self._buffers: Dict[int, SimpleNamespace] = {}

    def _static(self, e) -> SimpleNamespace:
        b = self._buffers.get(e.B)
        if b is None:
            f32 = dict(dtype=torch.float32, device=e.device)
            b = self._buffers[e.B] = SimpleNamespace(x=torch.empty(e.B, e.T, e.N, 3, **f32),
                                                     a=torch.empty(e.B, e.T, e.E, 1, **f32),
                                                     tau=torch.empty(e.B, e.K, **f32))
        return b

    def step(self, ds: WindowDataset, s0: int, e0: int, train: bool, tau_star, lambda_scheduler, log_sum):
        dist, rank, world = _dist_state()
        e = self.model.engine(e0 - s0)
        st = self._static(e)
        ds.fetch(s0, e0, out=(st.x, st.a))
        lam = float(lambda_scheduler.get_weight()) if (train and lambda_scheduler is not None) else 0.0
        use_tau = train and tau_star is not None and lam > 0.0
        if use_tau:
            st.tau.copy_(tau_star[s0:e0])
        items = [] if lambda_scheduler is None else \
            [(lambda_scheduler, _capi.H_LAMBDA_DISTILL, train, 1.0 if use_tau else 0.0)]

        def forward_backward():
            e.schedule_apply(items)
            e.vq_loss_grads(st.x, st.a, st.tau if use_tau else None, count=False)
            log_sum.add_(e.logs)

        key = (e.B, train, use_tau)
        if not train:
            self.graphs.run(key + ("val",), forward_backward)
        elif _dp_active(dist, world):
            _dp_step(self.graphs, key, None, forward_backward, e, dist, world)
        else:
            def whole():
                forward_backward()
                e.optimizer_step()
            self.graphs.run(key + ("train",), whole)
        e.count_vq_step()


def _fit_vqvae(train_ds: WindowDataset, val_ds: WindowDataset, adjacency_matrix: np.ndarray, common_cfg: CommonFitCfg,
               teacher_cfg: TurtleTeacherCfg, device=None, _engine_factory=None, shuffle: bool = True, trial=None):
    """training.py:1036-1263: Adam(lr, weight_decay 1e-4) on encoder + decoder + codebook (+ the distillation head
    when the TURTLE teacher is on), clip 0.75; best-val / best-score (alignment of the head with tau*) checkpoints."""
    dist, rank, world = _dist_state()
    is_main = rank == 0
    model = VQVAE(train_ds.x_shape, train_ds.a_shape, adjacency_matrix, common_cfg.latent_dim, common_cfg.n_components,
                  encoder_type=common_cfg.encoder_type, use_gnn=True, kmeans_loss=common_cfg.kmeans_loss,
                  interaction_regularization=common_cfg.interaction_regularization, batch_size=common_cfg.batch_size,
                  device=device, _engine_factory=_engine_factory)
    eng = model._base
    if world > 1:
        dist.broadcast(eng.params, src=0)
    rebuild_spec = {"model_name": "vqvae", "x_shape": train_ds.x_shape, "a_shape": train_ds.a_shape,
                    "adjacency_matrix": np.asarray(adjacency_matrix).astype("float32"),
                    "latent_dim": common_cfg.latent_dim, "n_components": common_cfg.n_components,
                    "encoder_type": common_cfg.encoder_type, "use_gnn": True,
                    "interaction_regularization": common_cfg.interaction_regularization}
    nb = n_batches(len(train_ds), common_cfg.batch_size, world)
    tau_star, lambda_scheduler = _generic_teacher(train_ds, common_cfg, teacher_cfg, eng, nb, is_main)
    eng.reset_optimizer()
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, common_cfg.learning_rate)
    eng.set_hyper(vq_beta=model.beta, km_latent=common_cfg.kmeans_loss, km_loss=1.0 if common_cfg.kmeans_loss else 0.0,
                  clip=0.75, wd=1e-4)
    eng.push_hyper()
    _, best_path_val, best_path_score, _ = ckpt_paths("vqvae", common_cfg)
    log_summary = init_log_summary("vqvae")
    selector = CheckpointSelector(common_cfg.epochs, rising_start=False)
    max_score = 0.0
    keys = ("total_loss", "enc_rec_loss", "reconstruct_loss", "vq_loss", "kmeans_loss",
            "number_of_populated_clusters", "distill_loss")

    stepper = VQVAEStepper(model)
    log_sum = torch.zeros(_capi.LOG_COUNT, dtype=torch.float64, device=eng.device)

    def run_epoch(ds, train):
        """One pass; every step = [schedule -> dof_vqvae_loss_grads -> log sum -> optimiser] replayed as a hipGraph."""
        model.train(train)
        log_sum.zero_()
        n_steps = 0
        ranges = ds.iter_ranges(common_cfg.batch_size, train and shuffle, common_cfg.seed if train else None,
                                world if train else 1, rank if train else 0)
        for s0, e0 in ranges:
            stepper.step(ds, s0, e0, train, tau_star, lambda_scheduler, log_sum)
            n_steps += 1
            if train and lambda_scheduler is not None:
                lambda_scheduler.step()
        if n_steps == 0:
            return {k: float("nan") for k in keys}
        m = (log_sum / n_steps).cpu().tolist()
        return {"total_loss": m[0], "enc_rec_loss": m[_capi.LOG_ENC_REC], "reconstruct_loss": m[1],
                "vq_loss": m[_capi.LOG_VQ], "kmeans_loss": m[4], "number_of_populated_clusters": m[_capi.LOG_POPULATED],
                "distill_loss": m[7]}

    for epoch in range(common_cfg.epochs):
        train_logs = run_epoch(train_ds, True)
        val_logs = run_epoch(val_ds, False)
        if tau_star is not None:
            zb = []
            for bi, s0 in enumerate(range(0, len(val_ds), common_cfg.batch_size)):
                if bi >= common_cfg.diag_max_batches:
                    break
                xv, av = val_ds.fetch(s0, min(s0 + common_cfg.batch_size, len(val_ds)))
                zb.append(model.encode(xv, av))
            val_logs.update(_head_diagnostics(eng, zb, common_cfg.n_components, tau_star, teacher_cfg))
        else:
            val_logs.update(alignment_score=float("nan"), conf_norm=float("nan"), bal_norm=float("nan"))
        v_total = float(val_logs["total_loss"])
        score_value = float(val_logs["alignment_score"])
        log_summary = _update_log_summary(log_summary, train_logs, val_logs)
        log_epoch_to_tensorboard(TB_WRITER, train_logs, val_logs, epoch, score_value,
                                 float(lambda_scheduler.get_weight()) if lambda_scheduler is not None else 0.0)
        if is_main:
            print(f"Epoch {epoch + 1}/{common_cfg.epochs} | train total={train_logs['total_loss']:.4f} "
                  f"recon={train_logs['reconstruct_loss']:.4f} codes={train_logs['number_of_populated_clusters']:.1f} "
                  f"distill={train_logs['distill_loss']:.4f} | val total={v_total:.4f} | align score={score_value:.3f}")
        improved_val, save_score = selector.update(epoch, v_total, score_value, has_score=tau_star is not None)
        max_score = _report_trial(trial, score_value, epoch)
        if save_score:
            if common_cfg.save_weights and is_main:
                save_model_info(best_path_score, stage="best_score", epoch=epoch, train_steps=(epoch + 1) * nb,
                                val_total=v_total, score_value=score_value, common_cfg=common_cfg, teacher_cfg=teacher_cfg,
                                model=model, log_summary=log_summary, rebuild_spec=rebuild_spec,
                                save_weights=common_cfg.save_weights)
        if improved_val:
            if common_cfg.save_weights and is_main:
                save_model_info(best_path_val, stage="best_val", epoch=epoch, train_steps=(epoch + 1) * nb,
                                val_total=v_total, common_cfg=common_cfg, teacher_cfg=teacher_cfg, model=model,
                                log_summary=log_summary, rebuild_spec=rebuild_spec, save_weights=common_cfg.save_weights)
    model_val, model_score = load_best_checkpoints(model, best_path_val, best_path_score, common_cfg.save_weights)
    if trial is not None:   # (training.py:1260-1261, 1516-1517: four values in tuning mode)
        return model_val, model_score, log_summary, max_score
    return model_val, model_score, None, log_summary


def _generic_teacher(train_ds, common_cfg, teacher_cfg, eng, nb, is_main):
    """Teacher + distillation schedule of fit_VQVAE / fit_contrastive (training.py:1098-1123, 1338-1363): PCA views
    only (no latent view), tf_sigmoid lambda schedule; returns (tau_star on the device or None, lambda schedule)."""
    if not teacher_cfg.use_turtle_teacher:
        return None, None
    from . import teacher as TT
    teacher_cfg.include_latent_view = False
    _t, tau_star, _views = TT.maybe_build_turtle_teacher(teacher_cfg=teacher_cfg, common_cfg=common_cfg,
                                                         train_dataset=train_ds, device=eng.device, latent_view=None,
                                                         lib=eng.lib)
    if tau_star is None:
        return None, None
    sched = DeviceSchedule(
        WeightSchedule(nb, mode="tf_sigmoid", warmup_epochs=0, at_max_epochs=teacher_cfg.lambda_decay_start,
                       max_weight=teacher_cfg.lambda_distill, cooldown_epochs=teacher_cfg.lambda_cooldown,
                       end_weight=teacher_cfg.lambda_end_weight), eng.device)
    eng.set_hyper(distill_T=teacher_cfg.generic_distill_sharpen_T,
                  conf_w=1.0 if teacher_cfg.generic_distill_conf_weight else 0.0,
                  conf_thr=teacher_cfg.generic_distill_conf_thresh)
    return _sync_from_rank0(tau_star.to(eng.device)), sched


def _head_diagnostics(eng, z_batches, n_components, tau_star, teacher_cfg):
    """conf_norm / bal_norm / alignment_score from q = softmax(distill_head(z)) over a few validation batches
    (logging.py:62-146 get_q_vqvae / get_q_contrastive + compute_diagnostics :148-301).  Diagnostics only."""
    W, b = eng.view("distill_head.fc.weight"), eng.view("distill_head.fc.bias")
    qs = []
    for z in z_batches:
        q = torch.softmax(z.float() @ W.T + b, dim=-1).clamp_min(1e-8)
        qs.append(q / q.sum(dim=-1, keepdim=True).clamp_min(1e-8))
    if not qs:
        return {"alignment_score": float("nan"), "conf_norm": float("nan"), "bal_norm": float("nan")}
    q = torch.cat(qs)
    log_k = math.log(float(n_components))
    conf = _clip01(1.0 - float(-(q * q.log()).sum(dim=-1).mean()) / max(1e-9, log_k))
    q_marg = q.mean(dim=0).clamp_min(1e-9)
    tau_marg = tau_star.to(q.device, q.dtype).mean(dim=0).clamp_min(1e-9)
    kl = max(0.0, float((q_marg * (q_marg.log() - tau_marg.log())).sum()))
    bal = _clip01(1.0 - kl / max(1e-9, log_k))
    return {"alignment_score": conf * bal, "conf_norm": conf, "bal_norm": bal}


class ContrastiveStepper:
    """One step of step_contrastive_distill (training.py:482-589) on the HIP path: both views of every window are
    built by dof_contrastive_views, encoded by the two plans of the batch size, scored by the all-pairs loss
    kernels, and back-propagated view by view into the shared gradient buffer."""

    def __init__(self, model: Contrastive, edge_index: np.ndarray, edge_index_local: np.ndarray,
                 cfg: ContrastiveCfg, seed: int = 0):
        from .augment import build_rotation_precomp
        self.model, self.cfg = model, cfg
        self.lib = model._base.lib
        self.edge_index = torch.from_numpy(np.ascontiguousarray(edge_index, dtype=np.int32)).to(model.device)
        self.precomp = build_rotation_precomp(edge_index_local.tolist(), model.input_n_nodes)
        self.gen = torch.Generator(device=model.device)
        self.gen.manual_seed(int(seed))
        self.host_gen = torch.Generator()
        self.host_gen.manual_seed(int(seed))
        self.graphs = StepGraphs(model.device)
        self._buffers: Dict[int, SimpleNamespace] = {}

    def views(self, x_full: torch.Tensor, draws: dict = None, out=None):
        """(x, a, x_aug, a_aug) of the full windows; ``out`` = the four buffers to fill."""
        from .augment import draw_augmentation
        from .engine import contrastive_views
        B, Tf, N, _ = x_full.shape
        if draws is None:
            draws = draw_augmentation(B, Tf, N, self.cfg, self.precomp, x_full.device, self.gen, self.host_gen)
        st = self.model._base._stream()
        x, a = contrastive_views(self.lib, x_full, self.edge_index, None, st, out=None if out is None else out[:2])
        xa, aa = contrastive_views(self.lib, x_full, self.edge_index, draws, st, out=None if out is None else out[2:])
        return x, a, xa, aa

    def loss_grads(self, x_full: torch.Tensor, draws: dict = None, want_grads: bool = True, teacher_tau=None):
        """Fills the shared grads (if want_grads) and e.logs; returns the central-view engine.  teacher_tau (B,K):
        targets of this batch for the distillation head (None = no distillation term)."""
        m = self.model
        x_full = x_full.to(m.device, torch.float32).contiguous()
        x, a, xa, aa = self.views(x_full, draws)
        B = x_full.shape[0]
        e1, e2 = m.engine(B), m.aug_engine(B)
        z = e1.contrastive_encode(x, a, train=want_grads)
        z_aug = e2.contrastive_encode(xa, aa, train=want_grads)
        dz, dza = e1.contrastive_loss(z, z_aug, m.similarity_function, m.loss_function, m.temperature, m.tau, m.beta,
                                      want_grads=want_grads, teacher_tau=teacher_tau)
        if want_grads:
            e1.contrastive_backward(dz, accumulate=False)
            e2.contrastive_backward(dza, accumulate=True)
        return e1

    # ---- the fit loop's step: views eagerly (their rotation choices are host draws baked into the launch), the
    # rest -- both encoder passes, the all-pairs loss, both backward passes, the optimiser -- as one hipGraph
    def _static(self, B: int) -> SimpleNamespace:
        b = self._buffers.get(B)
        if b is None:
            m = self.model
            f32 = dict(dtype=torch.float32, device=m.device)
            Tf, N, E, h = m.full_time_steps, m.input_n_nodes, m._base.E, m.window_size
            b = self._buffers[B] = SimpleNamespace(
                x_full=torch.empty(B, Tf, N, 3, **f32), a_full=torch.empty(B, Tf, E, 1, **f32),
                x=torch.empty(B, h, N, 3, **f32), a=torch.empty(B, h, E, 1, **f32),
                xa=torch.empty(B, h, N, 3, **f32), aa=torch.empty(B, h, E, 1, **f32),
                z=torch.empty(B, m.latent_dim, **f32), z_aug=torch.empty(B, m.latent_dim, **f32),
                dz=torch.empty(B, m.latent_dim, **f32), dza=torch.empty(B, m.latent_dim, **f32),
                tau=torch.empty(B, m.n_components, **f32))
        return b

    def step(self, dataset: WindowDataset, s: int, e: int, train: bool, tau_star, lambda_scheduler, log_sum):
        """One step of the fit loop on windows [s, e): adds the loss terms to ``log_sum``, updates the weights when
        ``train``.  The dataset's own edge tensor is discarded, as in the reference (Q14)."""
        dist, rank, world = _dist_state()
        m = self.model
        B = e - s
        st = self._static(B)
        dataset.fetch(s, e, out=(st.x_full, st.a_full))
        self.views(st.x_full, None, out=(st.x, st.a, st.xa, st.aa))
        e1, e2 = m.engine(B), m.aug_engine(B)
        lam = float(lambda_scheduler.get_weight()) if (train and lambda_scheduler is not None) else 0.0
        use_tau = train and tau_star is not None and lam > 0.0
        if use_tau:
            st.tau.copy_(tau_star[s:e])
        items = [] if lambda_scheduler is None else \
            [(lambda_scheduler, _capi.H_LAMBDA_DISTILL, train, 1.0 if use_tau else 0.0)]

        def forward_backward():
            e1.schedule_apply(items)
            e1.contrastive_encode(st.x, st.a, train=train, out=st.z, count=False)
            e2.contrastive_encode(st.xa, st.aa, train=train, out=st.z_aug, count=False)
            e1.contrastive_loss(st.z, st.z_aug, m.similarity_function, m.loss_function, m.temperature, m.tau, m.beta,
                                want_grads=train, teacher_tau=st.tau if use_tau else None, out=(st.dz, st.dza))
            if train:
                e1.contrastive_backward(st.dz, accumulate=False)
                e2.contrastive_backward(st.dza, accumulate=True)
            log_sum.add_(e1.logs)

        key = (B, train, use_tau)
        if not train:
            self.graphs.run(key + ("val",), forward_backward)
        elif _dp_active(dist, world):
            _dp_step(self.graphs, key, None, forward_backward, e1, dist, world)
        else:
            def whole():
                forward_backward()
                e1.optimizer_step()
            self.graphs.run(key + ("train",), whole)
        if train:
            e1._count_bn("", 2)  # two train-mode encoder passes (BatchNorm step counters of the TCN family)
            if lambda_scheduler is not None:
                lambda_scheduler.step()


def _fit_contrastive(train_ds: WindowDataset, val_ds: WindowDataset, adjacency_matrix: np.ndarray, meta_info: dict,
                     common_cfg: CommonFitCfg, teacher_cfg: TurtleTeacherCfg, contrastive_cfg: ContrastiveCfg,
                     device=None, _engine_factory=None, shuffle: bool = True, trial=None):
    """training.py:1266-1520: Adam(lr, weight_decay 1e-4) on the encoder (+ the distillation head when the TURTLE
    teacher is on), clip 0.75, best-val / best-score checkpointing."""
    from .augment import edge_index_from_meta
    dist, rank, world = _dist_state()
    is_main = rank == 0
    if meta_info is None:
        raise RuntimeError("meta_info (node_columns / edge_columns) is required for the contrastive augmentations")
    model = Contrastive(train_ds.x_shape, train_ds.a_shape, adjacency_matrix, latent_dim=common_cfg.latent_dim,
                        encoder_type=common_cfg.encoder_type, use_gnn=True,
                        similarity_function=contrastive_cfg.contrastive_similarity_function,
                        loss_function=contrastive_cfg.contrastive_loss_function,
                        temperature=contrastive_cfg.temperature, beta=contrastive_cfg.beta, tau=contrastive_cfg.tau,
                        batch_size=common_cfg.batch_size, device=device, _engine_factory=_engine_factory,
                        n_components=common_cfg.n_components)
    eng = model._base
    if world > 1:
        dist.broadcast(eng.params, src=0)
    rebuild_spec = {"model_name": "contrastive", "x_shape": train_ds.x_shape, "a_shape": train_ds.a_shape,
                    "adjacency_matrix": np.asarray(adjacency_matrix).astype("float32"),
                    "latent_dim": common_cfg.latent_dim, "n_components": common_cfg.n_components,
                    "encoder_type": common_cfg.encoder_type, "use_gnn": True,
                    "interaction_regularization": common_cfg.interaction_regularization}
    nb = n_batches(len(train_ds), common_cfg.batch_size, world)
    tau_star, lambda_scheduler = _generic_teacher(train_ds, common_cfg, teacher_cfg, eng, nb, is_main)
    ei_g, ei_l = edge_index_from_meta(meta_info, train_ds.x_shape[1])
    seed = common_cfg.seed if common_cfg.seed is not None else 0
    stepper = ContrastiveStepper(model, ei_g, ei_l, contrastive_cfg, seed=seed + 7919 * rank)
    eng.reset_optimizer()
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, common_cfg.learning_rate)
    eng.set_hyper(clip=0.75, wd=1e-4)
    eng.push_hyper()
    _, best_path_val, best_path_score, _ = ckpt_paths("contrastive", common_cfg)
    log_summary = init_log_summary("contrastive")
    selector = CheckpointSelector(common_cfg.epochs, rising_start=False)
    max_score = -float("inf")   # training.py:1384
    keys = ("total_loss", "pos_similarity", "neg_similarity", "distill_loss", "seperability")

    log_sum = torch.zeros(_capi.LOG_COUNT, dtype=torch.float64, device=eng.device)

    def run_epoch(ds, train):
        model.train(train)
        log_sum.zero_()
        n_steps = 0
        for s0, e0 in ds.iter_ranges(common_cfg.batch_size, train and shuffle, common_cfg.seed if train else None,
                                     world if train else 1, rank if train else 0):
            stepper.step(ds, s0, e0, train, tau_star, lambda_scheduler, log_sum)
            n_steps += 1
        if n_steps == 0:
            return {k: float("nan") for k in keys}
        m = (log_sum / n_steps).cpu().tolist()
        return {"total_loss": m[0], "pos_similarity": m[_capi.LOG_POS_SIM], "neg_similarity": m[_capi.LOG_NEG_SIM],
                "distill_loss": m[7], "seperability": 0.0}

    for epoch in range(common_cfg.epochs):
        train_logs = run_epoch(train_ds, True)
        val_logs = run_epoch(val_ds, False)
        if tau_star is not None:
            zb = []
            model.eval()
            for bi, s0 in enumerate(range(0, len(val_ds), common_cfg.batch_size)):
                if bi >= common_cfg.diag_max_batches:
                    break
                xv, _av = val_ds.fetch(s0, min(s0 + common_cfg.batch_size, len(val_ds)))
                xc, ac, _xa, _aa = stepper.views(xv.to(model.device, torch.float32).contiguous())
                zb.append(torch.nn.functional.normalize(model.embed(xc, ac), dim=1))
            val_logs.update(_head_diagnostics(eng, zb, common_cfg.n_components, tau_star, teacher_cfg))
        else:
            val_logs.update(alignment_score=float("nan"), conf_norm=float("nan"), bal_norm=float("nan"))
        v_total = float(val_logs["total_loss"])
        # Reference quirk (pinned by tests/golden/fit_traces.npz "rules::contrastive"): fit_contrastive never reads the
        # alignment score back (training.py:1447 is commented out), so score_value stays NaN and no best_score
        # checkpoint is ever written for this model; the score is still logged.
        score_value = float("nan")
        log_summary = _update_log_summary(log_summary, train_logs, val_logs)
        log_epoch_to_tensorboard(TB_WRITER, train_logs, val_logs, epoch, score_value,
                                 float(lambda_scheduler.get_weight()) if lambda_scheduler is not None else 0.0)
        if is_main:
            print(f"Epoch {epoch + 1}/{common_cfg.epochs} | train total={train_logs['total_loss']:.4f} "
                  f"pos={train_logs['pos_similarity']:.3f} neg={train_logs['neg_similarity']:.3f} "
                  f"distill={train_logs['distill_loss']:.4f} | val total={v_total:.4f} | "
                  f"align score={float(val_logs['alignment_score']):.3f}")
        improved_val, save_score = selector.update(epoch, v_total, score_value, has_score=tau_star is not None)
        max_score = _report_trial(trial, score_value, epoch, running_max=max_score)
        if save_score:
            if common_cfg.save_weights and is_main:
                save_model_info(best_path_score, stage="best_score", epoch=epoch, train_steps=(epoch + 1) * nb,
                                val_total=v_total, score_value=score_value, common_cfg=common_cfg, teacher_cfg=teacher_cfg,
                                contrastive_cfg=contrastive_cfg, model=model, log_summary=log_summary,
                                rebuild_spec=rebuild_spec, save_weights=common_cfg.save_weights)
        if improved_val:
            if common_cfg.save_weights and is_main:
                save_model_info(best_path_val, stage="best_val", epoch=epoch, train_steps=(epoch + 1) * nb,
                                val_total=v_total, common_cfg=common_cfg, teacher_cfg=teacher_cfg,
                                contrastive_cfg=contrastive_cfg, model=model, log_summary=log_summary,
                                rebuild_spec=rebuild_spec, save_weights=common_cfg.save_weights)
    model_val, model_score = load_best_checkpoints(model, best_path_val, best_path_score, common_cfg.save_weights)
    if trial is not None:   # (training.py:1260-1261, 1516-1517: four values in tuning mode)
        return model_val, model_score, log_summary, max_score
    return model_val, model_score, None, log_summary


# ------------------------------------------------------------------------------------------------
# public API
# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# fit_VADE / fit_VQVAE / fit_contrastive with the reference's signatures (training.py:1522-1532, 1036-1045, 1266-1277; the
# reference's own tests call them directly, tests/test_build_models.py:791, 1041, 1319)
# ------------------------------------------------------------------------------------------------
def _as_window_dataset(source, device) -> WindowDataset:
    """``source``: a WindowDataset (what train_deepof_model hands over: device-resident frame tables), or anything shaped
    like the reference's loaders -- an object with ``.dataset`` (or the dataset itself) whose items are
    ``(x (T,N,3), a (T,E,1), [angles,] idx, vid)`` and which carries ``x_shape`` / ``a_shape`` (dataset.py:16-181 of the
    reference; the in-memory stand-in of its tests).  Such a dataset is materialised once on the device, in index order."""
    if isinstance(source, WindowDataset):
        return source
    return WindowDataset.from_indexed(getattr(source, "dataset", source), device)


class _WriterScope:
    """The fit's event writer for the duration of one fit_* call: the caller's ``writer`` (the reference's SummaryWriter
    argument; flushed and closed at the end as the reference does, training.py:1257-1258) or the one already open."""

    def __init__(self, writer):
        self.writer, self.prev = writer, None

    def __enter__(self):
        global TB_WRITER
        self.prev = TB_WRITER
        if self.writer is not None:
            TB_WRITER = self.writer
        return self

    def __exit__(self, *exc):
        global TB_WRITER
        if self.writer is not None:
            self.writer.flush()
            self.writer.close()
        TB_WRITER = self.prev
        return False


def _fit_device(device, _engine_factory):
    if device is not None and str(device) != "cpu":
        return device
    # the reference's default is CPU; the HIP path has no CPU form, so its default is the ROCm device (an emulator engine
    # factory -- the CPU tests -- keeps "cpu")
    return "cpu" if _engine_factory is not None else None


def fit_VADE(train_loader, val_loader, preprocessed_train: dict, adjacency_matrix: np.ndarray, common_cfg: CommonFitCfg,
             teacher_cfg: TurtleTeacherCfg, vade_cfg: VaDECfg, writer=None, device=None, trial=None, *,
             _engine_factory=None, shuffle: bool = True):
    """training.py:1522-1918, same arguments in the same order.  ``preprocessed_train`` is accepted for signature parity:
    the teacher's views are computed from the training dataset itself (the frame tables it was built from)."""
    dev = _fit_device(device, _engine_factory)
    with _WriterScope(writer):
        return _fit_vade(_as_window_dataset(train_loader, dev or "cuda"), _as_window_dataset(val_loader, dev or "cuda"),
                         np.asarray(adjacency_matrix), common_cfg, teacher_cfg, vade_cfg, device=dev,
                         _engine_factory=_engine_factory, shuffle=shuffle, trial=trial)


def fit_VQVAE(train_loader, val_loader, preprocessed_train: dict, adjacency_matrix: np.ndarray, common_cfg: CommonFitCfg,
              teacher_cfg: TurtleTeacherCfg, writer=None, device=None, trial=None, *, _engine_factory=None, shuffle: bool = True):
    """training.py:1036-1263, same arguments in the same order."""
    dev = _fit_device(device, _engine_factory)
    with _WriterScope(writer):
        return _fit_vqvae(_as_window_dataset(train_loader, dev or "cuda"), _as_window_dataset(val_loader, dev or "cuda"),
                          np.asarray(adjacency_matrix), common_cfg, teacher_cfg, device=dev,
                          _engine_factory=_engine_factory, shuffle=shuffle, trial=trial)


def fit_contrastive(train_loader, val_loader, preprocessed_train: dict, adjacency_matrix: np.ndarray, meta_info: dict,
                    common_cfg: CommonFitCfg, teacher_cfg: TurtleTeacherCfg, contrastive_cfg: ContrastiveCfg, writer=None,
                    device=None, trial=None, *, _engine_factory=None, shuffle: bool = True):
    """training.py:1266-1520, same arguments in the same order."""
    dev = _fit_device(device, _engine_factory)
    with _WriterScope(writer):
        return _fit_contrastive(_as_window_dataset(train_loader, dev or "cuda"), _as_window_dataset(val_loader, dev or "cuda"),
                                np.asarray(adjacency_matrix), meta_info, common_cfg, teacher_cfg, contrastive_cfg, device=dev,
                                _engine_factory=_engine_factory, shuffle=shuffle, trial=trial)


def train_deepof_model_base(preprocessed_object, adjacency_matrix, meta_info, common_cfg: CommonFitCfg,
                            teacher_cfg: TurtleTeacherCfg, vade_cfg: VaDECfg, contrastive_cfg: ContrastiveCfg,
                            h5_dataset_folder: str = None, shuffle: bool = True, device: str = None,
                            bootstrap_training: bool = False, bootstrap_block_len: int = 250, _engine_factory=None):
    if common_cfg.pretrained:
        model, log_summary, _spec, _rep = load_model_from_ckpt(common_cfg.pretrained, _engine_factory=_engine_factory)
        return model, None, None, log_summary
    if device is not None and device not in ("cpu", "gpu"):
        raise ValueError("If a device is given, it needs to be either cpu or gpu!")
    if device == "cpu" and _engine_factory is None:
        raise RuntimeError("deepof_amd has no CPU path: the trainer runs on ROCm GPUs only (device='gpu' or None)")
    is_ddp, rank, world, local_rank = ddp_init_if_needed()
    torch.manual_seed(common_cfg.seed if common_cfg.seed is not None else 0)
    np.random.seed(common_cfg.seed if common_cfg.seed is not None else 0)
    model_name = common_cfg.model_name
    if model_name not in ("vade", "vqvae", "contrastive"):
        raise ValueError(f"Unsupported model: {model_name}")
    dev = None
    if _engine_factory is None:
        dev = torch.device(f"cuda:{local_rank}" if is_ddp else "cuda")
    data_dev = dev if dev is not None else torch.device("cpu")
    preprocessed_train, preprocessed_val = preprocessed_object
    # either the reference's {video: (node windows, edge windows, angles)} dicts, or window datasets over frame
    # tables that deepof_amd.preprocess.preprocess_tables left on the device (nothing is materialised on the host)
    # (the dicts' W-fold redundant windows are folded back into frame tables on the host and windows are gathered on the
    #  device, see WindowDataset.from_preprocessed; the emulator-backed test path keeps the materialised form)
    ingest_lib = None
    if dev is not None:
        from ._lib import load_hip_library
        ingest_lib = load_hip_library()
    train_ds = preprocessed_train if isinstance(preprocessed_train, WindowDataset) else \
        WindowDataset.from_preprocessed(preprocessed_train, data_dev, ingest_lib)
    val_ds = preprocessed_val if isinstance(preprocessed_val, WindowDataset) else \
        WindowDataset.from_preprocessed(preprocessed_val, data_dev, ingest_lib)
    # block bootstrap of the training batches (dataset.py:351-352, 604-614); validation is never bootstrapped
    train_ds.bootstrap_training, train_ds.bootstrap_block_len = bool(bootstrap_training), int(bootstrap_block_len)
    # TensorBoard event files under <output_path>/logs/<model>_run_<n> (training.py:977-982; rank 0, log_history only)
    global TB_WRITER
    TB_WRITER = open_writer(common_cfg, model_name, rank == 0)
    try:
        if model_name == "vqvae":
            return _fit_vqvae(train_ds, val_ds, np.asarray(adjacency_matrix), common_cfg, teacher_cfg, device=dev,
                              _engine_factory=_engine_factory, shuffle=shuffle)
        if model_name == "contrastive":
            return _fit_contrastive(train_ds, val_ds, np.asarray(adjacency_matrix), meta_info, common_cfg, teacher_cfg,
                                    contrastive_cfg, device=dev, _engine_factory=_engine_factory, shuffle=shuffle)
        return _fit_vade(train_ds, val_ds, np.asarray(adjacency_matrix), common_cfg, teacher_cfg, vade_cfg, device=dev,
                         _engine_factory=_engine_factory, shuffle=shuffle)
    finally:
        if TB_WRITER is not None:
            TB_WRITER.close()
            TB_WRITER = None
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        close_native_comm()   # the fit's RCCL communicator behind the C ABI (its step graphs died with the steppers)


def train_deepof_model(
    preprocessed_object: Tuple[dict, dict] = None, adjacency_matrix: np.ndarray = None, meta_info: dict = None,
    encoder_type: str = None, batch_size: int = None, latent_dim: int = None, epochs: int = None,
    output_path: str = None,
    n_clusters: int = 10, learning_rate: float = 1e-3, log_history: bool = True, data_path: str = ".",
    pretrained: Optional[str] = None, save_weights: bool = True, run: int = 0,
    reg_cat_clusters: float = 0.0, recluster: bool = False, freeze_gmm_epochs: int = 0, freeze_decoder_epochs: int = 0,
    prior_loss_weight: float = 0.0, gmm_learning_rate: float = 1e-3, learning_rate_pretrain: float = 1e-3,
    interaction_regularization: float = 0.0003, kmeans_loss: float = 0.0,
    num_workers: int = 0, prefetch_factor: int = 0, use_amp: bool = False,
    use_turtle_teacher: bool = True, teacher_gamma: float = 8.0, teacher_outer_steps: int = 500,
    teacher_inner_steps: int = 100, teacher_normalize_feats: bool = True, lambda_distill: float = 4.0,
    lambda_decay_start: int = 10, lambda_end_weight: float = 0.2, lambda_cooldown: int = 10,
    teacher_refresh_every: Optional[int] = False, teacher_freeze_at: Optional[int] = 10,
    teacher_head_temp: float = 0.5, teacher_task_temp: float = 0.5, teacher_alpha_sample_entropy: float = 2.0,
    teacher_batch_size: int = 2048,
    pretrain_epochs: int = 10, kmeans_loss_pretrain: float = 1.0, repel_weight_pretrain: float = 0.5,
    repel_length_scale_pretrain: float = 0.5, nonempty_weight_pretrain: float = 2e-2,
    nonempty_p_pretrain: float = 2.0, nonempty_floor_percent_pretrain: float = 0.05,
    kl_annealing_mode: str = "tf_sigmoid", kl_max_weight: float = 1, kl_warmup: int = 5, kl_end_weight: float = 0.2,
    kl_cooldown: int = 5, kl_annealing_mode_pretrain: str = "tf_sigmoid", kl_max_weight_pretrain: float = 0.2,
    kl_warmup_pretrain: int = 15, kl_end_weight_pretrain: float = 0.2, kl_cooldown_pretrain: int = 10,
    reg_scatter_weight: float = 0, temporal_cohesion_weight: float = 0, reg_scatter_beta: float = 1.0,
    repel_weight: float = 0, repel_length_scale: float = 1.0,
    main_clustering_loss: float = 0.0, nonempty_weight: float = 2e-2, nonempty_floor_percent: float = 0.05,
    nonempty_p: float = 2.0,
    distill_conf_weight: bool = False, distill_conf_thresh: float = 0.3, distill_sharpen_T: float = 0.5,
    include_edges_view: bool = False, include_nodes_view: bool = True, pca_nodes_dim: int = 32,
    pca_edges_dim: int = 32, include_angles_view: bool = False, pca_angles_dim: int = 32,
    reinit_gmm_on_refresh: bool = False,
    diag_max_batches: int = 4,
    model_name: str = "VaDE",
    generic_lambda_distill: float = 2.0, generic_distill_sharpen_T: float = 0.5,
    generic_distill_conf_weight: bool = True, generic_distill_conf_thresh: float = 0.6,
    generic_distill_warmup_epochs: int = 1, distill_class_reweight_beta: float = 1,
    distill_class_reweight_cap: float = 3,
    temperature: float = 0.1, contrastive_similarity_function: str = "cosine",
    contrastive_loss_function: str = "nce", beta: float = 0.1, tau: float = 0.1,
    aug_min_shift: int = 1, aug_max_shift: int = 3, aug_p_shift: int = 0.4, aug_max_rot: int = 30, aug_n_rot: int = 3,
    aug_p_rot: int = 0.8, aug_max_interp: int = 8, aug_min_interp: int = 3, aug_p_interp: float = 0.4,
    aug_noise_sigma: float = 0.03, aug_p_noise: float = 0.4,
    device: str = None, h5_dataset_folder: Optional[str] = None, bootstrap_training: Optional[bool] = False,
    bootstrap_block_len: int = 250,
    random_seed: int = 0,
    pca_backend: str = "device",
    _engine_factory=None,
):
    """Same signature / defaults / return value as the reference ``train_deepof_model`` (training.py:592-881); the
    two extra keyword arguments come after every reference argument, so positional calls bind as in the reference.
    ``pca_backend``: "device" (teacher PCA views on the GPU) or "sklearn" (the reference's host IncrementalPCA)."""
    model_name = str(model_name).lower()
    encoder_type = str(encoder_type).lower()
    kl_annealing_mode = str(kl_annealing_mode).lower()
    contrastive_similarity_function = str(contrastive_similarity_function).lower()
    contrastive_loss_function = str(contrastive_loss_function).lower()
    check_model_inputs(preprocessed_object, adjacency_matrix, meta_info, encoder_type, batch_size, latent_dim, epochs,
                       output_path, model_name, kl_annealing_mode, contrastive_similarity_function,
                       contrastive_loss_function, pretrained)
    common_cfg = CommonFitCfg(
        model_name=model_name, encoder_type=encoder_type, batch_size=batch_size, latent_dim=latent_dim, epochs=epochs,
        n_components=n_clusters, learning_rate=learning_rate, output_path=output_path, data_path=data_path,
        log_history=log_history, pretrained=pretrained, save_weights=save_weights, run=run, num_workers=num_workers,
        prefetch_factor=prefetch_factor, use_amp=use_amp, interaction_regularization=interaction_regularization,
        kmeans_loss=kmeans_loss, diag_max_batches=diag_max_batches, seed=random_seed)
    teacher_cfg = TurtleTeacherCfg(
        use_turtle_teacher=use_turtle_teacher, teacher_gamma=teacher_gamma, teacher_outer_steps=teacher_outer_steps,
        teacher_inner_steps=teacher_inner_steps, teacher_normalize_feats=teacher_normalize_feats,
        teacher_head_temp=teacher_head_temp, teacher_task_temp=teacher_task_temp,
        teacher_alpha_sample_entropy=teacher_alpha_sample_entropy, lambda_distill=lambda_distill,
        lambda_decay_start=lambda_decay_start, lambda_end_weight=lambda_end_weight, lambda_cooldown=lambda_cooldown,
        distill_sharpen_T=distill_sharpen_T, distill_conf_weight=distill_conf_weight,
        distill_conf_thresh=distill_conf_thresh, generic_lambda_distill=generic_lambda_distill,
        generic_distill_sharpen_T=generic_distill_sharpen_T, generic_distill_conf_weight=generic_distill_conf_weight,
        generic_distill_conf_thresh=generic_distill_conf_thresh,
        generic_distill_warmup_epochs=generic_distill_warmup_epochs,
        distill_class_reweight_beta=distill_class_reweight_beta, distill_class_reweight_cap=distill_class_reweight_cap,
        include_edges_view=include_edges_view, include_nodes_view=include_nodes_view,
        include_angles_view=include_angles_view, pca_nodes_dim=pca_nodes_dim, pca_edges_dim=pca_edges_dim,
        pca_angles_dim=pca_angles_dim, pca_backend=pca_backend,
        teacher_refresh_every=(None if teacher_refresh_every is False else teacher_refresh_every),
        teacher_freeze_at=teacher_freeze_at, reinit_gmm_on_refresh=reinit_gmm_on_refresh,
        teacher_batch_size=teacher_batch_size)
    vade_cfg = VaDECfg(
        reg_cat_clusters=reg_cat_clusters, recluster=recluster, freeze_gmm_epochs=freeze_gmm_epochs,
        freeze_decoder_epochs=freeze_decoder_epochs, gmm_learning_rate=gmm_learning_rate,
        learning_rate_pretrain=learning_rate_pretrain, prior_loss_weight=prior_loss_weight,
        pretrain_epochs=pretrain_epochs, reg_scatter_weight=reg_scatter_weight,
        temporal_cohesion_weight=temporal_cohesion_weight, reg_scatter_beta=reg_scatter_beta,
        repel_weight=repel_weight, repel_length_scale=repel_length_scale, tf_cluster_weight=main_clustering_loss,
        nonempty_weight=nonempty_weight, nonempty_floor_percent=nonempty_floor_percent, nonempty_p=nonempty_p,
        kmeans_loss_pretrain=kmeans_loss_pretrain, repel_weight_pretrain=repel_weight_pretrain,
        repel_length_scale_pretrain=repel_length_scale_pretrain, nonempty_weight_pretrain=nonempty_weight_pretrain,
        nonempty_p_pretrain=nonempty_p_pretrain, nonempty_floor_percent_pretrain=nonempty_floor_percent_pretrain,
        kl_annealing_mode=kl_annealing_mode, kl_max_weight=kl_max_weight, kl_warmup=kl_warmup,
        kl_end_weight=kl_end_weight, kl_cooldown=kl_cooldown, kl_annealing_mode_pretrain=kl_annealing_mode_pretrain,
        kl_max_weight_pretrain=kl_max_weight_pretrain, kl_warmup_pretrain=kl_warmup_pretrain,
        kl_end_weight_pretrain=kl_end_weight_pretrain, kl_cooldown_pretrain=kl_cooldown_pretrain)
    contrastive_cfg = ContrastiveCfg(
        temperature=temperature, contrastive_similarity_function=contrastive_similarity_function,
        contrastive_loss_function=contrastive_loss_function, beta=beta, tau=tau, aug_min_shift=aug_min_shift,
        aug_max_shift=aug_max_shift, aug_p_shift=aug_p_shift, aug_noise_sigma=aug_noise_sigma, aug_p_noise=aug_p_noise,
        aug_min_interp=aug_min_interp, aug_max_interp=aug_max_interp, aug_p_interp=aug_p_interp,
        aug_max_rot=aug_max_rot, aug_n_rot=aug_n_rot, aug_p_rot=aug_p_rot)
    return train_deepof_model_base(preprocessed_object, adjacency_matrix, meta_info, common_cfg=common_cfg,
                                   teacher_cfg=teacher_cfg, vade_cfg=vade_cfg, contrastive_cfg=contrastive_cfg,
                                   h5_dataset_folder=h5_dataset_folder, device=device,
                                   bootstrap_training=bootstrap_training, bootstrap_block_len=bootstrap_block_len,
                                   _engine_factory=_engine_factory)
