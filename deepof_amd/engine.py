"""Host-side driver of the libdeepof_hip VaDE step (device memory + streams are PyTorch plumbing).

``VadeEngine`` owns the flat parameter / gradient / Adam buffers and the workspace as torch
tensors, exposes them under the reference's ``state_dict`` names, and issues the C-ABI calls on
the current HIP stream.  Product code obtains it through :func:`create_vade_engine`, which
requires a ROCm device and the compiled ``libdeepof_hip.so`` and raises otherwise -- there is no
CPU or PyTorch fallback.  (``tests/`` may construct ``VadeEngine`` directly with the pytest-only
emulator build of the same kernels to check kernel logic in the GPU-less build container.)
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _capi
from .graph import censnet_operators

BUFFER_NAMES = ("encoder.laplacian", "encoder.edge_laplacian", "encoder.incidence",
                "latent_space.prior", "latent_space.pretrain")


def _param_shape(name: str, numel: int, N: int, E: int, L: int, K: int):
    leaf = name.split(".")[-1]
    if name.endswith("conv1d.weight"):
        if name.startswith("decoder"):
            return (2 * L, 4 * L, 5)
        return (2 * L, 3 if "node_recurrent" in name else 1, 5)
    if leaf.startswith("weight_ih") or leaf.startswith("weight_hh"):
        if name.startswith("decoder.gru1"):
            return (3 * L, L)
        if name.startswith("decoder.gru2") or ".gru1." in name:
            return (6 * L, 2 * L)
        return (3 * L, 4 * L) if leaf.startswith("weight_ih") else (3 * L, L)  # encoder gru2
    if name.endswith("projection.weight") and "loc_projection" not in name:
        return (2 * L, 2 * L)
    if name.endswith("node_kernel") or name.endswith("edge_kernel"):
        return (2 * L, L)
    if name.endswith("node_weights") or name.endswith("edge_weights"):
        return (2 * L, 1)
    if name == "encoder.final_dense.weight":
        return (L, (N + E) * L)
    if name.endswith("loc_projection.weight"):
        return (3 * N, 2 * L)
    if name in ("latent_space.gmm_means", "latent_space.gmm_log_vars"):
        return (K, L)
    if name == "vq_layer.codebook":
        return (L, K)
    if name.startswith("latent_space.") and leaf == "weight":
        return (L, L)
    return (numel,)


class VadeEngine:
    def __init__(self, lib, device, batch: int, window: int, adjacency: np.ndarray, latent_dim: int,
                 n_clusters: int, mc_samples: int = 32, graph_ops=None, shared: "VadeEngine" = None,
                 kind: str = "vade"):
        """``shared``: another engine (different batch size) whose parameter / gradient / Adam / hyper
        buffers this one borrows -- every batch size needs its own plan and workspace, not its own weights."""
        self.lib = lib
        self.device = torch.device(device)
        adjacency = np.asarray(adjacency, dtype=np.float32)
        lap, elap, inc = graph_ops if graph_ops is not None else censnet_operators(adjacency)
        self.adjacency = adjacency
        self.lap, self.elap, self.inc = (np.ascontiguousarray(m, dtype=np.float32) for m in (lap, elap, inc))
        self.N, self.E = self.inc.shape
        self.B, self.T, self.L, self.K, self.S = int(batch), int(window), int(latent_dim), int(n_clusters), int(mc_samples)
        dims = _capi.VadeDims(self.B, self.T, self.N, self.E, self.L, self.K, self.S)
        plan = C.c_void_p()
        assert kind in ("vade", "vqvae", "contrastive", "contrastive_tcn", "vade_tcn", "vqvae_tcn", "vade_tfm",
                        "vqvae_tfm", "contrastive_tfm")
        self.kind = kind
        create = {"vade": lib.dof_vade_plan_create, "vqvae": lib.dof_vqvae_plan_create,
                  "contrastive": lib.dof_contrastive_plan_create,
                  "contrastive_tcn": lib.dof_contrastive_tcn_plan_create,
                  "vade_tcn": lib.dof_vade_tcn_plan_create, "vqvae_tcn": lib.dof_vqvae_tcn_plan_create,
                  "vade_tfm": lib.dof_vade_tfm_plan_create, "vqvae_tfm": lib.dof_vqvae_tfm_plan_create,
                  "contrastive_tfm": lib.dof_contrastive_tfm_plan_create}[kind]
        _capi.check(lib, create(C.byref(dims), self.lap.ctypes.data, self.elap.ctypes.data, self.inc.ctypes.data,
                                C.byref(plan)), "dof_*_plan_create")
        self.plan = plan
        self.names = []
        self.layout: Dict[str, tuple] = {}
        for i in range(lib.dof_vade_param_count(plan)):
            name = lib.dof_vade_param_name(plan, i).decode()
            off, numel = lib.dof_vade_param_offset(plan, i), lib.dof_vade_param_numel(plan, i)
            self.names.append(name)
            dims4 = (C.c_int64 * 4)()
            rank = lib.dof_vade_param_shape(plan, i, dims4)
            shape = tuple(int(dims4[k]) for k in range(rank)) if rank else \
                _param_shape(name, numel, self.N, self.E, self.L, self.K)
            self.layout[name] = (off, numel, shape)
        self.index = {n: i for i, n in enumerate(self.names)}
        # BatchNorm bookkeeping of the TCN family: running buffers are entries of the flat buffer, the step
        # counters (num_batches_tracked, int64 in the reference state_dict) are host integers
        self.bn_layers = [n[: -len(".running_mean")] for n in self.names if n.endswith(".running_mean")]
        self.num_batches_tracked = {n: torch.zeros((), dtype=torch.int64) for n in self.bn_layers}
        self.bn_training = True
        total = lib.dof_vade_param_total(plan)
        f32 = dict(dtype=torch.float32, device=self.device)
        ws_bytes = lib.dof_vade_workspace_bytes(plan)
        self.workspace = torch.empty(ws_bytes // 4, **f32)
        _capi.check(lib, lib.dof_vade_bind(plan, self.workspace.data_ptr(), self._stream()), "dof_vade_bind")
        self._sync()
        if shared is not None:
            assert shared.params.numel() == total and shared.K == self.K and shared.L == self.L
            for attr in ("params", "grads", "adam_m", "adam_v", "prior", "hyper_host", "hyper", "logs", "teacher",
                         "opt_state", "num_batches_tracked"):
                setattr(self, attr, getattr(shared, attr))
            return
        self.params = torch.zeros(total, **f32)
        self.grads = torch.zeros(total, **f32)
        self.adam_m = torch.zeros(total, **f32)
        self.adam_v = torch.zeros(total, **f32)
        self.prior = torch.full((self.K,), 1.0 / self.K, **f32)
        self.hyper_host = torch.zeros(_capi.H_COUNT, dtype=torch.float32)  # host mirror; push_hyper() snapshots it
        self.hyper = torch.zeros(_capi.H_COUNT, **f32)
        self.logs = torch.zeros(_capi.LOG_COUNT, **f32)
        self.teacher = torch.zeros(2 * self.K, **f32)
        # Adam step count per optimiser segment, on the device: dof_optimizer_step advances it and derives the bias
        # corrections from it, so a captured step needs no host-written per-step value
        self.opt_state = torch.zeros(_capi.SEG_COUNT, dtype=torch.int32, device=self.device)
        self.set_hyper(logvar_lo=-8.0, logvar_hi=8.0, clip=0.75, wd=0.0, l1_act=0.1, distill_T=0.5)
        if "distill_head.fc.weight" in self.layout:  # DiscriminativeHead = nn.Linear(L, K) default init
            bound = 1.0 / math.sqrt(self.L)
            g = torch.Generator().manual_seed(0)  # own stream: the model's initialisation draws stay what they were
            for nm in ("distill_head.fc.weight", "distill_head.fc.bias"):
                v = self.view(nm)
                v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * bound)
        for s in range(_capi.SEG_COUNT):
            self.hyper_host[_capi.H_ACTIVE0 + s] = 1.0

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return 0

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def __del__(self):
        try:
            if getattr(self, "plan", None):
                self.lib.dof_vade_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, numel, shape = self.layout[name]
        return (self.params if buf is None else buf)[off:off + numel].view(shape)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Reference-compatible state_dict (parameter order + buffers; models_new.py VaDEPT)."""
        sd = {"encoder.laplacian": torch.from_numpy(self.lap.copy()),
              "encoder.edge_laplacian": torch.from_numpy(self.elap.copy()),
              "encoder.incidence": torch.from_numpy(self.inc.copy())}
        for n in self.names:
            if n.startswith("distill_head."):  # a separate module in the reference, not part of the model
                continue
            if n == "latent_space.encoder_mean.weight":
                sd["latent_space.prior"] = self.prior.detach().cpu().clone()
                sd["latent_space.pretrain"] = torch.tensor(0.0)
            sd[n] = self.view(n).detach().cpu().clone()
            if n.endswith(".running_var"):
                layer = n[: -len(".running_var")]
                sd[layer + ".num_batches_tracked"] = self.num_batches_tracked[layer].clone()
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        for n in self.names:
            if n in sd:
                self.view(n).copy_(torch.as_tensor(sd[n], dtype=torch.float32).reshape(self.layout[n][2]))
            elif strict and not n.startswith("distill_head."):
                raise KeyError(f"missing parameter {n}")
        if "latent_space.prior" in sd:
            self.prior.copy_(torch.as_tensor(sd["latent_space.prior"], dtype=torch.float32))
        for layer in self.bn_layers:
            if layer + ".num_batches_tracked" in sd:
                self.num_batches_tracked[layer].fill_(int(sd[layer + ".num_batches_tracked"]))

    def set_bn_training(self, training: bool):
        """module.train() / module.eval() for the plan's BatchNorm layers (no-op for the recurrent family)."""
        self.bn_training = bool(training)
        _capi.check(self.lib, self.lib.dof_vade_set_batchnorm_training(self.plan, 1 if training else 0),
                    "dof_vade_set_batchnorm_training")

    # ------------------------------------------------------------------ dropout (transformer family)
    def dropout_sites(self):
        """[(site name, byte offset, numel, p)] of a transformer plan in the order the reference's forward draws its
        dropout masks (embedding, then per layer: attention weights, the two residual dropouts; the decoder's four per
        layer); empty for the other families."""
        lib, plan = self.lib, self.plan
        return [(lib.dof_tfm_dropout_site_name(plan, i).decode(), lib.dof_tfm_dropout_site_offset(plan, i),
                 lib.dof_tfm_dropout_site_numel(plan, i), lib.dof_tfm_dropout_site_p(plan, i))
                for i in range(lib.dof_tfm_dropout_site_count(plan))]

    def set_dropout(self, masks: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0x2545F491):
        """Dropout source of a transformer plan.  masks=None: keep-masks come from the on-device counter hash seeded
        with ``seed`` (a fresh mask set per train-mode step).  masks={site name: 0/1 tensor in the reference's tensor
        shape}: those masks are used instead (parity tests replay the reference's recorded draws)."""
        inject = None
        if masks is not None:
            sites = self.dropout_sites()
            total = sum(n for _, _, n, _ in sites)
            buf = torch.ones(total, dtype=torch.uint8)
            for name, off, numel, _p in sites:
                if name in masks:
                    m = torch.as_tensor(masks[name]).reshape(-1)
                    assert m.numel() == numel, (name, m.numel(), numel)
                    buf[off:off + numel] = m.to(torch.uint8)
            inject = buf.to(self.device)
        self._drop_inject = inject  # keeps the device buffer alive while the plan points at it
        _capi.check(self.lib, self.lib.dof_tfm_set_dropout(self.plan, None if inject is None else inject.data_ptr(),
                                                           int(seed) & 0xFFFFFFFF), "dof_tfm_set_dropout")

    def set_dropout_counter(self, counter: Optional[torch.Tensor]):
        """Point a transformer plan at a caller-owned device step counter (int32 tensor of one element) shared by all
        plans of a model; None returns to the plan's own workspace slot."""
        if counter is not None:
            assert counter.dtype == torch.int32 and counter.numel() == 1 and counter.device.type == self.device.type
        self._drop_counter = counter
        _capi.check(self.lib, self.lib.dof_tfm_set_dropout_counter(self.plan, None if counter is None else counter.data_ptr()),
                    "dof_tfm_set_dropout_counter")

    def _count_bn(self, prefix: str, n: int):
        """num_batches_tracked of the BatchNorm layers under ``prefix`` after n train-mode passes."""
        if not self.bn_training:
            return
        for layer, t in self.num_batches_tracked.items():
            if layer.startswith(prefix):
                t += n

    def set_trainable(self, name: str, trainable: bool):
        """Take a parameter out of (or back into) the optimiser step (reference quirk Q11)."""
        rc = self.lib.dof_vade_set_trainable(self.plan, self.index[name], 1 if trainable else 0, self._stream())
        _capi.check(self.lib, rc, "dof_vade_set_trainable")

    _HYPER_IDX = dict(klw=_capi.H_KLW, lambda_distill=_capi.H_LAMBDA_DISTILL, km_latent=_capi.H_KM_LATENT,
                      km_loss=_capi.H_KM_LOSS, repel_w=_capi.H_REPEL_W, repel_ls=_capi.H_REPEL_LS,
                      nonempty_w=_capi.H_NONEMPTY_W, nonempty_floor=_capi.H_NONEMPTY_FLOOR,
                      nonempty_p=_capi.H_NONEMPTY_P, l1_act=_capi.H_L1_ACT, distill_T=_capi.H_DISTILL_T,
                      conf_w=_capi.H_CONF_W, conf_thr=_capi.H_CONF_THR, has_teacher=_capi.H_HAS_TEACHER,
                      logvar_lo=_capi.H_LOGVAR_LO, logvar_hi=_capi.H_LOGVAR_HI, clip=_capi.H_CLIP, wd=_capi.H_WD,
                      vq_beta=_capi.H_VQ_BETA, tf_w=_capi.H_TF_W, cat_w=_capi.H_CAT_W, temporal_w=_capi.H_TEMPORAL_W,
                      scatter_w=_capi.H_SCATTER_W, scatter_beta=_capi.H_SCATTER_BETA)

    def set_hyper(self, **kw):
        for k, v in kw.items():
            self.hyper_host[self._HYPER_IDX[k]] = float(v)

    def set_lr(self, seg: int, lr: float):
        self.hyper_host[_capi.H_LR0 + seg] = float(lr)

    def set_active(self, seg: int, active: bool):
        self.hyper_host[_capi.H_ACTIVE0 + seg] = 1.0 if active else 0.0

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.zero_()
        self.opt_state.zero_()

    def push_hyper(self):
        """Enqueue hyper_host -> hyper.  The copy reads an immutable pinned snapshot taken NOW (the caching host
        allocator keeps it alive until the copy has run), so later set_hyper() calls cannot leak into steps that
        are already enqueued.  Per-step values (KL weight, distillation lambda, Adam bias corrections) do not go
        through here at all in the fit loops: schedule_apply() / dof_optimizer_step produce them on the device."""
        if self.device.type == "cuda":
            self.hyper.copy_(self.hyper_host.pin_memory(), non_blocking=True)
        else:
            self.hyper.copy_(self.hyper_host)

    def schedule_apply(self, items):
        """dof_schedule_apply: items = [(DeviceSchedule, hyper index, advance, scale), ...] evaluated on the device at
        this point of the stream (graph-capturable)."""
        arr = (_capi.SchedItem * max(1, len(items)))()
        for i, (sched, index, advance, scale) in enumerate(items):
            arr[i] = _capi.SchedItem(sched.table.data_ptr(), sched.cursor.data_ptr(), int(sched.table.numel()), int(index),
                                     1 if advance else 0, float(scale))
        _capi.check(self.lib, self.lib.dof_schedule_apply(self.hyper.data_ptr(), arr, len(items), self._stream()),
                    "dof_schedule_apply")

    def step_begin(self, items, noise, seed: int, rng_state: torch.Tensor):
        """dof_step_begin: the schedule items of ``schedule_apply`` plus N(0,1) fills of the tensors in ``noise`` (at
        most two, contiguous fp32) in one launch; ``rng_state`` = device int32[2] call counter of this noise stream."""
        arr = (_capi.SchedItem * max(1, len(items)))()
        for i, (sched, index, advance, scale) in enumerate(items):
            arr[i] = _capi.SchedItem(sched.table.data_ptr(), sched.cursor.data_ptr(), int(sched.table.numel()), int(index),
                                     1 if advance else 0, float(scale))
        bufs = (_capi.NoiseBuf * max(1, len(noise)))()
        for i, t in enumerate(noise):
            assert t.is_contiguous() and t.dtype == torch.float32 and t.device == self.params.device
            bufs[i] = _capi.NoiseBuf(t.data_ptr(), int(t.numel()))
        assert rng_state.dtype == torch.int32 and rng_state.numel() == 2
        _capi.check(self.lib, self.lib.dof_step_begin(self.hyper.data_ptr(), arr, len(items), int(seed) & (2 ** 64 - 1),
                                                      rng_state.data_ptr(), bufs, len(noise), self._stream()),
                    "dof_step_begin")

    def set_log_accumulator(self, accum: Optional[torch.Tensor]):
        """dof_vade_set_log_accumulator: every loss_grads() also adds its logs to ``accum`` (device float64[LOG_COUNT])."""
        if accum is not None:
            assert accum.dtype == torch.float64 and accum.numel() == _capi.LOG_COUNT and accum.device == self.params.device
        self._log_accum = accum  # keeps the tensor alive while the plan points at it
        _capi.check(self.lib, self.lib.dof_vade_set_log_accumulator(self.plan, None if accum is None else accum.data_ptr()),
                    "dof_vade_set_log_accumulator")

    def configure_vade_phase(self, pretrain: bool, klw: float, tau: Optional[torch.Tensor] = None,
                             lambda_distill: float = 0.0, extra: Optional[dict] = None):
        """The reference's default VadeLoss configuration of one phase (training.py:640-668 signature defaults,
        losses.py:426-443 set_mode) with a fixed KL weight; ``tau`` (n, K) switches the distillation term on with the
        inverse-marginal class weights of losses.py:460-491 (beta 1, cap 3).  Pushes the values to the device."""
        K = self.K
        self.set_hyper(klw=klw, km_latent=1.0, km_loss=1.0 if pretrain else 0.0,
                       repel_w=0.5 if pretrain else 0.0, repel_ls=0.5 if pretrain else 1.0,
                       nonempty_w=0.02, nonempty_floor=max(1e-4, 0.05 / K), nonempty_p=2.0,
                       l1_act=0.1, distill_T=0.5, conf_w=0.0, conf_thr=0.3, lambda_distill=lambda_distill,
                       tf_w=0.0, cat_w=0.0, temporal_w=0.0, scatter_w=0.0, scatter_beta=1.0)
        if extra:
            self.set_hyper(**extra)
        if tau is not None:
            pi = tau.mean(dim=0).clamp_min(1e-8)
            w = pi.pow(-1.0)
            w = (w / w.mean()).clamp_max(3.0)
            self.set_teacher(w, pi)
        else:
            self.set_teacher(None, None)
        self.push_hyper()

    def set_teacher(self, class_weight: Optional[torch.Tensor], marginal: Optional[torch.Tensor]):
        if class_weight is None:
            self.set_hyper(has_teacher=0.0)
            return
        self.teacher[: self.K].copy_(class_weight.to(torch.float32))
        self.teacher[self.K:].copy_(marginal.to(torch.float32))
        self.set_hyper(has_teacher=1.0)

    # ------------------------------------------------------------------ compute
    def _chk_batch(self, x, a):
        assert tuple(x.shape) == (self.B, self.T, self.N, 3), (tuple(x.shape), (self.B, self.T, self.N, 3))
        assert tuple(a.shape) == (self.B, self.T, self.E, 1), tuple(a.shape)
        assert x.is_contiguous() and a.is_contiguous() and x.dtype == torch.float32 and a.dtype == torch.float32
        assert x.device == self.params.device and a.device == self.params.device

    def forward(self, x: torch.Tensor, a: torch.Tensor, eps: Optional[torch.Tensor] = None,
                want_loc: bool = True, want_enc: bool = False) -> Dict[str, torch.Tensor]:
        """VaDEPT.forward.  eps=None -> eval mode (z = z_mean)."""
        self._chk_batch(x, a)
        f32 = dict(dtype=torch.float32, device=self.device)
        out = {"z": torch.empty(self.B, self.L, **f32), "q": torch.empty(self.B, self.K, **f32),
               "z_mean": torch.empty(self.B, self.L, **f32), "z_log_var": torch.empty(self.B, self.L, **f32)}
        if want_loc:
            out["loc"] = torch.empty(self.B, self.T, 3 * self.N, **f32)
        if want_enc:
            out["enc"] = torch.empty(self.B, self.L, **f32)
        ptr = lambda k: out[k].data_ptr() if k in out else None
        rc = self.lib.dof_vade_forward(self.plan, self.params.data_ptr(), self.prior.data_ptr(), x.data_ptr(),
                                       a.data_ptr(), None if eps is None else eps.data_ptr(), ptr("z"), ptr("q"),
                                       ptr("z_mean"), ptr("z_log_var"), ptr("loc"), ptr("enc"), self._stream())
        _capi.check(self.lib, rc, "dof_vade_forward")
        if eps is not None:
            self._count_bn("encoder.", 1)
            if want_loc:
                self._count_bn("decoder.", 1)
        return out

    def loss_grads(self, x, a, eps, eps_mc=None, tau=None, pretrain: bool = True, count: bool = True):
        """Forward + VadeLoss + backward; fills self.grads and self.logs (device).  count=False leaves the host-side
        BatchNorm step counters alone (a graph replay does not run this Python, its caller counts instead)."""
        self._chk_batch(x, a)
        assert tuple(eps.shape) == (self.B, self.L) and eps.is_contiguous()
        if not pretrain:
            assert eps_mc is not None and tuple(eps_mc.shape) == (self.S, self.B, self.L) and eps_mc.is_contiguous()
        if tau is not None:
            assert tuple(tau.shape) == (self.B, self.K) and tau.is_contiguous()
        rc = self.lib.dof_vade_loss_grads(
            self.plan, self.params.data_ptr(), self.prior.data_ptr(), x.data_ptr(), a.data_ptr(), eps.data_ptr(),
            None if eps_mc is None else eps_mc.data_ptr(), None if tau is None else tau.data_ptr(),
            self.teacher.data_ptr() if self.hyper_host[_capi.H_HAS_TEACHER] != 0 else None,
            self.hyper.data_ptr(), 1 if pretrain else 0, self.grads.data_ptr(), self.logs.data_ptr(), self._stream())
        _capi.check(self.lib, rc, "dof_vade_loss_grads")
        if count:
            self._count_bn("", 1)

    # ------------------------------------------------------------------ VQ-VAE
    def vq_forward(self, x, a, want_loc: bool = True, want_soft: bool = True) -> Dict[str, torch.Tensor]:
        """VQVAEPT.forward(return_all_outputs=True): encoder output, quantised latents, soft counts, code indices and
        the reconstruction means from the quantised / raw latents."""
        self._chk_batch(x, a)
        f32 = dict(dtype=torch.float32, device=self.device)
        out = {"ze": torch.empty(self.B, self.L, **f32), "quantized": torch.empty(self.B, self.L, **f32),
               "idx": torch.empty(self.B, dtype=torch.int32, device=self.device)}
        if want_soft:
            out["soft_counts"] = torch.empty(self.B, self.K, **f32)
        if want_loc:
            out["loc_q"] = torch.empty(self.B, self.T, 3 * self.N, **f32)
            out["loc_e"] = torch.empty(self.B, self.T, 3 * self.N, **f32)
        ptr = lambda k: out[k].data_ptr() if k in out else None
        rc = self.lib.dof_vqvae_forward(self.plan, self.params.data_ptr(), x.data_ptr(), a.data_ptr(), ptr("ze"),
                                        ptr("quantized"), ptr("soft_counts"), ptr("idx"), ptr("loc_q"), ptr("loc_e"),
                                        self._stream())
        _capi.check(self.lib, rc, "dof_vqvae_forward")
        return out

    def vq_loss_grads(self, x, a, tau: Optional[torch.Tensor] = None, count: bool = True):
        """step_vqvae_distill + backward; fills self.grads / self.logs.  tau (B,K): teacher targets of the batch -> the
        distillation head term with hyper lambda_distill / distill_T / conf_w / conf_thr."""
        self._chk_batch(x, a)
        if tau is not None:
            assert tuple(tau.shape) == (self.B, self.K) and tau.is_contiguous()
        rc = self.lib.dof_vqvae_loss_grads(self.plan, self.params.data_ptr(), x.data_ptr(), a.data_ptr(),
                                           None if tau is None else tau.data_ptr(), self.hyper.data_ptr(),
                                           self.grads.data_ptr(), self.logs.data_ptr(), self._stream())
        _capi.check(self.lib, rc, "dof_vqvae_loss_grads")
        if count:
            self.count_vq_step()

    def count_vq_step(self):
        self._count_bn("encoder.", 1)
        self._count_bn("decoder.", 2)   # the decoder runs on the quantised and on the raw latents

    def read_vq_logs(self) -> Dict[str, float]:
        v = self.logs.detach().cpu().tolist()
        return {"total_loss": v[0], "enc_rec_loss": v[_capi.LOG_ENC_REC], "reconstruct_loss": v[1],
                "vq_loss": v[_capi.LOG_VQ], "kmeans_loss": v[4], "number_of_populated_clusters": v[_capi.LOG_POPULATED],
                "distill_loss": v[7]}

    # ------------------------------------------------------------------ contrastive
    def contrastive_encode(self, x, a, train: bool = False, out: Optional[torch.Tensor] = None,
                           count: bool = True) -> torch.Tensor:
        """ContrastivePT.forward on one view (half windows): (B, L) embeddings (into ``out`` when given)."""
        self._chk_batch(x, a)
        z = out if out is not None else torch.empty(self.B, self.L, dtype=torch.float32, device=self.device)
        rc = self.lib.dof_contrastive_encode(self.plan, self.params.data_ptr(), x.data_ptr(), a.data_ptr(),
                                             1 if train else 0, z.data_ptr(), self._stream())
        _capi.check(self.lib, rc, "dof_contrastive_encode")
        if train and count:
            self._count_bn("", 1)
        return z

    def contrastive_loss(self, z, z_aug, similarity="cosine", loss_fn="nce", temperature=0.1, tau=0.1, beta=0.1,
                         want_grads: bool = True, teacher_tau: Optional[torch.Tensor] = None, out=None):
        """Normalise + pairwise loss [+ distillation head on the normalised central embeddings when teacher_tau (B,K)
        is given]; fills self.logs, returns (dz, dz_aug) (the pair ``out`` when given) or (None, None)."""
        if loss_fn not in _capi.CONTRASTIVE_LOSSES:
            raise NotImplementedError(f"contrastive loss {loss_fn!r} is not built (available: "
                                      f"{sorted(_capi.CONTRASTIVE_LOSSES)})")
        assert tuple(z.shape) == (self.B, self.L) and tuple(z_aug.shape) == (self.B, self.L)
        assert z.is_contiguous() and z_aug.is_contiguous()
        dz, dza = out if (want_grads and out is not None) else (None, None)
        if want_grads and dz is None:
            dz, dza = torch.empty_like(z), torch.empty_like(z)
        rc = self.lib.dof_contrastive_loss(self.plan, z.data_ptr(), z_aug.data_ptr(), _capi.SIMILARITIES[similarity],
                                           _capi.CONTRASTIVE_LOSSES[loss_fn], float(temperature), float(tau),
                                           float(beta), self.params.data_ptr(),
                                           None if teacher_tau is None else teacher_tau.data_ptr(), self.hyper.data_ptr(),
                                           dz.data_ptr() if want_grads else None,
                                           dza.data_ptr() if want_grads else None, self.logs.data_ptr(), self._stream())
        _capi.check(self.lib, rc, "dof_contrastive_loss")
        return dz, dza

    def contrastive_backward(self, dz, accumulate: bool):
        assert tuple(dz.shape) == (self.B, self.L) and dz.is_contiguous()
        rc = self.lib.dof_contrastive_backward(self.plan, self.params.data_ptr(), dz.data_ptr(), self.grads.data_ptr(),
                                               1 if accumulate else 0, self._stream())
        _capi.check(self.lib, rc, "dof_contrastive_backward")

    def read_contrastive_logs(self) -> Dict[str, float]:
        v = self.logs.detach().cpu().tolist()
        return {"total_loss": v[0], "pos_similarity": v[_capi.LOG_POS_SIM], "neg_similarity": v[_capi.LOG_NEG_SIM],
                "distill_loss": v[7], "seperability": 0.0}

    def optimizer_step(self, grad_scale: float = 1.0):
        """clip + Adam on the flat buffer; advances the device-side step counters.  grad_scale = 1 / world after an
        all-reduce SUM (the averaging DDP does)."""
        rc = self.lib.dof_optimizer_step(self.plan, self.params.data_ptr(), self.grads.data_ptr(),
                                         self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.hyper.data_ptr(),
                                         self.opt_state.data_ptr(), float(grad_scale), self._stream())
        _capi.check(self.lib, rc, "dof_optimizer_step")

    def read_logs(self) -> Dict[str, float]:
        vals = self.logs.detach().cpu().tolist()
        return {k: vals[i] for i, k in enumerate(_capi.LOG_KEYS)}


def create_vade_engine(batch, window, adjacency, latent_dim, n_clusters, mc_samples=32, device=None, graph_ops=None,
                       shared=None, kind="vade"):
    """Product entry: requires a ROCm GPU and the compiled HIP library (no fallback)."""
    from ._lib import load_hip_library

    if not torch.cuda.is_available():
        raise RuntimeError("deepof_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
    lib = load_hip_library()
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    if dev.type != "cuda":
        raise RuntimeError(f"deepof_amd runs on ROCm devices only, got {dev}")
    return VadeEngine(lib, dev, batch, window, adjacency, latent_dim, n_clusters, mc_samples, graph_ops, shared, kind)


def contrastive_views(lib, x_full: torch.Tensor, edge_index: torch.Tensor, aug: Optional[dict] = None, stream=0,
                      out=None):
    """dof_contrastive_views: one (x, a) view of every full window.  ``aug`` = None gives the central view;
    otherwise a dict with the resolved draws (start, rot_pivot, rot_nodes, theta, interp_t0, interp_len, noise;
    tensors on x_full's device, rot_pivot / rot_nodes host lists).  ``out`` = (x, a) buffers to fill."""
    B, Tf, N, _ = x_full.shape
    E = edge_index.shape[0]
    assert x_full.is_contiguous() and x_full.dtype == torch.float32
    assert edge_index.dtype == torch.int32 and edge_index.is_contiguous() and edge_index.device == x_full.device
    half = Tf // 2
    if out is not None:
        x, a = out
        assert tuple(x.shape) == (B, half, N, 3) and tuple(a.shape) == (B, half, E, 1) and x.is_contiguous() and a.is_contiguous()
    else:
        x = torch.empty(B, half, N, 3, dtype=torch.float32, device=x_full.device)
        a = torch.empty(B, half, E, 1, dtype=torch.float32, device=x_full.device)
    arg = None
    keep = []
    if aug is not None:
        A = _capi.Augment()

        def dev(key, dtype, shape):
            t = aug.get(key)
            if t is None:
                return None
            t = t.to(device=x_full.device, dtype=dtype).contiguous()
            assert tuple(t.shape) == shape, (key, tuple(t.shape), shape)
            keep.append(t)
            return t.data_ptr()

        piv = list(aug.get("rot_pivot", []))
        A.n_rot = len(piv)
        assert A.n_rot <= _capi.MAX_ROT
        for r, (pv, nodes) in enumerate(zip(piv, aug.get("rot_nodes", []))):
            A.rot_pivot[r] = int(pv)
            m = 0
            for n in nodes:
                m |= 1 << int(n)
            A.rot_nodes[r] = m
        A.start = dev("start", torch.int32, (B,))
        A.theta = dev("theta", torch.float32, (A.n_rot, B)) if A.n_rot else None
        A.interp_t0 = dev("interp_t0", torch.int32, (B,))
        A.interp_len = dev("interp_len", torch.int32, (B,))
        A.noise = dev("noise", torch.float32, (B, N, 3))
        arg = C.byref(A)
    rc = lib.dof_contrastive_views(x_full.data_ptr(), edge_index.data_ptr(), B, Tf, N, E, arg, x.data_ptr(),
                                   a.data_ptr(), stream)
    _capi.check(lib, rc, "dof_contrastive_views")
    return x, a
