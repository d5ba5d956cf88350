"""Model objects with the reference's interface, backed by the HIP engine.

``VaDE`` mirrors what downstream DeepOF code touches on ``VaDEPT``
(/root/reference/deepof/clustering/models_new.py:1794-1975, consumed by ``embedding_per_video``
model_utils_new.py:542-617): an ``nn.Module`` with ``window_size``, ``encoder.spatial_gnn_block``
(whose ``str()`` is ``"CensNetConvPT()"``), ``latent_space.*``, the reference ``state_dict`` keys
and order (checkpoints are cross-loadable), and ``model(x, a, return_gmm_params=False) ->
(reconstruction_dist, z, q, kmeans_loss)``.  All tensors are views into the engine's flat fp32
buffer; all arithmetic happens in libdeepof_hip.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _capi
from .engine import VadeEngine, create_vade_engine


class _Box(nn.Module):
    """Plain container used to rebuild the reference's module tree (names only, no compute)."""


class CensNetConvPT(_Box):  # name matters: embedding_per_video checks str(block) == "CensNetConvPT()"
    pass


class ReconDistribution:
    """Independent Normal(loc, 1) over the 3N node features of each frame, masked by frame validity
    (models_new.py:686-710): .mean and .log_prob(x) as the reference's AffineTransformedDistribution."""

    def __init__(self, loc: torch.Tensor, valid: torch.Tensor):
        self.loc, self.valid = loc, valid

    @property
    def mean(self) -> torch.Tensor:
        return self.loc * self.valid.unsqueeze(-1).to(self.loc.dtype)

    def log_prob(self, x: torch.Tensor) -> torch.Tensor:
        d = x.shape[-1]
        lp = -0.5 * ((x - self.loc) ** 2).sum(dim=-1) - 0.5 * d * math.log(2.0 * math.pi)
        return torch.where(self.valid, lp, torch.full_like(lp, float("nan")))


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, CensNetConvPT() if p == "spatial_gnn_block" else _Box())
        mod = getattr(mod, p)
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def _attach_all(model: nn.Module, eng: VadeEngine, extra=None):
    """Register the engine's tensors on the module tree under the reference names: parameters, BatchNorm running
    buffers (views of the flat buffer) and their int64 step counters (shared with the engine), in state_dict order."""
    for name in eng.names:
        if name.startswith("distill_head."):  # the reference's DiscriminativeHead is not part of the model
            continue
        if extra is not None:
            extra(name)
        _attach(model, name, eng.view(name), buffer=".running_" in name)
        if name.endswith(".running_var"):
            layer = name[: -len(".running_var")]
            _attach(model, layer + ".num_batches_tracked", eng.num_batches_tracked[layer], buffer=True)


@torch.no_grad()
def _reset_tcn_family(model: nn.Module, latent_dim: int):
    """Initialisers of the TCN family (models_new.py:420-430 convs ~ N(0, 0.05) / zero bias; :596-601 and :764-766
    MLPs xavier-uniform / zero bias; BatchNorm identity; CensNet censNetConv_pt.py:62-84; GMM xavier-normal)."""
    for name, p in model.named_parameters():
        leaf = name.split(".")[-1]
        if ".bn" in name or name.startswith("encoder.head.2") or name.startswith("encoder.head.5"):
            p.fill_(1.0 if leaf == "weight" else 0.0)
        elif "spatial_gnn_block" in name:
            if leaf in ("node_kernel", "edge_kernel", "node_weights", "edge_weights"):
                nn.init.xavier_uniform_(p)
            else:
                bound = 1.0 / math.sqrt(latent_dim)
                p.uniform_(-bound, bound)
        elif ".head." in name or name.startswith("decoder.fc"):
            nn.init.xavier_uniform_(p) if leaf == "weight" else p.zero_()
        elif name in ("latent_space.gmm_means", "latent_space.gmm_log_vars"):
            nn.init.xavier_normal_(p)
        elif name == "vq_layer.codebook":
            p.uniform_(0.0, 1.0)
        elif "_tcn." in name or ".tcn." in name:
            p.normal_(0.0, 0.05) if leaf == "weight" else p.zero_()
        elif leaf == "weight":  # nn.Linear default: U(+-1/sqrt(fan_in))
            fan_in = int(np.prod(p.shape[1:]))
            p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        elif leaf == "bias":
            w = dict(model.named_parameters())[name[: -len("bias")] + "weight"]
            fan_in = int(np.prod(w.shape[1:]))
            p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
    for name, b in model.named_buffers():
        if name.endswith("running_mean"):
            b.zero_()
        elif name.endswith("running_var"):
            b.fill_(1.0)
        elif name.endswith("num_batches_tracked"):
            b.zero_()


@torch.no_grad()
def _reset_tfm_family(model: nn.Module, latent_dim: int):
    """Initialisers of the transformer family: every Linear of the encoder cores / head and of the whole decoder is
    xavier-uniform with zero bias (models_new.py:858-859, 909-912, 941-942, 1085-1088, 1227-1231, 1297-1302), LayerNorm
    and BatchNorm identity, CensNet as censNetConv_pt.py:62-84, GMM xavier-normal, latent heads nn.Linear default."""
    params = dict(model.named_parameters())
    for name, p in params.items():
        leaf = name.split(".")[-1]
        if ".norm" in name or name.startswith("encoder.head.2") or name.startswith("encoder.head.5"):
            p.fill_(1.0 if leaf == "weight" else 0.0)
        elif "spatial_gnn_block" in name:
            if leaf in ("node_kernel", "edge_kernel", "node_weights", "edge_weights"):
                nn.init.xavier_uniform_(p)
            else:
                bound = 1.0 / math.sqrt(latent_dim)
                p.uniform_(-bound, bound)
        elif name in ("latent_space.gmm_means", "latent_space.gmm_log_vars"):
            nn.init.xavier_normal_(p)
        elif name == "vq_layer.codebook":
            p.uniform_(0.0, 1.0)
        elif name.startswith("latent_space."):  # nn.Linear default: U(+-1/sqrt(fan_in))
            w = p if leaf == "weight" else params[name[: -len("bias")] + "weight"]
            fan_in = int(np.prod(w.shape[1:]))
            p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
        elif leaf == "weight":
            nn.init.xavier_uniform_(p)
        else:
            p.zero_()
    for name, b in model.named_buffers():
        if name.endswith("running_mean"):
            b.zero_()
        elif name.endswith("running_var"):
            b.fill_(1.0)
        elif name.endswith("num_batches_tracked"):
            b.zero_()


_FAMILY_SUFFIX = {"recurrent": "", "tcn": "_tcn", "transformer": "_tfm"}
_FAMILY_NAME = {"recurrent": "recurrent", "tcn": "TCN", "transformer": "transformer"}


def _encoder_family(encoder_type) -> str:
    """"recurrent", "tcn" or "transformer" (the reference's three encoder_type values, any case)."""
    enc = str(encoder_type).lower()
    if enc not in _FAMILY_SUFFIX:
        raise NotImplementedError(f'invalid encoder type {encoder_type!r}, try "recurrent", "TCN" or "transformer"')
    return enc


class VaDE(nn.Module):
    def __init__(self, input_shape, edge_feature_shape, adjacency_matrix: np.ndarray, latent_dim: int,
                 n_components: int, encoder_type: str = "recurrent", use_gnn: bool = True, kmeans_loss: float = 1.0,
                 interaction_regularization: float = 0.0, lens_enabled: bool = False, batch_size: int = 256,
                 device=None, _engine_factory: Optional[Callable[..., VadeEngine]] = None):
        super().__init__()
        self._family = _encoder_family(encoder_type)
        self._tcn = self._family != "recurrent"  # BatchNorm head + lazily built CensNet (quirk Q11): TCN and transformer
        self._KIND = "vade" + _FAMILY_SUFFIX[self._family]
        if not use_gnn:
            raise NotImplementedError("use_gnn=False is not implemented (the reference trainer always passes True)")
        time_steps, n_nodes, n_feat = (int(v) for v in input_shape)
        if n_feat != 3 or int(edge_feature_shape[-1]) != 1:
            raise ValueError("expected 3 features per node and 1 per edge")
        self.window_size = time_steps
        self.input_n_nodes = n_nodes
        self.input_n_features_per_node = n_feat
        self.latent_dim = int(latent_dim)
        self.n_components = int(n_components)
        self.encoder_type = _FAMILY_NAME[self._family]
        self.kmeans_weight = float(kmeans_loss)
        self.lens_enabled = False
        self._adjacency = np.asarray(adjacency_matrix, dtype=np.float32)
        self._factory = _engine_factory or (lambda **kw: create_vade_engine(device=device, **kw))
        self._engines: Dict[int, VadeEngine] = {}
        self._base = self._make_engine(int(batch_size), None)
        eng = self._base
        # --- reference module tree / state_dict (order follows registration)
        self.encoder = _Box()
        self.encoder.register_buffer("laplacian", torch.from_numpy(eng.lap.copy()))
        self.encoder.register_buffer("edge_laplacian", torch.from_numpy(eng.elap.copy()))
        self.encoder.register_buffer("incidence", torch.from_numpy(eng.inc.copy()))
        self.decoder = _Box()
        self.latent_space = _Box()

        def latent_buffers(name):
            if name == "latent_space.encoder_mean.weight":
                self.latent_space.register_buffer("prior", eng.prior)
                self.latent_space.register_buffer("pretrain", torch.tensor(0.0))

        _attach_all(self, eng, latent_buffers)
        self.reset_parameters()

    # ------------------------------------------------------------------ engines
    _KIND = "vade"

    def _make_engine(self, batch: int, shared):
        eng = self._factory(batch=batch, window=self.window_size, adjacency=self._adjacency,
                            latent_dim=self.latent_dim, n_clusters=self.n_components, shared=shared, kind=self._KIND)
        if getattr(self, "_family", "") == "transformer":
            # dropout masks: counter hash on the device, seeded from torch's generator (torch.manual_seed reproduces a run)
            # Every plan of the model (base, other batch sizes, the augmented-view plans) gets its own seed AND all of
            # them read ONE device step counter that each train-mode forward advances: the two contrastive views and the
            # ragged last batch draw masks no other forward of the run has used, as the reference's F.dropout calls do.
            if not hasattr(self, "_dropout_seed"):
                self._dropout_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
                self._dropout_plans = 0
                self._dropout_counter = torch.zeros(1, dtype=torch.int32, device=eng.device)
            eng.set_dropout(None, seed=self._dropout_seed + 7919 * self._dropout_plans)
            eng.set_dropout_counter(self._dropout_counter)
            self._dropout_plans += 1
        if getattr(self, "_tcn", False):
            eng.set_bn_training(self.training)
            if not getattr(self, "censnet_in_optimizer", False):
                # reference quirk Q11: the TCN encoder creates its CensNet tensors lazily at the first forward, i.e.
                # AFTER fit_VQVAE / fit_contrastive / fit_VADE's pre-training have built their optimiser: they get
                # gradients but are never updated (fit_VADE's main-phase optimiser does include them)
                for name in eng.names:
                    if ".spatial_gnn_block." in name:
                        eng.set_trainable(name, False)
        return eng

    def set_censnet_trainable(self, flag: bool):
        """Put the TCN encoder's CensNet tensors into / out of the optimiser on every plan of this model (Q11)."""
        self.censnet_in_optimizer = bool(flag)
        if getattr(self, "_tcn", False):
            for eng in self._all_engines():
                for name in eng.names:
                    if ".spatial_gnn_block." in name:
                        eng.set_trainable(name, flag)

    def _all_engines(self):
        out = [self._base] + list(self._engines.values())
        return out + list(getattr(self, "_aug_engines", {}).values())

    def train(self, mode: bool = True):
        """module.train()/eval(): the TCN family's BatchNorm layers follow the module mode in the step entries."""
        super().train(mode)
        if getattr(self, "_tcn", False) and hasattr(self, "_base"):
            for eng in self._all_engines():
                eng.set_bn_training(mode)
        return self

    def engine(self, batch: int) -> VadeEngine:
        """Plan + workspace for this batch size (parameters are shared across batch sizes)."""
        batch = int(batch)
        if batch == self._base.B:
            return self._base
        if batch not in self._engines:
            self._engines[batch] = self._make_engine(batch, self._base)
        return self._engines[batch]

    @property
    def device(self) -> torch.device:
        return self._base.device

    # ------------------------------------------------------------------ init (PyTorch default inits of the reference)
    @torch.no_grad()
    def reset_parameters(self):
        if getattr(self, "_family", "") == "transformer":
            return _reset_tfm_family(self, self.latent_dim)
        if getattr(self, "_tcn", False):
            return _reset_tcn_family(self, self.latent_dim)
        for name, p in self.named_parameters():
            leaf = name.split(".")[-1]
            if ".gru" in name:
                hid = p.shape[0] // 3
                p.uniform_(-1.0 / math.sqrt(hid), 1.0 / math.sqrt(hid))
            elif "norm" in name:
                p.fill_(1.0 if leaf == "weight" else 0.0)
            elif name in ("latent_space.gmm_means", "latent_space.gmm_log_vars"):
                nn.init.xavier_normal_(p)
            elif "spatial_gnn_block" in name:
                if leaf in ("node_kernel", "edge_kernel", "node_weights", "edge_weights"):
                    nn.init.xavier_uniform_(p)
                else:  # bias ~ U(+-1/sqrt(fan_in)), fan_in = channels (censNetConv_pt.py:75-84)
                    bound = 1.0 / math.sqrt(self.latent_dim)
                    p.uniform_(-bound, bound)
            elif leaf == "weight":  # Conv1d / Linear: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
                fan_in = int(np.prod(p.shape[1:]))
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            elif leaf == "bias":
                w = dict(self.named_parameters())[name[: -len("bias")] + "weight"]
                fan_in = int(np.prod(w.shape[1:]))
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))

    # ------------------------------------------------------------------ reference-facing API
    def set_pretrain_mode(self, pretrain_on: bool):
        self.latent_space.pretrain.fill_(1.0 if pretrain_on else 0.0)

    @property
    def get_gmm_params(self) -> dict:
        means, log_vars = self.latent_space.gmm_means, self.latent_space.gmm_log_vars
        return {"means": means, "log_vars": log_vars, "sigmas": torch.exp(0.5 * log_vars),
                "weights": self.latent_space.prior}

    def _run(self, x: torch.Tensor, a: torch.Tensor, eps=None, want_loc=True):
        x = x.to(self.device, torch.float32).contiguous()
        a = a.to(self.device, torch.float32).contiguous()
        eng = self.engine(x.shape[0])
        out = eng.forward(x, a, eps, want_loc=want_loc)
        return x, out

    def forward(self, x: torch.Tensor, a: torch.Tensor, return_gmm_params: bool = False):
        """(reconstruction_dist, z, q, kmeans_loss[, z_mean, z_log_var, gmm_params]).  Training mode draws the
        reparameterisation noise on device; eval mode uses z = z_mean (models_new.py:1761-1791)."""
        eps = None
        if self.training:
            eps = torch.randn(x.shape[0], self.latent_dim, device=self.device)
        x, out = self._run(x, a, eps)
        B, T = x.shape[:2]
        x_flat = x.reshape(B, T, -1)
        valid = ~torch.all(x_flat == 0.0, dim=2)
        dist = ReconDistribution(out["loc"], valid)
        kmeans = self._kmeans_value(out["z"])
        if return_gmm_params:
            gmm = {"means": self.latent_space.gmm_means, "log_vars": self.latent_space.gmm_log_vars,
                   "prior": self.latent_space.prior}
            return dist, out["z"], out["q"], kmeans, out["z_mean"], out["z_log_var"], gmm
        return dist, out["z"], out["q"], kmeans

    def _kmeans_value(self, z: torch.Tensor) -> torch.Tensor:
        """Gram-spectrum term of the latent layer (losses.py:257-287); host-side fp64 on an LxL matrix, value only
        (inside the training step the HIP path computes value and gradient on device)."""
        if not self.kmeans_weight > 0:
            return torch.zeros((), device=z.device)
        gram = ((z.T @ z) / float(z.shape[0])).double()
        if gram.device.type == "cuda":
            # on the device, no host round trip: the Gram matrix is symmetric positive semi-definite, so its singular
            # values are its eigenvalues (the reference's svdvals, losses.py:279)
            sv = torch.linalg.eigvalsh(gram).clamp_min(0.0)
        else:
            sv = torch.linalg.svdvals(gram)
        return (self.kmeans_weight * torch.sqrt(torch.clamp(sv, min=1e-9)).mean()).to(torch.float32)

    @torch.no_grad()
    def embed(self, x, a):
        """Latent embedding z (eval: z_mean).  (The reference's embed()/group() raise on a tuple-unpack bug, Q8.)"""
        was = self.training
        self.eval()
        _, out = self._run(x, a, None, want_loc=False)
        self.train(was)
        return out["z"]

    @torch.no_grad()
    def group(self, x, a):
        was = self.training
        self.eval()
        _, out = self._run(x, a, None, want_loc=False)
        self.train(was)
        return out["q"]

    @torch.no_grad()
    def encode_windows(self, x: torch.Tensor, a: torch.Tensor, batch: int = 256):
        """Batched inference over many windows -> (embeddings (n,L), soft_counts (n,K)) like the inner loop of
        embedding_per_video (model_utils_new.py:604-617).  The last ragged chunk uses its own plan."""
        n = x.shape[0]
        zs, qs = [], []
        for s in range(0, n, batch):
            xb, ab = x[s:s + batch], a[s:s + batch]
            _, out = self._run(xb, ab, None, want_loc=False)
            zs.append(out["z"])
            qs.append(out["q"])
        return torch.cat(zs), torch.cat(qs)


class VQVAE(VaDE):
    """VQ-VAE with the reference's interface (models_new.py:1507-1640, VQVAEPT): ``state_dict`` = encoder.*,
    decoder.*, vq_layer.codebook (L,K); ``model(x, a, return_losses=True, return_all_outputs=False)``."""

    _KIND = "vqvae"

    def __init__(self, input_shape, edge_feature_shape, adjacency_matrix, latent_dim: int, n_components: int,
                 encoder_type: str = "recurrent", use_gnn: bool = True, kmeans_loss: float = 0.0,
                 interaction_regularization: float = 0.0, beta: float = 1.0, batch_size: int = 256, device=None,
                 _engine_factory=None):
        nn.Module.__init__(self)
        self._family = _encoder_family(encoder_type)
        self._tcn = self._family != "recurrent"
        self._KIND = "vqvae" + _FAMILY_SUFFIX[self._family]
        if not use_gnn:
            raise NotImplementedError("use_gnn=False is not implemented (the reference trainer always passes True)")
        time_steps, n_nodes, n_feat = (int(v) for v in input_shape)
        self.window_size, self.input_n_nodes, self.input_n_features_per_node = time_steps, n_nodes, n_feat
        self.latent_dim, self.n_components = int(latent_dim), int(n_components)
        self.encoder_type, self.beta, self.kmeans_weight = _FAMILY_NAME[self._family], float(beta), float(kmeans_loss)
        self._adjacency = np.asarray(adjacency_matrix, dtype=np.float32)
        self._factory = _engine_factory or (lambda **kw: create_vade_engine(device=device, **kw))
        self._engines = {}
        self._base = self._make_engine(int(batch_size), None)
        eng = self._base
        self.encoder = _Box()
        self.encoder.register_buffer("laplacian", torch.from_numpy(eng.lap.copy()))
        self.encoder.register_buffer("edge_laplacian", torch.from_numpy(eng.elap.copy()))
        self.encoder.register_buffer("incidence", torch.from_numpy(eng.inc.copy()))
        self.decoder = _Box()
        self.vq_layer = _Box()
        _attach_all(self, eng)
        self.reset_parameters()
        with torch.no_grad():
            self.vq_layer.codebook.uniform_(0.0, 1.0)  # models_new.py:1348-1350

    def _quantise(self, x, a, want_loc=True):
        x = x.to(self.device, torch.float32).contiguous()
        a = a.to(self.device, torch.float32).contiguous()
        return x, self.engine(x.shape[0]).vq_forward(x, a, want_loc=want_loc)

    def forward(self, x, a, return_losses: bool = True, return_all_outputs: bool = False):
        x, out = self._quantise(x, a)
        B, T = x.shape[:2]
        valid = ~torch.all(x.reshape(B, T, -1) == 0.0, dim=2)
        enc_dist, rec_dist = ReconDistribution(out["loc_q"], valid), ReconDistribution(out["loc_e"], valid)
        losses = None
        if return_losses:
            sq = torch.mean((out["quantized"] - out["ze"]) ** 2)
            losses = {"vq_loss": (self.beta + 1.0) * sq}
            if self.kmeans_weight:
                losses["kmeans_loss"] = self._kmeans_value(out["ze"])
        if return_all_outputs:
            return enc_dist, rec_dist, out["quantized"], out["soft_counts"], out["ze"], losses
        return (enc_dist, rec_dist, losses) if return_losses else (enc_dist, rec_dist)

    @torch.no_grad()
    def encode(self, x, a):
        return self._quantise(x, a, want_loc=False)[1]["ze"]

    embed = encode

    @torch.no_grad()
    def group(self, x, a):
        return self._quantise(x, a, want_loc=False)[1]["soft_counts"]

    @torch.no_grad()
    def encode_windows(self, x, a, batch: int = 256):
        """(embeddings = encoder outputs (n,L), soft_counts (n,K)) -- outputs [4] and [3] of the 6-tuple that
        embedding_per_video reads for VQ-VAE (model_utils_new.py:614)."""
        zs, qs = [], []
        for s in range(0, x.shape[0], batch):
            _, out = self._quantise(x[s:s + batch], a[s:s + batch], want_loc=False)
            zs.append(out["ze"])
            qs.append(out["soft_counts"])
        return torch.cat(zs), torch.cat(qs)


class Contrastive(VaDE):
    """Contrastive embedding model with the reference's interface (models_new.py:1978-2075, ContrastivePT):
    the recurrent encoder built for HALF windows (``window_size = T // 2``); ``state_dict`` = encoder.* only;
    ``model(x_half, a_half)`` -> (B, latent_dim) embeddings.  Every batch size gets a pair of plans (central /
    augmented view) that share the parameters."""

    _KIND = "contrastive"

    def __init__(self, input_shape, edge_feature_shape, adjacency_matrix, latent_dim: int = 8,
                 encoder_type: str = "recurrent", use_gnn: bool = True, temperature: float = 0.1,
                 similarity_function: str = "cosine", loss_function: str = "nce", beta: float = 0.1, tau: float = 0.1,
                 interaction_regularization: float = 0.0, batch_size: int = 256, device=None, _engine_factory=None,
                 n_components: int = 1):
        nn.Module.__init__(self)
        self._family = _encoder_family(encoder_type)
        self._tcn = self._family != "recurrent"
        self._KIND = "contrastive" + _FAMILY_SUFFIX[self._family]
        if not use_gnn:
            raise NotImplementedError("use_gnn=False is not implemented (the reference trainer always passes True)")
        time_steps, n_nodes, n_feat = (int(v) for v in input_shape)
        if int(edge_feature_shape[0]) != time_steps:
            raise ValueError(f"Node and edge time dims must match: T={time_steps}, Te={edge_feature_shape[0]}")
        self.full_time_steps = time_steps
        self.window_size = time_steps // 2  # the encoder sees half windows (room for the time-shift augmentation)
        self.input_shape, self.edge_feature_shape = tuple(input_shape), tuple(edge_feature_shape)
        self.input_n_nodes, self.input_n_features_per_node = n_nodes, n_feat
        self.latent_dim, self.n_components = int(latent_dim), max(1, int(n_components))  # K of the distillation head
        self.encoder_type, self.use_gnn = _FAMILY_NAME[self._family], True
        self.temperature, self.similarity_function, self.loss_function = float(temperature), similarity_function, loss_function
        self.beta, self.tau = float(beta), float(tau)
        self.interaction_regularization = interaction_regularization
        self.kmeans_weight = 0.0
        self._adjacency = np.asarray(adjacency_matrix, dtype=np.float32)
        self.adjacency_matrix = self._adjacency
        self._factory = _engine_factory or (lambda **kw: create_vade_engine(device=device, **kw))
        self._engines = {}
        self._aug_engines = {}
        self._base = self._make_engine(int(batch_size), None)
        eng = self._base
        self.encoder = _Box()
        self.encoder.register_buffer("laplacian", torch.from_numpy(eng.lap.copy()))
        self.encoder.register_buffer("edge_laplacian", torch.from_numpy(eng.elap.copy()))
        self.encoder.register_buffer("incidence", torch.from_numpy(eng.inc.copy()))
        _attach_all(self, eng)
        self.reset_parameters()

    def aug_engine(self, batch: int) -> VadeEngine:
        """Second plan/workspace of this batch size: holds the augmented view's activations until its backward."""
        batch = int(batch)
        if batch not in self._aug_engines:
            self._aug_engines[batch] = self._make_engine(batch, self._base)
        return self._aug_engines[batch]

    def forward(self, x, a):
        """Embeddings of half windows.  With the TCN encoder a module in train() mode normalises with batch
        statistics and refreshes the running buffers, as the reference module does; eval() uses the buffers."""
        x = x.to(self.device, torch.float32).contiguous()
        a = a.to(self.device, torch.float32).contiguous()
        return self.engine(x.shape[0]).contrastive_encode(x, a, train=self.training and self._tcn)

    @torch.no_grad()
    def embed(self, x, a):
        was = self.training
        self.eval()
        z = self.forward(x, a)
        self.train(was)
        return z

    def group(self, x, a):
        raise NotImplementedError("the contrastive model has no cluster head (ContrastivePT defines none)")

    set_pretrain_mode = None
    get_gmm_params = None

    @torch.no_grad()
    def encode_windows(self, x, a, batch: int = 256):
        """Embeddings of many HALF windows (model_utils_new.py:604-617 feeds the sliced centre of each window)."""
        zs = [self.embed(x[s:s + batch], a[s:s + batch]) for s in range(0, x.shape[0], batch)]
        return torch.cat(zs), None
