"""TURTLE teacher: soft cluster targets tau* for distillation (reference deepof/clustering/teacher_model.py).

Host side of SURVEY section 8f row N3.  The optimisation itself (task encoder, per-view heads, bi-level
steps, prediction) runs in ``libdeepof_hip`` (``dof_turtle_*``, csrc/k_turtle.hip); this module mirrors the
reference's interface around it:

  TurtleTeacher                  teacher_model.py:152-350  (state_dict names, fit(loader), predict)
  run_turtle_teacher_on_views    :710-792   shuffled drop-last batches of the active views, then tau* in dataset order
  fit_nodes_pca / extract_pca_edges_view  :464-707  PCA views -- IncrementalPCA on the device (DeviceIncrementalPCA;
                                                    pca_backend="sklearn" = the host implementation), as in the
                                 reference (third-party arithmetic; fitted once per run)
  extract_latents                :354-391   z_mean of every training window
  initialize_gmm_from_teacher    :394-460   tau*-weighted moments -> GMM means / log-variances / prior
  maybe_build_turtle_teacher     :811-905
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _capi


class TurtleTeacher:
    def __init__(self, feature_dims: Sequence[int], n_components: int, gamma: float = 10.0,
                 alpha_sample_entropy: float = 0.1, inner_lr: float = 0.1, inner_steps: int = 100, head_wd: float = 1e-4,
                 head_temp: float = 0.5, task_temp: float = 0.5, normalize_feats: bool = True, lr_theta: float = 5e-3,
                 delta_death_barrier: float = 40.0, device=None, lib=None):
        if lib is None:
            from ._lib import load_hip_library
            if not torch.cuda.is_available():
                raise RuntimeError("deepof_amd needs a ROCm GPU (the TURTLE teacher has no CPU path)")
            lib = load_hip_library()
            device = device if device is not None else f"cuda:{torch.cuda.current_device()}"
        self.lib, self.device = lib, torch.device(device if device is not None else "cpu")
        self.feature_dims, self.n_components = [int(d) for d in feature_dims], int(n_components)
        self.hyper = _capi.TurtleHyper(float(gamma), float(alpha_sample_entropy), float(delta_death_barrier),
                                       float(head_temp), float(task_temp), float(inner_lr), float(head_wd),
                                       float(lr_theta), 0.04, int(inner_steps), 1 if normalize_feats else 0)
        self._dims = self._make_dims(2)
        total = lib.dof_turtle_param_total(C.byref(self._dims))
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(total, **f32)
        self.adam_m, self.adam_v = torch.zeros(total, **f32), torch.zeros(total, **f32)
        self.logs = torch.zeros(8, **f32)
        self._ws: Dict[int, torch.Tensor] = {}
        self.reset_parameters()

    # ---------------------------------------------------------------- plumbing
    def _make_dims(self, batch: int) -> _capi.TurtleDims:
        d = _capi.TurtleDims()
        d.batch, d.n_views, d.n_clusters = int(batch), len(self.feature_dims), self.n_components
        for v, fd in enumerate(self.feature_dims):
            d.view_dim[v] = fd
        return d

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def _view(self, task: bool, v: int, bias: bool) -> torch.Tensor:
        off = self.lib.dof_turtle_param_offset(C.byref(self._dims), 1 if task else 0, v, 1 if bias else 0)
        K, d = self.n_components, self.feature_dims[v]
        return self.params[off:off + K] if bias else self.params[off:off + K * d].view(K, d)

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """TurtleTeacher.state_dict() of the reference (heads first, then the task encoder)."""
        sd = {}
        for v in range(len(self.feature_dims)):
            sd[f"heads.heads.{v}.weight"] = self._view(False, v, False).detach().cpu().clone()
            sd[f"heads.heads.{v}.bias"] = self._view(False, v, True).detach().cpu().clone()
        for v in range(len(self.feature_dims)):
            sd[f"task_encoder.projs.{v}.weight"] = self._view(True, v, False).detach().cpu().clone()
            sd[f"task_encoder.projs.{v}.bias"] = self._view(True, v, True).detach().cpu().clone()
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for v in range(len(self.feature_dims)):
            for task, pre in ((False, f"heads.heads.{v}"), (True, f"task_encoder.projs.{v}")):
                self._view(task, v, False).copy_(torch.as_tensor(sd[pre + ".weight"], dtype=torch.float32))
                self._view(task, v, True).copy_(torch.as_tensor(sd[pre + ".bias"], dtype=torch.float32))

    @torch.no_grad()
    def reset_parameters(self):
        """nn.Linear defaults of the reference modules: U(+-1/sqrt(fan_in)) for weight and bias."""
        for v, d in enumerate(self.feature_dims):
            for task in (False, True):
                bound = 1.0 / math.sqrt(d)
                self._view(task, v, False).uniform_(-bound, bound)
                self._view(task, v, True).uniform_(-bound, bound)
        self.adam_m.zero_()
        self.adam_v.zero_()

    def _feat_ptrs(self, feats: List[torch.Tensor]):
        assert len(feats) == len(self.feature_dims)
        keep = []
        arr = (C.c_void_p * len(feats))()
        for v, f in enumerate(feats):
            f = f.to(self.device, torch.float32).contiguous()
            assert f.dim() == 2 and f.shape[1] == self.feature_dims[v], (tuple(f.shape), self.feature_dims[v])
            keep.append(f)
            arr[v] = f.data_ptr()
        return arr, keep

    # ---------------------------------------------------------------- reference-facing API
    def fit(self, loader: Iterable, outer_steps: int = 200, rho: float = 0.04, verbose: bool = True):
        """teacher_model.py:240-350.  ``loader`` yields lists of per-view batches and is cycled, as in the reference."""
        self.hyper.rho = float(rho)
        iterator = iter(loader)
        for step in range(int(outer_steps)):
            try:
                feats = next(iterator)
            except StopIteration:
                iterator = iter(loader)
                feats = next(iterator)
            arr, keep = self._feat_ptrs(list(feats))
            B = keep[0].shape[0]
            dims = self._make_dims(B)
            if B not in self._ws:
                self._ws[B] = torch.zeros(self.lib.dof_turtle_workspace_bytes(C.byref(dims)) // 4, dtype=torch.float32,
                                          device=self.device)
            rc = self.lib.dof_turtle_fit_step(C.byref(dims), C.byref(self.hyper), arr, self.params.data_ptr(),
                                              self.adam_m.data_ptr(), self.adam_v.data_ptr(), step, int(outer_steps),
                                              self._ws[B].data_ptr(), self.logs.data_ptr(), self._stream())
            _capi.check(self.lib, rc, "dof_turtle_fit_step")
            if verbose and (step % 20 == 0 or step == outer_steps - 1):
                lg = self.logs.cpu().tolist()
                print(f"[Teacher] step {step:03d} | loss {lg[0]:.4f} | CE {lg[1]:.4f} | E[H(tau)] {lg[2]:.4f} | "
                      f"H(marg) {lg[3]:.4f} | dead_pen {lg[4]:.3f}")

    @torch.no_grad()
    def predict_views(self, feats: List[torch.Tensor]) -> torch.Tensor:
        arr, keep = self._feat_ptrs(feats)
        n = keep[0].shape[0]
        tau = torch.empty(n, self.n_components, dtype=torch.float32, device=self.device)
        rc = self.lib.dof_turtle_predict(C.byref(self._dims), float(self.hyper.task_temp), arr, self.params.data_ptr(), n,
                                         tau.data_ptr(), self._stream())
        _capi.check(self.lib, rc, "dof_turtle_predict")
        return tau

    def predict(self, loader: Iterable) -> torch.Tensor:
        """Sequential pass -> tau* (N, K) on the host, in loader order (teacher_model.py:219-238)."""
        return torch.cat([self.predict_views(list(b)).cpu() for b in loader], dim=0)


def run_turtle_teacher_on_views(views_dict: dict, n_components: int, gamma: float = 6.0, alpha_sample_entropy: float = 1.0,
                                outer_steps: int = 200, inner_steps: int = 200, normalize_feats: bool = True,
                                verbose: bool = True, device=None, head_temp: float = 0.3, task_temp: float = 0.3,
                                batch_size: int = 2048, seed: Optional[int] = None, lib=None):
    """teacher_model.py:710-792: shuffled, drop-last mini-batches of the active views for ``outer_steps`` steps, then
    tau* for every row in dataset order.  The views stay on the device; batches are row gathers."""
    tensors = [v for v in views_dict.values() if v is not None]
    assert len(tensors) > 0, "No active views found."
    teacher = TurtleTeacher([t.shape[1] for t in tensors], n_components, gamma=gamma,
                            alpha_sample_entropy=alpha_sample_entropy, inner_lr=0.1, inner_steps=inner_steps, head_wd=1e-4,
                            head_temp=head_temp, task_temp=task_temp, normalize_feats=normalize_feats, lr_theta=1e-3,
                            device=device, lib=lib)
    dev = teacher.device
    tensors = [t.to(dev, torch.float32).contiguous() for t in tensors]
    n = tensors[0].shape[0]
    bs = min(int(batch_size), n)
    gen = torch.Generator()
    if seed is not None:
        gen.manual_seed(int(seed))

    class _Loader:  # DataLoader(TensorDataset(*views), batch_size, shuffle=True, drop_last=True)
        def __iter__(self_inner):
            perm = torch.randperm(n, generator=gen).to(dev)
            for s in range(0, n - bs + 1, bs):
                idx = perm[s:s + bs]
                yield [t.index_select(0, idx) for t in tensors]

    if verbose:
        print(f"--- Fitting TurtleTeacher (batch_size={bs}) ---")
    teacher.fit(_Loader(), outer_steps=outer_steps, rho=0.04, verbose=verbose)
    tau_star = torch.cat([teacher.predict_views([t[s:s + 2 * bs] for t in tensors]).cpu()
                          for s in range(0, n, 2 * bs)], dim=0)
    return teacher, tau_star


# ---------------------------------------------------------------------------------------- views
class DeviceIncrementalPCA:
    """``sklearn.decomposition.IncrementalPCA`` (partial_fit / transform, whiten=False) on torch tensors, wherever they
    live.  Same algorithm, same truncation to ``n_components`` after every batch: sklearn takes the SVD of the stacked
    matrix ``[S * Vt ; X - batch_mean ; mean_correction]`` ((k + b + 1) x d, LAPACK gesdd in float32 on the host); here
    its d x d Gram matrix ``(S Vt)^T (S Vt) + Xc^T Xc + c c^T`` is accumulated with one GEMM per batch and decomposed
    with one symmetric eigen-solve, all in float64 on the device -- no window ever leaves HBM.  At BASELINE C2's size
    (600k windows, 700 / 350 / 350 features) the host path needs 226 s for the three teacher views; see DESIGN.md.
    Signs follow ``svd_flip(u_based_decision=False)``: the largest-magnitude entry of every component is positive."""

    def __init__(self, n_components: int):
        self.n_components = int(n_components)
        self.n_seen = 0
        self.mean = None
        self.components = None        # (k, d)
        self.singular_values = None   # (k,)

    @torch.no_grad()
    def partial_fit(self, X: torch.Tensor) -> "DeviceIncrementalPCA":
        X = X.to(torch.float64)
        b, d = X.shape
        k = self.n_components
        if k > min(b, d):
            raise ValueError(f"n_components={k} must be less or equal to the batch number of samples {b} and features {d}")
        batch_sum = X.sum(dim=0)
        n_total = self.n_seen + b
        if self.n_seen == 0:
            col_mean = batch_sum / n_total
            Xc = X - col_mean
            G = Xc.T @ Xc
        else:
            col_mean = (self.mean * self.n_seen + batch_sum) / n_total
            batch_mean = batch_sum / b
            Xc = X - batch_mean
            corr = math.sqrt((self.n_seen / n_total) * b) * (self.mean - batch_mean)
            SV = self.singular_values[:, None] * self.components
            G = SV.T @ SV + Xc.T @ Xc + torch.outer(corr, corr)
        lam, vec = torch.linalg.eigh(G)                     # ascending eigenvalues, eigenvectors in columns
        Vt = vec[:, -k:].T.flip(0).contiguous()             # (k, d), descending
        lam_k = lam[-k:].flip(0).clamp_min(0.0)
        lead = Vt.abs().argmax(dim=1)
        signs = torch.sign(Vt[torch.arange(k, device=Vt.device), lead])
        signs = torch.where(signs == 0, torch.ones_like(signs), signs)
        self.components = Vt * signs[:, None]
        self.singular_values = lam_k.sqrt()
        self.mean, self.n_seen = col_mean, n_total
        return self

    @torch.no_grad()
    def transform(self, X: torch.Tensor) -> torch.Tensor:
        return ((X.to(torch.float64) - self.mean) @ self.components.T).float()


def _pca_two_pass(chunks_fn, n_components: int, backend: str = "device") -> torch.Tensor:
    """Two passes over the chunks (fit, then transform) like teacher_model.py:508-571.  ``backend="sklearn"`` is the
    reference's own host IncrementalPCA (chunks are copied to the host); ``"device"`` keeps everything where the
    chunks live (same algorithm, see DeviceIncrementalPCA).  Features are returned on the host, float32."""
    if backend == "sklearn":
        from sklearn.decomposition import IncrementalPCA
        ipca = IncrementalPCA(n_components=n_components)
        for X in chunks_fn():
            ipca.partial_fit(X.float().cpu().numpy())
        return torch.cat([torch.from_numpy(ipca.transform(X.float().cpu().numpy())).float() for X in chunks_fn()], dim=0)
    if backend != "device":
        raise ValueError("pca backend must be 'device' or 'sklearn'")
    ipca = DeviceIncrementalPCA(n_components)
    for X in chunks_fn():
        ipca.partial_fit(X)
    return torch.cat([ipca.transform(X).cpu() for X in chunks_fn()], dim=0)


def fit_nodes_pca(dataset, n_components_pos: int = 32, n_components_spd: int = 32, batch_size: int = 4096,
                  backend: str = "device"):
    """teacher_model.py:464-573: IncrementalPCA of the flattened (x, y) positions and of the flattened speeds of every
    window (two passes: partial_fit, transform).  Returns (feats_pos (N, n_pos), feats_spd (N, n_spd)) on the host."""
    def chunks(sl):
        def gen():
            for s in range(0, len(dataset), batch_size):
                x, _a = dataset.fetch(s, min(s + batch_size, len(dataset)))
                yield x[..., sl].reshape(x.shape[0], -1).float()
        return gen
    return (_pca_two_pass(chunks(slice(0, 2)), n_components_pos, backend),
            _pca_two_pass(chunks(slice(2, 3)), n_components_spd, backend))


def extract_pca_edges_view(dataset, n_components: int = 16, batch_size: int = 8192, backend: str = "device") -> torch.Tensor:
    """teacher_model.py:638-707."""
    def gen():
        for s in range(0, len(dataset), batch_size):
            _x, a = dataset.fetch(s, min(s + batch_size, len(dataset)))
            yield a.reshape(a.shape[0], -1).float()
    return _pca_two_pass(gen, n_components, backend)


def fit_angles_pca(dataset, n_components: int = 32, batch_size: int = 8192, backend: str = "device") -> torch.Tensor:
    """teacher_model.py:576-635: two-pass IncrementalPCA of the flattened angle windows (n, W*A)."""
    ang = getattr(dataset, "angles", None)
    if ang is None:
        raise RuntimeError("include_angles_view=True but the preprocessed data carries no angle tables")

    def gen():
        for s in range(0, ang.shape[0], batch_size):
            yield torch.as_tensor(ang[s:s + batch_size].reshape(min(batch_size, ang.shape[0] - s), -1)).float()
    return _pca_two_pass(gen, n_components, backend)


@torch.no_grad()
def extract_latents(model, dataset, batch_size: int = 2048) -> torch.Tensor:
    """z_mean of every window of the dataset, in order, on the host (teacher_model.py:354-391)."""
    was = model.training
    model.eval()
    zs = []
    for s in range(0, len(dataset), batch_size):
        x, a = dataset.fetch(s, min(s + batch_size, len(dataset)))
        _, out = model._run(x, a, None, want_loc=False)
        zs.append(out["z_mean"].cpu())
    model.train(was)
    return torch.cat(zs, dim=0)


def gmm_from_teacher(z_all: torch.Tensor, tau_star: torch.Tensor, min_var: float = 1e-4, min_mass: float = 1e-6):
    """tau*-weighted moments (teacher_model.py:430-450): (means (K,L), log_vars (K,L), prior (K)), float64 on the host."""
    z, tau = z_all.double().cpu().numpy(), tau_star.double().cpu().numpy()
    mass = tau.sum(axis=0) + min_mass
    prior = np.clip(mass / mass.sum(), 1e-8, 1.0)
    means = (tau.T @ z) / mass[:, None]
    second = (tau.T @ (z * z)) / mass[:, None] - 2.0 * means * ((tau.T @ z) / mass[:, None]) + means * means * (tau.sum(axis=0) / mass)[:, None]
    log_vars = np.log(np.maximum(second, min_var))
    tiny = mass <= 1e-4
    if tiny.any():
        means[tiny] = z.mean(axis=0)
        log_vars[tiny] = np.log(np.maximum(z.var(axis=0), min_var))
    return (torch.from_numpy(means).float(), torch.from_numpy(log_vars).float(), torch.from_numpy(prior).float())


@torch.no_grad()
def initialize_gmm_from_teacher(model, z_all: torch.Tensor, tau_star: torch.Tensor, min_var: float = 1e-4,
                                min_mass: float = 1e-6) -> None:
    means, log_vars, prior = gmm_from_teacher(z_all, tau_star, min_var, min_mass)
    model.latent_space.gmm_means.copy_(means)
    model.latent_space.gmm_log_vars.copy_(log_vars)
    model.latent_space.prior.copy_(prior)
    print("Initialized GMM from teacher tau*: "
          f"mean |mu|={means.norm(dim=1).mean():.3f}, mean var={log_vars.exp().mean():.5f}, "
          f"entropy(pi)={-(prior * prior.clamp_min(1e-9).log()).sum().item():.3f}")


def maybe_build_turtle_teacher(*, teacher_cfg, common_cfg, train_dataset, device=None,
                               latent_view: Optional[torch.Tensor] = None, lib=None):
    """teacher_model.py:811-905 -> (teacher, tau_star (N,K) host, views)."""
    if not teacher_cfg.use_turtle_teacher:
        return None, None, {}
    views = {"z": None, "pca_pos": None, "pca_spd": None, "pca_edges": None, "pca_angles": None}
    if teacher_cfg.include_latent_view:
        if latent_view is None:
            raise ValueError("include_latent_view=True but latent_view=None")
        views["z"] = latent_view
    backend = getattr(teacher_cfg, "pca_backend", "device")   # "sklearn" = the reference's host IncrementalPCA
    if teacher_cfg.include_nodes_view:
        print("\n--- Building PCA views for teacher (nodes) ---")
        views["pca_pos"], views["pca_spd"] = fit_nodes_pca(train_dataset, teacher_cfg.pca_nodes_dim, teacher_cfg.pca_nodes_dim,
                                                           teacher_cfg.batch_size_nodes, backend=backend)
    if teacher_cfg.include_edges_view:
        print("\n--- Building PCA views for teacher (edges) ---")
        views["pca_edges"] = extract_pca_edges_view(train_dataset, teacher_cfg.pca_edges_dim, teacher_cfg.batch_size_edges,
                                                    backend=backend)
    if teacher_cfg.include_angles_view:
        print("\n--- Building PCA views for teacher (angles) ---")
        views["pca_angles"] = fit_angles_pca(train_dataset, teacher_cfg.pca_angles_dim, teacher_cfg.batch_size_angles,
                                             backend=backend)
    print("\n--- Running TURTLE teacher on views ---")
    teacher, tau_star = run_turtle_teacher_on_views(
        views, common_cfg.n_components, gamma=teacher_cfg.teacher_gamma,
        alpha_sample_entropy=teacher_cfg.teacher_alpha_sample_entropy, outer_steps=teacher_cfg.teacher_outer_steps,
        inner_steps=teacher_cfg.teacher_inner_steps, normalize_feats=teacher_cfg.teacher_normalize_feats, verbose=True,
        device=device, head_temp=teacher_cfg.teacher_head_temp, task_temp=teacher_cfg.teacher_task_temp,
        batch_size=teacher_cfg.teacher_batch_size, seed=common_cfg.seed, lib=lib)
    return teacher, tau_star.detach(), views
