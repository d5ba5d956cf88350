"""CPU restatement of the reference's TURTLE teacher (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows /root/reference/deepof/clustering/teacher_model.py:
  soft_cross_entropy_logits   :32-40
  TurtleHeads                 :43-109   per-view linear heads, M inner SGD steps (lr 0.1, weight decay 1e-4) on
                                        soft-CE(head(normalize(f)) / T_head, tau)
  TaskEncoder                 :112-149  tau = softmax(mean_v (W_v f_v + c_v) / T_task)
  TurtleTeacher.fit           :240-350  outer step: tau -> inner fit -> loss(tau | heads) -> Adam(lr_theta) on the
                                        task encoder
  initialize_gmm_from_teacher :394-460
Everything random in the reference (nn.Linear initialisation, DataLoader shuffling) is an input here: the initial
weights and the list of batches are explicit, so the reference (run with the same ones), this restatement and the
HIP path can be compared step by step.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def soft_ce(logits, soft_targets, eps=1e-8):
    return -(torch.clamp(soft_targets, min=eps, max=1.0) * F.log_softmax(logits, dim=-1)).sum(dim=-1).mean()


def task_tau(P: Dict[str, torch.Tensor], feats: List[torch.Tensor], task_temp: float) -> torch.Tensor:
    logits = None
    for v, f in enumerate(feats):
        out = F.linear(f.float(), P[f"task_encoder.projs.{v}.weight"], P[f"task_encoder.projs.{v}.bias"]) / task_temp
        logits = out if logits is None else logits + out
    return F.softmax(logits / max(len(feats), 1), dim=-1)


def head_logits(P, feats, head_temp: float, normalize: bool = True):
    out = []
    for v, f in enumerate(feats):
        f = F.normalize(f.float(), dim=-1) if normalize else f.float()
        out.append(F.linear(f, P[f"heads.heads.{v}.weight"], P[f"heads.heads.{v}.bias"]) / head_temp)
    return out


def inner_fit(P, feats, tau, M: int, head_temp: float, lr: float = 0.1, wd: float = 1e-4, normalize: bool = True):
    """M plain-SGD steps per head (torch.optim.SGD: p <- p - lr * (grad + wd * p)); updates P in place."""
    tau = tau.detach().float()
    for v, f in enumerate(feats):
        fn = F.normalize(f.detach().float(), dim=-1) if normalize else f.detach().float()
        W = P[f"heads.heads.{v}.weight"].clone().requires_grad_(True)
        b = P[f"heads.heads.{v}.bias"].clone().requires_grad_(True)
        for _ in range(M):
            loss = soft_ce(F.linear(fn, W, b) / head_temp, tau)
            gW, gb = torch.autograd.grad(loss, [W, b])
            with torch.no_grad():
                W -= lr * (gW + wd * W)
                b -= lr * (gb + wd * b)
        P[f"heads.heads.{v}.weight"], P[f"heads.heads.{v}.bias"] = W.detach(), b.detach()


def outer_loss(tau, logits_k, step: int, outer_steps: int, n_components: int, gamma: float, alpha: float, delta: float,
               rho: float = 0.04):
    """teacher_model.py:283-332.  Returns (loss, dict of terms)."""
    def entropy(p, eps=1e-9):
        p = p.clamp_min(eps)
        return -(p * p.log()).sum(dim=-1)

    ce = sum(soft_ce(lg, tau) for lg in logits_k) / max(len(logits_k), 1)
    sample_entropy = entropy(tau).mean()
    marginal = tau.mean(dim=0)
    h_marg = entropy(marginal.unsqueeze(0)).mean()
    marg_gap = torch.relu(torch.tensor(math.log(n_components), dtype=tau.dtype) - h_marg)
    gamma_t = float(gamma) * (1.0 - float(step) / float(max(1, outer_steps)))
    dead_floor = max(1e-4, 0.1 / n_components)
    usage = (tau.clamp_min(1e-8) ** 2.0).mean(dim=0)
    dead_pen = torch.relu(dead_floor - usage).sum() / (dead_floor * n_components)
    delta_t = delta * max(0.5, 0.6 + 0.4 * (1.0 - step / float(max(1, outer_steps))))
    loss = ce + alpha * sample_entropy + gamma_t * marg_gap + delta_t * dead_pen
    if (step % 2) != 0 and rho > 0.0:
        loss = loss + rho * (tau[1:] - tau[:-1]).abs().sum(dim=-1).mean()
    return loss, dict(ce=float(ce), sample_entropy=float(sample_entropy), h_marginal=float(h_marg), dead_pen=float(dead_pen))


class Adam:
    def __init__(self, lr):
        self.lr, self.t, self.m, self.v = lr, 0, {}, {}

    def step(self, P, grads):
        self.t += 1
        for k, g in grads.items():
            m = self.m.get(k, torch.zeros_like(g)) * 0.9 + 0.1 * g
            v = self.v.get(k, torch.zeros_like(g)) * 0.999 + 0.001 * g * g
            self.m[k], self.v[k] = m, v
            P[k] = (P[k] - (self.lr / (1 - 0.9 ** self.t)) * m / (v.sqrt() / math.sqrt(1 - 0.999 ** self.t) + 1e-8)).detach()


def fit(P: Dict[str, torch.Tensor], batches: List[List[torch.Tensor]], n_components: int, outer_steps: int, inner_steps: int,
        gamma: float = 8.0, alpha: float = 2.0, delta: float = 40.0, head_temp: float = 0.35, task_temp: float = 0.35,
        lr_theta: float = 1e-3, normalize: bool = True, rho: float = 0.04):
    """TurtleTeacher.fit over the given batch sequence (cycled); P is updated in place.  Returns per-step losses."""
    opt = Adam(lr_theta)
    losses = []
    for step in range(outer_steps):
        feats = batches[step % len(batches)]
        keys = [k for k in P if k.startswith("task_encoder.")]
        leaf = {k: P[k].clone().requires_grad_(True) for k in keys}
        tau = task_tau({**P, **leaf}, feats, task_temp)
        inner_fit(P, feats, tau, inner_steps, head_temp, normalize=normalize)
        with torch.no_grad():
            lk = head_logits(P, feats, head_temp, normalize)
        loss, _ = outer_loss(tau, lk, step, outer_steps, n_components, gamma, alpha, delta, rho)
        gs = torch.autograd.grad(loss, [leaf[k] for k in keys])
        opt.step(P, dict(zip(keys, gs)))
        losses.append(float(loss))
    return losses


def predict(P, feats: List[torch.Tensor], task_temp: float):
    with torch.no_grad():
        return task_tau(P, feats, task_temp)


def gmm_from_teacher(z: torch.Tensor, tau: torch.Tensor, min_var: float = 1e-4, min_mass: float = 1e-6):
    """initialize_gmm_from_teacher (teacher_model.py:394-460): (means (K,L), log_vars (K,L), prior (K))."""
    mass = tau.sum(dim=0) + min_mass
    prior = (mass / mass.sum()).clamp(min=1e-8, max=1.0)
    means = (tau.T @ z) / mass.unsqueeze(1)
    diffs = z.unsqueeze(1) - means.unsqueeze(0)
    vars_ = ((tau.unsqueeze(-1) * diffs ** 2).sum(dim=0) / mass.unsqueeze(-1)).clamp(min=min_var)
    log_vars = vars_.log()
    tiny = mass <= 1e-4
    if tiny.any():
        means[tiny] = z.mean(dim=0)
        log_vars[tiny] = z.var(dim=0, unbiased=False).clamp(min=min_var).log()
    return means, log_vars, prior
