"""CPU restatement of the reference's contrastive step (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows, under /root/reference/deepof/clustering:
  recompute_edges               model_utils_new.py:332-363
  slice_time_per_sample         model_utils_new.py:751-763
  rotation triplets / branches  training.py:2064-2125  build_rotation_precomp
  augmentations                 training.py:2128-2165 (_augment_time_shift), :2167-2250 (_augment_angle_rotations),
                                :2299-2370 (_augment_linear_interpolate_segments), :2253-2296 (_augment_noise_xys),
                                :2373-2403 (_make_augmented_view)
  similarities / losses         losses.py:35-249 (cosine|dot|euclidean|edit x nce|dcl|fc|hard_dcl)
  step                          training.py:482-589 step_contrastive_distill (distillation head excluded: no teacher)
  model                         models_new.py:1978-2075 ContrastivePT (recurrent encoder on the half window)

The reference draws its augmentation randomness from the torch device generator inside the functions;
here every draw is an explicit argument (``AugDraws``), so the HIP path, this restatement and the
reference (replayed through recorded draws, tests/golden/make_golden.py) can be fed identical numbers.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import vade as ov


# ----------------------------------------------------------------------------- geometry helpers
def recompute_edges(x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """model_utils_new.py:332-363: a[..., e, 0] = sqrt(max(|p_i - p_j|^2, 1e-12))."""
    xy = x[..., 0:2]
    d = xy[:, :, edge_index[:, 0].long()] - xy[:, :, edge_index[:, 1].long()]
    return torch.sqrt(torch.clamp((d * d).sum(-1), min=1e-12)).unsqueeze(-1)


def slice_time(x: torch.Tensor, start: torch.Tensor, length: int) -> torch.Tensor:
    """model_utils_new.py:751-763."""
    t = start.long()[:, None] + torch.arange(length)[None, :]
    return x[torch.arange(x.shape[0])[:, None], t]


def rotation_triplets(edge_index_local, n_nodes: int):
    """training.py:2064-2125: every (a, b, c) with a, c neighbours of b (a before c in adjacency-list
    order), and for each the node sets reachable from a / from c without passing through b.
    Returns (triplets [(a,b,c)], branches_a [list[int]], branches_c [list[int]])."""
    adj = [[] for _ in range(n_nodes)]
    for u, v in [tuple(int(q) for q in e) for e in edge_index_local]:
        adj[u].append(v)
        adj[v].append(u)

    def reach(center, side):
        seen, stack = {side}, [side]
        while stack:
            u = stack.pop()
            for v in adj[u]:
                if v != center and v not in seen:
                    seen.add(v)
                    stack.append(v)
        return sorted(seen)

    trips, ba, bc = [], [], []
    for b in range(n_nodes):
        nb = adj[b]
        for i in range(len(nb)):
            for j in range(i + 1, len(nb)):
                trips.append((nb[i], b, nb[j]))
                ba.append(reach(b, nb[i]))
                bc.append(reach(b, nb[j]))
    return trips, ba, bc


# ----------------------------------------------------------------------------- augmentation
@dataclass
class AugDraws:
    """The resolved random choices of one call of _make_augmented_view (all per batch)."""
    start: torch.Tensor            # (B) int: slice start of the augmented view (base + shift, clamped)
    rot_pivot: List[int]           # chosen rotations, in application order: pivot node ...
    rot_nodes: List[List[int]]     # ... and the node set rotated about it
    theta: torch.Tensor            # (R, B) radians, already zero where the sample's rotation gate is off
    interp_t0: torch.Tensor        # (B) int first replaced frame
    interp_len: torch.Tensor       # (B) int segment length, 0 where the gate is off
    noise: torch.Tensor            # (B, N, 3) offsets on (x, y, speed), zero where the gate is off


def choose_rotations(perm, centers, n_nodes, n_rot):
    """training.py:2209-2221: walk the permutation, allow each centre at most twice, stop at n_rot."""
    count = [0] * n_nodes
    chosen = []
    for k in perm:
        b0 = centers[k]
        if count[b0] >= 2:
            continue
        count[b0] += 1
        chosen.append(int(k))
        if len(chosen) >= n_rot:
            break
    return chosen


def augmented_view(x_full: torch.Tensor, edge_index: torch.Tensor, d: AugDraws) -> Tuple[torch.Tensor, torch.Tensor]:
    """training.py:2373-2403 with the draws resolved: shift-slice -> rotations -> interpolation -> noise -> edges."""
    B, T = x_full.shape[0], x_full.shape[1]
    half = T // 2
    x = slice_time(x_full, d.start, half).clone()
    xy = x[..., 0:2].clone()
    for r, (b0, nodes) in enumerate(zip(d.rot_pivot, d.rot_nodes)):
        th = d.theta[r].to(x.dtype)
        c, s = torch.cos(th).view(B, 1, 1), torch.sin(th).view(B, 1, 1)
        pivot = xy[:, :, b0, :].unsqueeze(2)
        idx = torch.tensor(nodes, dtype=torch.long)
        rel = xy[:, :, idx, :] - pivot
        rx = rel[..., 0] * c - rel[..., 1] * s
        ry = rel[..., 0] * s + rel[..., 1] * c
        xy[:, :, idx, :] = torch.stack([rx, ry], -1) + pivot
    x[..., 0:2] = xy
    # linear interpolation of one segment per gated sample: endpoints are frames t0-1 and t0+len
    t0, ln = d.interp_t0.long(), d.interp_len.long()
    tt = torch.arange(half).view(1, half)
    b_idx = torch.arange(B)
    ln_safe = torch.where(ln > 0, ln, torch.ones_like(ln))
    first = x[b_idx, (t0 - 1).clamp(0, half - 1)].unsqueeze(1)
    last = x[b_idx, (t0 + ln_safe).clamp(0, half - 1)].unsqueeze(1)
    mask = (tt >= t0[:, None]) & (tt < (t0 + ln)[:, None]) & (ln > 0)[:, None]
    alpha = ((tt.to(x.dtype) - (t0[:, None].to(x.dtype) - 1.0)) / (ln_safe[:, None] + 1).to(x.dtype)).clamp(0.0, 1.0)
    alpha = alpha[:, :, None, None]
    x = torch.where(mask[:, :, None, None], (1.0 - alpha) * first + alpha * last, x)
    x = x + d.noise.unsqueeze(1).to(x.dtype)
    return x, recompute_edges(x, edge_index)


def central_view(x_full: torch.Tensor, edge_index: torch.Tensor):
    """training.py:517-522: the un-augmented half window starting at (T//2)//2, edges recomputed from nodes."""
    B, T = x_full.shape[0], x_full.shape[1]
    half = T // 2
    start = torch.full((B,), half // 2, dtype=torch.long)
    x = slice_time(x_full, start, half)
    return x, recompute_edges(x, edge_index)


# ----------------------------------------------------------------------------- similarities and losses
def similarity(x, y, kind: str):
    if kind == "cosine":
        return F.cosine_similarity(x.unsqueeze(1), y.unsqueeze(0), dim=2)
    if kind == "dot":
        return x @ y.t()
    if kind in ("euclidean", "edit"):
        d = torch.sqrt(torch.clamp(((x.unsqueeze(1) - y.unsqueeze(0)) ** 2).sum(2), min=0.0))
        return 1.0 / (1.0 + d)
    raise ValueError(kind)


def _off_diag(sim):
    n = sim.shape[0]
    keep = ~torch.eye(n, dtype=torch.bool)
    return sim[keep].reshape(n, n - 1)


def contrastive_loss(z, z_aug, sim_kind="cosine", loss_fn="nce", temperature=0.1, tau=0.1, beta=0.1,
                     elimination_topk=0.1):
    """losses.py:35-249.  Returns (loss, mean positive similarity, mean negative similarity)."""
    n = z.shape[0]
    sim = similarity(z, z_aug, sim_kind)
    pos = torch.diag(sim)
    neg = _off_diag(sim)
    if loss_fn == "nce":
        loss = F.cross_entropy(sim / temperature, torch.arange(n))
        return loss, pos.mean(), neg.mean()
    pos_e = torch.exp(pos / temperature)
    neg_e = torch.exp(neg / temperature)
    if loss_fn == "dcl":
        ng = (-tau * (n - 1) * pos_e + neg_e.sum(-1)) / (1.0 - tau)
        ng = torch.clamp(ng, min=(n - 1) * math.e ** (-1.0 / temperature), max=torch.finfo(z.dtype).max)
        return (-torch.log(pos_e / (pos_e + ng))).mean(), pos.mean(), neg.mean()
    if loss_fn == "hard_dcl":
        rew = torch.ones_like(neg_e) if beta == 0.0 else (beta * neg_e) / neg_e.mean(dim=1, keepdim=True)
        ng = (-tau * (n - 1) * pos_e + (rew * neg_e).sum(-1)) / (1.0 - tau)
        ng = torch.clamp(ng, min=math.e ** (-1.0 / temperature), max=torch.finfo(z.dtype).max)
        return (-torch.log(pos_e / (pos_e + ng))).mean(), pos.mean(), neg.mean()
    if loss_fn == "fc":
        k = max(int(math.ceil(min(elimination_topk, 0.5) * n)), 1)
        srt, _ = torch.sort(neg / temperature, dim=1)
        trimmed = srt[:, : max((n - 1) - k, 0)]
        neg_sum = torch.exp(trimmed).sum(1) if trimmed.numel() else torch.zeros(n, dtype=sim.dtype)
        loss = (-torch.log(pos_e / (pos_e + neg_sum))).mean()
        mneg = trimmed.mean() * temperature if trimmed.numel() else torch.tensor(0.0, dtype=sim.dtype)
        return loss, pos.mean(), mneg
    raise ValueError(loss_fn)


# ----------------------------------------------------------------------------- model + step
def encode(P, x, a, training=True, drop=None):
    """ContrastivePT.forward = the recurrent, TCN or transformer encoder on the half window (models_new.py:2038-2075)."""
    if "encoder.node_tf.embed.weight" in P:
        from . import tfm as otf
        return otf.tfm_encoder(x, a, P, training, drop)
    if "encoder.node_tcn.blocks.0.conv1.weight" in P:
        from . import tcn as ot
        return ot.tcn_encoder(x, a, P, training)
    return ov.encoder(x, a, P)


def contrastive_step(P, x_full, edge_index, draws: AugDraws, sim_kind="cosine", loss_fn="nce", temperature=0.1,
                     tau=0.1, beta=0.1, training=True, distill=None, drop=None):
    """training.py:482-589 without the distillation head.  Returns (total, logs, aux)."""
    x_aug, a_aug = augmented_view(x_full, edge_index, draws)
    x, a = central_view(x_full, edge_index)
    z = encode(P, x, a, training, drop)
    z_aug = encode(P, x_aug, a_aug, training, drop)
    zn, zan = F.normalize(z, dim=1), F.normalize(z_aug, dim=1)
    loss, pos, neg = contrastive_loss(zn, zan, sim_kind, loss_fn, temperature, tau, beta)
    dist = 0.0
    if distill is not None:  # head on the NORMALISED central embeddings (training.py:553-580)
        from .vqvae import distill_term
        dist = distill_term(zn, P["distill_head.fc.weight"], P["distill_head.fc.bias"], **distill)
        loss = loss + dist
    logs = {"total_loss": float(loss), "pos_similarity": float(pos), "neg_similarity": float(neg),
            "distill_loss": float(dist)}
    return loss, logs, {"x": x, "a": a, "x_aug": x_aug, "a_aug": a_aug, "z": z, "z_aug": z_aug}


def contrastive_grads(P, x_full, edge_index, draws, **kw):
    buffers = ("laplacian", "edge_laplacian", "incidence", "running_mean", "running_var", "num_batches_tracked")
    leaves = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and (k.startswith("encoder.") or k.startswith("distill_head.")) and
                                                   k.split(".")[-1] not in buffers)
              for k, v in P.items()}
    loss, logs, aux = contrastive_step(leaves, x_full, edge_index, draws, **kw)
    names = [k for k, v in leaves.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    aux["buffers"] = {k: v.detach() for k, v in leaves.items() if k.split(".")[-1] in buffers[3:]}
    return logs, {k: g for k, g in zip(names, gs)}, aux
