"""Oracle (test infrastructure): VaDE / recurrent encoder-decoder restatement, PyTorch-CPU fp32.

Functional restatement of the reference modules, operating on a plain ``dict`` of tensors that
uses the reference's ``state_dict`` key names (so reference checkpoints plug straight in):

* recurrent encoder block   /root/reference/deepof/clustering/models_new.py:184-278
* recurrent encoder         models_new.py:37-181
* CensNet layer             /root/reference/deepof/clustering/censNetConv_pt.py:92-136
* GMM latent                models_new.py:1679-1791
* recurrent decoder         models_new.py:281-373 + probabilistic head :677-710
* Gram/SVD "k-means" loss   /root/reference/deepof/clustering/losses.py:257-287
* VaDE loss                 losses.py:567-797
* train step                /root/reference/deepof/clustering/training.py:130-166 (+ step_vade :231-309)

Random draws (reparameterisation eps, the 32-sample MC-KL eps) are explicit arguments so the
HIP path, this oracle and the imported reference can be run on identical noise.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as Fn
from torch.nn.utils.rnn import PackedSequence, pack_padded_sequence, pad_packed_sequence

from .windows import group_scramble_index

Params = Dict[str, torch.Tensor]
LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
def group_scramble_t(x: torch.Tensor) -> torch.Tensor:
    """(B,T,G,F)->(B,G,T,F) scramble, models_new.py:120-138."""
    B, T, G, F = x.shape
    src = torch.from_numpy(group_scramble_index(T, G, F).reshape(-1))
    return x.reshape(B, -1)[:, src].reshape(B, G, T, F)


def _gru_weights(P: Params, prefix: str):
    names = []
    for sfx in ("", "_reverse"):
        names += [f"weight_ih_l0{sfx}", f"weight_hh_l0{sfx}", f"bias_ih_l0{sfx}", f"bias_hh_l0{sfx}"]
    return [P[f"{prefix}.{n}"] for n in names]


def bigru_prefix(x: torch.Tensor, lengths: torch.Tensor, P: Params, prefix: str):
    """Bidirectional GRU over the *first* ``lengths[i]`` steps of each row (packed semantics).

    x (S,T,IN), lengths (S,) int64 CPU.  Returns (out (S,T,2H) zero beyond length,
    h_n (S,2H) = [fwd final, bwd final], zero rows where length == 0).
    models_new.py:238-266 / :343-362.
    """
    S, T, _ = x.shape
    w = _gru_weights(P, prefix)
    H = w[1].shape[1]
    out = torch.zeros(S, T, 2 * H, dtype=x.dtype)
    h_n = torch.zeros(S, 2 * H, dtype=x.dtype)
    valid = torch.where(lengths > 0)[0]
    if valid.numel() == 0:
        return out, h_n
    packed = pack_padded_sequence(x[valid], lengths[valid], batch_first=True, enforce_sorted=False)
    hx = torch.zeros(2, int(packed.batch_sizes[0]), H, dtype=x.dtype)
    data, hid = torch._VF.gru(packed.data, packed.batch_sizes, hx, w, True, 1, 0.0, False, True)
    seq = PackedSequence(data, packed.batch_sizes, packed.sorted_indices, packed.unsorted_indices)
    unpacked, _ = pad_packed_sequence(seq, batch_first=True, total_length=T)
    hid = hid.index_select(1, packed.unsorted_indices)
    out = out.index_put((valid,), unpacked)
    h_n = h_n.index_put((valid,), hid.permute(1, 0, 2).reshape(valid.numel(), 2 * H))
    return out, h_n


def recurrent_block(xg: torch.Tensor, P: Params, prefix: str) -> torch.Tensor:
    """(B,G,T,F) -> (B,G,2L).  models_new.py:217-278."""
    B, G, T, F = xg.shape
    seq = xg.reshape(B * G, T, F).float()
    conv = Fn.relu(Fn.conv1d(seq.permute(0, 2, 1), P[f"{prefix}.conv1d.weight"], padding="same"))
    g1_in = conv.permute(0, 2, 1)  # (S,T,2H)
    mask = g1_in.abs().sum(dim=-1) > 0
    lengths = mask.sum(dim=1).to(torch.int64)
    o1, _ = bigru_prefix(g1_in, lengths, P, f"{prefix}.gru1")
    n1 = Fn.layer_norm(o1, (o1.shape[-1],), P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], 1e-3)
    _, h2 = bigru_prefix(n1, lengths, P, f"{prefix}.gru2")
    n2 = Fn.layer_norm(h2, (h2.shape[-1],), P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], 1e-3)
    out = n2.reshape(B, G, -1)
    w_ih2 = P[f"{prefix}.gru2.weight_hh_l0"]
    internal = w_ih2.shape[1]
    latent = P[f"{prefix}.projection.weight"].shape[0] // 2
    if internal != latent:  # models_new.py:274-275
        out = Fn.linear(out, P[f"{prefix}.projection.weight"], P[f"{prefix}.projection.bias"])
    return out


def censnet(xv: torch.Tensor, xe: torch.Tensor, P: Params, prefix: str, enc_prefix: str):
    """CensNet node/edge co-embedding with ReLU.  censNetConv_pt.py:92-136."""
    lap, elap, inc = P[f"{enc_prefix}.laplacian"], P[f"{enc_prefix}.edge_laplacian"], P[f"{enc_prefix}.incidence"]
    de = (xe @ P[f"{prefix}.edge_weights"]).squeeze(-1)  # (B,E)
    mv = torch.einsum("ne,be,me->bnm", inc, de, inc) * lap
    zv = Fn.relu(mv @ xv @ P[f"{prefix}.node_kernel"] + P[f"{prefix}.node_bias"])
    dv = (xv @ P[f"{prefix}.node_weights"]).squeeze(-1)  # (B,N)
    me = torch.einsum("ne,bn,nf->bef", inc, dv, inc) * elap
    ze = Fn.relu(me @ xe @ P[f"{prefix}.edge_kernel"] + P[f"{prefix}.edge_bias"])
    return zv, ze


def encoder(x: torch.Tensor, a: torch.Tensor, P: Params, prefix: str = "encoder") -> torch.Tensor:
    """x (B,T,N,3), a (B,T,E,1) -> (B,L).  models_new.py:140-181."""
    B = x.shape[0]
    nodes = recurrent_block(group_scramble_t(x), P, f"{prefix}.node_recurrent_block")
    edges = recurrent_block(group_scramble_t(a), P, f"{prefix}.edge_recurrent_block")
    zv, ze = censnet(nodes, edges, P, f"{prefix}.spatial_gnn_block", prefix)
    flat = torch.cat([zv.reshape(B, -1), ze.reshape(B, -1)], dim=-1)
    return Fn.linear(flat, P[f"{prefix}.final_dense.weight"], P[f"{prefix}.final_dense.bias"])


def kmeans_gram_loss(z: torch.Tensor, weight: float) -> torch.Tensor:
    """weight * mean_i sqrt(clamp(sv_i(Z^T Z / B), 1e-9)), fp64 svdvals.  losses.py:257-287."""
    gram = (z.T @ z) / float(z.shape[0])
    sv = torch.linalg.svdvals(gram.to(torch.float64))
    return weight * torch.sqrt(torch.clamp(sv, min=1e-9)).mean()


def gmm_posterior(z: torch.Tensor, P: Params, prefix: str = "latent_space") -> torch.Tensor:
    """softmax_c(log(pi_c + 1e-9) + sum_d log N(z_d; m_cd, max(exp(l_cd/2),1e-3))).  models_new.py:1745-1759."""
    std = torch.exp(0.5 * P[f"{prefix}.gmm_log_vars"]).clamp(min=1e-3)
    diff = (z.unsqueeze(1) - P[f"{prefix}.gmm_means"].unsqueeze(0)) / std.unsqueeze(0)
    logp = (-0.5 * diff**2 - torch.log(std).unsqueeze(0) - 0.5 * LOG_2PI).sum(dim=-1)
    return torch.softmax(torch.log(P[f"{prefix}.prior"] + 1e-9) + logp, dim=-1)


def gmm_latent(h: torch.Tensor, P: Params, training: bool, eps: Optional[torch.Tensor],
               kmeans_weight: float, prefix: str = "latent_space"):
    """models_new.py:1761-1791.  Returns dict(z, q, z_mean, z_log_var, kmeans, n_populated, confidence)."""
    z_mean = Fn.linear(h, P[f"{prefix}.encoder_mean.weight"], P[f"{prefix}.encoder_mean.bias"])
    z_log_var = Fn.softplus(Fn.linear(h, P[f"{prefix}.encoder_log_var.weight"], P[f"{prefix}.encoder_log_var.bias"]))
    if training:
        assert eps is not None
        z = z_mean + torch.exp(0.5 * z_log_var) * eps
    else:
        z = z_mean
    q = gmm_posterior(z, P, prefix)
    conf, hard = q.max(dim=1)
    km = torch.zeros((), dtype=torch.float32)
    if kmeans_weight > 0:
        km = kmeans_gram_loss(z, kmeans_weight)
    return dict(z=z, q=q, z_mean=z_mean, z_log_var=z_log_var, kmeans=km,
                n_populated=float(torch.unique(hard).numel()), confidence=conf.mean())


def decoder(z: torch.Tensor, x_flat: torch.Tensor, P: Params, prefix: str = "decoder"):
    """z (B,L), x_flat (B,T,3N) -> (loc (B,T,3N), valid (B,T) bool).  models_new.py:326-373, 686-710."""
    B, T, _ = x_flat.shape
    valid = ~torch.all(x_flat == 0.0, dim=2)
    lengths = valid.sum(dim=1).to(torch.int64)
    gen = z.unsqueeze(1).expand(-1, T, -1)
    o1, _ = bigru_prefix(gen, lengths, P, f"{prefix}.gru1")
    n1 = Fn.layer_norm(o1, (o1.shape[-1],), P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], 1e-3)
    o2, _ = bigru_prefix(n1, lengths, P, f"{prefix}.gru2")
    n2 = Fn.layer_norm(o2, (o2.shape[-1],), P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], 1e-3)
    conv = Fn.relu(Fn.conv1d(n2.permute(0, 2, 1), P[f"{prefix}.conv1d.weight"], padding="same")).permute(0, 2, 1)
    n3 = Fn.layer_norm(conv, (conv.shape[-1],), P[f"{prefix}.norm3.weight"], P[f"{prefix}.norm3.bias"], 1e-3)
    loc = Fn.linear(n3, P[f"{prefix}.prob_decoder.loc_projection.weight"], P[f"{prefix}.prob_decoder.loc_projection.bias"])
    loc = torch.nan_to_num(loc, nan=0.0, posinf=1e6, neginf=-1e6)
    return loc, valid


def recon_log_prob(loc: torch.Tensor, valid: torch.Tensor, x_flat: torch.Tensor) -> torch.Tensor:
    """log_prob of Independent(Normal(loc,1)) scaled by the validity mask: NaN on masked frames
    (scale 0), -0.5*||x-loc||^2 - D/2*log(2pi) elsewhere.  models_new.py:701-708."""
    D = x_flat.shape[-1]
    lp = -0.5 * ((x_flat - loc) ** 2).sum(dim=-1) - 0.5 * D * LOG_2PI
    return torch.where(valid, lp, torch.full_like(lp, float("nan")))


# --------------------------------------------------------------------------------------
# full model + loss
# --------------------------------------------------------------------------------------
def vade_forward(P: Params, x: torch.Tensor, a: torch.Tensor, training: bool,
                 eps: Optional[torch.Tensor] = None, kmeans_weight: float = 1.0, drop=None):
    """VaDEPT.forward, models_new.py:1841-1891.  Returns dict with loc/valid/z/q/z_mean/z_log_var/kmeans.
    drop: oracle.tfm.DropoutTape (transformer family, train mode) or None."""
    B, T = x.shape[:2]
    if "encoder.node_tf.embed.weight" in P:  # transformer family (models_new.py:832-1327)
        from . import tfm as otf
        h = otf.tfm_encoder(x, a, P, training, drop)
        lat = gmm_latent(h, P, training, eps, kmeans_weight)
        loc, valid = otf.tfm_decoder(lat["z"], x.reshape(B, T, -1), P, training, drop)
    elif "encoder.node_tcn.blocks.0.conv1.weight" in P:  # TCN family (models_new.py:518-819): BatchNorm follows `training`
        from . import tcn as ot
        h = ot.tcn_encoder(x, a, P, training)
        lat = gmm_latent(h, P, training, eps, kmeans_weight)
        loc, valid = ot.tcn_decoder(lat["z"], x.reshape(B, T, -1), P, training)
    else:
        h = encoder(x, a, P)
        lat = gmm_latent(h, P, training, eps, kmeans_weight)
        loc, valid = decoder(lat["z"], x.reshape(B, T, -1), P)
    lat.update(loc=loc, valid=valid, enc=h)
    return lat


class VadeLossCfg:
    """Active VadeLoss hyper-parameters (losses.py:383-457), one mode at a time."""

    def __init__(self, n_components: int, pretrain: bool, **kw):
        self.n_components = n_components
        self.pretrain = pretrain
        self.l1_activity_weight = kw.get("l1_activity_weight", 0.1)
        self.gmm_logvar_clamp = kw.get("gmm_logvar_clamp", (-8.0, 8.0))
        self.tf_cluster_weight = kw.get("tf_cluster_weight", 0.0)
        self.reg_cat_clusters = kw.get("reg_cat_clusters", 0.0)
        self.temporal_cohesion_weight = kw.get("temporal_cohesion_weight", 0.0)
        self.reg_scatter_weight = kw.get("reg_scatter_weight", 0.0)
        self.reg_scatter_beta = kw.get("reg_scatter_beta", 1.0)
        # mode-dependent (defaults = reference pretrain / main defaults, training.py:640-660)
        self.kmeans_loss_weight = kw.get("kmeans_loss_weight", 1.0 if pretrain else 0.0)
        self.repel_weight = kw.get("repel_weight", 0.5 if pretrain else 0.0)
        self.repel_length_scale = kw.get("repel_length_scale", 0.5 if pretrain else 1.0)
        self.nonempty_weight = kw.get("nonempty_weight", 0.02)
        floor_pct = kw.get("nonempty_floor_percent", 0.05)
        self.nonempty_floor = max(1e-4, floor_pct / n_components)
        self.nonempty_p = int(kw.get("nonempty_p", 2))
        # distillation
        self.lambda_distill = kw.get("lambda_distill", 0.0)
        self.distill_sharpen_T = kw.get("distill_sharpen_T", 0.5)
        self.distill_conf_weight = kw.get("distill_conf_weight", False)
        self.distill_conf_thresh = kw.get("distill_conf_thresh", 0.3)
        self.class_weight = kw.get("class_weight", None)
        self.teacher_marginal = kw.get("teacher_marginal", None)


def _log_normal_diag(x, mean, log_var):
    return -0.5 * torch.sum(LOG_2PI + log_var + (x - mean) ** 2 * torch.exp(-log_var), dim=-1)


def mc_kl(z_mean, z_log_var, means, log_vars, prior, eps_mc, clamp=(-8.0, 8.0)):
    """losses.py:525-545 with the (S,B,D) noise made explicit."""
    z_log_var = torch.clamp(z_log_var, min=-4.0, max=4.0)
    zs = z_mean.unsqueeze(0) + eps_mc * torch.exp(0.5 * z_log_var).unsqueeze(0)
    log_q = _log_normal_diag(zs, z_mean.unsqueeze(0), z_log_var.unsqueeze(0))
    lv = torch.clamp(log_vars, min=clamp[0], max=clamp[1])
    log_pzc = _log_normal_diag(zs.unsqueeze(2), means.view(1, 1, *means.shape), lv.view(1, 1, *lv.shape))
    log_p = torch.logsumexp(torch.log(torch.clamp(prior, min=1e-8)).view(1, 1, -1) + log_pzc, dim=-1)
    return torch.clamp((log_q - log_p).mean(), min=0.0)


def vade_loss(out: dict, x: torch.Tensor, P: Params, cfg: VadeLossCfg, klw: float,
              eps_mc: Optional[torch.Tensor] = None, tau_batch: Optional[torch.Tensor] = None):
    """VadeLoss.forward, losses.py:567-797.  ``tau_batch`` = tau_star[batch_indices] (or None)."""
    B, T = x.shape[:2]
    x_flat = x.reshape(B, T, -1).float()
    z, z_mean, z_log_var = out["z"], out["z_mean"], out["z_log_var"]
    recon = -(recon_log_prob(out["loc"], out["valid"], x_flat)).mean()
    q = out["q"].clamp_min(1e-8)
    q = q / q.sum(dim=-1, keepdim=True)
    activity = cfg.l1_activity_weight * z_log_var.abs().sum(dim=-1).mean()
    zlv = z_log_var.clamp(min=-4.0, max=2.0)
    means, log_vars, prior = P["latent_space.gmm_means"], P["latent_space.gmm_log_vars"], P["latent_space.prior"]
    zero = torch.zeros((), dtype=recon.dtype)
    if cfg.pretrain:
        kl = klw * (0.5 * (z_mean**2 + zlv.exp() - 1.0 - zlv).sum(dim=-1) / zlv.shape[-1]).mean()
    else:
        kl = klw * mc_kl(z_mean, zlv, means, log_vars, prior, eps_mc, cfg.gmm_logvar_clamp)
    kmeans = (cfg.kmeans_loss_weight * out["kmeans"]).to(recon.dtype)
    repel = zero
    if cfg.repel_weight > 0.0:
        qf = q.detach()
        pi_b = qf.sum(dim=0).clamp_min(1e-8)
        cen = (qf.t() @ z) / pi_b.unsqueeze(1)
        d2 = ((cen.unsqueeze(1) - cen.unsqueeze(0)) ** 2).sum(dim=-1)
        km = torch.exp(-d2 / max(1e-9, 2.0 * cfg.repel_length_scale**2))
        km = km - torch.diag(torch.diag(km))
        C = cen.shape[0]
        repel = cfg.repel_weight * (km.sum() / float(max(1, C * C - C)))
    nonempty = zero
    if cfg.nonempty_weight > 0.0:
        q_marg = q.mean(dim=0)
        floor_c = torch.full_like(q_marg, float(cfg.nonempty_floor))
        if cfg.teacher_marginal is not None:
            floor_c = torch.maximum(floor_c, 0.9 * cfg.teacher_marginal)
        nonempty = cfg.nonempty_weight * (floor_c - q_marg).clamp_min(0.0).pow(cfg.nonempty_p).sum()
    tf_cluster = prior_loss = cat = temporal = scatter = zero
    if not cfg.pretrain:
        lv = torch.clamp(log_vars, min=cfg.gmm_logvar_clamp[0], max=cfg.gmm_logvar_clamp[1])
        std = torch.exp(0.5 * lv).clamp(min=1e-3)
        dz = (z.unsqueeze(1) - means.unsqueeze(0)) / std.unsqueeze(0)
        logp = (-0.5 * dz**2 - torch.log(std).unsqueeze(0) - 0.5 * LOG_2PI).sum(dim=-1)
        tf_cluster = -(q * torch.softmax(logp, dim=-1)).sum(dim=-1).mean() * cfg.tf_cluster_weight
        prior_loss = -(q * math.log(1.0 / max(1, cfg.n_components))).sum(dim=-1).mean()
        if cfg.reg_cat_clusters > 0:
            mf = q.mean(dim=0)
            uni = torch.full_like(mf, 1.0 / mf.numel())
            # KLDivLoss(batchmean) on a 1-D input divides by input.size(0) == C (losses.py:354-359)
            cat = cfg.reg_cat_clusters * (uni * (uni.log() - torch.log(mf + 1e-9))).sum() / mf.numel()
        if cfg.temporal_cohesion_weight > 0 and B > 1:
            temporal = cfg.temporal_cohesion_weight * (q[1:] - q[:-1]).abs().sum(dim=-1).mean()
        if cfg.reg_scatter_weight > 0:
            pi_b = q.sum(dim=0).clamp_min(1e-8)
            mu = (q.t() @ z_mean) / pi_b.unsqueeze(1)
            diff = z_mean.unsqueeze(1) - mu.unsqueeze(0)
            scat = (q.unsqueeze(-1) * diff.pow(2)).sum(dim=0) / pi_b.unsqueeze(1)
            w = ((pi_b / pi_b.mean()).pow(-cfg.reg_scatter_beta)).unsqueeze(1)
            scatter = cfg.reg_scatter_weight * (w * scat).mean()
    distill = zero
    if cfg.lambda_distill > 0.0 and tau_batch is not None:
        tb = tau_batch
        if cfg.distill_sharpen_T is not None and cfg.distill_sharpen_T > 0.0:
            tb = torch.softmax(tb.clamp_min(1e-8).log() / float(cfg.distill_sharpen_T), dim=-1)
        ce = -(tb * q.clamp_min(1e-8).log()).sum(dim=-1)
        w_conf = None
        if cfg.distill_conf_weight:
            conf = tb.max(dim=1).values
            thr = float(cfg.distill_conf_thresh)
            w_conf = ((conf - thr) / max(1e-6, 1.0 - thr)).clamp(0.0, 1.0).detach()
        w_total = w_conf
        if cfg.class_weight is not None:
            w_class = tb @ cfg.class_weight
            w_class = (w_class / w_class.mean().clamp_min(1e-8)).detach()
            w_total = w_class if w_conf is None else w_class * w_conf
        distill = cfg.lambda_distill * ((w_total * ce).mean() if w_total is not None else ce.mean())
    total = (recon + kl + cat + temporal + nonempty + tf_cluster + prior_loss + kmeans
             + activity + scatter + repel + distill)
    return dict(total_loss=total, reconstruct_loss=recon, kl_div=kl, cat_clust_loss=cat,
                kmeans_loss=kmeans, activity_l1=activity, prior_loss=prior_loss, distill_loss=distill,
                tf_clust_loss=tf_cluster, nonempty_loss=nonempty, temporal_loss=temporal,
                scatter_loss=scatter, repel_loss=repel)


# --------------------------------------------------------------------------------------
# train step (autograd + clip_grad_value_ + Adam), training.py:159-166, losses.py:817-833
# --------------------------------------------------------------------------------------
GMM_KEYS = ("latent_space.gmm_means", "latent_space.gmm_log_vars")
BUFFER_SUFFIXES = ("laplacian", "edge_laplacian", "incidence", "prior", "pretrain", "running_mean", "running_var",
                   "num_batches_tracked")


def trainable_keys(P: Params):
    return [k for k in P if k.split(".")[-1] not in BUFFER_SUFFIXES]


class AdamState:
    """torch.optim.Adam(betas=(0.9,0.999), eps=1e-8) on a param dict; grads that are None are skipped."""

    def __init__(self):
        self.t: Dict[str, int] = {}
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}

    def step(self, P: Params, grads: Dict[str, Optional[torch.Tensor]], lr_base: float, lr_gmm: float,
             weight_decay: float = 0.0):
        b1, b2, eps = 0.9, 0.999, 1e-8
        for k, g in grads.items():
            if g is None:
                continue
            if weight_decay:
                g = g + weight_decay * P[k]
            t = self.t.get(k, 0) + 1
            self.t[k] = t
            m = self.m.get(k, torch.zeros_like(g)) * b1 + (1 - b1) * g
            v = self.v.get(k, torch.zeros_like(g)) * b2 + (1 - b2) * g * g
            self.m[k], self.v[k] = m, v
            lr = lr_gmm if k in GMM_KEYS else lr_base
            step_size = lr / (1 - b1**t)
            denom = v.sqrt() / math.sqrt(1 - b2**t) + eps
            P[k] = (P[k] - step_size * m / denom).detach()


def vade_grads(P: Params, x, a, cfg: VadeLossCfg, klw: float, eps, eps_mc=None, tau_batch=None,
               kmeans_weight: float = 1.0, drop=None):
    """Forward + loss + autograd.  Returns (loss dict, grads dict (None for unused params), out)."""
    keys = trainable_keys(P)
    leaf = {k: (P[k].detach().clone().requires_grad_(True) if k in keys else P[k]) for k in P}
    out = vade_forward(leaf, x, a, training=True, eps=eps, kmeans_weight=kmeans_weight, drop=drop)
    losses = vade_loss(out, x, leaf, cfg, klw, eps_mc, tau_batch)
    gl = torch.autograd.grad(losses["total_loss"], [leaf[k] for k in keys], allow_unused=True)
    return losses, dict(zip(keys, gl)), out


def vade_train_step(P: Params, opt: AdamState, x, a, cfg: VadeLossCfg, klw: float, lr_base: float,
                    lr_gmm: float, eps, eps_mc=None, tau_batch=None, clip: float = 0.75,
                    kmeans_weight: float = 1.0):
    losses, grads, out = vade_grads(P, x, a, cfg, klw, eps, eps_mc, tau_batch, kmeans_weight)
    grads = {k: (None if g is None else g.clamp(-clip, clip)) for k, g in grads.items()}
    opt.step(P, grads, lr_base, lr_gmm)
    return {k: float(v) for k, v in losses.items()}, grads, out
