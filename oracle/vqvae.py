"""Oracle (test infrastructure): VQ-VAE restatement, PyTorch-CPU fp32.

* VectorQuantizerPT            /root/reference/deepof/clustering/models_new.py:1330-1423
* VQVAEPT.forward              models_new.py:1575-1635  (two decoder passes: quantised + raw z_e)
* step_vqvae_distill           /root/reference/deepof/clustering/training.py:312-389
  (vq_loss / kmeans_loss enter the total as detached Python floats; no straight-through)
* build_optimizer_generic      /root/reference/deepof/clustering/losses.py:805-814 (Adam, weight_decay 1e-4)

Shares the recurrent encoder / decoder restatement with oracle/vade.py.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import vade as OV

Params = Dict[str, torch.Tensor]


def vq_layer(ze: torch.Tensor, codebook: torch.Tensor, beta: float = 1.0, kmeans_weight: float = 0.0):
    """ze (B,L), codebook (L,K) -> dict(quantized, idx, soft_counts, vq_loss, kmeans_loss, distances)."""
    sim = ze @ codebook
    dist = (ze**2).sum(dim=1, keepdim=True) + (codebook**2).sum(dim=0) - 2 * sim
    idx = torch.argmin(dist, dim=1)
    inv = (1.0 / dist) ** 2
    soft = inv / inv.sum(dim=1, keepdim=True)
    onehot = torch.nn.functional.one_hot(idx, codebook.shape[1]).float()
    quantized = onehot @ codebook.T
    vq_loss = beta * torch.mean((quantized.detach() - ze) ** 2) + torch.mean((quantized - ze.detach()) ** 2)
    km = torch.zeros(())
    if kmeans_weight:
        km = OV.kmeans_gram_loss(ze, kmeans_weight)
    return dict(quantized=quantized, idx=idx, soft_counts=soft, vq_loss=vq_loss, kmeans_loss=km, distances=dist)


def vqvae_forward(P: Params, x: torch.Tensor, a: torch.Tensor, beta: float = 1.0, kmeans_weight: float = 0.0,
                  training: bool = True, drop=None):
    B, T = x.shape[:2]
    x_flat = x.reshape(B, T, -1)
    if "encoder.node_tf.embed.weight" in P:  # transformer family: the two decoder passes draw their own dropout masks
        from . import tfm as otf
        ze = otf.tfm_encoder(x, a, P, training, drop)
        vq = vq_layer(ze, P["vq_layer.codebook"], beta, kmeans_weight)
        loc_q, valid = otf.tfm_decoder(vq["quantized"], x_flat, P, training, drop, site="dec")
        loc_e, _ = otf.tfm_decoder(ze, x_flat, P, training, drop, site="dec2")
        vq.update(ze=ze, loc_q=loc_q, loc_e=loc_e, valid=valid)
        return vq
    if "encoder.node_tcn.blocks.0.conv1.weight" in P:  # TCN family: encoder, then the decoder on q, then on z_e
        from . import tcn as ot
        ze = ot.tcn_encoder(x, a, P, training)
        vq = vq_layer(ze, P["vq_layer.codebook"], beta, kmeans_weight)
        loc_q, valid = ot.tcn_decoder(vq["quantized"], x_flat, P, training)
        loc_e, _ = ot.tcn_decoder(ze, x_flat, P, training)
        vq.update(ze=ze, loc_q=loc_q, loc_e=loc_e, valid=valid)
        return vq
    ze = OV.encoder(x, a, P)
    vq = vq_layer(ze, P["vq_layer.codebook"], beta, kmeans_weight)
    loc_q, valid = OV.decoder(vq["quantized"], x_flat, P)
    loc_e, _ = OV.decoder(ze, x_flat, P)
    vq.update(ze=ze, loc_q=loc_q, loc_e=loc_e, valid=valid)
    return vq


def distill_term(z, W, b, tau_b, lam: float, T: float = 0.5, conf_weight: bool = False, thr: float = 0.6):
    """Generic distillation head term of step_vqvae_distill / step_contrastive_distill (training.py:344-372, 553-580):
    lam * mean_b w_b * soft-CE(W z_b + b, sharpened tau_b)."""
    logits = torch.nn.functional.linear(z, W, b)
    if T > 0.0:
        tau_b = torch.softmax(tau_b.clamp_min(1e-8).log() / T, dim=-1)
    per = -(torch.clamp(tau_b, min=1e-8, max=1.0) * torch.nn.functional.log_softmax(logits, dim=-1)).sum(dim=-1)
    if conf_weight:
        w = ((tau_b.max(dim=1).values - thr) / max(1e-6, 1.0 - thr)).clamp(0.0, 1.0).detach()
        return lam * (w * per).mean()
    return lam * per.mean()


def vqvae_loss(out: dict, x: torch.Tensor, distill=None):
    B, T = x.shape[:2]
    x_flat = x.reshape(B, T, -1).float()
    enc_rec = -(OV.recon_log_prob(out["loc_q"], out["valid"], x_flat)).mean()
    rec = -(OV.recon_log_prob(out["loc_e"], out["valid"], x_flat)).mean()
    const = float(out["vq_loss"]) + float(out["kmeans_loss"])
    dist = torch.zeros(())
    if distill is not None:
        dist = distill_term(out["ze"], **distill)
    total = enc_rec + rec + const + dist
    populated = float(out["soft_counts"].argmax(dim=-1).unique().numel())
    return dict(total_loss=total, enc_rec_loss=enc_rec, reconstruct_loss=rec, vq_loss=float(out["vq_loss"]),
                kmeans_loss=float(out["kmeans_loss"]), number_of_populated_clusters=populated,
                distill_loss=dist)


def vqvae_grads(P: Params, x, a, beta: float = 1.0, kmeans_weight: float = 0.0, distill=None, drop=None):
    """distill: dict(tau_b, lam, T, conf_weight, thr) -- the head weights are P["distill_head.fc.weight" / ".bias"]."""
    keys = OV.trainable_keys(P)
    leaf = {k: (P[k].detach().clone().requires_grad_(True) if k in keys else P[k]) for k in P}
    out = vqvae_forward(leaf, x, a, beta, kmeans_weight, drop=drop)
    if distill is not None:
        distill = dict(distill, W=leaf["distill_head.fc.weight"], b=leaf["distill_head.fc.bias"])
    losses = vqvae_loss(out, x, distill)
    gl = torch.autograd.grad(losses["total_loss"], [leaf[k] for k in keys], allow_unused=True)
    return losses, dict(zip(keys, gl)), out


def vqvae_train_step(P: Params, opt: OV.AdamState, x, a, lr: float, weight_decay: float = 1e-4, clip: float = 0.75,
                     beta: float = 1.0, kmeans_weight: float = 0.0):
    losses, grads, out = vqvae_grads(P, x, a, beta, kmeans_weight)
    grads = {k: (None if g is None else g.clamp(-clip, clip)) for k, g in grads.items()}
    opt.step(P, grads, lr, lr, weight_decay=weight_decay)
    return {k: float(v) for k, v in losses.items()}, grads, out
