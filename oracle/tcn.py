"""CPU restatement of the reference's TCN encoder (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows, under /root/reference/deepof/clustering/models_new.py:
  TemporalBlockPT   :376-443   causal pad -> Conv1d(k=4, dilation d, bias) -> BatchNorm1d(eps 1e-3) -> ReLU, twice;
                               residual (1x1 conv when channels differ) ; out = ReLU(y + res) ; skip = y
  TCN1DPT           :446-506   2 stacks x dilations (1,2,4,8); skip-sum of the blocks' post-conv2 activations,
                               final ReLU, last time step (return_sequences=False)
  BatchNorm1dKerasFP32 :508-516  eps 1e-3, momentum 0.01
  TCNEncoderPT      :518-657   TF-style group scramble -> per-node / per-edge TCN -> CensNet(32 -> latent) -> ReLU ->
                               flatten -> x / max(rms, 1) -> clamp +-1e4 -> Linear -> ReLU -> BN -> Linear -> ReLU ->
                               BN -> Linear
Functional over a dict with the reference's state_dict names.  ``training=True`` uses batch statistics and
updates the running buffers of ``P`` in place (momentum 0.1 for the TCN BatchNorms -- the nn.BatchNorm1d
default -- and 0.01 for the head), exactly as module.train() does in the reference.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import vade as ov

DILATIONS = (1, 2, 4, 8, 1, 2, 4, 8)
KERNEL = 4


def _bn(y, P, prefix, training, momentum):
    rm, rv = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    out = F.batch_norm(y, rm, rv, P[prefix + ".weight"], P[prefix + ".bias"], training, momentum, 1e-3)
    if training and (prefix + ".num_batches_tracked") in P:
        P[prefix + ".num_batches_tracked"] += 1
    return out


def temporal_block(x, P, prefix, dilation, training):
    """x (S, C_in, T) -> (out, skip), both (S, C, T)."""
    pad = (KERNEL - 1) * dilation
    y = F.conv1d(F.pad(x, (pad, 0)), P[prefix + ".conv1.weight"], P[prefix + ".conv1.bias"], dilation=dilation)
    y = torch.relu(_bn(y, P, prefix + ".bn1", training, 0.1))
    y = F.conv1d(F.pad(y, (pad, 0)), P[prefix + ".conv2.weight"], P[prefix + ".conv2.bias"], dilation=dilation)
    y = torch.relu(_bn(y, P, prefix + ".bn2", training, 0.1))
    res = x
    if (prefix + ".downsample.weight") in P:
        res = F.conv1d(x, P[prefix + ".downsample.weight"], P[prefix + ".downsample.bias"])
    return torch.relu(y + res), y


def tcn(x, P, prefix, training, return_sequences=False):
    """x (S, T, C_in) -> (S, C) features of the last time step (or (S, T, C))."""
    y = x.transpose(1, 2).float()
    skip_sum = None
    for b, d in enumerate(DILATIONS if (prefix + ".blocks.7.conv1.weight") in P else (8, 4, 2, 1)):
        y, skip = temporal_block(y, P, f"{prefix}.blocks.{b}", d, training)
        skip_sum = skip if skip_sum is None else skip_sum + skip
    out = torch.relu(skip_sum).transpose(1, 2)
    return out if return_sequences else out[:, -1, :]


def censnet_wide(xv, xe, P, prefix="encoder"):
    """CensNetConvPT with 32 input channels per node / edge (censNetConv_pt.py:26-175), then the encoder's extra ReLU."""
    return ov.censnet(xv, xe, P, prefix + ".spatial_gnn_block", prefix)


def tcn_encoder(x, a, P, training, prefix="encoder"):
    """TCNEncoderPT.forward (models_new.py:603-657): x (B,T,N,3), a (B,T,E,1) -> (B, latent)."""
    B, T, N, Fn = x.shape
    E = a.shape[2]
    xn = ov.group_scramble_t(x).reshape(B * N, T, Fn)
    xe = ov.group_scramble_t(a).reshape(B * E, T, a.shape[3])
    hn = tcn(xn, P, prefix + ".node_tcn", training).view(B, N, -1)
    he = tcn(xe, P, prefix + ".edge_tcn", training).view(B, E, -1)
    gn, ge = censnet_wide(hn, he, P, prefix)
    enc = torch.cat([torch.relu(gn).reshape(B, -1), torch.relu(ge).reshape(B, -1)], dim=-1).float()
    rms = enc.pow(2).mean(dim=1, keepdim=True).sqrt()
    h = (enc / rms.clamp(min=1.0)).clamp(min=-1e4, max=1e4)
    h = torch.nan_to_num(h, nan=0.0, posinf=1e4, neginf=-1e4)
    h = torch.relu(F.linear(h, P[prefix + ".head.0.weight"], P[prefix + ".head.0.bias"]))
    h = _bn(h, P, prefix + ".head.2", training, 0.01)
    h = torch.relu(F.linear(h, P[prefix + ".head.3.weight"], P[prefix + ".head.3.bias"]))
    h = _bn(h, P, prefix + ".head.5", training, 0.01)
    return F.linear(h, P[prefix + ".head.6.weight"], P[prefix + ".head.6.bias"])


def tcn_decoder(z, x_flat, P, training, prefix="decoder"):
    """TCNDecoderPT.forward (models_new.py:713-819): z (B,L), x_flat (B,T,3N) -> (loc (B,T,3N), valid (B,T)).
    RMS guard -> Linear -> BN -> Linear -> ReLU -> BN -> Linear -> ReLU -> BN -> repeat over T -> TCN(64 filters,
    dilations 8,4,2,1, return_sequences) -> Linear(64 -> 3N)."""
    B, T, _ = x_flat.shape
    valid = ~torch.all(x_flat == 0.0, dim=-1)
    g = z.float()
    rms = g.pow(2).mean(dim=1, keepdim=True).sqrt()
    g = torch.nan_to_num((g / rms.clamp(min=1.0)).clamp(min=-1e4, max=1e4), nan=0.0, posinf=1e4, neginf=-1e4)
    h = _bn(F.linear(g, P[prefix + ".fc0.weight"], P[prefix + ".fc0.bias"]), P, prefix + ".bn0", training, 0.01)
    h = _bn(torch.relu(F.linear(h, P[prefix + ".fc1.weight"], P[prefix + ".fc1.bias"])), P, prefix + ".bn1", training, 0.01)
    h = _bn(torch.relu(F.linear(h, P[prefix + ".fc2.weight"], P[prefix + ".fc2.bias"])), P, prefix + ".bn2", training, 0.01)
    hidden = tcn(h.unsqueeze(1).repeat(1, T, 1), P, prefix + ".tcn", training, return_sequences=True)
    loc = F.linear(hidden, P[prefix + ".prob_decoder.loc_projection.weight"], P[prefix + ".prob_decoder.loc_projection.bias"])
    return torch.nan_to_num(loc, nan=0.0, posinf=1e6, neginf=-1e6), valid
