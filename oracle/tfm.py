"""CPU restatement of the reference's transformer encoder / decoder (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Follows, under /root/reference/deepof/clustering/models_new.py:
  sinusoidal_positional_encoding :832-840
  MultiHeadAttentionPT           :843-890   q/k/v/out projections without bias, scaled dot-product attention with a
                                            key-padding mask (-inf on padded keys), dropout on the attention weights
  TransformerEncoderLayerPT      :893-919   post-norm: x = LN(x + drop(mha(x))); x = LN(x + drop(ffn(x))), eps 1e-6
  TransformerCorePT              :922-982   Linear embed -> ReLU -> * sqrt(key_dim) -> + PE -> dropout -> layers ->
                                            LAST time step
  TFMEncoderPT                   :985-1164  TF-style group scramble -> per-node / per-edge transformer ->
                                            CensNet(key_dim -> latent) -> ReLU -> flatten -> x / max(rms, 1) -> clamp ->
                                            Linear -> ReLU -> BN -> Linear -> ReLU -> BN -> Linear; in train mode the
                                            batch of outputs is standardised (unbiased std clamped at 0.1, :1160-1162)
  TFMDecoderPT                   :1167-1267 latent-expand MLP (GELU) -> repeat over T -> + PE -> causal pre-norm
                                            layers -> Linear(4L -> 3N) -> ProbabilisticDecoderPT(3N -> 3N)
  CausalSelfAttentionLayer       :1270-1327 x = x + drop(out(attn(LN(x)))); x = x + drop(ffn(LN(x))), GELU + dropout
                                            inside the ffn

Dropout is the only random element; it is explicit here: ``drop`` is a :class:`DropoutTape` that hands out the
keep-masks (already divided by 1 - p) in the order the reference's forward draws them, so that the HIP path, this
oracle and the imported reference run on identical masks.  ``drop=None`` means no dropout (eval mode).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import vade as ov
from .tcn import _bn

ENC_LAYERS, ENC_HEADS, ENC_DFF, ENC_DROP = 2, 4, 128, 0.1
DEC_LAYERS, DEC_HEADS, DEC_DFF, DEC_DROP = 2, 8, 128, 0.2


class DropoutTape:
    """Keep-masks in draw order.  ``masks`` = list of 0/1 arrays (reference tensor shapes); ``named`` collects
    (site name, mask, p) as they are consumed."""

    def __init__(self, masks: List[torch.Tensor]):
        self.masks = list(masks)
        self.pos = 0
        self.named: List[tuple] = []

    def take(self, site: str, shape, p: float) -> torch.Tensor:
        m = self.masks[self.pos]
        self.pos += 1
        assert tuple(m.shape) == tuple(shape), (site, tuple(m.shape), tuple(shape))
        self.named.append((site, m, p))
        return m.to(torch.float32) / (1.0 - p)


def _drop(x, drop: Optional[DropoutTape], site: str, p: float):
    if drop is None or p == 0.0:
        return x
    return x * drop.take(site, x.shape, p)


def positional_encoding(T: int, d: int) -> torch.Tensor:
    """models_new.py:832-840 -> (T, d)."""
    pe = torch.zeros(T, d)
    pos = torch.arange(0, T, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)[:, : pe[:, 1::2].shape[1]]
    return pe


def attention(q, k, v, add_mask, drop, site, p):
    """softmax(q k^T / sqrt(dh) + add_mask) -> dropout -> @ v.   q, k, v (S, H, T, dh)."""
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    if add_mask is not None:
        s = s + add_mask
    w = _drop(torch.softmax(s, dim=-1), drop, site, p)
    return w @ v


def encoder_layer(x, pad, P, prefix, heads, drop, site, p):
    """TransformerEncoderLayerPT.forward (post-norm).  x (S,T,D), pad (S,T) bool True = padded key."""
    S, T, D = x.shape
    dh = D // heads
    split = lambda w: F.linear(x, w).view(S, T, heads, dh).transpose(1, 2)
    q, k, v = (split(P[f"{prefix}.mha.{n}_proj.weight"]) for n in ("q", "k", "v"))
    add = torch.zeros(S, 1, 1, T).masked_fill(pad.view(S, 1, 1, T), float("-inf"))
    att = attention(q, k, v, add, drop, site + ".attn", p).transpose(1, 2).reshape(S, T, D)
    att = F.linear(att, P[f"{prefix}.mha.out_proj.weight"])
    x = F.layer_norm(x + _drop(att, drop, site + ".drop1", p), (D,), P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], 1e-6)
    ff = F.linear(torch.relu(F.linear(x, P[f"{prefix}.ffn.0.weight"], P[f"{prefix}.ffn.0.bias"])),
                  P[f"{prefix}.ffn.2.weight"], P[f"{prefix}.ffn.2.bias"])
    return F.layer_norm(x + _drop(ff, drop, site + ".drop2", p), (D,), P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], 1e-6)


def transformer_core(seq, P, prefix, drop, site, heads=ENC_HEADS, p=ENC_DROP):
    """TransformerCorePT.forward: seq (S,T,F) -> (S,D) (last time step)."""
    S, T, _ = seq.shape
    D = P[f"{prefix}.embed.weight"].shape[0]
    pad = torch.all(seq == 0.0, dim=-1)
    y = torch.relu(F.linear(seq, P[f"{prefix}.embed.weight"], P[f"{prefix}.embed.bias"])) * (D ** 0.5)
    y = _drop(y + positional_encoding(T, D), drop, site + ".embed", p)
    l = 0
    while f"{prefix}.layers.{l}.norm1.weight" in P:
        y = encoder_layer(y, pad, P, f"{prefix}.layers.{l}", heads, drop, f"{site}.l{l}", p)
        l += 1
    return y[:, -1, :]


def tfm_encoder(x, a, P, training, drop: Optional[DropoutTape] = None, prefix="encoder"):
    """TFMEncoderPT.forward (models_new.py:1090-1164): x (B,T,N,3), a (B,T,E,1) -> (B, latent)."""
    B, T, N, Fn_ = x.shape
    E = a.shape[2]
    drop = drop if training else None
    xn = ov.group_scramble_t(x).reshape(B * N, T, Fn_)
    xe = ov.group_scramble_t(a).reshape(B * E, T, a.shape[3])
    hn = transformer_core(xn, P, prefix + ".node_tf", drop, "enc.node").view(B, N, -1)
    he = transformer_core(xe, P, prefix + ".edge_tf", drop, "enc.edge").view(B, E, -1)
    gn, ge = ov.censnet(hn, he, P, prefix + ".spatial_gnn_block", prefix)
    enc = torch.cat([torch.relu(gn).reshape(B, -1), torch.relu(ge).reshape(B, -1)], dim=-1).float()
    rms = enc.pow(2).mean(dim=1, keepdim=True).sqrt()
    h = (enc / rms.clamp(min=1.0)).clamp(min=-1e4, max=1e4)
    h = torch.nan_to_num(h, nan=0.0, posinf=1e4, neginf=-1e4)
    h = torch.relu(F.linear(h, P[prefix + ".head.0.weight"], P[prefix + ".head.0.bias"]))
    h = _bn(h, P, prefix + ".head.2", training, 0.01)
    h = torch.relu(F.linear(h, P[prefix + ".head.3.weight"], P[prefix + ".head.3.bias"]))
    h = _bn(h, P, prefix + ".head.5", training, 0.01)
    out = F.linear(h, P[prefix + ".head.6.weight"], P[prefix + ".head.6.bias"])
    if training and B > 1:  # :1160-1162
        out = (out - out.mean(dim=0, keepdim=True)) / out.std(dim=0, keepdim=True).clamp(min=0.1)
    return out


def causal_layer(x, P, prefix, heads, drop, site, p):
    """CausalSelfAttentionLayer.forward (pre-norm).  x (B,T,D)."""
    B, T, D = x.shape
    dh = D // heads
    xn = F.layer_norm(x, (D,), P[f"{prefix}.norm1.weight"], P[f"{prefix}.norm1.bias"], 1e-6)
    split = lambda w: F.linear(xn, w).view(B, T, heads, dh).transpose(1, 2)
    q, k, v = (split(P[f"{prefix}.{n}_proj.weight"]) for n in ("q", "k", "v"))
    causal = torch.full((T, T), float("-inf")).triu(1)
    att = attention(q, k, v, causal, drop, site + ".attn", p).transpose(1, 2).reshape(B, T, D)
    x = x + _drop(F.linear(att, P[f"{prefix}.out_proj.weight"]), drop, site + ".drop1", p)
    xn = F.layer_norm(x, (D,), P[f"{prefix}.norm2.weight"], P[f"{prefix}.norm2.bias"], 1e-6)
    ff = _drop(F.gelu(F.linear(xn, P[f"{prefix}.ffn.0.weight"], P[f"{prefix}.ffn.0.bias"])), drop, site + ".ffn", p)
    ff = F.linear(ff, P[f"{prefix}.ffn.3.weight"], P[f"{prefix}.ffn.3.bias"])
    return x + _drop(ff, drop, site + ".drop2", p)


def tfm_decoder(z, x_flat, P, training, drop: Optional[DropoutTape] = None, prefix="decoder", site="dec"):
    """TFMDecoderPT.forward (models_new.py:1233-1267): z (B,L), x_flat (B,T,3N) -> (loc (B,T,3N), valid (B,T))."""
    B, T, _ = x_flat.shape
    valid = ~torch.all(x_flat == 0.0, dim=-1)
    g = z
    for i in (0, 2, 4):
        g = F.gelu(F.linear(g, P[f"{prefix}.latent_expand.{i}.weight"], P[f"{prefix}.latent_expand.{i}.bias"]))
    D = g.shape[1]
    h = g.unsqueeze(1).expand(-1, T, -1) + positional_encoding(T, D)
    l = 0
    while f"{prefix}.layers.{l}.norm1.weight" in P:
        h = causal_layer(h, P, f"{prefix}.layers.{l}", DEC_HEADS, drop if training else None, f"{site}.l{l}", DEC_DROP)
        l += 1
    h = F.linear(h, P[f"{prefix}.output_proj.weight"], P[f"{prefix}.output_proj.bias"])
    loc = F.linear(h, P[f"{prefix}.prob_decoder.loc_projection.weight"], P[f"{prefix}.prob_decoder.loc_projection.bias"])
    return torch.nan_to_num(loc, nan=0.0, posinf=1e6, neginf=-1e6), valid


def is_tfm(P: Dict[str, torch.Tensor]) -> bool:
    return "encoder.node_tf.embed.weight" in P
