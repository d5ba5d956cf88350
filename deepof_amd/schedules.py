"""Per-iteration loss-weight curves (KL weight, distillation lambda) as precomputed tables.

The curve is the reference's (``Dynamic_weight_manager``, /root/reference/deepof/clustering/losses.py:290-351):
iterations ``0..w`` ramp ``0 -> top`` through a shape function of ``t/w``, an optional plateau of ``h``
iterations stays at ``top``, ``c`` iterations blend linearly ``top -> tail``, afterwards ``tail`` for good
(w, h, c = warm-up / plateau / cool-down epochs x batches per epoch; w is at least one iteration).

Here the whole curve is evaluated once, vectorised, into a float64 table: a step is then a table lookup, and the
table can be handed to the device in one piece when a captured training step wants to index it with its own
iteration counter.
"""
from __future__ import annotations

import numpy as np


def ramp_shape(mode: str, p: np.ndarray) -> np.ndarray:
    """Shape functions on p in [0, 1]: identity, logistic 1/(1+e^{-12(p-1/2)}), and the "tf_sigmoid"
    sigma((2p-1)/max(0.01, p-p^2)) (losses.py:311-324); unknown modes fall back to the identity like the reference."""
    p = np.clip(np.asarray(p, dtype=np.float64), 0.0, 1.0)
    if mode == "sigmoid":
        arg = 12.0 * (p - 0.5)
    elif mode == "tf_sigmoid":
        arg = (2.0 * p - 1.0) / np.maximum(1e-2, p - p * p)
    else:
        return p
    with np.errstate(over="ignore"):
        return 1.0 / (1.0 + np.exp(-arg))


def weight_table(n_batches_per_epoch: int, mode: str, warmup_epochs: float, max_weight: float, at_max_epochs: float,
                 cooldown_epochs: float, end_weight: float) -> np.ndarray:
    """table[t] for t = 0 .. total (table[total] = the value held forever after)."""
    warm = max(1, int(warmup_epochs * n_batches_per_epoch))
    hold = max(0, int(at_max_epochs * n_batches_per_epoch))
    cool = max(0, int(cooldown_epochs * n_batches_per_epoch))
    top, tail = float(max_weight), float(end_weight)
    t = np.arange(warm + hold + cool + 1, dtype=np.float64)
    out = np.empty_like(t)
    ramp = t <= warm
    out[ramp] = top * ramp_shape(mode, t[ramp] / warm)
    if hold > 0:  # the plateau claims iteration `warm` itself (the ramp reaches top there only for monotone shapes)
        out[(t >= warm) & (t < warm + hold)] = top
    late = t > warm if hold == 0 else t >= warm + hold
    if cool > 0:
        frac = (t[late] - (warm + hold)) / cool
        out[late] = (1.0 - frac) * top + frac * tail
    else:
        out[late] = top
    out[-1] = tail
    return out


class WeightSchedule:
    """Cursor over a weight table; ``get_weight()`` / ``step()`` are what the fit loops call once per batch."""

    def __init__(self, n_batches_per_epoch: int, mode: str = "sigmoid", warmup_epochs: int = 15,
                 max_weight: float = 1.0, at_max_epochs: int = 0, cooldown_epochs: int = 15, end_weight: float = 1.0):
        self.table = weight_table(n_batches_per_epoch, mode, warmup_epochs, max_weight, at_max_epochs, cooldown_epochs,
                                  end_weight)
        self.max_weight = float(max_weight)
        self.position = 0

    @classmethod
    def constant(cls, value: float) -> "WeightSchedule":
        """One-entry table: ``value`` from iteration 0 on (no ramp; ``weight_table`` clamps a warm-up to >= 1 step)."""
        self = cls.__new__(cls)
        self.table = np.array([float(value)], dtype=np.float64)
        self.max_weight = float(value)
        self.position = 0
        return self

    def at(self, iteration: int) -> float:
        return float(self.table[min(int(iteration), len(self.table) - 1)])

    def get_weight(self) -> float:
        return self.at(self.position)

    def step(self) -> None:
        self.position += 1
