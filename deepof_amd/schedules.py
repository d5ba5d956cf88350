"""Iteration-based loss-weight schedules (KL weight, distillation lambda).

Same curve as the reference's Dynamic_weight_manager
(/root/reference/deepof/clustering/losses.py:290-351): warm-up 0 -> max with a shape function,
optional hold at max, linear cool-down max -> end, then `end` forever.  Shapes: linear,
logistic 1/(1+e^{-12(p-1/2)}), and "tf_sigmoid" sigma((2p-1)/max(0.01, p-p^2)).
"""
from __future__ import annotations

import math


def _shape(mode: str, p: float) -> float:
    p = min(1.0, max(0.0, float(p)))
    if mode == "sigmoid":
        return 1.0 / (1.0 + math.exp(-12.0 * (p - 0.5)))
    if mode == "tf_sigmoid":
        return 1.0 / (1.0 + math.exp(-(2.0 * p - 1.0) / max(1e-2, p - p * p)))
    return p


class WeightSchedule:
    def __init__(self, n_batches_per_epoch: int, mode: str = "sigmoid", warmup_epochs: int = 15,
                 max_weight: float = 1.0, at_max_epochs: int = 0, cooldown_epochs: int = 15, end_weight: float = 1.0):
        self.mode = mode
        self.warmup_iters = max(1, warmup_epochs * n_batches_per_epoch)
        self.at_max_iters = max(0, at_max_epochs * n_batches_per_epoch)
        self.cooldown_iters = max(0, cooldown_epochs * n_batches_per_epoch)
        self.total_iters = self.warmup_iters + self.at_max_iters + self.cooldown_iters
        self.current_iteration = 0
        self.max_weight = float(max_weight)
        self.end_weight = float(end_weight)

    def get_weight(self) -> float:
        t = self.current_iteration
        if t >= self.total_iters:
            return self.end_weight
        hold_end = self.warmup_iters + self.at_max_iters
        if self.at_max_iters > 0 and self.warmup_iters <= t < hold_end:
            return self.max_weight
        if t <= self.warmup_iters:
            return self.max_weight * _shape(self.mode, t / self.warmup_iters)
        if self.cooldown_iters <= 0:
            return self.max_weight
        pc = (t - hold_end) / self.cooldown_iters
        return (1.0 - pc) * self.max_weight + pc * self.end_weight

    def step(self):
        self.current_iteration += 1
