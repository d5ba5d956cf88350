"""Deterministic noise streams shared by the reference-side golden generator (tests/golden/make_golden_fit.py) and the
replay through deepof_amd.training (tests/): noise(kind, k, shape) is the k-th draw of stream `kind`."""
import torch

_randn = torch.randn  # (callers may patch torch.randn while they route a reference draw through here)
_KIND_ID = {"eps_train": 1, "mc_train": 2, "mc_val": 3}


def noise(kind: str, k: int, shape) -> torch.Tensor:
    g = torch.Generator().manual_seed(7_000_003 * _KIND_ID[kind] + int(k))
    return _randn(tuple(int(v) for v in shape), generator=g)
