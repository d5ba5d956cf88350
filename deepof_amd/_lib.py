"""Loader of the compiled HIP library.  Fails loudly: the product has no CPU / PyTorch fallback."""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must load first: the library shares torch's HIP runtime, streams and pointers)

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdeepof_hip.so")
_lib = None


def load_hip_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C deepof_amd/csrc` (or __graft_entry__.build()); "
                "deepof_amd has no fallback path without its HIP kernels")
        _lib = _capi.bind(ctypes.CDLL(LIB_PATH))
    return _lib
