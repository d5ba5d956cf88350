"""The data-parallel exchange through the C ABI (``dof_comm_*`` / ``dof_flat_allreduce``, include/deepof_hip.h).

The reference wraps its models in ``DistributedDataParallel`` (/root/reference/deepof/clustering/training.py:1087-1096,
1321-1330, 1567-1576).  Here the gradient is one flat buffer, so the exchange is one RCCL all-reduce enqueued on the
step's own stream.  ``NativeComm`` is the host side of that: rank 0 draws the RCCL unique id, the 128 bytes travel
over whatever channel the launcher already has (a ``torch.distributed`` group of any backend, or a callable supplied by
the caller), every rank creates its communicator on its device.  ``training._dp_step`` uses it when
the process group runs on RCCL (the default; ``DOF_DP_NATIVE=0`` keeps ``torch.distributed.all_reduce``).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi


class NativeComm:
    """One RCCL communicator of this process's device, owned by the C library."""

    def __init__(self, lib, rank: int, world: int, unique_id: bytes):
        if len(unique_id) != _capi.COMM_ID_BYTES:
            raise ValueError(f"unique id must be {_capi.COMM_ID_BYTES} bytes")
        self.lib, self.rank, self.world = lib, int(rank), int(world)
        handle = C.c_void_p()
        buf = C.create_string_buffer(unique_id, _capi.COMM_ID_BYTES)
        _capi.check(lib, lib.dof_comm_create(buf, self.rank, self.world, C.byref(handle)), "dof_comm_create")
        self._h = handle

    @staticmethod
    def unique_id(lib) -> bytes:
        buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
        _capi.check(lib, lib.dof_comm_unique_id(buf), "dof_comm_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls, lib, dist=None):
        """Rank 0's unique id broadcast over the initialised ``torch.distributed`` default group (any backend)."""
        if dist is None:
            import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        # Creation is collective, so it must fail on every rank or on none: rank 0 ALWAYS broadcasts (ok, id or error text)
        # -- a rank 0 that raised before the broadcast would leave the others blocked in it -- and after ncclCommInitRank
        # the ranks agree on a success flag before anyone uses (or gives up on) the communicator.
        box = [None]
        if rank == 0:
            try:
                box[0] = (True, cls.unique_id(lib))
            except Exception as exc:
                box[0] = (False, f"{type(exc).__name__}: {exc}")
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        ok, payload = box[0]
        if not ok:
            raise RuntimeError(f"rank 0 could not draw the RCCL unique id ({payload})")
        comm, err = None, ""
        try:
            comm = cls(lib, rank, world, payload)
        except Exception as exc:
            err = f"{type(exc).__name__}: {exc}"
        if world > 1:
            flags = [None] * world
            dist.all_gather_object(flags, err)
            bad = [(r, e) for r, e in enumerate(flags) if e]
        else:
            bad = [(rank, err)] if err else []
        if bad:
            if comm is not None:
                comm.abort()
            raise RuntimeError("RCCL communicator creation failed on rank(s) " +
                               "; ".join(f"{r}: {e}" for r, e in bad))
        return comm

    def _stream(self, stream):
        return torch.cuda.current_stream().cuda_stream if stream is None else stream

    def all_reduce_(self, t: torch.Tensor, stream=None) -> torch.Tensor:
        """In-place SUM over the ranks, enqueued on ``stream`` (default: torch's current stream); no host wait."""
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("all_reduce_: a contiguous float32 device tensor is required")
        _capi.check(self.lib, self.lib.dof_flat_allreduce(self._h, t.data_ptr(), t.numel(), self._stream(stream)),
                    "dof_flat_allreduce")
        return t

    def broadcast_(self, t: torch.Tensor, root: int = 0, stream=None) -> torch.Tensor:
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("broadcast_: a contiguous float32 device tensor is required")
        _capi.check(self.lib, self.lib.dof_comm_broadcast(self._h, t.data_ptr(), t.numel(), int(root), self._stream(stream)),
                    "dof_comm_broadcast")
        return t

    def close(self):
        if self._h is not None and self._h.value:
            self.lib.dof_comm_destroy(self._h)
        self._h = None

    def abort(self):
        """``ncclCommAbort``: drop the communicator without waiting for collectives in flight."""
        if self._h is not None and self._h.value:
            self.lib.dof_comm_abort(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
