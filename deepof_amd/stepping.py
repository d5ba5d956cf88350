"""hipGraph plumbing of the fit loops: device-resident weight schedules and a cache of captured step bodies.

A training step of the hot path is ~110 small kernel launches (VaDE recurrent, C2) whose launch overhead rivals
their run time, so the fit loops of ``deepof_amd.training`` replay each step as ONE hipGraph.  That only works when
a step is a pure function of device memory:

* per-step loss weights (KL weight, distillation lambda: the reference's ``Dynamic_weight_manager`` pair
  ``get_weight()`` / ``step()``, /root/reference/deepof/clustering/losses.py:290-351, driven once per batch by
  training.py:167-181) come from ``DeviceSchedule``: the whole curve as a device table plus a device cursor that
  ``dof_schedule_apply`` reads and advances inside the captured step;
* Adam's step count lives on the device (``dof_optimizer_step``);
* the batch and the noise live in static buffers that the step overwrites in place.

``StepGraphs.run(key, body)`` runs ``body`` eagerly the first time a key is seen (real work: it also lets the HIP
runtime load every code object outside of a capture), captures it the second time, and replays it from then on.
A new batch size (the ragged last batch), phase or teacher switch is a new key, hence a new graph.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Hashable

import torch

from .schedules import WeightSchedule


class DeviceSchedule:
    """A ``WeightSchedule`` mirrored on the device.  The host cursor keeps serving ``get_weight()`` for console
    output; the device cursor is the one the kernels see.  Both advance once per training step (``step()`` on the
    host, ``dof_schedule_apply(advance=1)`` on the device), so they agree by construction."""

    _next_uid = 0

    def __init__(self, schedule: WeightSchedule, device):
        DeviceSchedule._next_uid += 1
        self.uid = DeviceSchedule._next_uid  # graph-cache key (id() can be recycled after garbage collection)
        self.host = schedule
        self.table = torch.from_numpy(schedule.table).to(device=device, dtype=torch.float32).contiguous()
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)

    @property
    def max_weight(self) -> float:
        return self.host.max_weight

    def get_weight(self) -> float:
        return self.host.get_weight()

    def step(self) -> None:
        self.host.step()

    def device_position(self) -> int:
        """(synchronises) cursor value on the device -- tests / diagnostics only."""
        return int(self.cursor.item())


def constant_schedule(value: float, device) -> DeviceSchedule:
    """A one-entry table: a fixed weight, in force from the first step, that still goes through dof_schedule_apply."""
    return DeviceSchedule(WeightSchedule.constant(value), device)


class StepGraphs:
    """Capture-once / replay cache of step bodies on one device."""

    def __init__(self, device, enabled: bool = None):
        self.device = torch.device(device)
        if enabled is None:
            enabled = self.device.type == "cuda" and os.environ.get("DOF_NO_GRAPH", "0") != "1"
        self.enabled = bool(enabled)
        self._seen = set()
        self._graphs: Dict[Hashable, "torch.cuda.CUDAGraph"] = {}
        self._slot_key: Dict[Hashable, Hashable] = {}
        self._pool = None
        self.replays = 0

    def run(self, key: Hashable, body: Callable[[], None], slot: Hashable = None) -> None:
        """``slot``: the part of ``key`` that names the step variant (batch size, phase, train / val ...); the rest of
        the key names the objects the captured body points at (schedule tables ...).  When a slot is run under a new
        key -- a schedule was replaced, the teacher refreshed -- the graph captured under its previous key can never
        be replayed again and is dropped with its pool memory: one live graph per slot."""
        if not self.enabled:
            body()
            return
        if slot is not None:
            old = self._slot_key.get(slot)
            if old is not None and old != key:
                self._graphs.pop(old, None)
                self._seen.discard(old)
            self._slot_key[slot] = key
        g = self._graphs.get(key)
        if g is None:
            if key not in self._seen:  # first occurrence: eager (loads code objects, builds lazily created plans)
                self._seen.add(key)
                body()
                return
            g = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            # thread_local: RCCL's watchdog thread may poll events while this thread captures
            with torch.cuda.graph(g, pool=self._pool, capture_error_mode="thread_local"):
                body()
            self._graphs[key] = g
        g.replay()
        self.replays += 1

    def clear(self) -> None:
        self._graphs.clear()
        self._seen.clear()
        self._slot_key.clear()
