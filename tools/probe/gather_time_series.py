"""Per-launch durations of the C2 materialisation gather over 120 back-to-back launches (HIP events around every launch):
does the rate drift within a process?  python tools/probe/gather_time_series.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth_tables_fast  # noqa: E402
from deepof_amd import _capi  # noqa: E402
from deepof_amd._lib import load_hip_library  # noqa: E402

lib = load_hip_library()
F, T, N, E = 600_000, 25, 14, 14
dev = torch.device("cuda")
tn, te = synth_tables_fast(F, N, E, 0, dev)
nw = F - T + 1
x = torch.empty(nw, T, N, 3, device=dev)
a = torch.empty(nw, T, E, 1, device=dev)
n = 120
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
evs[0].record()
for i in range(n):
    _capi.check(lib, lib.dof_window_gather_range(tn.data_ptr(), te.data_ptr(), 0, 1, nw, T, N, E, x.data_ptr(), a.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream))
    evs[i + 1].record()
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
print("launch ms, groups of 10:", [round(sum(ms[i:i + 10]) / 10, 3) for i in range(0, n, 10)])
print("min / median / max:", round(min(ms), 3), round(sorted(ms)[n // 2], 3), round(max(ms), 3))
