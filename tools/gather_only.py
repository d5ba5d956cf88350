"""The C2 materialisation launches of bench.py's roofline section alone (one process that runs nothing but
k_window_gather), for the rocprofv3 PMC passes of tools/gather_pmc.sh."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    _capi.check(lib, lib.dof_window_gather_range(tn.data_ptr(), te.data_ptr(), 0, 1, nw, T, N, E, x.data_ptr(), a.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream))
if len(sys.argv) > 2 and sys.argv[2] == "bf16":   # + the bf16-storage variant of the same launch
    xb = torch.empty(nw, T, N, 3, device=dev, dtype=torch.bfloat16)
    ab = torch.empty(nw, T, E, 1, device=dev, dtype=torch.bfloat16)
    for _ in range(int(sys.argv[1])):
        _capi.check(lib, lib.dof_window_gather_bf16(tn.data_ptr(), te.data_ptr(), None, 0, 1, nw, T, N, E, xb.data_ptr(), ab.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("gathered", nw, "windows x", sys.argv[1] if len(sys.argv) > 1 else 6)
