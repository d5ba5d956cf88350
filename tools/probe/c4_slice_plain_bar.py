"""contrastive_tcn14_b64 golden against the HIP path WITHOUT any flip attribution: per-tensor gradient error in units of
the standard bar (5e-5 + 5e-4 max|ref|).  python tools/probe/c4_b128_plain_bar.py   (GPU box)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_common as PC  # noqa: E402
from deepof_amd._lib import load_hip_library  # noqa: E402
from deepof_amd.engine import VadeEngine, contrastive_views  # noqa: E402

lib = load_hip_library()
d = PC.load_golden(os.path.join(ROOT, "tests", "golden"), "contrastive_tcn14_b64.npz")
pfx, device = "c0::", "cuda"
x_full = torch.from_numpy(d["x_full"]).to(device)
ei = torch.from_numpy(d["edge_index"]).to(device)
B, Tf, N, _ = x_full.shape
L = d[pfx + "sd::encoder.head.6.bias"].shape[0]
e1 = VadeEngine(lib, device, B, Tf // 2, d["adj"], L, 1, kind="contrastive_tcn")
e2 = VadeEngine(lib, device, B, Tf // 2, d["adj"], L, 1, kind="contrastive_tcn", shared=e1)
e1.load_state_dict(PC.params_from(d, pfx + "sd::"))
xc, ac = contrastive_views(lib, x_full, ei, None)
xa, aa = contrastive_views(lib, x_full, ei, PC.aug_from_golden(d, pfx, device))
z = e1.contrastive_encode(xc, ac, train=True)
z_aug = e2.contrastive_encode(xa, aa, train=True)
print("z err", float(np.abs(z.cpu().numpy() - d[pfx + "z"]).max()), "z_aug err", float(np.abs(z_aug.cpu().numpy() - d[pfx + "z_aug"]).max()))
dz, dza = e1.contrastive_loss(z, z_aug, "cosine", "nce", 0.1, 0.1, 0.1)
print({k: (v, float(d[pfx + f"log::{k}"])) for k, v in e1.read_contrastive_logs().items() if pfx + f"log::{k}" in d})
e1.contrastive_backward(dz, accumulate=False)
e2.contrastive_backward(dza, accumulate=True)
rows = []
for k in d:
    if k.startswith(pfx + "grad::"):
        name = k[len(pfx) + 6:]
        if PC.math_zero_gradient(name):
            continue
        g = e1.view(name, e1.grads).cpu().numpy()
        ref = d[k].reshape(g.shape)
        bar = 5e-5 + 5e-4 * float(np.abs(ref).max())
        rows.append((float(np.abs(g - ref).max()) / bar, name, float(np.abs(ref).max())))
rows.sort(reverse=True)
print("tensors:", len(rows), "beyond the plain bar:", sum(r[0] > 1 for r in rows))
for r in rows[:25]:
    print(f"{r[0]:8.3f}  scale {r[2]:.3e}  {r[1]}")
