"""Development aid: the one-pass fixture through the HIP path with DOF_TCN_ONEPASS as set in the environment; prints the
loss terms next to the reference's and saves gradients + train-mode forward outputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from deepof_amd._lib import load_hip_library  # noqa: E402
from deepof_amd.engine import VadeEngine  # noqa: E402
from parity_common import configure_phase, load_golden, params_from  # noqa: E402

tag = sys.argv[1]
lib = load_hip_library()
d = load_golden(os.path.join(ROOT, "tests", "golden"), "vade_tcn14_onepass.npz")
x, a = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["a"]).cuda()
B, T, N, _ = x.shape
K, L = d["sd::latent_space.gmm_means"].shape
eng = VadeEngine(lib, "cuda", B, T, d["adj"], L, K, kind="vade_tcn")
sd0 = params_from(d)
eps = torch.from_numpy(d["eps"]).cuda()
eng.load_state_dict(sd0)
eng.set_bn_training(True)
configure_phase(eng, K, True, 0.13, None, 0.0)
eng.loss_grads(x, a, eps, None, None, pretrain=True)
logs = eng.read_logs()
out = {}
for k, v in logs.items():
    key = f"pre::loss::{k}"
    if key in d:
        print(f"{k:28s} hip {v:.8f} ref {float(d[key]):.8f} rel {abs(v - float(d[key])) / (abs(float(d[key])) + 1e-12):.2e}")
for name in eng.names:
    out[name] = eng.view(name, eng.grads).cpu().numpy().copy()
sd1 = eng.state_dict()
worst = 0.0
for k in d:
    if k.startswith("pre::sd_after::") and "running" in k:
        name = k[len("pre::sd_after::"):]
        worst = max(worst, float(np.abs(sd1[name].numpy() - d[k]).max() / (np.abs(d[k]).max() + 1e-6)))
print("worst refreshed running buffer, relative:", worst)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"onepass_diag_{tag}.npz"), **out)
