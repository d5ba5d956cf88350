"""Child process of test_tcn_onepass_opt_in_deviation_gpu: the one-pass fixture through the HIP path with whatever
DOF_TCN_ONEPASS the parent set; prints one JSON line of deviations from the reference golden."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from deepof_amd._lib import load_hip_library  # noqa: E402
from deepof_amd.engine import VadeEngine  # noqa: E402
from parity_common import configure_phase, load_golden, math_zero_gradient, params_from  # noqa: E402

d = load_golden(os.path.join(HERE, "golden"), "vade_tcn14_onepass.npz")
x, a = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["a"]).cuda()
B, T, N, _ = x.shape
K, L = d["sd::latent_space.gmm_means"].shape
eng = VadeEngine(load_hip_library(), "cuda", B, T, d["adj"], L, K, kind="vade_tcn")
eng.load_state_dict(params_from(d))
eng.set_bn_training(True)
configure_phase(eng, K, True, 0.13, None, 0.0)
eng.loss_grads(x, a, torch.from_numpy(d["eps"]).cuda(), None, None, pretrain=True)
logs = eng.read_logs()
loss_rel = max(abs(v - float(d[f"pre::loss::{k}"])) / (abs(float(d[f"pre::loss::{k}"])) + 1e-3) for k, v in logs.items()
               if f"pre::loss::{k}" in d)
sd1 = eng.state_dict()
buf_rel = max(float(np.abs(sd1[k[len("pre::sd_after::"):]].numpy() - d[k]).max() / (np.abs(d[k]).max() + 1e-6))
              for k in d if k.startswith("pre::sd_after::") and "running" in k)
worst, beyond, n = 0.0, 0, 0
for k in d:
    if k.startswith("pre::grad::"):
        name = k[len("pre::grad::"):]
        if math_zero_gradient(name):
            continue
        g = eng.view(name, eng.grads).cpu().numpy()
        r = d[k].reshape(g.shape)
        sc = float(np.abs(r).max())
        err = float(np.abs(g - r).max())
        worst = max(worst, err / (sc + 1e-12))
        beyond += err > 5e-5 + 5e-4 * sc
        n += 1
print("PROBE " + json.dumps({"worst_loss_rel": loss_rel, "worst_buffer_rel": buf_rel, "worst_grad_over_scale": worst,
                             "beyond_standard_bar": int(beyond), "n_grads": n}))
