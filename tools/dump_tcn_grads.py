"""Development aid: run the TCN-family reference fixtures on the GPU and save the HIP path's gradients
(gpurun_out/tcn_grads.npz) so that the ReLU-kink attribution of tests/parity_common.py can be studied offline."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from deepof_amd import _capi  # noqa: E402
from deepof_amd._lib import load_hip_library  # noqa: E402
from deepof_amd.engine import VadeEngine  # noqa: E402
from parity_common import configure_phase, load_golden, params_from  # noqa: E402

lib = load_hip_library()
G = os.path.join(ROOT, "tests", "golden")
out = {}
for fixture in ("vade_tcn14_b64.npz", "vade_tcn14_onepass.npz"):
    d = load_golden(G, fixture)
    x, a = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["a"]).cuda()
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = VadeEngine(lib, "cuda", B, T, d["adj"], L, K, kind="vade_tcn")
    sd0 = params_from(d)
    eps, eps_mc = torch.from_numpy(d["eps"]).cuda(), torch.from_numpy(d["eps_mc"]).cuda()
    tau = torch.from_numpy(d["tau"]).cuda()
    for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
        eng.load_state_dict(sd0)
        eng.set_bn_training(True)
        configure_phase(eng, K, phase == "pre", klw, tau if teacher else None, 1.7 if teacher else 0.0)
        eng.loss_grads(x, a, eps, None if phase == "pre" else eps_mc, tau if teacher else None, pretrain=phase == "pre")
        for name in eng.names:
            out[f"{fixture[:-4]}::{phase}::{name}"] = eng.view(name, eng.grads).cpu().numpy().copy()
d = load_golden(G, "vqvae_tcn14.npz")
x, a = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["a"]).cuda()
B, T, N, _ = x.shape
L, K = d["sd::vq_layer.codebook"].shape
eng = VadeEngine(lib, "cuda", B, T, d["adj"], L, K, kind="vqvae_tcn")
eng.load_state_dict(params_from(d))
eng.set_bn_training(True)
eng.set_hyper(vq_beta=1.0, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
for seg in range(_capi.SEG_COUNT):
    eng.set_lr(seg, 1e-3)
eng.push_hyper()
eng.vq_loss_grads(x, a)
for name in eng.names:
    out[f"vqvae_tcn14::step1::{name}"] = eng.view(name, eng.grads).cpu().numpy().copy()
eng.load_state_dict({k: v for k, v in params_from(d, "sd_step1::").items()})
eng.vq_loss_grads(torch.from_numpy(d["step2::x"]).cuda(), torch.from_numpy(d["step2::a"]).cuda())
for name in eng.names:
    out[f"vqvae_tcn14::step2::{name}"] = eng.view(name, eng.grads).cpu().numpy().copy()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "tcn_grads.npz"), **out)
print("saved", len(out), "tensors")
