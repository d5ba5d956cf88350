import os, sys
import numpy as np, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from deepof_amd._lib import load_hip_library
from deepof_amd.engine import VadeEngine
from parity_common import configure_phase, load_golden, params_from
tag = sys.argv[1]
lib = load_hip_library()
d = load_golden(os.path.join(ROOT, "tests", "golden"), "vade_tcn14_onepass.npz")
x, a = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["a"]).cuda()
B, T, N, _ = x.shape
K, L = d["sd::latent_space.gmm_means"].shape
eng = VadeEngine(lib, "cuda", B, T, d["adj"], L, K, kind="vade_tcn")
eng.load_state_dict(params_from(d))
eng.set_bn_training(True)
configure_phase(eng, K, True, 0.13, None, 0.0)
eps = torch.from_numpy(d["eps"]).cuda()
o = eng.forward(x, a, eps, want_loc=True, want_enc=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"fwd_{tag}.npz"), **{k: v.cpu().numpy() for k, v in o.items()})
