import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from deepof_amd._lib import load_hip_library as emu_lib
from parity_common import load_golden, params_from, aug_from_golden
from deepof_amd.engine import VadeEngine, contrastive_views
d = load_golden("/root/repo/tests/golden", "contrastive_tcn14_b64.npz")
pfx = "c0::"
lib = emu_lib()
x_full = torch.from_numpy(d["x_full"]).cuda(); ei = torch.from_numpy(d["edge_index"]).cuda()
B, Tf, N, _ = x_full.shape
L = d[pfx + "sd::encoder.head.6.bias"].shape[0]
e1 = VadeEngine(lib, "cuda", B, Tf // 2, d["adj"], L, 1, kind="contrastive_tcn")
e1.load_state_dict(params_from(d, pfx + "sd::"))
xc, ac = contrastive_views(lib, x_full, ei, None)
z = e1.contrastive_encode(xc, ac, train=True)
sd1 = e1.state_dict()
worst = 0
for k in d:
    if k.startswith(pfx + "sd_after::") and ("running_mean" in k or "running_var" in k):
        name = k[len(pfx) + 10:]
        err = float(np.abs(sd1[name].numpy() - d[k]).max() / (np.abs(d[k]).max() + 1e-30))
        worst = max(worst, err)
print(os.environ.get("DOF_TCN_NCW8"), "z err", float(np.abs(z.cpu().numpy() - d[pfx + "z"]).max()), "worst running-stat rel err", worst)
# the second view and the complete state comparison
xa, aa = contrastive_views(lib, x_full, ei, aug_from_golden(d, pfx, "cuda"))
e2 = VadeEngine(lib, "cuda", B, Tf // 2, d["adj"], L, 1, kind="contrastive_tcn", shared=e1)
za = e2.contrastive_encode(xa, aa, train=True)
sd1 = e1.state_dict()
errs = []
for k in d:
    if k.startswith(pfx + "sd_after::"):
        name = k[len(pfx) + 10:]
        errs.append((float(np.abs(sd1[name].numpy() - d[k]).max() / (np.abs(d[k]).max() + 1e-30)), name))
errs.sort(reverse=True)
print("z_aug err", float(np.abs(za.cpu().numpy() - d[pfx + "z_aug"]).max()), "worst buffers", errs[:4])
