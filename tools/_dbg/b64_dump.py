import sys, ctypes, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from deepof_amd import _capi
import parity_common as PC
from parity_common import *
def run(libpath, tag):
    lib = _capi.bind(ctypes.CDLL(libpath))
    d = PC.load_golden('/root/repo/tests/golden', "vade_tcn14_b64.npz")
    device='cuda'
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = PC.VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vade_tcn")
    sd0 = PC.params_from(d)
    eng.load_state_dict(sd0)
    eng.set_bn_training(True)
    eps = torch.from_numpy(d["eps"]).to(device)
    PC.configure_phase(eng, K, True, 0.13, None, 0.0)
    eng.loss_grads(x, a, eps, None, None, pretrain=True)
    out = {}
    for k in d:
        if k.startswith("pre::grad::"):
            name = k.split("::")[-1]
            out[name] = eng.view(name, eng.grads).cpu().numpy().copy()
            out["ref::" + name] = d[k].reshape(out[name].shape)
    np.savez('/root/repo/gpurun_out/grads_%s.npz' % tag, **out)
run(sys.argv[1], sys.argv[2])
