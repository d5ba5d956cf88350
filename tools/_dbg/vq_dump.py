import sys, ctypes, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from deepof_amd import _capi
import parity_common as PC
def run(libpath, tag):
    lib = _capi.bind(ctypes.CDLL(libpath))
    d = PC.load_golden('/root/repo/tests/golden', "vqvae_tcn14.npz")
    device='cuda'
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    print("shapes", x.shape, a.shape)
    L, K = d["sd::vq_layer.codebook"].shape
    eng = PC.VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vqvae_tcn")
    sd0 = PC.params_from(d)
    eng.load_state_dict(sd0)
    eng.set_bn_training(True)
    for name in eng.names:
        if ".spatial_gnn_block." in name:
            eng.set_trainable(name, False)
    eng.reset_optimizer()
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, 1e-3)
    eng.set_hyper(vq_beta=1.0, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
    eng.push_hyper()
    eng.vq_loss_grads(x, a)
    out = {}
    for k in d:
        if k.startswith("grad::"):
            name = k[len("grad::"):]
            out[name] = eng.view(name, eng.grads).cpu().numpy().copy()
            out["ref::" + name] = d[k].reshape(out[name].shape)
    np.savez('/root/repo/gpurun_out/vqgrads_%s.npz' % tag, **out)
run(sys.argv[1], sys.argv[2])
