"""Helper of test_tcn_record_statistics_vs_two_pass_gpu: one VaDE-TCN train step whose BatchNorm running means equal the
batch means (the hardest case for a one-pass variance); the parent runs it with the default record statistics and with
DOF_TCN_STAT_RECORDS=0 (sum pass + centred second pass).  Prints one JSON line (loss terms, gradient checksums, refreshed
running variances)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from deepof_amd._lib import load_hip_library
    from parity_common import VadeEngine, configure_phase
    lib = load_hip_library()
    B, T, L, K = 48, 25, 8, 5
    adj = np.zeros((5, 5), np.float32)
    for i in range(4):
        adj[i, i + 1] = adj[i + 1, i] = 1.0
    eng = VadeEngine(lib, "cuda", B, T, adj, L, K, kind="vade_tcn")
    g = torch.Generator().manual_seed(11)
    P = eng.state_dict()
    for n, v in P.items():
        if not v.dtype.is_floating_point or n.split(".")[-1] in ("laplacian", "edge_laplacian", "incidence", "prior", "pretrain"):
            continue
        if n.endswith("running_var"):
            P[n] = torch.ones(v.shape)
        elif n.endswith("running_mean"):
            P[n] = torch.zeros(v.shape)
        elif (".bn" in n or "head.2" in n or "head.5" in n) and n.endswith("weight"):
            P[n] = 0.6 + 0.8 * torch.rand(v.shape, generator=g)
        elif ".bn" in n and n.endswith("bias"):
            P[n] = 0.3 * torch.randn(v.shape, generator=g)
        else:
            P[n] = torch.randn(v.shape, generator=g) * (0.3 if v.dim() > 1 else 0.1)
    x = (torch.randn(B, T, 5, 3, generator=g).cumsum(1) * 0.3).cuda()
    a = torch.randn(B, T, 4, 1, generator=g).cuda()
    eps = torch.randn(B, L, generator=g).cuda()
    configure_phase(eng, K, True, 0.2, None, 0.0)
    # probe step: running_mean 0 -> momentum * batch mean, i.e. the batch means of every layer (they do not depend on
    # the running statistics in train mode)
    eng.load_state_dict(P)
    eng.loss_grads(x, a, eps, None, None, pretrain=True)
    after = eng.state_dict()
    for n in P:
        if n.endswith("running_mean") and "_tcn.blocks." in n:
            P[n] = after[n] * 10.0      # momentum 0.1
    eng.load_state_dict(P)
    eng.loss_grads(x, a, eps, None, None, pretrain=True)
    logs = eng.read_logs()
    sd = eng.state_dict()
    out = {"logs": {k: float(v) for k, v in logs.items()}, "grads": {}, "rvar": {}}
    for n in eng.names:
        if n in eng.layout and eng.layout[n][2] is not None and "running" not in n:
            gv = eng.view(n, eng.grads).double()
            out["grads"][n] = [float(gv.abs().max()), float(gv.sum()), float((gv * gv).sum())]
    for n, v in sd.items():
        if n.endswith("running_var") and "_tcn.blocks." in n:
            out["rvar"][n] = [float(t) for t in v.double().flatten()[:4]]
    print("PROBE " + json.dumps(out))


if __name__ == "__main__":
    main()
