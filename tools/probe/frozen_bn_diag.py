"""Development aid: gradient of the contrastive TCN encoder with frozen BatchNorm statistics, one launch of 2 Bc windows
against the sum of two launches of Bc windows; prints per-tensor max error and the ratio of norms."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepof_amd import graph as G
from deepof_amd._lib import load_hip_library
from deepof_amd.engine import contrastive_views, create_vade_engine
hip = load_hip_library()
nodes, edges = G.bodypart_graph([""])
adj = G.adjacency_from_graph(nodes, edges)
ei, _ = G.edge_index_from_graph(nodes, edges)
Bc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, Tf, L, N = 2 * Bc, 50, 8, len(nodes)
g = torch.Generator().manual_seed(11)
x_full = (torch.randn(B, Tf, N, 3, generator=g).cumsum(1) * 0.1).contiguous().cuda()
eid = torch.from_numpy(ei).cuda()
big = create_vade_engine(B, Tf // 2, adj, L, 1, kind="contrastive_tcn")
small = create_vade_engine(Bc, Tf // 2, adj, L, 1, kind="contrastive_tcn")
for n in big.names:
    shape = big.layout[n][2]
    if n.endswith("running_var"):
        v = 0.5 + torch.rand(shape, generator=g)
    elif n.endswith("running_mean"):
        v = torch.randn(shape, generator=g) * 0.2
    elif (".bn" in n and n.endswith("weight")) or n in ("encoder.head.2.weight", "encoder.head.5.weight"):
        v = 1.0 + torch.randn(shape, generator=g) * 0.1
    elif n.endswith("bias"):
        v = torch.randn(shape, generator=g) * 0.05
    elif ".head." in n or "spatial_gnn_block" in n:
        v = torch.randn(shape, generator=g) * 0.3
    else:
        v = torch.randn(shape, generator=g) * 0.05
    big.view(n).copy_(v)
small.params.copy_(big.params)
frozen = (sys.argv[2] if len(sys.argv) > 2 else "frozen") == "frozen"
big.set_bn_training(not frozen)
small.set_bn_training(not frozen)
x, a = contrastive_views(hip, x_full, eid, None)
dz = (torch.randn(B, L, generator=g) * 0.1).cuda()
z_big = big.contrastive_encode(x, a, train=True, count=False)
big.contrastive_backward(dz, accumulate=False)
acc = torch.zeros_like(small.grads, dtype=torch.float64)
for c in range(2):
    sl = slice(c * Bc, (c + 1) * Bc)
    z_c = small.contrastive_encode(x[sl].contiguous(), a[sl].contiguous(), train=True, count=False)
    print("chunk", c, "max |z - z_big|", float((z_c - z_big[sl]).abs().max()))
    small.contrastive_backward(dz[sl].contiguous(), accumulate=False)
    acc += small.grads.double()
tot = acc.float()
for n in big.names:
    if n not in big.layout or "running" in n or n.startswith("distill_head."):
        continue
    gb, gs = big.view(n, big.grads).cpu().numpy(), small.view(n, tot).cpu().numpy()
    sc = float(np.abs(gs).max())
    err = float(np.abs(gb - gs).max())
    flag = "" if err <= 1e-4 + 2e-3 * sc else "   <<<<"
    print(f"{n:55s} err {err:10.3e} scale {sc:10.3e} |big|/|sum| {np.linalg.norm(gb) / (np.linalg.norm(gs) + 1e-30):8.4f}{flag}")
