"""Pin the CPU oracle against fixtures produced by the imported reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import vade as OV
from oracle import windows as OW
from deepof_amd import graph as G

torch.set_num_threads(2)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def _params(d, prefix="sd::"):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


def test_scramble_index(golden_dir):
    d = _load(golden_dir, "scramble.npz")
    for k, ref in d.items():
        T, Gn, F = (int(s) for s in k.split("_")[1:])
        np.testing.assert_array_equal(OW.group_scramble_index(T, Gn, F), ref)
        x = np.random.default_rng(0).standard_normal((2, T, Gn, F)).astype(np.float32)
        np.testing.assert_array_equal(OW.group_scramble(x)[1], x[1].reshape(-1)[ref.reshape(-1)].reshape(Gn, T, F))


def test_graph_ops(golden_dir):
    d = _load(golden_dir, "graph_ops.npz")
    for tag, ids in [("single", [""]), ("pair", ["B", "W"])]:
        nodes, edges = G.bodypart_graph(ids)
        adj = G.adjacency_from_graph(nodes, edges)
        np.testing.assert_array_equal(adj, d[f"{tag}_adj"])
        lap, elap, inc = G.censnet_operators(adj)
        np.testing.assert_array_equal(inc, d[f"{tag}_inc"])
        np.testing.assert_allclose(lap, d[f"{tag}_lap"], atol=1e-7)
        np.testing.assert_allclose(elap, d[f"{tag}_elap"], atol=1e-7)
    nodes, edges = G.bodypart_graph([""])
    assert len(nodes) == 14 and len(edges) == 14
    assert nodes[0] == "Center" and edges[0] == ("Center", "Left_fhip")
    nodes, edges = G.bodypart_graph(["B", "W"])
    assert len(nodes) == 28 and len(edges) == 32


def test_window_build_matches_as_strided():
    rng = np.random.default_rng(1)
    N, E, W, Fr = 5, 4, 7, 40
    nodes, edges = rng.standard_normal((Fr, 3 * N)), rng.standard_normal((Fr, E))
    ref = np.lib.stride_tricks.sliding_window_view(nodes, W, axis=0).transpose(0, 2, 1)
    np.testing.assert_array_equal(OW.rolling_window(nodes, W, 1), ref)
    np.testing.assert_array_equal(OW.rolling_window(nodes, W, 3), ref[::3])
    assert OW.rolling_window(nodes, W, 1).shape[0] == (Fr - W) // 1 + 1
    x, a = OW.gather_windows(nodes, edges, np.array([0, 5, 33]), W)
    assert x.shape == (3, W, N, 3) and a.shape == (3, W, E, 1)
    np.testing.assert_array_equal(x[1, 2, 3], nodes[7, [3, N + 3, 2 * N + 3]].astype(np.float32))
    np.testing.assert_array_equal(a[2, 6, 1, 0], np.float32(edges[39, 1]))


def test_window_build_matches_reference_golden(golden_dir):
    """oracle/windows.py and the host helper deepof_amd.dataset.reorder_and_reshape against the outputs of the
    reference's own rolling_window (utils.py:3354) + reorder_and_reshape (clustering/dataset.py:16-26)."""
    from deepof_amd.dataset import reorder_and_reshape
    d = _load(golden_dir, "windows_graph.npz")
    for ci in range(int(d["n_window_cases"])):
        F, W, step, N, E = (int(v) for v in d[f"w{ci}::cfg"])
        nt, et = d[f"w{ci}::node_table"], d[f"w{ci}::edge_table"]
        wn = OW.rolling_window(nt, W, step)
        np.testing.assert_array_equal(wn, d[f"w{ci}::node_windows"])
        np.testing.assert_array_equal(OW.rolling_window(et, W, step), d[f"w{ci}::edge_windows"])
        assert wn.shape[0] == (F - W) // step + 1                       # reference tests/test_utils.py:543
        np.testing.assert_array_equal(OW.node_windows_to_x(wn), d[f"w{ci}::x"])
        np.testing.assert_array_equal(OW.edge_windows_to_a(d[f"w{ci}::edge_windows"]), d[f"w{ci}::a"])
        np.testing.assert_array_equal(reorder_and_reshape(wn).astype(np.float32), d[f"w{ci}::x"])
        starts = np.arange(0, F - W + 1, step)
        x, a = OW.gather_windows(nt, et, starts, W)
        np.testing.assert_array_equal(x, d[f"w{ci}::x"])
        np.testing.assert_array_equal(a, d[f"w{ci}::a"])


def test_bodypart_graphs_match_connect_mouse(golden_dir):
    """deepof_amd.graph presets vs the reference's connect_mouse (utils.py:416-508) in get_graph_dataset's sorted
    node / edge order with its adjacency matrix (data.py:2791-2793)."""
    d = _load(golden_dir, "windows_graph.npz")
    for gi in range(int(d["n_graph_cases"])):
        ids, preset = [str(v) for v in d[f"g{gi}::ids"]], str(d[f"g{gi}::preset"])
        nodes, edges = G.bodypart_graph(ids, preset)
        assert nodes == [str(v) for v in d[f"g{gi}::nodes"]], (ids, preset)
        assert [tuple(e) for e in edges] == [tuple(str(v) for v in e) for e in d[f"g{gi}::edges"]], (ids, preset)
        np.testing.assert_array_equal(G.adjacency_from_graph(nodes, edges), d[f"g{gi}::adj"])


@pytest.mark.parametrize("tag", ["node", "edge", "node_l6"])
def test_recurrent_block(golden_dir, tag):
    d = _load(golden_dir, "recurrent_block.npz")
    P = {"blk." + k: v for k, v in _params(d, f"{tag}::sd::").items()}
    leaf = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    y = OV.recurrent_block(torch.from_numpy(d[f"{tag}::x"]), leaf, "blk")
    np.testing.assert_allclose(y.detach().numpy(), d[f"{tag}::y"], atol=2e-6, rtol=1e-5)
    (y * torch.from_numpy(d[f"{tag}::up"])).sum().backward()
    for k, v in d.items():
        if k.startswith(f"{tag}::grad::"):
            name = "blk." + k.split("::grad::")[1]
            np.testing.assert_allclose(leaf[name].grad.numpy(), v, atol=5e-5, rtol=1e-4, err_msg=name)


@pytest.mark.parametrize("tag", ["rec14", "rec28", "c5l8", "rec14l16", "rec14l32", "rec14l4", "rec14l5", "rec14l6", "rec14l7", "rec14l9", "rec14l14", "rec14l10", "rec14l12", "rec14l20", "rec14l24"])
def test_vade_eval_forward(golden_dir, tag):
    d = _load(golden_dir, f"vade_{tag}.npz")
    P = _params(d)
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    with torch.no_grad():
        out = OV.vade_forward(P, x, a, training=False)
    np.testing.assert_allclose(out["enc"].numpy(), d["eval_enc"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(out["z"].numpy(), d["eval_z"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(out["q"].numpy(), d["eval_q"], atol=2e-6, rtol=1e-4)
    np.testing.assert_allclose(out["loc"].numpy(), d["eval_loc"], atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(float(out["kmeans"]), float(d["eval_kmeans"]), rtol=1e-6)


PHASES = {
    "pre": dict(klw=0.13, pretrain=True, teacher=False, kw={}),
    "main": dict(klw=0.7, pretrain=False, teacher=False, kw={}),
    "mainT": dict(klw=0.7, pretrain=False, teacher=True, kw={}),
    "mainX": dict(klw=0.45, pretrain=False, teacher=True,
                  kw=dict(repel_weight=0.3, reg_scatter_weight=0.2, temporal_cohesion_weight=0.1,
                          reg_cat_clusters=0.5, tf_cluster_weight=0.7, kmeans_loss_weight=0.5,
                          distill_conf_weight=True)),
}


def make_cfg(K, phase, tau):
    spec = PHASES[phase]
    kw = dict(spec["kw"])
    if spec["teacher"]:
        pi = tau.mean(dim=0).clamp_min(1e-8)
        w = pi.pow(-1.0)
        w = (w / w.mean()).clamp_max(3.0)
        kw.update(lambda_distill=1.7, class_weight=w, teacher_marginal=pi)
    return OV.VadeLossCfg(K, spec["pretrain"], **kw), spec["klw"]


@pytest.mark.parametrize("tag", ["rec14", "rec28", "c5l8", "rec14l16", "rec14l32", "rec14l4", "rec14l5", "rec14l6", "rec14l7", "rec14l9", "rec14l14", "rec14l10", "rec14l12", "rec14l20", "rec14l24"])
@pytest.mark.parametrize("phase", list(PHASES))
def test_vade_train_loss_and_grads(golden_dir, tag, phase):
    d = _load(golden_dir, f"vade_{tag}.npz")
    P = _params(d)
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    tau = torch.from_numpy(d["tau"])
    K = tau.shape[1]
    cfg, klw = make_cfg(K, phase, tau)
    losses, grads, out = OV.vade_grads(P, x, a, cfg, klw, torch.from_numpy(d["eps"]),
                                       torch.from_numpy(d["eps_mc"]),
                                       tau if PHASES[phase]["teacher"] else None)
    np.testing.assert_allclose(out["z"].detach().numpy(), d[f"{phase}::z"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(out["q"].detach().numpy(), d[f"{phase}::q"], atol=3e-6, rtol=1e-4)
    for k, v in losses.items():
        np.testing.assert_allclose(float(v), float(d[f"{phase}::loss::{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    n_checked = 0
    for k, v in d.items():
        if k.startswith(f"{phase}::grad::"):
            name = k.split("::grad::")[1]
            assert grads[name] is not None, name
            np.testing.assert_allclose(grads[name].numpy(), v, atol=2e-5, rtol=2e-4, err_msg=name)
            n_checked += 1
    assert n_checked >= 80
    # params the reference leaves without a gradient must be unused here too
    for name, g in grads.items():
        if f"{phase}::grad::{name}" not in d:
            assert g is None, name


def test_vade_train_trace(golden_dir):
    d = _load(golden_dir, "vade_train_trace.npz")
    P = _params(d, "sd0::")
    K = P["latent_space.gmm_means"].shape[0]
    opt = None
    last_phase = None
    for s in range(6):
        phase = str(d[f"step{s}::phase"])
        if phase != last_phase:
            opt, last_phase = OV.AdamState(), phase
        cfg = OV.VadeLossCfg(K, phase == "pre")
        lr_b, lr_g = d[f"step{s}::lr"]
        logs, _, _ = OV.vade_train_step(
            P, opt, torch.from_numpy(d[f"step{s}::x"]), torch.from_numpy(d[f"step{s}::a"]), cfg,
            float(d[f"step{s}::klw"]), float(lr_b), float(lr_g), torch.from_numpy(d[f"step{s}::eps"]),
            torch.from_numpy(d[f"step{s}::eps_mc"]))
        for k, v in logs.items():
            np.testing.assert_allclose(v, float(d[f"step{s}::log::{k}"]), rtol=5e-4, atol=5e-5, err_msg=f"{s}:{k}")
        keys = OV.trainable_keys(P)
        pn = float(torch.sqrt(sum((P[k] ** 2).sum() for k in keys)))
        np.testing.assert_allclose(pn, float(d[f"step{s}::pnorm"]), rtol=1e-5)
    for k, v in _params(d, "sd_final::").items():
        np.testing.assert_allclose(P[k].numpy(), v.numpy(), atol=2e-4, rtol=1e-3, err_msg=k)


def test_kmeans_value_and_grad(golden_dir):
    d = _load(golden_dir, "schedules_kmeans.npz")
    z = torch.from_numpy(d["km_z"]).requires_grad_(True)
    km = OV.kmeans_gram_loss(z, 1.3)
    km.backward()
    np.testing.assert_allclose(float(km), float(d["km_val"]), rtol=1e-9)
    np.testing.assert_allclose(z.grad.numpy(), d["km_grad"], atol=1e-7, rtol=1e-5)


@pytest.mark.parametrize("tag", ["rec14", "rec28", "c5l8", "c3k512", "rec14l16", "rec14l32", "rec14l4", "rec14l5", "rec14l6", "rec14l12", "rec14l24", "rec14l14"])
def test_vqvae_forward_loss_grads_trace(golden_dir, tag):
    from oracle import vqvae as OQ
    d = _load(golden_dir, f"vqvae_{tag}.npz")
    P = _params(d)
    km = float(d["kmeans"])
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    losses, grads, out = OQ.vqvae_grads(P, x, a, 1.0, km)
    np.testing.assert_array_equal(out["idx"].numpy(), d["idx"])
    np.testing.assert_allclose(out["ze"].detach().numpy(), d["ze"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(out["quantized"].detach().numpy(), d["quantized"], atol=1e-6)
    np.testing.assert_allclose(out["soft_counts"].detach().numpy(), d["soft_counts"], atol=1e-6, rtol=2e-4)
    np.testing.assert_allclose(out["loc_q"].detach().numpy(), d["loc_q"], atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(out["loc_e"].detach().numpy(), d["loc_e"], atol=5e-6, rtol=1e-5)
    for k, v in losses.items():
        np.testing.assert_allclose(float(v), float(d[f"log::{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    n = 0
    for k in d:
        if k.startswith("grad::"):
            name = k[6:]
            np.testing.assert_allclose(grads[name].numpy(), d[k], atol=2e-5, rtol=2e-4, err_msg=name)
            n += 1
    assert n >= 70
    for name, g in grads.items():
        if f"grad::{name}" not in d:
            assert g is None, name
    opt = OV.AdamState()
    for i in range(3):
        logs, _, _ = OQ.vqvae_train_step(P, opt, torch.from_numpy(d[f"step{i}::x"]), torch.from_numpy(d[f"step{i}::a"]),
                                         1e-3, 1e-4, 0.75, 1.0, km)
        for k, v in logs.items():
            np.testing.assert_allclose(v, float(d[f"step{i}::log::{k}"]), rtol=5e-4, atol=5e-5, err_msg=f"{i}:{k}")
    for k, v in _params(d, "sd_final::").items():
        np.testing.assert_allclose(P[k].numpy(), v.numpy(), atol=2e-4, rtol=1e-3, err_msg=k)


# ----------------------------------------------------------------------------- contrastive (R13/R14)
def _aug_draws(d, pfx):
    from oracle import contrastive as OC
    mask = d[pfx + "aug::rot_mask"]
    return OC.AugDraws(start=torch.from_numpy(d[pfx + "aug::start"]),
                       rot_pivot=[int(v) for v in d[pfx + "aug::rot_pivot"]],
                       rot_nodes=[np.nonzero(m)[0].tolist() for m in mask],
                       theta=torch.from_numpy(d[pfx + "aug::theta"]),
                       interp_t0=torch.from_numpy(d[pfx + "aug::interp_t0"]),
                       interp_len=torch.from_numpy(d[pfx + "aug::interp_len"]),
                       noise=torch.from_numpy(d[pfx + "aug::noise"]))


@pytest.mark.parametrize("tag", ["rec14", "rec28", "c5l8", "rec14l16", "rec14l32", "rec14l4", "rec14l5", "rec14l6", "rec14l12", "rec14l24", "rec14l7"])
def test_contrastive_losses_match_reference(golden_dir, tag):
    from oracle import contrastive as OC
    d = _load(golden_dir, f"contrastive_{tag}.npz")
    for sim in ("cosine", "dot", "euclidean", "edit"):
        for lf in ("nce", "dcl", "fc", "hard_dcl"):
            z = torch.from_numpy(d["loss_z"]).requires_grad_(True)
            za = torch.from_numpy(d["loss_za"]).requires_grad_(True)
            l, p, n = OC.contrastive_loss(z, za, sim, lf, 0.1, 0.1, 0.1)
            np.testing.assert_allclose([float(l), float(p), float(n)], d[f"loss::{sim}::{lf}"], rtol=1e-5, atol=1e-6,
                                       err_msg=f"{sim}/{lf}")
            g = torch.autograd.grad(l, [z, za])
            np.testing.assert_allclose(np.stack([g[0].numpy(), g[1].numpy()]), d[f"loss_grad::{sim}::{lf}"],
                                       rtol=1e-4, atol=1e-6, err_msg=f"{sim}/{lf}")


@pytest.mark.parametrize("tag,ids", [("rec14", [""]), ("rec28", ["B", "W"]), ("c5l8", ["B", "W"]), ("rec14l16", [""]), ("rec14l32", [""]), ("rec14l4", [""]), ("rec14l5", [""]), ("rec14l6", [""]), ("rec14l7", [""]), ("rec14l12", [""]), ("rec14l24", [""])])
def test_contrastive_views_and_step_match_reference(golden_dir, tag, ids):
    from oracle import contrastive as OC
    d = _load(golden_dir, f"contrastive_{tag}.npz")
    nodes, edges = G.bodypart_graph(ids)
    ei, ei_local = G.edge_index_from_graph(nodes, edges)
    np.testing.assert_array_equal(ei, d["edge_index"])
    np.testing.assert_array_equal(ei_local, d["edge_index_local"])
    x_full = torch.from_numpy(d["x_full"])
    eit = torch.from_numpy(d["edge_index"]).long()
    for ci in range(3):
        pfx = f"c{ci}::"
        dr = _aug_draws(d, pfx)
        assert (d[pfx + "aug::interp_len"] > 0).any() and np.abs(d[pfx + "aug::theta"]).max() > 0
        xa, aa = OC.augmented_view(x_full, eit, dr)
        np.testing.assert_allclose(xa.numpy(), d[pfx + "x_aug"], atol=1e-6)
        np.testing.assert_allclose(aa.numpy(), d[pfx + "a_aug"], atol=2e-6)
        xc, ac = OC.central_view(x_full, eit)
        np.testing.assert_array_equal(xc.numpy(), d[pfx + "x"])
        np.testing.assert_allclose(ac.numpy(), d[pfx + "a"], atol=1e-7)
        P = _params(d, pfx + "sd::")
        logs, grads, aux = OC.contrastive_grads(P, x_full, eit, dr, sim_kind=str(d[pfx + "sim"]),
                                                loss_fn=str(d[pfx + "loss_fn"]), temperature=0.1, tau=0.1, beta=0.1)
        np.testing.assert_allclose(aux["z"].detach().numpy(), d[pfx + "z"], atol=3e-6, rtol=1e-5)
        np.testing.assert_allclose(aux["z_aug"].detach().numpy(), d[pfx + "z_aug"], atol=3e-6, rtol=1e-5)
        for k in ("total_loss", "pos_similarity", "neg_similarity"):
            np.testing.assert_allclose(logs[k], float(d[pfx + f"log::{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
        n = 0
        for k in d:
            if k.startswith(pfx + "grad::"):
                name = k[len(pfx) + 6:]
                np.testing.assert_allclose(grads[name].numpy(), d[k], atol=2e-5, rtol=3e-4, err_msg=name)
                n += 1
        assert n >= 40


def test_rotation_triplets_choice():
    from oracle import contrastive as OC
    trips, ba, bc = OC.rotation_triplets([(0, 1), (1, 2), (1, 3), (3, 4)], 5)
    assert trips == [(0, 1, 2), (0, 1, 3), (2, 1, 3), (1, 3, 4)]
    assert ba[1] == [0] and bc[1] == [3, 4] and ba[3] == [0, 1, 2] and bc[3] == [4]
    assert OC.choose_rotations([0, 1, 2, 3], [t[1] for t in trips], 5, 3) == [0, 1, 3]


@pytest.mark.parametrize("fixture", ["contrastive_tcn14.npz", "contrastive_tcn14l16.npz"])
def test_tcn_contrastive_matches_reference(golden_dir, fixture):
    """TCN encoder (R12) inside the contrastive step: train-mode embeddings, BatchNorm running buffers, eval-mode
    embeddings, loss and every gradient; then the two recorded optimiser steps (CensNet frozen, quirk Q11)."""
    from oracle import contrastive as OC
    from oracle import tcn as OT
    from parity_common import math_zero_gradient
    d = _load(golden_dir, fixture)
    pfx = "c0::"
    x_full = torch.from_numpy(d["x_full"])
    eit = torch.from_numpy(d["edge_index"]).long()
    P = _params(d, pfx + "sd::")
    with torch.no_grad():
        z_eval = OT.tcn_encoder(torch.from_numpy(d[pfx + "x"]), torch.from_numpy(d[pfx + "a"]),
                                {k: v.clone() for k, v in P.items()}, False)
    np.testing.assert_allclose(z_eval.numpy(), d[pfx + "z_eval"], atol=2e-6, rtol=1e-5)
    logs, grads, aux = OC.contrastive_grads(P, x_full, eit, _aug_draws(d, pfx), sim_kind="cosine", loss_fn="nce")
    np.testing.assert_allclose(aux["z"].detach().numpy(), d[pfx + "z"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(aux["z_aug"].detach().numpy(), d[pfx + "z_aug"], atol=3e-6, rtol=1e-5)
    for k, v in aux["buffers"].items():
        np.testing.assert_allclose(v.numpy(), d[pfx + "sd_after::" + k], atol=1e-6, rtol=1e-5, err_msg=k)
    for k in ("total_loss", "pos_similarity", "neg_similarity"):
        np.testing.assert_allclose(logs[k], float(d[pfx + f"log::{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    n = 0
    for k in d:
        if k.startswith(pfx + "grad::"):
            name = k[len(pfx) + 6:]
            # (conv biases feeding a BatchNorm have an exactly-zero true gradient: pure rounding noise there, only
            # bounded -- parity_common.math_zero_gradient, the same rule as the device checks)
            if math_zero_gradient(name):
                assert max(np.abs(grads[name].numpy()).max(), np.abs(d[k]).max()) < 3e-4, name
            else:
                np.testing.assert_allclose(grads[name].numpy(), d[k], atol=6e-5, rtol=5e-4, err_msg=name)
            n += 1
    assert n == 148


@pytest.mark.parametrize("fixture", ["vade_tcn14.npz", "vade_tcn14w50.npz"])
def test_vade_tcn_matches_reference(golden_dir, fixture):
    """VaDE with the TCN encoder AND decoder (R12): eval forward on running statistics (bit-stable), then the
    train-mode step for the pre-training and the main (+teacher) objective.  BatchNorm over 6 windows amplifies
    fp32 rounding to ~1e-4 relative in the gradients, so the golden holds the reference evaluated in float64 plus
    the reference's own fp32 deviation from it per tensor ("noise"); the oracle must sit within a few noise units."""
    d = _load(golden_dir, fixture)
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    K, L = d["sd::latent_space.gmm_means"].shape
    P0 = _params(d)
    with torch.no_grad():
        out = OV.vade_forward({k: v.clone() for k, v in P0.items()}, x, a, training=False)
    np.testing.assert_allclose(out["enc"].numpy(), d["eval_enc"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(out["z"].numpy(), d["eval_z"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(out["q"].numpy(), d["eval_q"], atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(out["loc"].numpy(), d["eval_loc"], atol=1e-5, rtol=1e-5)
    eps, eps_mc, tau = (torch.from_numpy(d[k]) for k in ("eps", "eps_mc", "tau"))
    for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
        P = {k: v.clone() for k, v in P0.items()}
        kw = {}
        if teacher:
            pi = tau.mean(0).clamp_min(1e-8)
            w = pi.pow(-1.0)
            kw = dict(lambda_distill=1.7, class_weight=(w / w.mean()).clamp_max(3.0), teacher_marginal=pi)
        cfg = OV.VadeLossCfg(K, phase == "pre", **kw)
        losses, grads, out = OV.vade_grads(P, x, a, cfg, klw, eps, None if phase == "pre" else eps_mc,
                                           tau if teacher else None)
        for key in ("z", "loc"):
            err = np.abs(out[key].detach().numpy() - d[f"{phase}::{key}"]).max()
            assert err <= 3.0 * float(d[f"{phase}::noise::{key}"]) + 1e-6, (phase, key, err)
        for k in d:
            if k.startswith(f"{phase}::loss::"):
                name = k.split("::")[-1]
                if name in losses:
                    np.testing.assert_allclose(float(losses[name]), float(d[k]), rtol=2e-5, atol=2e-6, err_msg=k)
        n = 0
        for k in d:
            if k.startswith(f"{phase}::grad::"):
                name = k.split("::")[-1]
                err = np.abs(grads[name].numpy() - d[k]).max()
                assert err <= 3.0 * float(d[f"{phase}::gnoise::{name}"]) + 1e-6 * np.abs(d[k]).max() + 1e-7, (phase, name, err)
                n += 1
        assert n >= (200 if phase == "pre" else 10)
        if phase == "pre":
            for k in d:
                if k.startswith("pre::sd_after::"):
                    np.testing.assert_allclose(P[k[len("pre::sd_after::"):]].numpy(), d[k], atol=2e-6, rtol=2e-5, err_msg=k)


def _turtle_inputs(d):
    K, B, nb, inner, outer = (int(v) for v in d["cfg"])
    dims = [int(v) for v in d["dims"]]
    P = {k[6:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("init::")}
    batches = [[torch.from_numpy(d[f"batch{i}::{v}"]) for v in range(len(dims))] for i in range(nb)]
    return K, B, inner, outer, dims, P, batches


def test_turtle_teacher_matches_reference(golden_dir):
    """TURTLE teacher (N3): 7 outer x 12 inner steps from recorded initial weights over a fixed batch list, the
    prediction pass, and the GMM initialisation from tau*."""
    from oracle import turtle as OT
    d = _load(golden_dir, "turtle.npz")
    K, B, inner, outer, dims, P, batches = _turtle_inputs(d)
    OT.fit(P, batches, K, outer, inner, gamma=8.0, alpha=2.0, delta=40.0, head_temp=0.35, task_temp=0.35, lr_theta=1e-3)
    for k, v in d.items():
        if k.startswith("final::"):
            np.testing.assert_allclose(P[k[7:]].numpy(), v, atol=2e-6, rtol=2e-5, err_msg=k)
    tau = torch.cat([OT.predict(P, batches[0], 0.35), OT.predict(P, batches[1], 0.35)])
    np.testing.assert_allclose(tau.numpy(), d["tau_star"], atol=1e-6, rtol=1e-5)
    z = torch.cat([batches[0][0], batches[1][0]])
    m, lv, pr = OT.gmm_from_teacher(z, tau, min_var=0.01)
    np.testing.assert_allclose(m.numpy(), d["gmm_means"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(lv.numpy(), d["gmm_log_vars"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(pr.numpy(), d["gmm_prior"], atol=1e-6, rtol=1e-5)


def test_distillation_head_matches_reference(golden_dir):
    """Generic distillation head (DiscriminativeHead) inside the VQ-VAE and the contrastive step: logged terms and
    every gradient incl. the head's own."""
    from oracle import vqvae as OQ
    from oracle import contrastive as OC
    d = _load(golden_dir, "vqvae_rec28.npz")
    P = _params(d, "sd_final::")
    P.update({"distill_head." + k: v for k, v in _params(d, "dist::head::").items()})
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    dist = dict(tau_b=torch.from_numpy(d["dist::tau"]), lam=1.3, T=0.5, conf_weight=True, thr=0.2)
    losses, grads, _ = OQ.vqvae_grads(P, x, a, 1.0, float(d["kmeans"]), distill=dist)
    for k in ("total_loss", "distill_loss", "reconstruct_loss", "enc_rec_loss"):
        np.testing.assert_allclose(float(losses[k]), float(d[f"dist::log::{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    for k in d:
        if k.startswith("dist::grad::"):
            np.testing.assert_allclose(grads[k[12:]].numpy(), d[k], atol=2e-5, rtol=3e-4, err_msg=k)
    d = _load(golden_dir, "contrastive_rec28.npz")
    pfx = "c0::"
    P = _params(d, pfx + "sd::")
    P.update({"distill_head." + k: v for k, v in _params(d, "dist::head::").items()})
    dist = dict(tau_b=torch.from_numpy(d["dist::tau"]), lam=0.9, T=0.5, conf_weight=True, thr=0.2)
    logs, grads, _ = OC.contrastive_grads(P, torch.from_numpy(d["x_full"]), torch.from_numpy(d["edge_index"]).long(),
                                          _aug_draws(d, pfx), sim_kind=str(d[pfx + "sim"]), loss_fn=str(d[pfx + "loss_fn"]),
                                          distill=dist)
    for k in ("total_loss", "distill_loss", "pos_similarity"):
        np.testing.assert_allclose(logs[k], float(d[f"dist::log::{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    for k in d:
        if k.startswith("dist::grad::"):
            np.testing.assert_allclose(grads[k[12:]].numpy(), d[k], atol=2e-5, rtol=3e-4, err_msg=k)


# ---- pose-table preprocessing ---------------------------------------------------------
def load_preprocess_golden():
    import json
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess.npz"))
    cases = json.loads(str(g["cases"]))
    data = {}
    for tag in ("pair", "single"):
        cols = [tuple(c) if isinstance(c, list) else c for c in json.loads(str(g[f"{tag}::columns"]))]
        tabs = {k.split("::")[-1]: g[k] for k in g.files if k.startswith(f"{tag}::raw::")}
        data[tag] = (cols, json.loads(str(g[f"{tag}::animal_ids"])), tabs)
    return g, cases, data


def test_preprocess_oracle_matches_reference():
    from oracle import preprocess as op
    g, cases, data = load_preprocess_golden()
    for c in cases:
        cols, aids, tabs = data[c["data"]]
        out, gs = op.preprocess(tabs, cols, aids, samples_max=c["samples_max"], dist_standardize=c["dist"],
                                speed_standardize=c["speed"], coord_standardize=c["coord"], log_distances=c["log"],
                                interpolate_normalized=c["clip"])
        exp = {k.split("::")[-1]: g[k] for k in g.files if k.startswith(c["case"] + "::out::")}
        assert sorted(out) == sorted(exp), c["case"]
        for k in exp:
            assert np.isfinite(out[k]).all()
            np.testing.assert_allclose(out[k], exp[k], rtol=1e-11, atol=1e-11, err_msg=f"{c['case']} {k}")
        for part in ("speed", "dist", "dist_inner", "dist_intra", "coord"):
            key = f"{c['case']}::scaler::{part}::mean"
            assert (key in g.files) == (part in gs), (c["case"], part)
            if part in gs:
                np.testing.assert_allclose(np.atleast_1d(gs[part][0]), g[key], rtol=1e-11, atol=1e-12)
                np.testing.assert_allclose(np.atleast_1d(gs[part][1]), g[key.replace("mean", "scale")], rtol=1e-11, atol=1e-12)
    # pretrained scaler re-applied to new videos
    cols, aids, _ = data["pair"]
    c = cases[0]
    _, gs = op.preprocess(data["pair"][2], cols, aids, dist_standardize=c["dist"], speed_standardize=c["speed"],
                          coord_standardize=c["coord"])
    new = {k.split("::")[-1]: g[k] for k in g.files if k.startswith("pair::pre::raw::")}
    out, _ = op.preprocess(new, cols, aids, dist_standardize=c["dist"], speed_standardize=c["speed"], coord_standardize=c["coord"],
                           pretrained_scaler=gs)
    for k in new:
        np.testing.assert_allclose(out[k], g[f"pair::pre::out::{k}"], rtol=1e-11, atol=1e-11)


def test_scale_table_oracle_matches_reference():
    from oracle import preprocess as op
    g, _, data = load_preprocess_golden()
    cols, aids, tabs = data["pair"]
    t = tabs["vid0"]
    runs = {"size_only": dict(standardize=False), "geom": dict(inter_scale="geom", standardize=False), "full_pc": dict(),
            "full_gw": dict(dist_standardize="groupwise", speed_standardize="groupwise", coord_standardize="groupwise"),
            "infer_ids": dict(animal_ids=None)}
    for nm, kw in runs.items():
        kw = dict(kw)
        aid = kw.pop("animal_ids", aids)
        np.testing.assert_allclose(op.scale_table(t, cols, aid, **kw), g[f"scale_table::{nm}"], rtol=1e-12, atol=1e-12,
                                   equal_nan=True, err_msg=nm)


def test_preprocess_r03_oracle_matches_reference(golden_dir):
    """scale="minmax" and filter_low_variance (tests/golden/make_golden_preprocess_r03.py: the reference's own functions)."""
    from oracle import preprocess as op
    import parity_common as PC
    g, cases, data = PC.load_preprocess_r03_golden(golden_dir)
    for c in cases:
        cols, aids, tabs = data[c["data"]]
        out, gs = op.preprocess(tabs, cols, aids, samples_max=c["samples_max"], dist_standardize=c["dist"],
                                speed_standardize=c["speed"], coord_standardize=c["coord"], log_distances=c["log"],
                                interpolate_normalized=c["clip"], scale=c["scale"], filter_low_variance=c["filter"])
        exp = {k.split("::")[-1]: g[k] for k in g.files if k.startswith(c["case"] + "::out::")}
        assert sorted(out) == sorted(exp), c["case"]
        for k in exp:
            assert np.isfinite(out[k]).all()
            np.testing.assert_allclose(out[k], exp[k], rtol=1e-11, atol=1e-11, err_msg=f"{c['case']} {k}")
        for part in ("speed", "dist", "dist_inner", "dist_intra", "coord"):
            if c["scale"] == "minmax":
                key = f"{c['case']}::scaler::{part}::data_min"
                assert (key in g.files) == (part in gs), (c["case"], part)
                if part in gs:     # oracle pair = (scale_, min_) of sklearn's MinMaxScaler
                    rng = g[key.replace("data_min", "data_range")].copy()
                    rng[rng < 10 * np.finfo(np.float64).eps] = 1.0
                    np.testing.assert_allclose(np.atleast_1d(gs[part][0]), 1.0 / rng, rtol=1e-12, atol=0)
                    np.testing.assert_allclose(np.atleast_1d(gs[part][1]), -g[key] / rng, rtol=1e-11, atol=1e-14)
            else:
                key = f"{c['case']}::scaler::{part}::" + ("center" if c["scale"] == "robust" else "mean")
                assert (key in g.files) == (part in gs), (c["case"], part)
                if part in gs:
                    np.testing.assert_allclose(np.atleast_1d(gs[part][0]), g[key], rtol=1e-11, atol=1e-12)
                    np.testing.assert_allclose(np.atleast_1d(gs[part][1]), g[key.rsplit("::", 1)[0] + "::scale"], rtol=1e-11, atol=1e-12)
    cols, aids, tabs = data["pair"]
    for nm, kw in {"mm_pc": dict(), "mm_gw": dict(dist_standardize="groupwise", speed_standardize="groupwise",
                                                  coord_standardize="groupwise")}.items():
        np.testing.assert_allclose(op.scale_table(tabs["vid0"], cols, aids, scale="minmax", **kw), g[f"scale_table::{nm}"],
                                   rtol=1e-12, atol=1e-12, equal_nan=True, err_msg=nm)
    c = cases[0]
    kw = dict(dist_standardize=c["dist"], speed_standardize=c["speed"], coord_standardize=c["coord"], scale="minmax")
    _, gs = op.preprocess(tabs, cols, aids, **kw)
    new = {k.split("::")[-1]: g[k] for k in g.files if k.startswith("pair::pre::raw::")}
    out, _ = op.preprocess(new, cols, aids, pretrained_scaler=gs, **kw)
    for k in new:
        np.testing.assert_allclose(out[k], g[f"pair::pre::out::{k}"], rtol=1e-11, atol=1e-11)


from parity_common import math_zero_gradient  # noqa: E402


@pytest.mark.parametrize("fixture", ["vade_tcn14_b64.npz", "vade_tcn14_onepass.npz"])
def test_vade_tcn_b64_matches_reference(golden_dir, fixture):
    """oracle/tcn.py + oracle/vade.py against the B = 64 trained-like VaDE-TCN goldens (reference fp32 values); the
    "onepass" fixture has every BatchNorm running mean at the batch mean of the recorded step (make_golden_r03.py)."""
    d = _load(golden_dir, fixture)
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    K, L = d["sd::latent_space.gmm_means"].shape
    P0 = _params(d)
    with torch.no_grad():
        out = OV.vade_forward({k: v.clone() for k, v in P0.items()}, x, a, training=False)
    np.testing.assert_allclose(out["z"].numpy(), d["eval_z"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(out["q"].numpy(), d["eval_q"], atol=2e-6, rtol=2e-4)
    np.testing.assert_allclose(out["loc"].numpy(), d["eval_loc"], atol=1e-5, rtol=1e-5)
    eps, eps_mc, tau = (torch.from_numpy(d[k]) for k in ("eps", "eps_mc", "tau"))
    for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
        P = {k: v.clone() for k, v in P0.items()}
        kw = {}
        if teacher:
            pi = tau.mean(0).clamp_min(1e-8)
            w = pi.pow(-1.0)
            kw = dict(lambda_distill=1.7, class_weight=(w / w.mean()).clamp_max(3.0), teacher_marginal=pi)
        losses, grads, out = OV.vade_grads(P, x, a, OV.VadeLossCfg(K, phase == "pre", **kw), klw, eps,
                                           None if phase == "pre" else eps_mc, tau if teacher else None)
        for k in d:
            if k.startswith(f"{phase}::loss::") and k.split("::")[-1] in losses:
                np.testing.assert_allclose(float(losses[k.split("::")[-1]]), float(d[k]), rtol=2e-5, atol=2e-6, err_msg=k)
        n = 0
        for k in d:
            if k.startswith(f"{phase}::grad::"):
                name = k.split("::")[-1]
                ref = d[k]
                scale = np.abs(ref).max()
                if math_zero_gradient(name):  # rounding noise of a mathematical zero (bias in front of a BatchNorm)
                    assert scale < 3e-4 and np.abs(grads[name].numpy()).max() < 3e-4, name
                else:
                    assert np.abs(grads[name].numpy() - ref).max() <= 2e-5 + 3e-4 * scale, (phase, name)
                n += 1
        assert n >= (200 if phase == "pre" else 20)


def test_vqvae_tcn_matches_reference(golden_dir):
    """oracle/vqvae.py (TCN family) against the reference VQVAEPT(encoder_type="TCN") golden."""
    from oracle import vqvae as OQ
    d = _load(golden_dir, "vqvae_tcn14.npz")
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    P0 = _params(d)
    with torch.no_grad():
        ev = OQ.vqvae_forward({k: v.clone() for k, v in P0.items()}, x, a, training=False)
    np.testing.assert_array_equal(ev["idx"].numpy(), d["eval_idx"])
    np.testing.assert_allclose(ev["ze"].numpy(), d["eval_ze"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(ev["loc_q"].numpy(), d["eval_loc_q"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ev["loc_e"].numpy(), d["eval_loc_e"], atol=1e-5, rtol=1e-5)
    P = {k: v.clone() for k, v in P0.items()}
    losses, grads, _ = OQ.vqvae_grads(P, x, a, 1.0, 0.0)
    for k in ("total_loss", "enc_rec_loss", "reconstruct_loss", "vq_loss"):
        np.testing.assert_allclose(float(losses[k]), float(d[f"log::{k}"]), rtol=2e-5, atol=2e-6, err_msg=k)
    n = 0
    for k in d:
        if k.startswith("grad::"):
            name, ref = k[len("grad::"):], d[k]
            scale = np.abs(ref).max()
            if math_zero_gradient(name):
                assert scale < 3e-4 and np.abs(grads[name].numpy()).max() < 3e-4, name
            else:
                assert np.abs(grads[name].numpy() - ref).max() <= 2e-5 + 3e-4 * scale, name
            n += 1
    assert n >= 190
    for k in d:  # BatchNorm buffers after the train-mode step (encoder once, decoder twice)
        if k.startswith("sd_after::") and "running_" in k:
            np.testing.assert_allclose(P[k[len("sd_after::"):]].numpy(), d[k], atol=2e-6, rtol=2e-5, err_msg=k)


# ------------------------------------------------------------------------------------------------
# transformer family (R17): oracle/tfm.py against the reference goldens of tests/golden/make_golden_tfm.py
# ------------------------------------------------------------------------------------------------
def _tape(d, prefix=""):
    from oracle.tfm import DropoutTape
    keys = sorted(k for k in d if k.startswith(prefix + "drop::"))
    return DropoutTape([torch.from_numpy(d[k]) for k in keys])


def _check_grads(d, grads, prefix, min_count, zero_names=("encoder.head.6.bias", "encoder.head.5.bias")):
    n = 0
    for k in d:
        if k.startswith(prefix + "grad::"):
            name, ref = k.split("::")[-1], d[k]
            scale = np.abs(ref).max()
            got = grads[name].numpy()
            if name in zero_names:  # a constant in front of the batch standardisation: rounding noise on both sides
                assert scale < 1e-4 and np.abs(got).max() < 1e-4, name
            else:
                noise = float(d[prefix + "gnoise::" + name])
                kink = float(d[prefix + "gkink::" + name])  # ReLU inputs within 2e-6 of zero: undecidable signs
                assert np.abs(got - ref).max() <= 2e-5 + 3e-4 * scale + 4 * noise + kink, (prefix, name)
            n += 1
    assert n >= min_count, n


def test_vade_tfm_matches_reference(golden_dir):
    """VaDEPT(encoder_type="transformer"): eval forward (with padded keys / masked frames), two train steps on the
    recorded dropout masks: every loss term, all gradients, BatchNorm buffers."""
    d = _load(golden_dir, "vade_tfm14.npz")
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    K, L = d["sd::latent_space.gmm_means"].shape
    P0 = _params(d)
    with torch.no_grad():
        out = OV.vade_forward({k: v.clone() for k, v in P0.items()}, x, a, training=False)
        outm = OV.vade_forward({k: v.clone() for k, v in P0.items()}, torch.from_numpy(d["xm"]), torch.from_numpy(d["am"]),
                               training=False)
    for o, pfx in ((out, "eval_"), (outm, "evalm_")):
        np.testing.assert_allclose(o["z"].numpy(), d[pfx + "z"], atol=5e-6, rtol=1e-5)
        np.testing.assert_allclose(o["q"].numpy(), d[pfx + "q"], atol=2e-6, rtol=2e-4)
        np.testing.assert_allclose(o["loc"].numpy(), d[pfx + "loc"], atol=2e-5, rtol=1e-5)
    assert np.abs(d["evalm_z"] - d["eval_z"]).max() > 1e-3  # the zeroed frames matter
    eps, eps_mc, tau = (torch.from_numpy(d[k]) for k in ("eps", "eps_mc", "tau"))
    for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
        P = {k: v.clone() for k, v in P0.items()}
        kw = {}
        if teacher:
            pi = tau.mean(0).clamp_min(1e-8)
            w = pi.pow(-1.0)
            kw = dict(lambda_distill=1.7, class_weight=(w / w.mean()).clamp_max(3.0), teacher_marginal=pi)
        tape = _tape(d, phase + "::")
        losses, grads, out = OV.vade_grads(P, x, a, OV.VadeLossCfg(K, phase == "pre", **kw), klw, eps,
                                           None if phase == "pre" else eps_mc, tau if teacher else None, drop=tape)
        assert tape.pos == len(tape.masks) == 2 * 7 + 2 * 4
        for k in d:
            if k.startswith(f"{phase}::loss::") and k.split("::")[-1] in losses:
                np.testing.assert_allclose(float(losses[k.split("::")[-1]]), float(d[k]), rtol=3e-5, atol=3e-6, err_msg=k)
        np.testing.assert_allclose(out["z"].detach().numpy(), d[f"{phase}::z"], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(out["loc"].detach().numpy(), d[f"{phase}::loc"], atol=5e-5, rtol=1e-4)
        _check_grads(d, grads, phase + "::", 100)
        if phase == "pre":
            for k in d:
                if k.startswith("pre::sd_after::") and "running_" in k:
                    np.testing.assert_allclose(P[k[len("pre::sd_after::"):]].numpy(), d[k], atol=2e-6, rtol=2e-5, err_msg=k)


def test_vqvae_tfm_matches_reference(golden_dir):
    from oracle import vqvae as OQ
    d = _load(golden_dir, "vqvae_tfm14.npz")
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    P0 = _params(d)
    with torch.no_grad():
        ev = OQ.vqvae_forward({k: v.clone() for k, v in P0.items()}, x, a, training=False)
    np.testing.assert_array_equal(ev["idx"].numpy(), d["eval_idx"])
    np.testing.assert_allclose(ev["ze"].numpy(), d["eval_ze"], atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(ev["loc_q"].numpy(), d["eval_loc_q"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(ev["loc_e"].numpy(), d["eval_loc_e"], atol=2e-5, rtol=1e-5)
    P = {k: v.clone() for k, v in P0.items()}
    tape = _tape(d)
    losses, grads, _ = OQ.vqvae_grads(P, x, a, 1.0, 0.0, drop=tape)
    assert tape.pos == len(tape.masks) == 2 * 7 + 2 * 2 * 4
    for k in ("total_loss", "enc_rec_loss", "reconstruct_loss", "vq_loss"):
        np.testing.assert_allclose(float(losses[k]), float(d[f"log::{k}"]), rtol=3e-5, atol=3e-6, err_msg=k)
    _check_grads(d, grads, "", 100)
    for k in d:
        if k.startswith("sd_after::") and "running_" in k:
            np.testing.assert_allclose(P[k[len("sd_after::"):]].numpy(), d[k], atol=2e-6, rtol=2e-5, err_msg=k)


def test_contrastive_tfm_matches_reference(golden_dir):
    from oracle import contrastive as OC
    d = _load(golden_dir, "contrastive_tfm14.npz")
    P0 = _params(d)
    x, a, xa, aa = (torch.from_numpy(d[k]) for k in ("x", "a", "x_aug", "a_aug"))
    with torch.no_grad():
        z = OC.encode({k: v.clone() for k, v in P0.items()}, x, a, training=False)
    np.testing.assert_allclose(z.numpy(), d["eval_z"], atol=5e-6, rtol=1e-5)
    buffers = ("laplacian", "edge_laplacian", "incidence", "running_mean", "running_var", "num_batches_tracked")
    P = {k: (v.clone().requires_grad_(True) if k.split(".")[-1] not in buffers else v.clone()) for k, v in P0.items()}
    tape = _tape(d)
    z = OC.encode(P, x, a, True, tape)
    za = OC.encode(P, xa, aa, True, tape)
    assert tape.pos == len(tape.masks) == 2 * 2 * 7
    np.testing.assert_allclose(z.detach().numpy(), d["z"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(za.detach().numpy(), d["z_aug"], atol=2e-5, rtol=1e-4)
    loss, pos, neg = OC.contrastive_loss(torch.nn.functional.normalize(z, dim=1), torch.nn.functional.normalize(za, dim=1),
                                         "cosine", "nce", 0.1, 0.1, 0.1)
    np.testing.assert_allclose([float(loss), float(pos), float(neg)], d["loss"], rtol=3e-5, atol=3e-6)
    names = [k for k, v in P.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
    _check_grads(d, dict(zip(names, gs)), "", 60)
    for k in d:
        if k.startswith("sd_after::") and "running_" in k:
            np.testing.assert_allclose(P[k[len("sd_after::"):]].detach().numpy(), d[k], atol=2e-6, rtol=2e-5, err_msg=k)


def test_philox_known_answers():
    """oracle.noise's Philox-4x32-10 against the Random123 known-answer vectors (kat_vectors: zero, all ones, digits of
    pi) -- the generator dof_step_begin's noise is compared with."""
    from oracle.noise import philox4x32_10, normal_fill
    kat = [([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0),
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        got = philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert [int(v) for v in got] == want
    x = normal_fill(200_000, 7, 0, 3).astype(np.float64)
    assert abs(x.mean()) < 0.01 and abs(x.var() - 1.0) < 0.01 and abs((x ** 4).mean() - 3.0) < 0.1
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.01
    assert not np.array_equal(x[:100], normal_fill(100, 7, 0, 4)) and not np.array_equal(x[:100], normal_fill(100, 7, 1, 3))
