"""Shared parity checks (used with the emulator on CPU and with the real library on the GPU)."""
import os

import numpy as np
import pytest
import torch

from deepof_amd import _capi
from deepof_amd.engine import VadeEngine
from oracle import windows as OW


def load_golden(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


def params_from(d, prefix="sd::"):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items() if k.startswith(prefix)}


PHASES = {
    "pre": dict(klw=0.13, pretrain=True, teacher=False),
    "main": dict(klw=0.7, pretrain=False, teacher=False),
    "mainT": dict(klw=0.7, pretrain=False, teacher=True),
    # every optional regulariser switched on at once (reference defaults are 0 for these)
    "mainX": dict(klw=0.45, pretrain=False, teacher=True,
                  extra=dict(repel_w=0.3, repel_ls=1.0, scatter_w=0.2, scatter_beta=1.0, temporal_w=0.1, cat_w=0.5,
                             tf_w=0.7, km_loss=0.5, conf_w=1.0, conf_thr=0.3)),
}


def configure_phase(eng, K, pretrain, klw, tau=None, lambda_distill=0.0, extra=None):
    """Reference defaults of VadeLoss per phase (VadeEngine.configure_vade_phase)."""
    assert K == eng.K
    eng.configure_vade_phase(pretrain, klw, tau, lambda_distill, extra)


def gather_check(lib, device):
    rng = np.random.default_rng(0)
    # (staged-in-LDS path: consecutive / clustered starts incl. a ragged last workgroup and both buffer classes;
    #  direct path: starts too far apart for the staging buffer)
    for (N, E, W, Fr, starts) in [(14, 14, 25, 90, [0, 1, 2, 3, 40, 65, 7]), (5, 4, 7, 31, list(range(25))),
                                  (28, 32, 50, 120, [70, 0, 33]), (14, 14, 25, 120, list(range(3, 3 + 37))),
                                  (28, 32, 50, 130, list(range(60, 60 + 21))), (14, 14, 25, 90, [9, 4, 11, 4, 30, 12]),
                                  (42, 47, 50, 140, list(range(0, 70, 3))),
                                  # > 8192 windows: the 16-windows-per-workgroup variant (smaller launches use 4)
                                  (5, 4, 7, 8400, list(range(1, 1 + 8213)))]:
        nodes = rng.standard_normal((Fr, 3 * N)).astype(np.float32)
        edges = rng.standard_normal((Fr, E)).astype(np.float32)
        # non-finite table entries travel unchanged (fp32) / as Inf and the canonical NaN (bf16 storage)
        nodes.view(np.uint32)[5, :5] = [0x7F800001, 0x7FFFFFFF, 0xFFFFFFFF, 0x7F800000, 0xFF800000]
        edges.view(np.uint32)[4, :3] = [0x7FC00000, 0xFF800001, 0x7F7FFFFF]
        x_ref, a_ref = OW.gather_windows(nodes, edges, np.array(starts), W)
        tn, te = torch.from_numpy(nodes).to(device), torch.from_numpy(edges).to(device)
        st = torch.tensor(starts, dtype=torch.int64, device=device)
        x = torch.full((len(starts), W, N, 3), float("nan"), device=device)
        a = torch.full((len(starts), W, E, 1), float("nan"), device=device)
        stream = torch.cuda.current_stream().cuda_stream if device != "cpu" else 0
        _capi.check(lib, lib.dof_window_gather(tn.data_ptr(), te.data_ptr(), st.data_ptr(), len(starts), W, N, E,
                                               x.data_ptr(), a.data_ptr(), stream))
        np.testing.assert_array_equal(x.cpu().numpy(), x_ref)
        np.testing.assert_array_equal(a.cpu().numpy(), a_ref)
        # arithmetic starts (stride-2 windows) through the _range entry point
        nw = (Fr - W) // 2 + 1
        x2 = torch.empty((nw, W, N, 3), device=device)
        a2 = torch.empty((nw, W, E, 1), device=device)
        _capi.check(lib, lib.dof_window_gather_range(tn.data_ptr(), te.data_ptr(), 0, 2, nw, W, N, E,
                                                     x2.data_ptr(), a2.data_ptr(), stream))
        xr, ar = OW.gather_windows(nodes, edges, np.arange(nw) * 2, W)
        np.testing.assert_array_equal(x2.cpu().numpy(), xr)
        np.testing.assert_array_equal(a2.cpu().numpy(), ar)
        # bf16 storage (BASELINE configs[1]): the same windows rounded to nearest-even bf16 by the gather itself, bit for bit;
        # widened back exactly.  (W*3N and W*E must be even: 16-byte aligned runs.)
        if (W * 3 * N) % 2 == 0 and (W * E) % 2 == 0:
            xb = torch.zeros((len(starts), W, N, 3), dtype=torch.bfloat16, device=device)
            ab = torch.zeros((len(starts), W, E, 1), dtype=torch.bfloat16, device=device)
            _capi.check(lib, lib.dof_window_gather_bf16(tn.data_ptr(), te.data_ptr(), st.data_ptr(), 0, 0, len(starts), W, N, E,
                                                        xb.data_ptr(), ab.data_ptr(), stream))
            want_x, want_a = torch.from_numpy(x_ref).to(torch.bfloat16), torch.from_numpy(a_ref).to(torch.bfloat16)

            def same_bits(got, want):   # bit for bit, a NaN wherever the source holds one (payload bits are not compared)
                got, nan = got.cpu(), torch.isnan(want)
                return torch.equal(torch.isnan(got), nan) and torch.equal(got.view(torch.int16)[~nan], want.view(torch.int16)[~nan])
            assert same_bits(xb, want_x) and same_bits(ab, want_a)
            xb2 = torch.zeros((nw, W, N, 3), dtype=torch.bfloat16, device=device)
            ab2 = torch.zeros((nw, W, E, 1), dtype=torch.bfloat16, device=device)
            _capi.check(lib, lib.dof_window_gather_bf16(tn.data_ptr(), te.data_ptr(), None, 0, 2, nw, W, N, E,
                                                        xb2.data_ptr(), ab2.data_ptr(), stream))
            assert same_bits(xb2, torch.from_numpy(xr).to(torch.bfloat16)) and same_bits(ab2, torch.from_numpy(ar).to(torch.bfloat16))
            wide = torch.full((len(starts), W, N, 3), float("nan"), device=device)
            _capi.check(lib, lib.dof_widen_bf16(xb.data_ptr(), wide.data_ptr(), xb.numel(), stream))
            np.testing.assert_array_equal(wide.cpu().numpy(), want_x.float().numpy())
        else:
            xb = torch.zeros((len(starts), W, N, 3), dtype=torch.bfloat16, device=device)
            assert lib.dof_window_gather_bf16(tn.data_ptr(), te.data_ptr(), st.data_ptr(), 0, 0, len(starts), W, N, E,
                                              xb.data_ptr(), xb.data_ptr(), stream) != 0


def run_phase_check(lib, device, golden_dir, tag, phase, atol_g=5e-5, rtol_g=5e-4):
    d = load_golden(golden_dir, f"vade_{tag}.npz")
    spec = PHASES[phase]
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K)
    eng.load_state_dict(params_from(d))
    tau = torch.from_numpy(d["tau"]) if spec["teacher"] else None
    configure_phase(eng, K, spec["pretrain"], spec["klw"], tau, 1.7 if spec["teacher"] else 0.0, spec.get("extra"))
    eng.loss_grads(x, a, torch.from_numpy(d["eps"]).to(device), torch.from_numpy(d["eps_mc"]).to(device),
                   None if tau is None else tau.to(device), pretrain=spec["pretrain"])
    logs = eng.read_logs()
    for k, v in logs.items():
        if k == "kl_weight":
            continue
        ref = float(d[f"{phase}::loss::{k}"])
        np.testing.assert_allclose(v, ref, rtol=1e-4, atol=1e-5, err_msg=f"{tag}/{phase}: {k}")
    worst = []
    for k in d:
        if k.startswith(f"{phase}::grad::"):
            name = k.split("::grad::")[1]
            g = eng.view(name, eng.grads).cpu().numpy()
            ref = d[k].reshape(g.shape)
            err = np.abs(g - ref).max() / (np.abs(ref).max() + 1e-12)
            worst.append((err, name))
            np.testing.assert_allclose(g, ref, atol=atol_g, rtol=rtol_g, err_msg=f"{tag}/{phase}: grad {name}")
    assert len(worst) >= 80
    # parameters the reference leaves without gradient stay exactly zero here
    for name in eng.names:
        if f"{phase}::grad::{name}" not in d:
            assert float(eng.view(name, eng.grads).abs().max()) == 0.0, name
    return sorted(worst)[-3:]


def run_trace_check(lib, device, golden_dir):
    d = load_golden(golden_dir, "vade_train_trace.npz")
    sd0 = params_from(d, "sd0::")
    K, L = sd0["latent_space.gmm_means"].shape
    x0 = d["step0::x"]
    B, T, N, _ = x0.shape
    nodes_adj = load_golden(golden_dir, "graph_ops.npz")["single_adj"]
    eng = VadeEngine(lib, device, B, T, nodes_adj, L, K)
    eng.load_state_dict(sd0)
    last = None
    travel = [0.0, 0.0]   # sum of the steps' learning rates: base segments, GMM segment
    lr_min = 1.0
    for s in range(6):
        phase = str(d[f"step{s}::phase"])
        if phase != last:
            eng.reset_optimizer()
            last = phase
        lr_b, lr_g = (float(v) for v in d[f"step{s}::lr"])
        travel[0] += lr_b
        travel[1] += lr_g
        lr_min = min(lr_min, lr_b, lr_g)
        for seg in (_capi.SEG_ENCODER, _capi.SEG_DECODER, _capi.SEG_HEADS):
            eng.set_lr(seg, lr_b)
        eng.set_lr(_capi.SEG_GMM, lr_g)
        configure_phase(eng, K, phase == "pre", float(d[f"step{s}::klw"]))
        t = lambda k: torch.from_numpy(d[f"step{s}::{k}"]).to(device)
        eng.loss_grads(t("x"), t("a"), t("eps"), t("eps_mc"), None, pretrain=(phase == "pre"))
        eng.optimizer_step()
        logs = eng.read_logs()
        for k, v in logs.items():
            if k == "kl_weight":
                continue
            np.testing.assert_allclose(v, float(d[f"step{s}::log::{k}"]), rtol=2e-3, atol=2e-4, err_msg=f"step {s}: {k}")
        pn = float(torch.sqrt((eng.params.double() ** 2).sum()))
        np.testing.assert_allclose(pn, float(d[f"step{s}::pnorm"]), rtol=2e-5)
    for k, v in params_from(d, "sd_final::").items():
        if k in eng.layout:
            got, ref = eng.view(k).cpu().numpy(), v.numpy().reshape(eng.layout[k][2])
            is_gmm = k.startswith("latent_space.gmm_")
            assert_trace_params(k, got, ref, sd0[k].numpy().reshape(got.shape), travel[1 if is_gmm else 0], lr_min)


def assert_trace_params(name, got, ref, init, full_travel, lr_min=1e-3):
    """Final parameters of an Adam trace against the reference's.  Strict bar (5e-4 abs + 2e-3 rel) for every element
    whose reference value travelled (nearly) the full sum of the steps' learning rates: its gradient kept one sign well above rounding
    noise, so both implementations must have taken the same steps.  An element that travelled less had a gradient near
    zero at some step -- Adam normalises such an element into +-lr steps whose sign is rounding noise -- and only those
    may sit off the strict bar: at most max(1, 0.05 %) of a tensor, by at most three steps (3.2e-3)."""
    err = np.abs(got - ref)
    bad = err > 5e-4 + 2e-3 * np.abs(ref)
    steady = np.abs(ref - init) >= full_travel - 0.5 * lr_min
    assert not (bad & steady).any(), (name, "steady-gradient elements off the strict bar", int((bad & steady).sum()), float(err[steady].max()))
    assert bad.sum() <= max(1, int(5e-4 * bad.size)) and err.max() <= 3.2e-3, (name, int(bad.sum()), float(err.max()))


def run_vqvae_check(lib, device, golden_dir, tag):
    """VQ-VAE forward outputs, step losses, all gradients and a 3-step Adam(+weight decay) trace vs the reference."""
    d = load_golden(golden_dir, f"vqvae_{tag}.npz")
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    L, K = d["sd::vq_layer.codebook"].shape
    km = float(d["kmeans"])
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vqvae")
    sd_init = params_from(d)
    eng.load_state_dict(sd_init)
    out = eng.vq_forward(x, a)
    np.testing.assert_array_equal(out["idx"].cpu().numpy(), d["idx"])
    np.testing.assert_allclose(out["ze"].cpu().numpy(), d["ze"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(out["quantized"].cpu().numpy(), d["quantized"], atol=1e-6)
    np.testing.assert_allclose(out["soft_counts"].cpu().numpy(), d["soft_counts"], atol=2e-6, rtol=2e-3)
    np.testing.assert_allclose(out["loc_q"].cpu().numpy(), d["loc_q"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["loc_e"].cpu().numpy(), d["loc_e"], atol=2e-5, rtol=1e-4)
    eng.set_hyper(vq_beta=1.0, km_latent=km, km_loss=1.0 if km else 0.0, clip=0.75, wd=1e-4)
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, 1e-3)
    eng.push_hyper()
    eng.vq_loss_grads(x, a)
    logs = eng.read_vq_logs()
    for k, v in logs.items():
        np.testing.assert_allclose(v, float(d[f"log::{k}"]), rtol=1e-4, atol=1e-5, err_msg=f"{tag}: {k}")
    n = 0
    for k in d:
        if k.startswith("grad::"):
            name = k[6:]
            g = eng.view(name, eng.grads).cpu().numpy()
            np.testing.assert_allclose(g, d[k].reshape(g.shape), atol=5e-5, rtol=5e-4, err_msg=f"{tag}: grad {name}")
            n += 1
    assert n >= 70
    for name in eng.names:
        if f"grad::{name}" not in d:
            assert float(eng.view(name, eng.grads).abs().max()) == 0.0, name
    eng.reset_optimizer()
    for i in range(3):
        eng.push_hyper()
        xs, as_ = torch.from_numpy(d[f"step{i}::x"]).to(device), torch.from_numpy(d[f"step{i}::a"]).to(device)
        eng.vq_loss_grads(xs, as_)
        eng.optimizer_step()
        for k, v in eng.read_vq_logs().items():
            np.testing.assert_allclose(v, float(d[f"step{i}::log::{k}"]), rtol=2e-3, atol=2e-4, err_msg=f"step {i}: {k}")
    for k, v in params_from(d, "sd_final::").items():
        if k in eng.layout:
            got, ref = eng.view(k).cpu().numpy(), v.numpy().reshape(eng.layout[k][2])
            assert_trace_params(k, got, ref, sd_init[k].numpy().reshape(got.shape), 3 * 1e-3)


def aug_from_golden(d, pfx, device):
    mask = d[pfx + "aug::rot_mask"]
    return dict(start=torch.from_numpy(d[pfx + "aug::start"]).to(device),
                rot_pivot=[int(v) for v in d[pfx + "aug::rot_pivot"]],
                rot_nodes=[np.nonzero(m)[0].tolist() for m in mask],
                theta=torch.from_numpy(d[pfx + "aug::theta"]).to(device),
                interp_t0=torch.from_numpy(d[pfx + "aug::interp_t0"]).to(device),
                interp_len=torch.from_numpy(d[pfx + "aug::interp_len"]).to(device),
                noise=torch.from_numpy(d[pfx + "aug::noise"]).to(device))


def run_contrastive_loss_check(lib, device, golden_dir, tag):
    """Pairwise losses + gradients wrt both (unnormalised) embeddings vs the reference loss table."""
    from deepof_amd.engine import VadeEngine
    d = load_golden(golden_dir, f"contrastive_{tag}.npz")
    z, za = torch.from_numpy(d["loss_z"]).to(device), torch.from_numpy(d["loss_za"]).to(device)
    B, L = z.shape
    eng = VadeEngine(lib, device, B, 12, d["adj"], L, 1, kind="contrastive")
    for sim in ("cosine", "dot", "euclidean", "edit"):
        for lf in ("nce", "dcl", "hard_dcl", "fc"):
            dz, dza = eng.contrastive_loss(z, za, sim, lf, 0.1, 0.1, 0.1)
            logs = eng.read_contrastive_logs()
            ref = d[f"loss::{sim}::{lf}"]
            np.testing.assert_allclose([logs["total_loss"], logs["pos_similarity"], logs["neg_similarity"]], ref,
                                       rtol=2e-5, atol=2e-6, err_msg=f"{sim}/{lf}")
            # the reference table differentiates wrt unit-norm inputs; the kernel adds the F.normalize
            # backward (projection off the radial direction, norms are 1): compare against the projected grads
            for got, g, v in ((dz, d[f"loss_grad::{sim}::{lf}"][0], z), (dza, d[f"loss_grad::{sim}::{lf}"][1], za)):
                g = torch.from_numpy(g)
                vc = v.cpu()
                proj = g - (g * vc).sum(1, keepdim=True) * vc
                np.testing.assert_allclose(got.cpu().numpy(), proj.numpy(), rtol=2e-4, atol=2e-6,
                                           err_msg=f"{sim}/{lf}")
    with np.testing.assert_raises(NotImplementedError):
        eng.contrastive_loss(z, za, "cosine", "triplet")


def run_contrastive_check(lib, device, golden_dir, tag):
    """Views (central + augmented, identical draws), embeddings, step logs and encoder gradients vs the reference."""
    from deepof_amd.engine import VadeEngine, contrastive_views
    d = load_golden(golden_dir, f"contrastive_{tag}.npz")
    x_full = torch.from_numpy(d["x_full"]).to(device)
    ei = torch.from_numpy(d["edge_index"]).to(device)
    B, Tf, N, _ = x_full.shape
    half = Tf // 2
    for ci in range(3):
        pfx = f"c{ci}::"
        sim, lf = str(d[pfx + "sim"]), str(d[pfx + "loss_fn"])
        L = d[pfx + "sd::encoder.final_dense.bias"].shape[0]
        xc, ac = contrastive_views(lib, x_full, ei, None)
        np.testing.assert_array_equal(xc.cpu().numpy(), d[pfx + "x"])
        np.testing.assert_allclose(ac.cpu().numpy(), d[pfx + "a"], atol=1e-7)
        xa, aa = contrastive_views(lib, x_full, ei, aug_from_golden(d, pfx, device))
        np.testing.assert_allclose(xa.cpu().numpy(), d[pfx + "x_aug"], atol=2e-6)
        np.testing.assert_allclose(aa.cpu().numpy(), d[pfx + "a_aug"], atol=3e-6)
        e1 = VadeEngine(lib, device, B, half, d["adj"], L, 1, kind="contrastive")
        e2 = VadeEngine(lib, device, B, half, d["adj"], L, 1, kind="contrastive", shared=e1)
        e1.load_state_dict(params_from(d, pfx + "sd::"))
        assert all(n.startswith("encoder.") or n.startswith("distill_head.") for n in e1.names)
        assert all(n.startswith("encoder.") for n in e1.state_dict() if "." in n)
        z = e1.contrastive_encode(xc, ac, train=True)
        z_aug = e2.contrastive_encode(xa, aa, train=True)
        np.testing.assert_allclose(z.cpu().numpy(), d[pfx + "z"], atol=1e-5, rtol=1e-4)
        np.testing.assert_allclose(z_aug.cpu().numpy(), d[pfx + "z_aug"], atol=2e-5, rtol=1e-4)
        dz, dza = e1.contrastive_loss(z, z_aug, sim, lf, 0.1, 0.1, 0.1)
        logs = e1.read_contrastive_logs()
        for k in ("total_loss", "pos_similarity", "neg_similarity"):
            np.testing.assert_allclose(logs[k], float(d[pfx + f"log::{k}"]), rtol=1e-4, atol=1e-5, err_msg=f"{ci}: {k}")
        e1.contrastive_backward(dz, accumulate=False)
        e2.contrastive_backward(dza, accumulate=True)
        n = 0
        for k in d:
            if k.startswith(pfx + "grad::"):
                name = k[len(pfx) + 6:]
                g = e1.view(name, e1.grads).cpu().numpy()
                np.testing.assert_allclose(g, d[k].reshape(g.shape), atol=5e-5, rtol=1e-3, err_msg=f"{ci}: grad {name}")
                n += 1
        assert n >= 40
        for name in e1.names:
            if pfx + f"grad::{name}" not in d:
                assert float(e1.view(name, e1.grads).abs().max()) == 0.0, name   # incl. the unused distillation head


def run_contrastive_tcn_check(lib, device, golden_dir, fixture="contrastive_tcn14.npz"):
    """Contrastive step with the TCN encoder (R12): eval-mode embeddings (running statistics), train-mode embeddings,
    BatchNorm buffers after the two passes, loss and all gradients, then the reference's two optimiser steps
    (Adam + weight decay 1e-4, clip 0.75, CensNet tensors outside the optimiser -- quirk Q11)."""
    from deepof_amd.engine import VadeEngine, contrastive_views
    d = load_golden(golden_dir, fixture)
    pfx = "c0::"
    x_full = torch.from_numpy(d["x_full"]).to(device)
    ei = torch.from_numpy(d["edge_index"]).to(device)
    B, Tf, N, _ = x_full.shape
    L = d[pfx + "sd::encoder.head.6.bias"].shape[0]
    e1 = VadeEngine(lib, device, B, Tf // 2, d["adj"], L, 1, kind="contrastive_tcn")
    e2 = VadeEngine(lib, device, B, Tf // 2, d["adj"], L, 1, kind="contrastive_tcn", shared=e1)
    sd0 = params_from(d, pfx + "sd::")
    e1.load_state_dict(sd0)
    assert list(e1.state_dict().keys()) == list(sd0.keys())
    xc, ac = contrastive_views(lib, x_full, ei, None)
    xa, aa = contrastive_views(lib, x_full, ei, aug_from_golden(d, pfx, device))
    z_eval = e1.contrastive_encode(xc, ac, train=False)
    np.testing.assert_allclose(z_eval.cpu().numpy(), d[pfx + "z_eval"], atol=2e-5, rtol=1e-4)
    z = e1.contrastive_encode(xc, ac, train=True)
    z_aug = e2.contrastive_encode(xa, aa, train=True)
    np.testing.assert_allclose(z.cpu().numpy(), d[pfx + "z"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(z_aug.cpu().numpy(), d[pfx + "z_aug"], atol=2e-5, rtol=1e-4)
    sd1 = e1.state_dict()
    nbuf = 0
    for k in d:
        if k.startswith(pfx + "sd_after::"):
            name = k[len(pfx) + 10:]
            np.testing.assert_allclose(sd1[name].numpy(), d[k], atol=2e-6, rtol=2e-5, err_msg=name)
            nbuf += 1
    assert nbuf == 3 * 34
    dz, dza = e1.contrastive_loss(z, z_aug, "cosine", "nce", 0.1, 0.1, 0.1)
    logs = e1.read_contrastive_logs()
    for k in ("total_loss", "pos_similarity", "neg_similarity"):
        np.testing.assert_allclose(logs[k], float(d[pfx + f"log::{k}"]), rtol=1e-4, atol=1e-5, err_msg=k)
    e1.contrastive_backward(dz, accumulate=False)
    e2.contrastive_backward(dza, accumulate=True)
    # C4's shape (B = 64, window 50 -> 25, both views): the standard bar + the explicit attribution of ReLU-branch flips
    # (tcn_kinks.npz, make_golden_r04.py): a tensor no identified flip reaches is held to the plain bar
    # (round 6: the latent-16 fixture too -- tests/golden/make_golden_r06.py -- since the bf16-piece convolutions sum in another
    #  order than the fp32-MFMA ones did and one of its 69 near-zero pre-activations changes branch on the MI355X)
    kinks, flips = None, []
    if fixture in ("contrastive_tcn14_b64.npz", "contrastive_tcn14l16.npz"):
        kinks = KinkAttribution(golden_dir, fixture[:-4] + "::c0::")
        flips = kinks.identify(lambda t: e1.view(t, e1.grads).cpu().numpy(), lambda t: d[pfx + "grad::" + t])
        # ... which is an inference from the residual; the device's own tensors say which candidates DID change branch, and
        # those set the bars (round 6: the pursuit had named three that did not and missed one below a bar)
        observed, _ = confirm_flips_on_device(kinks, [e1, e2], flips)
        kinks.use_observed(observed)
    n, worst = 0, 0.0
    for k in d:
        if k.startswith(pfx + "grad::"):
            name = k[len(pfx) + 6:]
            g = e1.view(name, e1.grads).cpu().numpy()
            if math_zero_gradient(name):  # a bias in front of a BatchNorm: both sides hold rounding noise, only bounded
                assert max(float(np.abs(g).max()), float(np.abs(d[k]).max())) < 3e-4, name
            elif fixture == "contrastive_tcn14.npz":  # the round-1 fixture keeps its elementwise check
                np.testing.assert_allclose(g, d[k].reshape(g.shape), atol=1e-4, rtol=2e-3, err_msg=f"grad {name}")
            else:  # latent 16: gradients of O(5); the standard per-tensor bar (5e-5 + 5e-4 max|ref|) of the TCN checks
                worst = max(worst, _grad_bar(g, d[k].reshape(g.shape), name, extra=kinks.extra(name) if kinks else 0.0))
            n += 1
    assert n == 148
    if kinks is not None:
        print("contrastive TCN B = 64: worst gradient error / tensor scale", worst, "flips observed on the device",
              [int(o) for o in kinks.first_ordinal[kinks.flipped]])
    # optimiser: step 1 on these gradients, step 2 = a full step with the second set of recorded draws
    for name in e1.names:
        if ".spatial_gnn_block." in name:
            e1.set_trainable(name, False)
    e1.reset_optimizer()
    for seg in range(_capi.SEG_COUNT):
        e1.set_lr(seg, 1e-3)
    e1.set_hyper(clip=0.75, wd=1e-4)
    e1.push_hyper()
    e1.optimizer_step()
    aug2 = {k.replace("aug2::", "aug::"): v for k, v in d.items() if k.startswith(pfx + "aug2::")}
    xa2, aa2 = contrastive_views(lib, x_full, ei, aug_from_golden(aug2, pfx, device))
    z = e1.contrastive_encode(xc, ac, train=True)
    z_aug = e2.contrastive_encode(xa2, aa2, train=True)
    dz, dza = e1.contrastive_loss(z, z_aug, "cosine", "nce", 0.1, 0.1, 0.1)
    for k, v in e1.read_contrastive_logs().items():
        if f"{pfx}log2::{k}" in d and k != "seperability":
            # (after one Adam step the rounding-noise-driven +-lr moves of the zero-gradient biases are in the weights)
            np.testing.assert_allclose(v, float(d[f"{pfx}log2::{k}"]), rtol=1e-2, atol=1e-3, err_msg=f"step 2: {k}")
    e1.contrastive_backward(dz, accumulate=False)
    e2.contrastive_backward(dza, accumulate=True)
    e1.push_hyper()
    e1.optimizer_step()
    sd2 = e1.state_dict()
    for k, v in params_from(d, pfx + "sd_step2::").items():
        if v.dtype == torch.int64:
            assert int(sd2[k]) == int(v), k
        elif k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            # a bias in front of a BatchNorm has an exactly-zero gradient; Adam normalises the rounding noise of
            # either implementation into +-lr steps, so only the bound is comparable
            assert float(sd2[k].abs().max()) <= 2.1e-3 and float(v.abs().max()) <= 2.1e-3, k
        elif k in e1.layout:
            # (the running mean of a conv BatchNorm contains 0.1 x that noise-driven bias)
            got, ref = sd2[k].numpy(), v.numpy().reshape(sd2[k].shape)
            atol = 6e-4 if k.endswith("running_mean") else 3e-4
            bad = np.abs(got - ref) > atol + 2e-3 * np.abs(ref)
            # Adam turns the rounding noise of a (mathematically) zero gradient into +-lr steps -- e.g. a bias whose
            # unit is active for the whole batch in front of a BatchNorm: such elements are only bounded, and
            # isolated near-zero-gradient elements elsewhere are tolerated
            # "near zero" is measured against what the gradient check itself allows: Adam's second update is
            # (0.9 g1 + 0.1 g2) / 0.19 / sqrt(v), whose sensitivity to an error dg in g2 is ~0.5 dg / |g1| of a step, so an element
            # whose step-1 gradient is below twice the tensor's gradient bar (standard bar + the observed flips' measured change)
            # may move by more than 0.3 lr although both gradients are inside their bars
            if pfx + "grad::" + k in d:
                gref = np.abs(d[pfx + "grad::" + k].reshape(got.shape))
                bad &= gref > 2.0 * (5e-5 + 5e-4 * float(gref.max()) + (kinks.extra(k) if kinks else 0.0))
            # (0.75 % of the elements, at least three: the first block's edge convolution has 128 weights.  The elements
            #  counted here carry step-1 gradients ABOVE the gradient bar -- measured, round 6 -- so they come from step 2:
            #  its ReLU flips are not attributed (tcn_kinks.npz covers the recorded step only) and one flip moves a few
            #  weights of its block by more than their step-1 gradient, i.e. by a visible fraction of an Adam step.
            #  Worst tensors on the MI355X with the round-6 kernels: 22 of 4096, and 3 of the 128 weights of the edge stream's
            #  first convolution (deviations <= 0.63 lr) -- hence "at least three".)
            assert bad.sum() <= max(3, int(0.0075 * bad.size)) and np.abs(got - ref).max() <= 4.2e-3, \
                (k, bad.sum(), np.abs(got - ref).max())
    np.testing.assert_array_equal(sd2["encoder.spatial_gnn_block.node_kernel"].numpy(),
                                  d[pfx + "sd::encoder.spatial_gnn_block.node_kernel"])


# noise units allowed on the 6-window VaDE-TCN golden (measured: median 1.5, worst 5.5 with the two-pass and 9 with the
# one-pass BatchNorm statistics; the oracle-based twin of this check uses the same 10)
TCN_NOISE_BAR = 10.0
# ... or this fraction of the tensor scale, whichever is larger: the fixture is a FRESHLY INITIALISED model over 6 windows
# (beta = 0, bias = 0: whole rows of BatchNorm outputs sit at 0 +- rounding and their ReLU masks are decided by the
# rounding; the reference's own fp32 gradients deviate 0.6 % (median) to 2 % from its float64 evaluation there, see
# make_golden_r02._trained_like_state), and the per-tensor "noise" is ONE sample of that deviation.  Measured with the
# one-pass BatchNorm statistics: median 3.5 noise units, worst 57 = 1.5e-3 of the tensor scale.  The standard bar
# (5e-5 + 5e-4 scale) is held by the well-conditioned B = 64 fixture (run_vade_tcn_b64_check), in both of its phases.
TCN_B6_RTOL = 5e-3


def run_vade_tcn_check(lib, device, golden_dir, fixture="vade_tcn14.npz"):
    """VaDE with the TCN encoder and decoder (R12) vs the reference golden: eval forward on the running statistics,
    then (from the same initial state each) train-mode loss terms, all gradients and the refreshed BatchNorm buffers.
    fixture="vade_tcn14w50.npz" (round 4, make_golden_r04.py w50): the same model at window 50 -- the 8-sequence form of
    the time-resident convolutions and the 2-sequence weight-gradient chunks."""
    d = load_golden(golden_dir, fixture)
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vade_tcn")
    sd0 = params_from(d)
    eng.load_state_dict(sd0)
    assert list(eng.state_dict().keys()) == list(sd0.keys())
    out = eng.forward(x, a, None, want_loc=True, want_enc=True)
    np.testing.assert_allclose(out["enc"].cpu().numpy(), d["eval_enc"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["z"].cpu().numpy(), d["eval_z"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["q"].cpu().numpy(), d["eval_q"], atol=1e-5, rtol=1e-3)
    np.testing.assert_allclose(out["loc"].cpu().numpy(), d["eval_loc"], atol=5e-5, rtol=1e-4)
    eps, eps_mc = torch.from_numpy(d["eps"]).to(device), torch.from_numpy(d["eps_mc"]).to(device)
    tau = torch.from_numpy(d["tau"]).to(device)
    for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
        eng.load_state_dict(sd0)
        configure_phase(eng, K, phase == "pre", klw, tau if teacher else None, 1.7 if teacher else 0.0)
        eng.loss_grads(x, a, eps, None if phase == "pre" else eps_mc, tau if teacher else None, pretrain=phase == "pre")
        logs = eng.read_logs()
        for k, v in logs.items():
            key = f"{phase}::loss::{k}"
            if key in d:
                np.testing.assert_allclose(v, float(d[key]), rtol=2e-4, atol=2e-5, err_msg=key)
        # gradients: the golden holds the reference evaluated in float64 and, per tensor, the reference's own fp32
        # deviation from it ("gnoise": ~3e-4 of the tensor scale here -- BatchNorm over 6 windows is ill-conditioned).
        # Bar: within TCN_NOISE_BAR noise units of the fp64 value (measured: median 1.5, worst 5.5 / 9).
        n, worst = 0, 0.0
        for k in d:
            if k.startswith(f"{phase}::grad::"):
                name = k.split("::")[-1]
                g = eng.view(name, eng.grads).cpu().numpy()
                ref = d[k].reshape(g.shape)
                err, noise = np.abs(g - ref).max(), float(d[f"{phase}::gnoise::{name}"])
                assert err <= max(TCN_NOISE_BAR * noise, TCN_B6_RTOL * np.abs(ref).max()) + 2e-6 * np.abs(ref).max() + 1e-7, (phase, name, err, noise)
                worst = max(worst, err / (noise + 2e-6 * np.abs(ref).max() + 1e-7))
                n += 1
        assert n >= (200 if phase == "pre" else 10)
        if phase == "pre":
            sd1 = eng.state_dict()
            nb = 0
            for k in d:
                if k.startswith("pre::sd_after::"):
                    name = k[len("pre::sd_after::"):]
                    np.testing.assert_allclose(sd1[name].numpy(), d[k], atol=5e-6, rtol=5e-5, err_msg=name)
                    nb += 1
            assert nb == 2 * (34 + 3 + 8)


def math_zero_gradient(name) -> bool:
    """Parameters whose gradient is mathematically zero in the TCN family: a bias added right in front of a
    BatchNorm (conv1 / conv2 of every temporal block, decoder.fc0) -- the normalisation removes any constant.  Both
    implementations return rounding noise there (1e-7 .. 1e-4 depending on how many terms cancel)."""
    n = name[-1] if isinstance(name, tuple) else name
    return (("_tcn.blocks." in n or ".tcn.blocks." in n) and n.endswith(("conv1.bias", "conv2.bias"))) or n == "decoder.fc0.bias"


def _grad_bar(got, ref, name, atol=5e-5, rtol=5e-4, extra=0.0):
    """Standard fp32 gradient bar: |got - ref| <= atol + rtol * max|ref| (+ ``extra``: the measured consequence of
    identified ReLU-branch flips, see KinkAttribution) per tensor; mathematically-zero gradients (math_zero_gradient)
    are only bounded."""
    scale = float(np.abs(ref).max())
    if math_zero_gradient(name):
        assert scale < 3e-4 and float(np.abs(got).max()) < 3e-4, (name, scale, float(np.abs(got).max()))
        return 0.0
    err = float(np.abs(got - ref).max())
    assert err <= atol + rtol * scale + extra, (name, err, scale, extra)
    return err / max(scale, 1e-30)


class KinkAttribution:
    """Explicit attribution of ReLU-branch flips in the TCN family (tests/golden/make_golden_r03.py, tcn_kinks.npz).

    A train step of the B = 64 fixtures evaluates ~23 M BatchNorm+ReLU pre-activations of O(1) spread; ~150 of them lie
    within 5e-6 of zero, where fp32 rounding decides the ReLU branch, so two correct fp32 implementations disagree on a
    few, and one flipped branch moves the gradients of its block by up to 5 x the standard bar.  The fixture holds, for
    every such candidate, the REFERENCE's own gradient change when exactly that element takes the other branch.  Here
    the flipped elements are NAMED: the gradient error at each significant candidate's most affected element ("probe")
    is explained as A c with c_i in {0, 1}; a coefficient that is neither fails the test.  A tensor's bar is then
    standard bar + (changes of the identified flips [+ the harmless candidates' summed changes]) for the tensors an
    identified flip reaches, and the plain standard bar for every other tensor -- nothing else."""

    MAX_NAMED = 8

    def __init__(self, golden_dir, prefix):
        k = load_golden(golden_dir, "tcn_kinks.npz")
        self.prefix, self.golden_dir = prefix, golden_dir
        self.tensors = [str(t) for t in k[prefix + "tensors"]]
        self.col = {t: i for i, t in enumerate(self.tensors)}
        self.harmless = k[prefix + "harmless"]
        self.first_ordinal = k[prefix + "first_ordinal"]
        self.probe_tensor = k[prefix + "probe_tensor"]       # (candidates, probes) index into self.tensors
        self.probe_index = k[prefix + "probe_index"]
        self.probe_value = k[prefix + "probe_value"].astype(np.float64)
        self.maxd = k[prefix + "maxd"]
        self.rtol, self.atol = float(k[prefix + "rtol"]), float(k[prefix + "atol"])
        self.flipped = np.zeros(len(self.first_ordinal), dtype=bool)
        self._has_loc = (prefix + "loc_call") in k

    def has_locations(self):
        return self._has_loc

    def identify(self, got_of, ref_of):
        """got_of / ref_of: name -> gradient array.  Matching pursuit over the significant candidates: the one whose
        own probes best carry its change with coefficient 1 is named and its change subtracted from the residual at
        those elements, until none qualifies.  Returns the first ordinals (ReLU call order x flat index in the
        reference run) of the named flips."""
        n = len(self.first_ordinal)
        if n == 0:
            return []
        err, bars = {}, {}

        def residual(t, idx):
            key = (int(t), int(idx))
            if key not in err:
                name = self.tensors[key[0]]
                ref = ref_of(name).reshape(-1)
                if key[0] not in bars:
                    bars[key[0]] = self.atol + self.rtol * float(np.abs(ref).max())
                err[key] = (float(got_of(name).reshape(-1)[key[1]]) - float(ref[key[1]])) / bars[key[0]]
            return err[key]

        R = np.array([[residual(t, i) for t, i in zip(self.probe_tensor[c], self.probe_index[c])] for c in range(n)])
        V = self.probe_value
        vv = (V * V).sum(1)
        for _ in range(n):
            coef = np.where(vv >= 0.25, (R * V).sum(1) / np.maximum(vv, 1e-30), 0.0)   # (below half a bar: rounding)
            gain = (R * R).sum(1) - ((R - V) ** 2).sum(1)
            ok = (~self.flipped) & (coef >= 0.6) & (coef <= 1.4) & (gain > 0)
            if not ok.any():
                break
            best = int(np.argmax(np.where(ok, gain, -np.inf)))
            self.flipped[best] = True
            for t, i, v in zip(self.probe_tensor[best], self.probe_index[best], V[best]):
                err[(int(t), int(i))] -= v
            R = np.array([[err[(int(t), int(i))] for t, i in zip(self.probe_tensor[c], self.probe_index[c])] for c in range(n)])
        named = self.first_ordinal[self.flipped].tolist()
        # a ceiling on the attribution: MI355X names 0 - 4 flips on these fixtures; matching pursuit over ~1,700 candidates
        # must not be able to explain an arbitrary error away by naming many
        print(f"KinkAttribution[{self.prefix}]: {len(named)} named flip(s) of {n} candidates: {named}")
        # (with candidate locations in the fixture the caller replaces this inference by the flips OBSERVED on the device --
        #  use_observed() -- and the ceiling applies to those)
        assert self._has_loc or len(named) <= self.MAX_NAMED, (self.prefix, len(named), named)
        return named

    def use_observed(self, candidate_indices, unlocated=()):
        """Replace the inferred flips by the ones observed on the device (confirm_flips_on_device); same ceiling.  A flip the
        pursuit inferred on a ReLU the device lookup cannot reach (``unlocated`` ordinals: the decoder's, CensNet's and the
        head's) stays inferred."""
        unloc = set(int(o) for o in unlocated)
        inferred = [ci for ci in np.nonzero(self.flipped)[0].tolist() if int(self.first_ordinal[ci]) in unloc]
        self.flipped[:] = False
        self.flipped[list(candidate_indices) + inferred] = True
        assert int(self.flipped.sum()) <= self.MAX_NAMED, (self.prefix, int(self.flipped.sum()))
        return [int(self.first_ordinal[ci]) for ci in inferred]

    def extra(self, name):
        """Additional bar of tensor ``name``: 1.25 x the identified flips' measured changes there, plus -- ONLY where a
        named flip reaches the tensor -- the summed changes of the harmless candidates (each below a quarter of the
        bar; a branch that flipped upstream moves pre-activations of its block by ~1e-5, enough to take further
        near-zero elements along).  A tensor no named flip reaches is held to the plain bar: without an identified
        flip the whole check is the standard one."""
        if name not in self.col:
            return 0.0
        i = self.col[name]
        named = float(self.maxd[self.flipped, i].sum())
        if named <= 0.0:
            return 0.0
        return float(self.harmless[i]) + 1.25 * named


def confirm_flips_on_device(kinks, engines, named):
    """The attribution's other half: KinkAttribution NAMES flips from the gradient residual; here the device's own tensors say
    which branch each candidate took.  ``engines`` = the plans of the step in the reference's forward order (contrastive:
    central view, augmented view).  tcn_kinks.npz holds, per candidate ordinal, the ReLU call it belongs to, its flat index
    in that call's (S, 32, T) input and the reference's pre-activation value (tests/golden/make_golden_r06.py kinkloc); the
    reference runs 25 ReLU calls per TCN stream -- block b: ReLU(BN1(conv1)), ReLU(BN2(conv2)), ReLU(y + res); then the
    skip-sum's (models_new.py:431-446, 492-505) -- node stream first (models_new.py:617-632).  The device branch of a
    candidate is the sign of the same quantity recomputed from the plan's workspace (dof_vade_ws_tensor): fma(y, scale,
    shift) of the stored convolution output and BatchNorm record, or the stored block output.
    Returns (observed, unlocated): the indices (into kinks.first_ordinal) of the significant candidates whose branch on the
    device differs from the reference's -- OBSERVED flips, which the caller uses for the bars instead of the ones the
    matching pursuit inferred from the gradient residual -- and the ordinals it cannot look up (the CensNet / head ReLUs,
    the skip-sum's ReLU, the last block's unused output).  Asserts that the looked-up quantities ARE the reference's (every
    candidate's device value within 2e-5 of the recorded one)."""
    import ctypes as C
    k = load_golden(kinks.golden_dir, "tcn_kinks.npz")
    pfx = kinks.prefix
    call, flat, val, shapes = (k[pfx + n] for n in ("loc_call", "loc_flat", "loc_value", "loc_shapes"))
    per_fwd = len(shapes) // len(engines)
    assert per_fwd * len(engines) == len(shapes)
    cache = {}

    def tensor(eng, name):
        key = (id(eng), name)
        if key not in cache:
            off, sp = C.c_int64(), C.c_int64()
            _capi.check(eng.lib, eng.lib.dof_vade_ws_tensor(eng.plan, name.encode(), C.byref(off), C.byref(sp)), "dof_vade_ws_tensor")
            cache[key] = (off.value, sp.value)
        return cache[key]

    last, dev = [None], [0.0]

    def device_positive(o):
        """None = a ReLU whose input the workspace does not keep (last block's output, the skip-sum before its last step)"""
        c = int(call[o])
        eng = engines[c // per_fwd]
        kk = c % per_fwd
        if kk >= 50:
            return None
        stream, kk = ("n" if kk < 25 else "e"), kk % 25
        S, Cc, T = (int(v) for v in shapes[c])
        j = int(flat[o])
        s_i, c_i, t_i = j // (Cc * T), (j // T) % Cc, j % T
        ws = eng.workspace
        if kk == 24:
            return None
        b, which = kk // 3, kk % 3
        if which == 2:
            if b == 7:
                return None
            off, sp = tensor(eng, f"{stream}.out.{b}")
            last[0] = (c, stream, b, which, s_i, c_i, t_i, float(ws[off + (t_i * sp + s_i) * 32 + c_i]))
            return float(ws[off + (t_i * sp + s_i) * 32 + c_i]) > 0.0
        off, sp = tensor(eng, f"{stream}.{'y1' if which == 0 else 'y2'}.{b}")
        boff, _ = tensor(eng, f"{stream}.{'bnp1' if which == 0 else 'bnp2'}.{b}")
        y = np.float64(float(ws[off + (t_i * sp + s_i) * 32 + c_i]))
        scale, shift = np.float64(float(ws[boff + 64 + c_i])), np.float64(float(ws[boff + 96 + c_i]))
        last[0] = (c, stream, b, which, s_i, c_i, t_i, float(y), float(scale), float(shift), float(y * scale + shift))
        dev[0] = max(dev[0], abs(float(y * scale + shift) - float(val[o])))   # the lookup itself: same quantity as the reference's
        return float(y * scale + shift) > 0.0   # (exact sign of the device's fmaf: the product of two floats is exact in double)

    named = set(int(o) for o in named)
    observed, unlocated = [], []
    # first_ordinal lists the SIGNIFICANT candidates (a flip moves some gradient by >= a quarter of its bar); probe_value = the
    # candidate's largest changes in bar units
    for ci, o in enumerate(kinks.first_ordinal.tolist()):
        pos = device_positive(o)
        if pos is None:
            unlocated.append(o)
        elif pos != (float(val[o]) > 0.0):
            observed.append(ci)
    assert dev[0] < 2e-5, ("the device's pre-activations at the candidates are not the reference's quantities", dev[0])
    obs_ord = [int(kinks.first_ordinal[ci]) for ci in observed]
    print(f"confirm_flips_on_device[{pfx}]: {len(kinks.first_ordinal) - len(unlocated)} significant candidates looked up in the "
          f"device's workspace (largest |device - reference| pre-activation {dev[0]:.2e}); {len(obs_ord)} took the other branch: "
          f"{[(o, round(float(np.abs(kinks.probe_value[ci]).max()), 2)) for o, ci in zip(obs_ord, observed)]} (ordinal, largest "
          f"change in bars); the gradient residual had named {sorted(named)}; not kept in the workspace: {len(unlocated)}")
    return observed, unlocated


def run_vade_tcn_b64_check(lib, device, golden_dir, fixture="vade_tcn14_b64.npz", min_main=20):
    """VaDE with the TCN encoder and decoder at B = 64 in a trained-like state (tests/golden/make_golden_r02.py)
    against the REFERENCE's fp32 values at the standard bars: eval forward, then for the pre-training and the main
    (+ teacher) objective every loss term, the train-mode outputs, every stored gradient (all 200 parameter tensors
    for "pre") and the refreshed BatchNorm buffers / step counters.  Gradients: standard bar + the measured consequence
    of the ReLU-branch flips KinkAttribution identifies (usually none or one).

    fixture="vade_tcn14_onepass.npz" (make_golden_r03.py): the same model with every BatchNorm running mean equal to
    the batch mean of the recorded step, so every channel of every layer takes the ONE-PASS (shifted-sum) statistics
    form of the convolution epilogues -- the steady-state path of a fit -- and is compared elementwise."""
    d = load_golden(golden_dir, fixture)
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vade_tcn")
    sd0 = params_from(d)
    eng.load_state_dict(sd0)
    eng.set_bn_training(False)
    out = eng.forward(x, a, None, want_loc=True, want_enc=True)
    np.testing.assert_allclose(out["enc"].cpu().numpy(), d["eval_enc"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["z"].cpu().numpy(), d["eval_z"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["q"].cpu().numpy(), d["eval_q"], atol=1e-5, rtol=1e-3)
    np.testing.assert_allclose(out["loc"].cpu().numpy(), d["eval_loc"], atol=5e-5, rtol=1e-4)
    eng.set_bn_training(True)
    eps, eps_mc = torch.from_numpy(d["eps"]).to(device), torch.from_numpy(d["eps_mc"]).to(device)
    tau = torch.from_numpy(d["tau"]).to(device)
    worst = {}
    for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
        eng.load_state_dict(sd0)
        configure_phase(eng, K, phase == "pre", klw, tau if teacher else None, 1.7 if teacher else 0.0)
        eng.loss_grads(x, a, eps, None if phase == "pre" else eps_mc, tau if teacher else None, pretrain=phase == "pre")
        logs = eng.read_logs()
        n_terms = 0
        for k, v in logs.items():
            key = f"{phase}::loss::{k}"
            if key in d:
                np.testing.assert_allclose(v, float(d[key]), rtol=1e-4, atol=1e-5, err_msg=key)
                n_terms += 1
        assert n_terms >= 12
        kinks = KinkAttribution(golden_dir, f"{fixture[:-4]}::{phase}::")
        flips = kinks.identify(lambda t: eng.view(t, eng.grads).cpu().numpy(), lambda t: d[f"{phase}::grad::{t}"])
        if kinks.has_locations():   # the encoder's candidates: observed in the device's workspace instead of inferred
            observed, unlocated = confirm_flips_on_device(kinks, [eng], flips)
            still_inferred = kinks.use_observed(observed, unlocated)
            flips = [int(o) for o in kinks.first_ordinal[kinks.flipped]]
            print(f"VaDE TCN {phase}: flips used for the bars {flips} (inferred, on ReLUs outside the encoder blocks: {still_inferred})")
        n, w = 0, 0.0
        for k in d:
            if k.startswith(f"{phase}::grad::"):
                name = k.split("::")[-1]
                g = eng.view(name, eng.grads).cpu().numpy()
                w = max(w, _grad_bar(g, d[k].reshape(g.shape), (phase, name), extra=kinks.extra(name)))
                n += 1
        worst[phase] = (w, flips)
        assert n >= (200 if phase == "pre" else min_main), n
        if phase == "pre":
            sd1 = eng.state_dict()
            nb = 0
            for k in d:
                if k.startswith("pre::sd_after::"):
                    name = k[len("pre::sd_after::"):]
                    if name.endswith("num_batches_tracked"):
                        assert int(sd1[name]) == int(d[k]), name
                    else:
                        np.testing.assert_allclose(sd1[name].numpy(), d[k], atol=5e-6, rtol=5e-5, err_msg=name)
                    nb += 1
            assert nb == 3 * (34 + 3 + 8)
    return worst


# VQ-VAE with the TCN family: the decoder (9 BatchNorm layers) is differentiated twice and both passes feed the 17
# BatchNorm layers of each encoder stream.  Measured worst error / tensor scale on MI355X: 2.6e-4 at step 1; at step 2
# 7e-4 .. 2.3e-3 on the three tensors of ONE block (edge stream, block 1: conv2.weight 2.3e-3, conv1.weight 8.4e-4,
# bn2.bias 7.4e-4), every other tensor <= 6e-4, deterministic.  The oracle's own fp32-vs-fp64 deviation there is
# 4e-4 at worst, so this is fp32 accumulation inside our BatchNorm-backward / weight-gradient kernels (ATen's CPU
# BatchNorm accumulates in double), not conditioning: the bar is 3e-3 for this model and DESIGN.md lists the finding.
VQ_TCN_RTOL = 3e-3


def run_vqvae_tcn_ref_check(lib, device, golden_dir):
    """VQVAEPT(encoder_type="TCN") vs the REFERENCE golden (make_golden_r02.py): eval forward (code indices bit-exact),
    one step_vqvae_distill step (logs, all gradients, BatchNorm buffers and counters: encoder +1, decoder +2), then the
    weights after two optimiser steps of the generic optimiser (Q11: the CensNet tensors stay untouched)."""
    d = load_golden(golden_dir, "vqvae_tcn14.npz")
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    L, K = d["sd::vq_layer.codebook"].shape
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vqvae_tcn")
    sd0 = params_from(d)
    eng.load_state_dict(sd0)
    assert [k for k in eng.state_dict() if k not in sd0] == [] and [k for k in sd0 if k not in eng.state_dict()] == []
    eng.set_bn_training(False)
    out = eng.vq_forward(x, a)
    np.testing.assert_array_equal(out["idx"].cpu().numpy(), d["eval_idx"])
    np.testing.assert_allclose(out["ze"].cpu().numpy(), d["eval_ze"], atol=2e-5, rtol=1e-4)
    np.testing.assert_array_equal(out["quantized"].cpu().numpy(), d["eval_quantized"])
    np.testing.assert_allclose(out["soft_counts"].cpu().numpy(), d["eval_soft_counts"], rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(out["loc_q"].cpu().numpy(), d["eval_loc_q"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(out["loc_e"].cpu().numpy(), d["eval_loc_e"], atol=5e-5, rtol=1e-4)
    # ---- the train step
    eng.set_bn_training(True)
    for name in eng.names:  # fit_VQVAE builds its optimiser before the first forward (Q11)
        if ".spatial_gnn_block." in name:
            eng.set_trainable(name, False)
    eng.reset_optimizer()
    for seg in range(_capi.SEG_COUNT):
        eng.set_lr(seg, 1e-3)
    eng.set_hyper(vq_beta=1.0, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
    eng.push_hyper()
    eng.vq_loss_grads(x, a)
    logs = eng.read_vq_logs()
    for k in ("total_loss", "enc_rec_loss", "reconstruct_loss", "vq_loss", "number_of_populated_clusters"):
        np.testing.assert_allclose(logs[k], float(d[f"log::{k}"]), rtol=1e-4, atol=1e-5, err_msg=k)
    kinks1 = KinkAttribution(golden_dir, "vqvae_tcn14::step1::")
    flips1 = kinks1.identify(lambda t: eng.view(t, eng.grads).cpu().numpy(), lambda t: d["grad::" + t])
    n, worst = 0, 0.0
    for k in d:
        if k.startswith("grad::"):
            name = k[len("grad::"):]
            g = eng.view(name, eng.grads).cpu().numpy()
            worst = max(worst, _grad_bar(g, d[k].reshape(g.shape), name, rtol=VQ_TCN_RTOL, extra=kinks1.extra(name)))
            n += 1
    assert n >= 190, n
    sd1 = eng.state_dict()
    nb = 0
    for k in d:
        if k.startswith("sd_after::"):
            name = k[len("sd_after::"):]
            if name.endswith("num_batches_tracked"):
                assert int(sd1[name]) == int(d[k]), (name, int(sd1[name]), int(d[k]))
            else:
                np.testing.assert_allclose(sd1[name].numpy(), d[k], atol=5e-6, rtol=5e-5, err_msg=name)
            nb += 1
    assert nb == 3 * (34 + 3 + 8)
    # ---- optimiser step 1 on these gradients.  Adam's first step is lr * sign(g): where the gradient is below what
    # fp32 resolves (4 x the gradient bar) the sign is arbitrary in either implementation, elsewhere it must match.
    g1_hip = {k: eng.view(k, eng.grads).cpu().numpy().copy() for k in eng.names if "grad::" + k in d}
    eng.optimizer_step()
    sd1 = eng.state_dict()
    ref1 = params_from(d, "sd_step1::")
    lr = 1e-3
    for k in eng.names:
        if k.endswith("running_mean") or k.endswith("running_var") or k.startswith("distill_head."):
            continue
        got, ref, start = sd1[k].numpy(), ref1[k].numpy().reshape(sd1[k].shape), sd0[k].numpy().reshape(sd1[k].shape)
        if ".spatial_gnn_block." in k:
            np.testing.assert_array_equal(got, start)       # not in the optimiser (Q11)
            np.testing.assert_array_equal(ref, start)
            continue
        assert float(np.abs(got - start).max()) <= lr * 1.001, k
        g1 = np.abs(d["grad::" + k].reshape(got.shape))
        # (an identified ReLU-branch flip moved this tensor's gradient by up to kinks1.extra(k): smaller elements
        # have no resolved sign either)
        weak = g1 < 2e-3 * max(float(g1.max()), 1e-30) + 1e-6 + kinks1.extra(k)
        if math_zero_gradient(k):
            weak[:] = True
        np.testing.assert_allclose(got[~weak], ref[~weak], atol=2e-6, rtol=1e-5, err_msg=k)
    # ---- step 2 from the REFERENCE's post-step-1 weights (teacher forcing: a free-running trace is chaotic, see
    # make_golden_r02.py), our own Adam moments: gradients at the standard bar, then the weights
    eng.load_state_dict({k: v for k, v in ref1.items()})
    x2, a2 = torch.from_numpy(d["step2::x"]).to(device), torch.from_numpy(d["step2::a"]).to(device)
    eng.vq_loss_grads(x2, a2)
    logs2 = eng.read_vq_logs()
    for k in ("total_loss", "enc_rec_loss", "reconstruct_loss", "vq_loss"):
        np.testing.assert_allclose(logs2[k], float(d[f"step2::log::{k}"]), rtol=1e-4, atol=1e-5, err_msg="step 2: " + k)
    kinks2 = KinkAttribution(golden_dir, "vqvae_tcn14::step2::")
    flips2 = kinks2.identify(lambda t: eng.view(t, eng.grads).cpu().numpy(), lambda t: d["grad2::" + t])
    n2 = 0
    for k in d:
        if k.startswith("grad2::"):
            name = k[len("grad2::"):]
            g = eng.view(name, eng.grads).cpu().numpy()
            worst = max(worst, _grad_bar(g, d[k].reshape(g.shape), name, rtol=VQ_TCN_RTOL, extra=kinks2.extra(name)))
            n2 += 1
    assert n2 >= 190, n2
    # The second Adam update of an element is lr * f(g1, g2) of the value-clipped gradients with |df| <= (|dg1| + |dg2|) /
    # sqrt(g1^2 + g2^2) (bias corrections of t = 2; measured constant 0.87): the weights may differ from the reference's by exactly what the two
    # gradient differences -- each already judged against its bar above -- propagate to, element by element, plus
    # 0.3 % of a step for the update arithmetic.  No outlier allowance.
    g2_hip = {k: eng.view(k, eng.grads).cpu().numpy().copy() for k in eng.names if "grad2::" + k in d}
    eng.optimizer_step()
    sd2 = eng.state_dict()
    for k, v in params_from(d, "sd_step2::").items():
        if k not in eng.layout or k.endswith("running_mean") or k.endswith("running_var") or ".spatial_gnn_block." in k:
            continue
        got, ref = sd2[k].numpy(), v.numpy().reshape(sd2[k].shape)
        start = ref1[k].numpy().reshape(got.shape)
        assert float(np.abs(got - start).max()) <= lr * 1.05, k   # |m_hat| / sqrt(v_hat) peaks just above 1 at t = 2
        if "grad2::" + k not in d or "grad::" + k not in d or math_zero_gradient(k):
            continue
        def clamp(g):   # clip_grad_value_(0.75) before both optimiser steps (make_golden_r02.py; the engine's `clip`)
            return np.clip(g.reshape(got.shape).astype(np.float64), -0.75, 0.75)

        r1, r2 = clamp(d["grad::" + k]), clamp(d["grad2::" + k])
        e1, e2 = np.abs(clamp(g1_hip[k]) - r1), np.abs(clamp(g2_hip[k]) - r2)
        rel = (e1 + e2) / np.maximum(np.sqrt(r1 ** 2 + r2 ** 2), 1e-30)
        weak = rel > 0.25        # the gradients themselves are not resolved there (within their bars): direction open
        dev = np.abs(got - ref)
        tol = 3e-6 + 1e-5 * np.abs(ref) + lr * rel
        over = (~weak) & (dev > tol)
        assert not over.any(), (k, int(over.sum()), over.size, float((dev - tol)[over].max()))
        flipped = kinks1.extra(k) > 0 or kinks2.extra(k) > 0
        assert (~weak).mean() > (0.25 if flipped else 0.5) or math_zero_gradient(k), (k, float((~weak).mean()))
    return worst, flips1, flips2


def _oracle_truth(fn, P, *tensors):
    """Run an oracle function in fp32 and in fp64 (its .float() casts redirected to double): (out32, out64)."""
    r32 = fn({k: v.clone() for k, v in P.items()}, *tensors)
    P64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in P.items()}
    orig = torch.Tensor.float
    torch.Tensor.float = lambda self: self.double()
    try:
        r64 = fn(P64, *[t.double() if t.dtype == torch.float32 else t for t in tensors])
    finally:
        torch.Tensor.float = orig
    return r32, r64


def run_vqvae_tcn_check(lib, device, golden_dir, fixture="vade_tcn14.npz"):
    """VQ-VAE with the TCN encoder/decoder: weights = the VaDE-TCN golden's encoder/decoder + a random codebook.
    Eval forward vs the oracle directly; the train step (encoder once, decoder on the quantised and on the raw
    latents, BatchNorm in train mode) vs the oracle evaluated in fp64, in units of the oracle's own fp32 noise."""
    from oracle import vqvae as OQ
    d = load_golden(golden_dir, fixture)
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    B, T, N, _ = x.shape
    L, K = 8, 12
    P = {k: v for k, v in params_from(d).items() if not k.startswith("latent_space")}
    P["vq_layer.codebook"] = torch.randn(L, K, generator=torch.Generator().manual_seed(5)) * 0.5
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vqvae_tcn")
    assert [n for n in eng.state_dict() if n not in P] == []
    eng.load_state_dict(P)
    out = eng.vq_forward(x.to(device), a.to(device))
    with torch.no_grad():
        ref = OQ.vqvae_forward({k: v.clone() for k, v in P.items()}, x, a, training=False)
    np.testing.assert_array_equal(out["idx"].cpu().numpy(), ref["idx"].numpy())
    np.testing.assert_allclose(out["ze"].cpu().numpy(), ref["ze"].numpy(), atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(out["loc_q"].cpu().numpy(), ref["loc_q"].numpy(), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["loc_e"].cpu().numpy(), ref["loc_e"].numpy(), atol=2e-5, rtol=1e-4)
    eng.set_hyper(vq_beta=1.0, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
    eng.push_hyper()
    eng.vq_loss_grads(x.to(device), a.to(device))
    logs = eng.read_vq_logs()
    (l32, g32, _), (l64, g64, _) = _oracle_truth(lambda Pq, xx, aa: OQ.vqvae_grads(Pq, xx, aa, 1.0, 0.0), P, x, a)
    for k in ("total_loss", "enc_rec_loss", "reconstruct_loss", "vq_loss"):
        np.testing.assert_allclose(logs[k], float(l64[k]), rtol=2e-4, atol=2e-5, err_msg=k)
    n = 0
    for name, t in g64.items():
        if t is None:
            assert float(eng.view(name, eng.grads).abs().max()) == 0.0, name
            continue
        g = eng.view(name, eng.grads).cpu().numpy().astype(np.float64)
        t = t.numpy().reshape(g.shape)
        noise = np.abs(g32[name].numpy().astype(np.float64).reshape(g.shape) - t).max()
        err = np.abs(g - t).max()
        # (noise: one fp32 evaluation of the oracle on the host running the test -- it moves with the host's BLAS, so the bar
        # has run_vade_tcn_check's scale term as its floor)
        assert err <= max(8.0 * noise, TCN_B6_RTOL * np.abs(t).max()) + 2e-6 * np.abs(t).max() + 1e-7, (name, err, noise)
        n += 1
    assert n >= 180
    sd = eng.state_dict()
    assert int(sd["decoder.bn0.num_batches_tracked"]) == int(P["decoder.bn0.num_batches_tracked"]) + 2
    assert int(sd["encoder.head.2.num_batches_tracked"]) == int(P["encoder.head.2.num_batches_tracked"]) + 1


def run_turtle_check(lib, device, golden_dir):
    """TURTLE teacher (N3) vs the reference golden: 7 outer x 12 inner steps from the recorded initial weights over the
    recorded batch list, tau* of the prediction pass, GMM initialisation from tau*."""
    from deepof_amd.teacher import TurtleTeacher, gmm_from_teacher
    d = load_golden(golden_dir, "turtle.npz")
    K, B, nb, inner, outer = (int(v) for v in d["cfg"])
    dims = [int(v) for v in d["dims"]]
    t = TurtleTeacher(dims, K, gamma=8.0, alpha_sample_entropy=2.0, inner_lr=0.1, inner_steps=inner, head_wd=1e-4,
                      head_temp=0.35, task_temp=0.35, normalize_feats=True, lr_theta=1e-3, device=device, lib=lib)
    init = {k[6:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("init::")}
    assert list(t.state_dict().keys()) == list(init.keys())
    t.load_state_dict(init)
    batches = [[torch.from_numpy(d[f"batch{i}::{v}"]).to(device) for v in range(len(dims))] for i in range(nb)]
    t.fit(batches, outer_steps=outer, rho=0.04, verbose=False)
    sd = t.state_dict()
    for k, v in d.items():
        if k.startswith("final::"):
            np.testing.assert_allclose(sd[k[7:]].numpy(), v, atol=2e-5, rtol=2e-4, err_msg=k)
    tau = t.predict([batches[0], batches[1]])
    np.testing.assert_allclose(tau.numpy(), d["tau_star"], atol=2e-5, rtol=2e-4)
    np.testing.assert_allclose(tau.sum(1).numpy(), 1.0, atol=1e-5)
    z = torch.cat([batches[0][0], batches[1][0]]).cpu()
    m, lv, pr = gmm_from_teacher(z, torch.from_numpy(d["tau_star"]), min_var=0.01)
    np.testing.assert_allclose(m.numpy(), d["gmm_means"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(lv.numpy(), d["gmm_log_vars"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(pr.numpy(), d["gmm_prior"], atol=1e-6, rtol=1e-5)


def run_distill_head_check(lib, device, golden_dir):
    """Generic distillation head (VQ-VAE on z_e, contrastive on the normalised central embeddings) vs the reference."""
    from deepof_amd.engine import VadeEngine, contrastive_views
    # ---- VQ-VAE
    d = load_golden(golden_dir, "vqvae_rec28.npz")
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    L, K = d["sd::vq_layer.codebook"].shape
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vqvae")
    sd = params_from(d, "sd_final::")
    sd.update({"distill_head." + k: v for k, v in params_from(d, "dist::head::").items()})
    eng.load_state_dict(sd)
    km = float(d["kmeans"])
    eng.set_hyper(vq_beta=1.0, km_latent=km, km_loss=1.0 if km else 0.0, clip=0.75, wd=1e-4, lambda_distill=1.3,
                  distill_T=0.5, conf_w=1.0, conf_thr=0.2)
    eng.push_hyper()
    eng.vq_loss_grads(x, a, torch.from_numpy(d["dist::tau"]).to(device))
    logs = eng.read_vq_logs()
    for k in ("total_loss", "distill_loss", "reconstruct_loss", "enc_rec_loss"):
        np.testing.assert_allclose(logs[k], float(d[f"dist::log::{k}"]), rtol=1e-4, atol=1e-5, err_msg=k)
    for k in d:
        if k.startswith("dist::grad::"):
            g = eng.view(k[12:], eng.grads).cpu().numpy()
            np.testing.assert_allclose(g, d[k].reshape(g.shape), atol=5e-5, rtol=5e-4, err_msg=k)
    # ---- contrastive
    d = load_golden(golden_dir, "contrastive_rec28.npz")
    pfx = "c0::"
    x_full = torch.from_numpy(d["x_full"]).to(device)
    ei = torch.from_numpy(d["edge_index"]).to(device)
    B, Tf, N, _ = x_full.shape
    L = d[pfx + "sd::encoder.final_dense.bias"].shape[0]
    Kd = d["dist::tau"].shape[1]
    e1 = VadeEngine(lib, device, B, Tf // 2, d["adj"], L, Kd, kind="contrastive")
    e2 = VadeEngine(lib, device, B, Tf // 2, d["adj"], L, Kd, kind="contrastive", shared=e1)
    sd = params_from(d, pfx + "sd::")
    sd.update({"distill_head." + k: v for k, v in params_from(d, "dist::head::").items()})
    e1.load_state_dict(sd)
    e1.set_hyper(lambda_distill=0.9, distill_T=0.5, conf_w=1.0, conf_thr=0.2)
    e1.push_hyper()
    xc, ac = contrastive_views(lib, x_full, ei, None)
    xa, aa = contrastive_views(lib, x_full, ei, aug_from_golden(d, pfx, device))
    z, z_aug = e1.contrastive_encode(xc, ac, train=True), e2.contrastive_encode(xa, aa, train=True)
    dz, dza = e1.contrastive_loss(z, z_aug, str(d[pfx + "sim"]), str(d[pfx + "loss_fn"]), 0.1, 0.1, 0.1,
                                  teacher_tau=torch.from_numpy(d["dist::tau"]).to(device))
    logs = e1.read_contrastive_logs()
    for k in ("total_loss", "distill_loss", "pos_similarity", "neg_similarity"):
        np.testing.assert_allclose(logs[k], float(d[f"dist::log::{k}"]), rtol=1e-4, atol=1e-5, err_msg=k)
    e1.contrastive_backward(dz, accumulate=False)
    e2.contrastive_backward(dza, accumulate=True)
    n = 0
    for k in d:
        if k.startswith("dist::grad::"):
            g = e1.view(k[12:], e1.grads).cpu().numpy()
            np.testing.assert_allclose(g, d[k].reshape(g.shape), atol=5e-5, rtol=1e-3, err_msg=k)
            n += 1
    assert n >= 42


def run_step_begin_check(lib, device, sizes=(1029, 37), seed=0x1234_5678_9ABC_DEF1, calls=3):
    """dof_step_begin: noise fills vs the numpy Philox / Box-Muller restatement (oracle.noise) call after call, the
    device-side call counter, and the schedule items it applies in the same launch."""
    import ctypes as C
    from oracle import noise as ON
    hyper = torch.zeros(_capi.H_COUNT, dtype=torch.float32, device=device)
    table = torch.tensor([0.25, 0.5, 0.75], dtype=torch.float32, device=device)
    cursor = torch.zeros(1, dtype=torch.int32, device=device)
    state = torch.zeros(2, dtype=torch.int32, device=device)
    outs = [torch.full((n,), float("nan"), dtype=torch.float32, device=device) for n in sizes]
    stream = torch.cuda.current_stream().cuda_stream if device != "cpu" else 0
    items = (_capi.SchedItem * 1)(_capi.SchedItem(table.data_ptr(), cursor.data_ptr(), 3, _capi.H_KLW, 1, 2.0))
    bufs = (_capi.NoiseBuf * len(sizes))(*[_capi.NoiseBuf(o.data_ptr(), o.numel()) for o in outs])
    for call in range(calls):
        _capi.check(lib, lib.dof_step_begin(hyper.data_ptr(), items, 1, seed, state.data_ptr(), bufs, len(sizes), stream))
        for i, o in enumerate(outs):
            ref = ON.normal_fill(o.numel(), seed, i, call)
            np.testing.assert_allclose(o.cpu().numpy(), ref, atol=2e-5, rtol=1e-5)
        assert state.cpu().tolist() == [call + 1, 0]
        assert float(hyper[_capi.H_KLW]) == 2.0 * [0.25, 0.5, 0.75][min(call, 2)]
    assert int(cursor.item()) == calls
    # items only (validation step without noise): nothing but the schedule moves
    _capi.check(lib, lib.dof_step_begin(hyper.data_ptr(), items, 1, seed, None, None, 0, stream))
    assert state.cpu().tolist() == [calls, 0] and int(cursor.item()) == calls + 1


def run_vade_rec_vs_oracle(lib, device, K, L=8, B=21, T=9, S=5, seed=11):
    """Recurrent VaDE, main phase with the teacher and every optional regulariser switched on, at component counts the
    goldens do not have (K = 25: two components per lane of the row kernels; K = 40: the one-thread-per-window
    kernels) and a batch that is not a multiple of 16: logged loss terms and all gradients vs autograd of the oracle."""
    from oracle import vade as OV
    adj = np.zeros((5, 5), np.float32)
    for i, j in ((0, 1), (1, 2), (2, 3), (3, 4), (1, 3)):
        adj[i, j] = adj[j, i] = 1.0
    E = 5
    eng = VadeEngine(lib, device, B, T, adj, L, K, mc_samples=S)
    g = torch.Generator().manual_seed(seed)
    P = eng.state_dict()
    for n, v in P.items():
        if v.dtype.is_floating_point and n.split(".")[-1] not in ("laplacian", "edge_laplacian", "incidence", "prior", "pretrain"):
            P[n] = torch.randn(v.shape, generator=g) * (0.3 if v.dim() > 1 else 0.05)
    P["latent_space.gmm_means"] = torch.randn(K, L, generator=g) * 0.7
    P["latent_space.gmm_log_vars"] = torch.randn(K, L, generator=g) * 0.5
    P["latent_space.prior"] = torch.softmax(torch.randn(K, generator=g), 0)
    eng.load_state_dict(P)
    P = eng.state_dict()
    x = torch.randn(B, T, 5, 3, generator=g).cumsum(1) * 0.3
    a = torch.randn(B, T, E, 1, generator=g)
    eps = torch.randn(B, L, generator=g)
    eps_mc = torch.randn(S, B, L, generator=g)
    tau = torch.softmax(torch.randn(B, K, generator=g) * 2, dim=-1)
    extra = PHASES["mainX"]["extra"]
    klw, lam = 0.45, 1.7
    configure_phase(eng, K, False, klw, tau, lam, extra)
    dev = lambda t: t.to(device)
    eng.loss_grads(dev(x), dev(a), dev(eps), dev(eps_mc), dev(tau), pretrain=False)
    pi = tau.mean(dim=0).clamp_min(1e-8)
    w = pi.pow(-1.0)
    cfg = OV.VadeLossCfg(K, False, lambda_distill=lam, class_weight=(w / w.mean()).clamp_max(3.0), teacher_marginal=pi,
                         repel_weight=extra["repel_w"], repel_length_scale=extra["repel_ls"],
                         reg_scatter_weight=extra["scatter_w"], reg_scatter_beta=extra["scatter_beta"],
                         temporal_cohesion_weight=extra["temporal_w"], reg_cat_clusters=extra["cat_w"],
                         tf_cluster_weight=extra["tf_w"], kmeans_loss_weight=extra["km_loss"],
                         distill_conf_weight=True, distill_conf_thresh=extra["conf_thr"])
    ref, grads, _ = OV.vade_grads(P, x, a, cfg, klw, eps, eps_mc, tau)
    logs = eng.read_logs()
    for k, v in ref.items():
        np.testing.assert_allclose(logs[k], float(v.detach()), rtol=2e-4, atol=2e-5, err_msg=k)
    checked = 0
    for name, gr in grads.items():
        if gr is None:
            continue
        got = eng.view(name, eng.grads).cpu().numpy()
        np.testing.assert_allclose(got, gr.numpy().reshape(got.shape), atol=5e-5, rtol=5e-4, err_msg=name)
        checked += 1
    assert checked >= 75, checked


RELU_TIE_MARGIN = 1e-5


def _relu_tie_margin(fn):
    """Smallest |x| over every ReLU input while fn() runs (the oracle's torch.relu is patched for the call)."""
    orig = torch.relu
    seen = [float("inf")]

    def spy(t):
        v = t.detach().abs()
        v = v[v > 0]  # exact zeros are structural (both sides evaluate them identically: mask 0)
        if v.numel():
            seen[0] = min(seen[0], float(v.min()))
        return orig(t)
    torch.relu = spy
    try:
        fn()
    finally:
        torch.relu = orig
    return seen[0]


def run_vade_tcn_vs_oracle(lib, device, L=4, K=3, B=6, T=10, seed=3):
    """VaDE-TCN at a latent size whose decoder input (4L channels) needs the zero-padded MFMA operand path
    (4L < 32): eval forward and train-step gradients vs the CPU oracle, fp64-anchored as in run_vade_tcn_check.

    The draw is TIE-FREE: over 6 windows one flipped ReLU mask moves a block's gradients by 2-8 % (DESIGN.md section 3,
    "ReLU-mask ties"), and which side of zero a pre-activation within ~1e-6 of it lands on is decided by the rounding
    of whichever fp32 implementation evaluates it.  So the seed is advanced until the oracle's float64 train-mode
    forward has no ReLU input within RELU_TIE_MARGIN of zero (a property of the oracle and the draw only -- the same
    seed on the emulator and on the GPU); about one draw in twenty qualifies."""
    from oracle import vade as OV
    adj = np.zeros((4, 4), np.float32)
    for i in range(3):
        adj[i, i + 1] = adj[i + 1, i] = 1.0
    eng = VadeEngine(lib, device, B, T, adj, L, K, kind="vade_tcn")
    cfg = OV.VadeLossCfg(K, True)
    for attempt in range(400):
        g = torch.Generator().manual_seed(seed + 1000 * attempt)
        P = eng.state_dict()
        for n, v in P.items():
            if not v.dtype.is_floating_point or n.split(".")[-1] in ("laplacian", "edge_laplacian", "incidence", "prior", "pretrain"):
                continue
            if n.endswith("running_var"):
                P[n] = torch.rand(v.shape, generator=g) + 0.5
            elif n.endswith("running_mean"):
                P[n] = torch.randn(v.shape, generator=g) * 0.1
            elif (".bn" in n or "head.2" in n or "head.5" in n) and n.endswith("weight"):
                P[n] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
            else:
                P[n] = torch.randn(v.shape, generator=g) * (0.3 if v.dim() > 1 else 0.05)
        x = torch.randn(B, T, 4, 3, generator=g).cumsum(1) * 0.3
        a = torch.randn(B, T, 3, 1, generator=g)
        eps = torch.randn(B, L, generator=g)

        def train_forward():
            P64 = {k: (v.double() if v.dtype == torch.float32 else v.clone()) for k, v in P.items()}
            orig = torch.Tensor.float
            torch.Tensor.float = lambda self: self.double()
            try:
                with torch.no_grad():
                    OV.vade_forward(P64, x.double(), a.double(), training=True, eps=eps.double())
            finally:
                torch.Tensor.float = orig
        if _relu_tie_margin(train_forward) >= RELU_TIE_MARGIN:
            break
    else:
        raise AssertionError("no tie-free draw found")
    eng.load_state_dict(P)
    out = eng.forward(x.to(device), a.to(device), None, want_loc=True, want_enc=True)
    with torch.no_grad():
        ref = OV.vade_forward({k: v.clone() for k, v in P.items()}, x, a, training=False)
    sc_enc, sc_loc = max(1.0, float(ref["enc"].abs().max())), max(1.0, float(ref["loc"].abs().max()))
    np.testing.assert_allclose(out["enc"].cpu().numpy(), ref["enc"].numpy(), atol=2e-5 * sc_enc, rtol=1e-4)
    np.testing.assert_allclose(out["loc"].cpu().numpy(), ref["loc"].numpy(), atol=5e-5 * sc_loc, rtol=2e-4)
    configure_phase(eng, K, True, 0.2, None, 0.0)
    eng.loss_grads(x.to(device), a.to(device), eps.to(device), None, None, pretrain=True)
    (l32, g32, _), (l64, g64, _) = _oracle_truth(lambda Pq, xx, aa, ee: OV.vade_grads(Pq, xx, aa, cfg, 0.2, ee, None, None),
                                                 P, x, a, eps)
    np.testing.assert_allclose(eng.read_logs()["total_loss"], float(l64["total_loss"]), rtol=2e-4)
    n = 0
    for name, t in g64.items():
        if t is None or name not in eng.layout:
            continue
        got = eng.view(name, eng.grads).cpu().numpy().astype(np.float64)
        t = t.numpy().reshape(got.shape)
        noise = np.abs(g32[name].numpy().astype(np.float64).reshape(got.shape) - t).max()
        err = np.abs(got - t).max()
        # a bias in front of a BatchNorm has a mathematically zero gradient: what is left is the rounding of a sum of
        # the terms that also make up its weight's gradient, so the floor scales with those (2 fp32 ulps), not with |t|
        sib = g64.get(name[:-len("bias")] + "weight") if name.endswith(".bias") else None
        floor = 2.5e-7 * float(sib.abs().max()) if sib is not None else 0.0
        # noise is ONE fp32 evaluation's deviation on the host running the test (it moves with the host's BLAS): the bar is
        # run_vade_tcn_check's -- 10 noise units or TCN_B6_RTOL of the tensor scale, whichever is larger
        assert err <= max(10.0 * noise, TCN_B6_RTOL * np.abs(t).max()) + 5e-6 * np.abs(t).max() + 1e-6 + floor, (name, err, noise)
        n += 1
    assert n >= 200


# ---- pose-table preprocessing ---------------------------------------------------------
def load_preprocess_golden(golden_dir):
    import json
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    cases = json.loads(str(g["cases"]))
    data = {}
    for tag in ("pair", "single"):
        cols = [tuple(c) if isinstance(c, list) else c for c in json.loads(str(g[f"{tag}::columns"]))]
        tabs = {k.split("::")[-1]: g[k] for k in g.files if k.startswith(f"{tag}::raw::")}
        data[tag] = (cols, json.loads(str(g[f"{tag}::animal_ids"])), tabs)
    return g, cases, data


def preprocess_output_columns(cols):
    """Frame-table columns the way get_graph_dataset orders them: sorted nodes [x | y | speed], a subset of the
    distance columns as 'edges', all angles."""
    nodes = sorted({c[0] for c in cols if isinstance(c, tuple) and len(c) == 2 and c[1] in ("x", "y")})
    node_cols = [(n, "x") for n in nodes] + [(n, "y") for n in nodes] + nodes
    dist = [c for c in cols if isinstance(c, tuple) and len(c) == 2 and c[1] not in ("x", "y")]
    edge_cols = sorted(dist[::3])
    angle_cols = [c for c in cols if isinstance(c, tuple) and len(c) == 3]
    return node_cols, edge_cols, angle_cols


def _check_tables(res, exp, cols, node_cols, edge_cols, angle_cols, what):
    pos = {c: i for i, c in enumerate(cols)}
    assert res.keys == sorted(exp), what
    for i, k in enumerate(res.keys):
        lo, hi = int(res.video_off[i]), int(res.video_off[i + 1])
        for name, tab, sel in (("node", res.node_table, node_cols), ("edge", res.edge_table, edge_cols),
                               ("angle", res.angle_table, angle_cols)):
            if not sel:
                continue
            got = tab[lo:hi].cpu().numpy()
            want = exp[k][:, [pos[c] for c in sel]].astype(np.float32)
            assert got.dtype == np.float32 and np.isfinite(got).all()
            # float64 pipeline on both sides, compared after the fp32 cast: 1 fp32 ulp of slack
            np.testing.assert_allclose(got, want, rtol=2.5e-7, atol=1e-7, err_msg=f"{what} {k} {name}")


def run_preprocess_check(lib, device, golden_dir):
    from deepof_amd.preprocess import preprocess_tables
    g, cases, data = load_preprocess_golden(golden_dir)
    first_scaler = None
    for c in cases:
        cols, aids, tabs = data[c["data"]]
        node_cols, edge_cols, angle_cols = preprocess_output_columns(cols)
        res = preprocess_tables(tabs, cols, aids, node_cols, edge_cols, angle_cols, samples_max=c["samples_max"],
                                dist_standardize=c["dist"], speed_standardize=c["speed"], coord_standardize=c["coord"],
                                log_distances=c["log"], interpolate_normalized=c["clip"], device=device, lib=lib)
        exp = {k.split("::")[-1]: g[k] for k in g.files if k.startswith(c["case"] + "::out::")}
        _check_tables(res, exp, cols, node_cols, edge_cols, angle_cols, c["case"])
        for part in ("speed", "dist", "dist_inner", "dist_intra", "coord"):
            key = f"{c['case']}::scaler::{part}::mean"
            have = res.global_scaler is not None and res.global_scaler.get(part) is not None
            assert have == (key in g.files), (c["case"], part)
            if have:
                np.testing.assert_allclose(res.global_scaler[part][0], g[key], rtol=1e-9, atol=1e-12, err_msg=key)
                np.testing.assert_allclose(res.global_scaler[part][1], g[key.replace("mean", "scale")], rtol=1e-9, atol=1e-12)
        if first_scaler is None:
            first_scaler = (c, res.global_scaler)
    # the fitted scalers re-applied to other videos (pretrained_scaler path)
    c, gs = first_scaler
    cols, aids, _ = data["pair"]
    node_cols, edge_cols, angle_cols = preprocess_output_columns(cols)
    new = {k.split("::")[-1]: g[k] for k in g.files if k.startswith("pair::pre::raw::")}
    res = preprocess_tables(new, cols, aids, node_cols, edge_cols, angle_cols, dist_standardize=c["dist"],
                            speed_standardize=c["speed"], coord_standardize=c["coord"], pretrained_scaler=gs, device=device, lib=lib)
    _check_tables(res, {k: g[f"pair::pre::out::{k}"] for k in new}, cols, node_cols, edge_cols, angle_cols, "pretrained")


def load_preprocess_r03_golden(golden_dir):
    import json
    g = np.load(os.path.join(golden_dir, "preprocess_r03.npz"))
    cases = json.loads(str(g["cases"]))
    data = {}
    for tag in ("pair", "single", "filt", "ragged"):
        cols = [tuple(c) if isinstance(c, list) else c for c in json.loads(str(g[f"{tag}::columns"]))]
        tabs = {k.split("::")[-1]: g[k] for k in g.files if k.startswith(f"{tag}::raw::")}
        data[tag] = (cols, json.loads(str(g[f"{tag}::animal_ids"])), tabs)
    return g, cases, data


def run_preprocess_r03_check(lib, device, golden_dir):
    """scale="minmax" and filter_low_variance against the reference's own outputs (make_golden_preprocess_r03.py)."""
    from deepof_amd.preprocess import preprocess_tables
    g, cases, data = load_preprocess_r03_golden(golden_dir)
    first_mm = None
    for c in cases:
        cols, aids, tabs = data[c["data"]]
        node_cols, edge_cols, angle_cols = preprocess_output_columns(cols)
        kw = dict(samples_max=c["samples_max"], dist_standardize=c["dist"], speed_standardize=c["speed"],
                  coord_standardize=c["coord"], log_distances=c["log"], interpolate_normalized=c["clip"], scale=c["scale"],
                  filter_low_variance=c["filter"], device=device, lib=lib)
        res = preprocess_tables(tabs, cols, aids, node_cols, edge_cols, angle_cols, **kw)
        exp = {k.split("::")[-1]: g[k] for k in g.files if k.startswith(c["case"] + "::out::")}
        _check_tables(res, exp, cols, node_cols, edge_cols, angle_cols, c["case"])
        suffix = {"minmax": ("data_min", "data_range"), "robust": ("center", "scale"), "standard": ("mean", "scale")}[c["scale"]]
        for part in ("speed", "dist", "dist_inner", "dist_intra", "coord"):
            key = f"{c['case']}::scaler::{part}::{suffix[0]}"
            have = res.global_scaler is not None and res.global_scaler.get(part) is not None
            assert have == (key in g.files), (c["case"], part)
            if have:
                want1 = g[key.replace(suffix[0], suffix[1])].copy()
                if c["scale"] == "minmax":
                    want1[want1 < 10 * np.finfo(np.float64).eps] = 1.0
                np.testing.assert_allclose(res.global_scaler[part][0], g[key], rtol=1e-9, atol=1e-12, err_msg=key)
                np.testing.assert_allclose(res.global_scaler[part][1], want1, rtol=1e-9, atol=1e-12, err_msg=key)
        assert res.global_scaler["kind"] == c["scale"]
        if first_mm is None and c["scale"] == "minmax":
            first_mm = (c, res.global_scaler)
        if c["filter"] and c["dropped"] and "per_column" in (c["dist"], c["speed"], c["coord"]) and c["scale"] != "robust":
            # a scaler fitted on FILTERED tables (per-column sections hold the kept columns only) is reusable as the
            # pretrained scaler of tables that drop the same columns, like the reference's sklearn scalers
            again = preprocess_tables(tabs, cols, aids, node_cols, edge_cols, angle_cols, pretrained_scaler=res.global_scaler, **kw)
            _check_tables(again, exp, cols, node_cols, edge_cols, angle_cols, c["case"] + " (own scaler as pretrained)")
            with pytest.raises(ValueError):   # a scaler of another kind is refused, not silently applied
                preprocess_tables(tabs, cols, aids, node_cols, edge_cols, angle_cols, pretrained_scaler=res.global_scaler,
                                  **dict(kw, scale="standard" if c["scale"] != "standard" else "minmax"))
    c, gs = first_mm
    cols, aids, _ = data["pair"]
    node_cols, edge_cols, angle_cols = preprocess_output_columns(cols)
    new = {k.split("::")[-1]: g[k] for k in g.files if k.startswith("pair::pre::raw::")}
    res = preprocess_tables(new, cols, aids, node_cols, edge_cols, angle_cols, dist_standardize=c["dist"], scale="minmax",
                            speed_standardize=c["speed"], coord_standardize=c["coord"], pretrained_scaler=gs, device=device, lib=lib)
    _check_tables(res, {k: g[f"pair::pre::out::{k}"] for k in new}, cols, node_cols, edge_cols, angle_cols, "pretrained minmax")
    # what the reference refuses, refused the same way
    cols, aids, tabs = data["ragged"]
    node_cols, edge_cols, angle_cols = preprocess_output_columns(cols)
    with pytest.raises(ValueError):      # per-column global scalers over videos that kept different columns
        preprocess_tables(tabs, cols, aids, node_cols, edge_cols, angle_cols, dist_standardize="per_column",
                          speed_standardize="per_column", coord_standardize="per_column", filter_low_variance=1.2, device=device, lib=lib)
    with pytest.raises(ValueError):
        preprocess_tables(tabs, cols, aids, node_cols, edge_cols, angle_cols, scale="quantile", device=device, lib=lib)


def synth_raw_tables(n_videos, frames, bodyparts, seed, nan_rate=0.002):
    """Random-walk pose tables (coords, speeds, all pairwise distances) with sparse gaps; float64 (frames, C) per video."""
    from itertools import combinations
    rng = np.random.default_rng(seed)
    cols = [(bp, ax) for bp in bodyparts for ax in ("x", "y")] + list(bodyparts) + [tuple(p) for p in combinations(bodyparts, 2)]
    pairs = [c for c in cols if isinstance(c, tuple) and c[1] not in ("x", "y")]
    ia = np.array([bodyparts.index(p[0]) for p in pairs])
    ib = np.array([bodyparts.index(p[1]) for p in pairs])
    tabs = {}
    for v in range(n_videos):
        n = frames if np.isscalar(frames) else frames[v]
        pos = np.cumsum(rng.standard_normal((n, len(bodyparts), 2)), axis=0) + rng.uniform(-30, 30, (len(bodyparts), 2)) * (1 + 0.3 * v)
        speed = np.concatenate([np.zeros((1, len(bodyparts))), np.linalg.norm(np.diff(pos, axis=0), axis=2)])
        dist = np.linalg.norm(pos[:, ia] - pos[:, ib], axis=2)
        t = np.concatenate([pos.reshape(n, -1), speed, dist], axis=1)
        t[rng.random(t.shape) < nan_rate] = np.nan
        tabs[f"v{v:03d}"] = t
    return tabs, cols


def run_preprocess_vs_oracle(lib, device, n_videos=3, frames=(300, 97, 161), seed=5, samples_max=227272, parts=5, edge_stride=3,
                             n_angles=0, nan_rate=0.02, **modes):
    """Device path against the (reference-pinned) oracle on larger random tables; tiles, strips and videos ragged.
    ``parts`` body parts per animal (2 animals), every ``edge_stride``-th distance column is an output edge,
    ``n_angles`` angle columns (interpolated only)."""
    from deepof_amd.preprocess import preprocess_tables
    from oracle import preprocess as op
    names = ["Nose", "Center", "Tail_base", "Left_ear", "Right_ear"] + [f"Spine_{i}" for i in range(40)]
    bps = [f"{a}_{p}" for a in ("B", "W") for p in names[:parts]]
    tabs, cols = synth_raw_tables(n_videos, frames, bps, seed, nan_rate=nan_rate)
    rng = np.random.default_rng(seed + 1)
    if n_angles:
        cols = cols + [(bps[i], bps[i + 1], bps[i + 2]) for i in range(n_angles)]
        for k in tabs:
            ang = rng.uniform(0, np.pi, (tabs[k].shape[0], n_angles))
            ang[rng.random(ang.shape) < 0.05] = np.nan
            tabs[k] = np.concatenate([tabs[k], ang], axis=1)
    k0 = sorted(tabs)[0]
    n0 = tabs[k0].shape[0]
    if n0 > 60:
        tabs[k0][:40, 3] = np.nan                          # a leading gap longer than a tile
        tabs[k0][n0 // 3: n0 // 3 + min(130, n0 // 3), 7] = np.nan   # an interior gap spanning several tiles
        tabs[k0][-50:, len(bps) * 2 + 1] = np.nan          # a trailing gap
    node_cols, _, angle_cols = preprocess_output_columns(cols)
    dist = [c for c in cols if isinstance(c, tuple) and len(c) == 2 and c[1] not in ("x", "y")]
    edge_cols = sorted(dist[::edge_stride])
    kw = dict(dist_standardize=modes.get("dist", "groupwise"), speed_standardize=modes.get("speed", "groupwise"),
              coord_standardize=modes.get("coord", "groupwise"))
    kw["scale"] = modes.get("scale", "standard")
    want, gs = op.preprocess(tabs, cols, ["B", "W"], samples_max=samples_max, **kw)
    res = preprocess_tables(tabs, cols, ["B", "W"], node_cols, edge_cols, angle_cols, samples_max=samples_max, device=device, lib=lib, **kw)
    _check_tables(res, want, cols, node_cols, edge_cols, angle_cols, f"oracle {modes}")
    sf = res.size_factors.cpu().numpy()
    for i, k in enumerate(res.keys):
        s_by, dflt = op.size_factors(tabs[k], cols, ["B", "W"])
        np.testing.assert_allclose(sf[i], [s_by["B"], s_by["W"], dflt], rtol=1e-15, atol=0)   # exact selection
    return res


# ------------------------------------------------------------------------------------------------
# R16: replay of the reference's fit_* traces (tests/golden/make_golden_fit.py) through deepof_amd.training.fit_*
# ------------------------------------------------------------------------------------------------
def run_fit_trace_check(engine_factory, device, golden_dir, model_name, rtol=0.03):
    """Same data, initial weights, batch order, injected noise and configuration as the reference run; compares the
    learning rates held in every training epoch (Q22), the KL weight reported per epoch, the epochs saved as
    best_val / best_score (Q19) exactly, and every column of the per-epoch log_summary within ``rtol``."""
    import tempfile
    from deepof_amd import training as TR
    from deepof_amd.config import CommonFitCfg, ContrastiveCfg, TurtleTeacherCfg, VaDECfg
    from deepof_amd.dataset import WindowDataset
    from deepof_amd.graph import bodypart_graph, make_meta_info
    from deepof_amd import models as MD
    from noise_streams import noise
    d = load_golden(golden_dir, "fit_traces.npz")
    p = f"trace::{model_name}::"
    T_full, L, K, BS, n_train, n_val, epochs = (int(v) for v in d[p + "cfg"])
    x, a, adj = d[p + "x"], d[p + "a"], d[p + "adj"]
    sd0 = params_from(d, p + "sd0::")

    def dataset(lo, hi):
        ds = WindowDataset(device)
        ds.x, ds.a = torch.from_numpy(x[lo:hi]).to(device), torch.from_numpy(a[lo:hi]).to(device)
        ds.video_idx = np.zeros(hi - lo, dtype=np.int32)
        ds.length, ds.keys = hi - lo, ["v"]
        ds.x_shape, ds.a_shape = tuple(x.shape[1:]), tuple(a.shape[1:])
        ds.angles = None
        return ds

    train_ds, val_ds = dataset(0, n_train), dataset(n_train, n_train + n_val)
    out_dir = tempfile.mkdtemp()
    common = CommonFitCfg(model_name=model_name, encoder_type="recurrent", batch_size=BS, latent_dim=L, epochs=epochs,
                          n_components=K, output_path=out_dir, save_weights=True, seed=0, diag_max_batches=4,
                          learning_rate=3e-4)   # the dataclass default the reference run used
    teacher = TurtleTeacherCfg(use_turtle_teacher=False)
    vade = VaDECfg(pretrain_epochs=2, kl_warmup=2, kl_cooldown=1, kl_warmup_pretrain=2, kl_cooldown_pretrain=1)
    ccfg = ContrastiveCfg(aug_p_shift=0.0, aug_p_rot=0.0, aug_p_interp=0.0, aug_p_noise=0.0)
    cls = {"vade": MD.VaDE, "vqvae": MD.VQVAE, "contrastive": MD.Contrastive}[model_name]
    rec = {"saves": [], "lrs": [], "klw": []}
    orig_init, orig_save, orig_epoch = cls.__init__, TR.save_model_info, TR.VadeStepper.train_epoch

    def init(self, *args, **kw):
        orig_init(self, *args, **kw)
        self.load_state_dict(sd0, strict=False)

    def save(path, *args, stage=None, epoch=None, **kw):
        rec["saves"].append((str(stage), int(epoch)))
        return orig_save(path, *args, stage=stage, epoch=epoch, **kw)

    def train_epoch(self, dataset_, seed, shuffle=True):
        h = self.model._base.hyper_host
        rec["lrs"].append([float(h[_capi.H_LR0 + _capi.SEG_ENCODER]), float(h[_capi.H_LR0 + _capi.SEG_GMM])])
        res = orig_epoch(self, dataset_, seed, shuffle)
        rec["klw"].append(float(res[1]))
        return res

    cls.__init__, TR.save_model_info, TR.VadeStepper.train_epoch, TR.NOISE_HOOK = init, save, train_epoch, noise
    torch.manual_seed(0)
    np.random.seed(0)
    try:
        if model_name == "vade":
            res = TR.fit_VADE(train_ds, val_ds, {}, adj, common, teacher, vade, device=device, _engine_factory=engine_factory)
        elif model_name == "vqvae":
            res = TR.fit_VQVAE(train_ds, val_ds, {}, adj, common, teacher, device=device, _engine_factory=engine_factory)
        else:
            nodes, edges = bodypart_graph([""])
            res = TR.fit_contrastive(train_ds, val_ds, {}, adj, make_meta_info(nodes, edges), common, teacher, ccfg,
                                     device=device, _engine_factory=engine_factory)
    finally:
        cls.__init__, TR.save_model_info, TR.VadeStepper.train_epoch, TR.NOISE_HOOK = orig_init, orig_save, orig_epoch, None
    log_summary = res[3]
    # ---- exact items
    ref_saves = list(zip([str(v) for v in d[p + "saves_stage"]], [int(v) for v in d[p + "saves_epoch"]]))
    if model_name == "vade":
        np.testing.assert_allclose(np.array(rec["lrs"]), d[p + "lrs"], rtol=1e-6)       # pretrain lr / 0, then 5e-4 / 2e-4
        np.testing.assert_allclose(np.array(rec["klw"]), d[p + "klw"], rtol=1e-6, atol=1e-9)
    report = {}
    for split in ("train", "val"):
        for key, ours in log_summary[split].items():
            ref = d[p + f"log::{split}::{key}"]
            ours = np.array([float(v) for v in ours], dtype=np.float64)
            assert ours.shape == ref.shape, (split, key, ours.shape, ref.shape)
            assert np.array_equal(np.isnan(ours), np.isnan(ref)), (split, key, ours, ref)
            ok = ~np.isnan(ref)
            if ok.any():
                err = np.abs(ours[ok] - ref[ok]) / (np.abs(ref[ok]) + 1e-3)
                report[f"{split}::{key}"] = float(err.max())
    bad = {k: v for k, v in report.items() if v > rtol}
    assert not bad, (bad, report)
    assert rec["saves"] == ref_saves, (rec["saves"], ref_saves)
    return report


def run_checkpoint_rules_check(golden_dir):
    """deepof_amd.training.CheckpointSelector against the epochs the REFERENCE's fit_* functions saved when their epoch
    functions returned scripted validation totals / alignment scores (make_golden_fit.py, part B)."""
    from deepof_amd.training import CheckpointSelector
    d = load_golden(golden_dir, "fit_traces.npz")
    n = 0
    for model_name in ("vade", "vqvae", "contrastive"):
        for sname in ("a", "b"):
            p = f"rules::{model_name}::{sname}::"
            val, score = d[p + "val"], d[p + "score"]
            sel = CheckpointSelector(len(val), rising_start=model_name == "vade")
            saves = []
            for ep in range(len(val)):
                # the reference never reads the contrastive model's alignment score (training.py:1447 is commented
                # out): its best_score checkpoint is never written
                sc = float("nan") if model_name == "contrastive" else float(score[ep])
                sv, ss = sel.update(ep, float(val[ep]), sc, has_score=True)
                if sv:
                    saves.append(("best_val", ep))
                if ss:
                    saves.append(("best_score", ep))
            ref = list(zip([str(v) for v in d[p + "saves_stage"]], [int(v) for v in d[p + "saves_epoch"]]))
            if model_name != "vade":   # those two loops test the validation loss first, the score second, like VaDE
                pass
            assert sorted(saves) == sorted(ref), (model_name, sname, saves, ref)
            n += 1
    return n


# ------------------------------------------------------------------------------------------------
# transformer family (R17): the HIP path against the REFERENCE goldens of tests/golden/make_golden_tfm.py
# ------------------------------------------------------------------------------------------------
def tfm_masks(d, prefix, site_names):
    """The golden's recorded keep-masks (draw order) keyed by the plan's dropout-site names (same order)."""
    keys = sorted(k for k in d if k.startswith(prefix + "drop::"))
    assert len(keys) == len(site_names), (len(keys), site_names)
    return {name: torch.from_numpy(d[k]) for name, k in zip(site_names, keys)}


def _tfm_grad_check(eng, d, prefix, min_count, zero_names=("encoder.head.6.bias", "encoder.head.5.bias")):
    n, worst = 0, 0.0
    for k in d:
        if k.startswith(prefix + "grad::"):
            name = k.split("::")[-1]
            g = eng.view(name, eng.grads).cpu().numpy()
            ref = d[k].reshape(g.shape)
            scale = float(np.abs(ref).max())
            if name in zero_names:  # a constant in front of the batch standardisation: rounding noise on both sides
                assert scale < 1e-4 and float(np.abs(g).max()) < 1e-4, name
            else:
                # + the reference's own sensitivity to the ReLU inputs whose sign is decided by fp32 rounding
                # (|x| < 2e-6, about nine elements per step; tests/golden/make_golden_tfm.py ReluKinkFlipper)
                kink = float(d[prefix + "gkink::" + name])
                err = float(np.abs(g - ref).max())
                assert err <= 5e-5 + 5e-4 * scale + kink, (prefix, name, err, scale, kink)
                worst = max(worst, err / max(scale, 1e-30))
            n += 1
    assert n >= min_count, n
    return worst


def run_vade_tfm_check(lib, device, golden_dir):
    """VaDEPT(encoder_type="transformer"): eval forward (also with padded keys / masked frames), the pre-training and
    the main (+ teacher) objective on the reference's recorded dropout masks: every loss term, the train-mode outputs,
    all 108 gradients, BatchNorm buffers and step counters."""
    d = load_golden(golden_dir, "vade_tfm14.npz")
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vade_tfm")
    sd0 = params_from(d)
    eng.load_state_dict(sd0)
    assert list(eng.state_dict().keys()) == list(sd0.keys())
    eng.set_bn_training(False)
    for pfx, xx, aa in (("eval_", x, a), ("evalm_", torch.from_numpy(d["xm"]).to(device), torch.from_numpy(d["am"]).to(device))):
        out = eng.forward(xx, aa, None, want_loc=True, want_enc=True)
        if pfx == "eval_":
            np.testing.assert_allclose(out["enc"].cpu().numpy(), d["eval_enc"], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(out["z"].cpu().numpy(), d[pfx + "z"], atol=2e-5, rtol=1e-4)
        np.testing.assert_allclose(out["q"].cpu().numpy(), d[pfx + "q"], atol=1e-5, rtol=1e-3)
        np.testing.assert_allclose(out["loc"].cpu().numpy(), d[pfx + "loc"], atol=5e-5, rtol=1e-4)
    eng.set_bn_training(True)
    eps, eps_mc = torch.from_numpy(d["eps"]).to(device), torch.from_numpy(d["eps_mc"]).to(device)
    tau = torch.from_numpy(d["tau"]).to(device)
    sites = [s[0] for s in eng.dropout_sites()]
    assert len(sites) == 22
    worst = {}
    for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
        eng.load_state_dict(sd0)
        eng.set_dropout(tfm_masks(d, phase + "::", sites))
        configure_phase(eng, K, phase == "pre", klw, tau if teacher else None, 1.7 if teacher else 0.0)
        eng.loss_grads(x, a, eps, None if phase == "pre" else eps_mc, tau if teacher else None, pretrain=phase == "pre")
        logs = eng.read_logs()
        n_terms = 0
        for k, v in logs.items():
            key = f"{phase}::loss::{k}"
            if key in d:
                np.testing.assert_allclose(v, float(d[key]), rtol=1e-4, atol=1e-5, err_msg=key)
                n_terms += 1
        assert n_terms >= 12
        worst[phase] = _tfm_grad_check(eng, d, phase + "::", 108)
        if phase == "pre":
            sd1 = eng.state_dict()
            nb = 0
            for k in d:
                if k.startswith("pre::sd_after::"):
                    name = k[len("pre::sd_after::"):]
                    if name.endswith("num_batches_tracked"):
                        assert int(sd1[name]) == int(d[k]), name
                    else:
                        np.testing.assert_allclose(sd1[name].numpy(), d[k], atol=5e-6, rtol=5e-5, err_msg=name)
                    nb += 1
            assert nb == 6
    eng.set_dropout(None)
    return worst


def run_vqvae_tfm_check(lib, device, golden_dir):
    """VQVAEPT(encoder_type="transformer"): eval forward (code indices exact), one step_vqvae_distill step: logs and
    all gradients; the two decoder passes run on their own recorded masks."""
    d = load_golden(golden_dir, "vqvae_tfm14.npz")
    x, a = torch.from_numpy(d["x"]).to(device), torch.from_numpy(d["a"]).to(device)
    B, T, N, _ = x.shape
    L, K = d["sd::vq_layer.codebook"].shape
    eng = VadeEngine(lib, device, B, T, d["adj"], L, K, kind="vqvae_tfm")
    sd0 = params_from(d)
    eng.load_state_dict(sd0)
    assert list(eng.state_dict().keys()) == list(sd0.keys())
    eng.set_bn_training(False)
    out = eng.vq_forward(x, a)
    np.testing.assert_array_equal(out["idx"].cpu().numpy(), d["eval_idx"])
    np.testing.assert_allclose(out["ze"].cpu().numpy(), d["eval_ze"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(out["soft_counts"].cpu().numpy(), d["eval_soft_counts"], atol=1e-5, rtol=2e-3)
    np.testing.assert_allclose(out["loc_q"].cpu().numpy(), d["eval_loc_q"], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(out["loc_e"].cpu().numpy(), d["eval_loc_e"], atol=5e-5, rtol=1e-4)
    eng.set_bn_training(True)
    sites = [s[0] for s in eng.dropout_sites()]
    assert len(sites) == 30
    eng.set_dropout(tfm_masks(d, "", sites))
    eng.set_hyper(vq_beta=1.0, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
    eng.push_hyper()
    eng.vq_loss_grads(x, a)
    logs = eng.read_vq_logs()
    for k in ("total_loss", "enc_rec_loss", "reconstruct_loss", "vq_loss"):
        np.testing.assert_allclose(logs[k], float(d[f"log::{k}"]), rtol=1e-4, atol=1e-5, err_msg=k)
    worst = _tfm_grad_check(eng, d, "", 100)
    sd1 = eng.state_dict()
    for k in d:
        if k.startswith("sd_after::") and "running_" in k:
            np.testing.assert_allclose(sd1[k[len("sd_after::"):]].numpy(), d[k], atol=5e-6, rtol=5e-5, err_msg=k)
    eng.set_dropout(None)
    return worst


def run_contrastive_tfm_check(lib, device, golden_dir):
    """ContrastivePT(encoder_type="transformer"): eval embeddings, train-mode embeddings of two views on the recorded
    masks, NCE / cosine loss and every gradient (both views accumulate)."""
    d = load_golden(golden_dir, "contrastive_tfm14.npz")
    x, a, xa, aa = (torch.from_numpy(d[k]).to(device) for k in ("x", "a", "x_aug", "a_aug"))
    B, T, N, _ = x.shape
    L = d["sd::encoder.head.6.bias"].shape[0]
    e1 = VadeEngine(lib, device, B, T, d["adj"], L, 1, kind="contrastive_tfm")
    e2 = VadeEngine(lib, device, B, T, d["adj"], L, 1, kind="contrastive_tfm", shared=e1)
    sd0 = params_from(d)
    e1.load_state_dict(sd0)
    assert list(e1.state_dict().keys()) == list(sd0.keys())
    np.testing.assert_allclose(e1.contrastive_encode(x, a, train=False).cpu().numpy(), d["eval_z"], atol=2e-5, rtol=1e-4)
    sites = [s[0] for s in e1.dropout_sites()]
    assert len(sites) == 14
    keys = sorted(k for k in d if k.startswith("drop::"))
    assert len(keys) == 28
    e1.set_dropout({n: torch.from_numpy(d[k]) for n, k in zip(sites, keys[:14])})
    e2.set_dropout({n: torch.from_numpy(d[k]) for n, k in zip(sites, keys[14:])})
    z = e1.contrastive_encode(x, a, train=True)
    z_aug = e2.contrastive_encode(xa, aa, train=True)
    np.testing.assert_allclose(z.cpu().numpy(), d["z"], atol=5e-5, rtol=2e-4)
    np.testing.assert_allclose(z_aug.cpu().numpy(), d["z_aug"], atol=5e-5, rtol=2e-4)
    dz, dza = e1.contrastive_loss(z, z_aug, "cosine", "nce", 0.1, 0.1, 0.1)
    logs = e1.read_contrastive_logs()
    np.testing.assert_allclose([logs["total_loss"], logs["pos_similarity"], logs["neg_similarity"]], d["loss"], rtol=1e-4, atol=1e-5)
    e1.contrastive_backward(dz, accumulate=False)
    e2.contrastive_backward(dza, accumulate=True)
    worst = _tfm_grad_check(e1, d, "", 68)
    sd1 = e1.state_dict()
    for k in d:
        if k.startswith("sd_after::") and "running_" in k:
            np.testing.assert_allclose(sd1[k[len("sd_after::"):]].numpy(), d[k], atol=5e-6, rtol=5e-5, err_msg=k)
    return worst


def run_tfm_widths_vs_oracle(lib, device, n_nodes, latent, B=6, T=10, K=5, seed=0, kind="vade"):
    """Transformer family at other widths than the goldens' (key_dim 24 / 32 / 48 / 64, latent 4 / 6, decoder widths
    16 / 24): eval forward and one train step (all gradients) against the oracle (pinned to the reference at key_dim
    40, latent 8) on random keep-masks injected into both.  Chain graph of n_nodes body parts."""
    from oracle import tfm as OT
    from oracle import vade as OV
    from oracle import vqvae as OQ
    N = n_nodes
    adj = np.zeros((N, N), np.float32)
    for i in range(N - 1):
        adj[i, i + 1] = adj[i + 1, i] = 1
    E = N - 1
    eng = VadeEngine(lib, device, B, T, adj, latent, K, kind=f"{kind}_tfm")
    g = torch.Generator().manual_seed(seed)
    for n in eng.names:
        shape = eng.layout[n][2]
        v = torch.randn(shape, generator=g) * (0.25 if len(shape) > 1 else 0.1)
        if (".norm" in n or ".head.2" in n or ".head.5" in n) and n.endswith("weight"):
            v = 1.0 + v
        if n.endswith("running_var"):
            v = 0.5 + torch.rand(shape, generator=g)
        eng.view(n).copy_(v)
    P = {k: v.clone() for k, v in eng.state_dict().items()}
    x = torch.randn(B, T, N, 3, generator=g)
    a = torch.randn(B, T, E, 1, generator=g)
    x[1, 2] = 0.0
    a[1, 2] = 0.0                       # one padded / masked frame
    eng.set_bn_training(False)
    if kind == "vade":
        out = eng.forward(x.to(device), a.to(device), None, want_loc=True)
        with torch.no_grad():
            ref = OV.vade_forward({k: v.clone() for k, v in P.items()}, x, a, training=False)
        np.testing.assert_allclose(out["z"].cpu().numpy(), ref["z"].numpy(), atol=3e-5, rtol=2e-4)
        ref_l = ref["loc"].numpy()
        np.testing.assert_allclose(out["loc"].cpu().numpy(), ref_l, atol=max(1e-4, 5e-6 * float(np.abs(ref_l).max())), rtol=2e-4)
    else:
        out = eng.vq_forward(x.to(device), a.to(device))
        with torch.no_grad():
            ref = OQ.vqvae_forward({k: v.clone() for k, v in P.items()}, x, a, training=False)
        np.testing.assert_allclose(out["ze"].cpu().numpy(), ref["ze"].numpy(), atol=3e-5, rtol=2e-4)
        # (decoder outputs of this random initialisation reach |36| at latent 16: the absolute term follows the tensor
        # scale -- 5e-6 of it, i.e. about two fp32 ulps of the largest values -- and stays 1e-4 up to a scale of 20)
        ref_e = ref["loc_e"].numpy()
        np.testing.assert_allclose(out["loc_e"].cpu().numpy(), ref_e, atol=max(1e-4, 5e-6 * float(np.abs(ref_e).max())),
                                   rtol=2e-4)
    # train step on random masks (no masked frames: a masked frame makes the reconstruction loss NaN)
    x[1, 2] = torch.randn(N, 3, generator=g)
    a[1, 2] = torch.randn(E, 1, generator=g)
    eng.set_bn_training(True)
    sites = eng.dropout_sites()
    shapes = {}
    for name, _off, numel, p in sites:
        shapes[name] = (torch.rand(numel, generator=g) >= p).to(torch.uint8)
    eng.set_dropout(shapes)

    class Tape(OT.DropoutTape):  # hands out the named masks in the oracle's own draw order
        def __init__(self):
            super().__init__([])

        def take(self, site, shape, p):
            m = shapes[site].reshape(tuple(shape))
            self.named.append((site, m, p))
            return m.to(torch.float32) / (1.0 - p)

    eps = torch.randn(B, latent, generator=g)
    if kind == "vade":
        configure_phase(eng, K, True, 0.2, None, 0.0)
        eng.loss_grads(x.to(device), a.to(device), eps.to(device), None, None, pretrain=True)
        losses, grads, _ = OV.vade_grads({k: v.clone() for k, v in P.items()}, x, a, OV.VadeLossCfg(K, True), 0.2, eps,
                                         drop=Tape())
        np.testing.assert_allclose(eng.read_logs()["total_loss"], float(losses["total_loss"]), rtol=2e-4)
    else:
        eng.set_hyper(vq_beta=1.0, km_latent=0.0, km_loss=0.0, clip=0.75, wd=1e-4)
        eng.push_hyper()
        eng.vq_loss_grads(x.to(device), a.to(device))
        losses, grads, _ = OQ.vqvae_grads({k: v.clone() for k, v in P.items()}, x, a, 1.0, 0.0, drop=Tape())
        np.testing.assert_allclose(eng.read_vq_logs()["total_loss"], float(losses["total_loss"]), rtol=2e-4)
    worst, n = 0.0, 0
    for name, ref_g in grads.items():
        if ref_g is None or name in ("encoder.head.6.bias", "encoder.head.5.bias"):
            continue
        got = eng.view(name, eng.grads).cpu().numpy()
        r = ref_g.numpy().reshape(got.shape)
        scale = float(np.abs(r).max())
        err = float(np.abs(got - r).max())
        # small batches: a ReLU input within rounding of zero may flip (see ReluKinkFlipper); bound, do not equate
        assert err <= 1e-4 + 3e-3 * scale, (name, err, scale)
        worst = max(worst, err / max(scale, 1e-30))
        n += 1
    assert n >= 60
    eng.set_dropout(None)
    return worst
