"""Generate golden fixtures by running the REFERENCE (imported in place from /root/reference).

Run in the build container only:  ``python tests/golden/make_golden.py``.
Outputs small ``.npz`` files next to this script; they are data (inputs + expected outputs),
committed, and are what pins ``oracle/`` (tests/test_oracle_golden.py) and the HIP path.
The reference itself never travels: nothing here is imported by the package, the GPU tests,
``smoke()`` or ``bench.py``.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _ref_import import load_reference  # noqa: E402
from deepof_amd.graph import adjacency_from_graph, bodypart_graph  # noqa: E402

R = load_reference()
torch.set_num_threads(1)


def sd_np(model, prefix="sd::"):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def synth_batch(B, T, N, E, seed):
    """Smooth, standardised synthetic trajectories; no all-zero frames."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T + 8, N, 3)).astype(np.float32)
    a = rng.standard_normal((B, T + 8, E, 1)).astype(np.float32)
    k = np.ones(9, dtype=np.float32) / 3.0
    x = np.stack([np.convolve(x[b, :, n, f], k, mode="valid") for b in range(B) for n in range(N) for f in range(3)])
    a = np.stack([np.convolve(a[b, :, e, 0], k, mode="valid") for b in range(B) for e in range(E)])
    x = x.reshape(B, N, 3, T).transpose(0, 3, 1, 2).copy()
    a = a.reshape(B, E, 1, T).transpose(0, 3, 1, 2).copy()
    return x.astype(np.float32), a.astype(np.float32)


# ------------------------------------------------------------------------------------
def gen_scramble():
    out = {}
    for (T, G, F) in [(25, 14, 3), (25, 14, 1), (50, 28, 3), (50, 32, 1), (24, 11, 3), (7, 3, 2)]:
        src = torch.arange(T * G * F, dtype=torch.float32).reshape(1, T, G, F)
        y = R.M.RecurrentEncoderPT.tf_style_group_reshape(src, G, F)
        out[f"idx_{T}_{G}_{F}"] = y[0].numpy().astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "scramble.npz"), **out)


def gen_graph_ops():
    out = {}
    for tag, ids in [("single", [""]), ("pair", ["B", "W"])]:
        nodes, edges = bodypart_graph(ids)
        adj = adjacency_from_graph(nodes, edges)
        lap, elap, inc = R.C.CensNetConvPT.preprocess(torch.tensor(adj))
        out[f"{tag}_adj"] = adj
        out[f"{tag}_lap"] = lap.float().numpy()
        out[f"{tag}_elap"] = elap.float().numpy()
        out[f"{tag}_inc"] = inc.float().numpy()
    np.savez_compressed(os.path.join(HERE, "graph_ops.npz"), **out)


def gen_recurrent_block():
    """RecurrentBlockPT forward + parameter grads, incl. sequences with masked (all-zero conv) rows."""
    out = {}
    for tag, F, L in [("node", 3, 8), ("edge", 1, 8), ("node_l6", 3, 6)]:
        torch.manual_seed(11)
        blk = R.M.RecurrentBlockPT(input_features=F, latent_dim=L)
        B, G, T = 3, 5, 25
        rng = np.random.default_rng(5)
        x = rng.standard_normal((B, G, T, F)).astype(np.float32)
        # Q2: zero runs long enough (>= 5+k) that some conv rows vanish -> length < T
        x[0, 1, 8:16] = 0.0
        x[1, 3, 0:9] = 0.0
        x[2, 0, :] = 0.0  # whole sequence masked -> dropped from the packed GRU
        x[2, 4, 17:] = 0.0
        xt = torch.from_numpy(x)
        y = blk(xt)
        up = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        (y * up).sum().backward()
        out.update(sd_np(blk, f"{tag}::sd::"))
        out[f"{tag}::x"] = x
        out[f"{tag}::y"] = y.detach().numpy()
        out[f"{tag}::up"] = up.numpy()
        for n, p in blk.named_parameters():
            if p.grad is not None:
                out[f"{tag}::grad::{n}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "recurrent_block.npz"), **out)


def _cfgs(K, L, **vade_kw):
    common = R.U.CommonFitCfg(n_components=K, latent_dim=L, kmeans_loss=0.0)
    vade = R.U.VaDECfg(**vade_kw)
    teacher = R.U.TurtleTeacherCfg()
    return common, vade, teacher


def gen_vade(tag, ids, T, L, K, B, seed):
    nodes, edges = bodypart_graph(ids)
    adj = adjacency_from_graph(nodes, edges)
    N, E = len(nodes), len(edges)
    torch.manual_seed(seed)
    model = R.M.VaDEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type="recurrent", kmeans_loss=1.0)
    with torch.no_grad():  # make the GMM heads / prior-free bits non-trivial
        model.latent_space.gmm_means.mul_(3.0)
    x, a = synth_batch(B, T, N, E, seed + 1)
    xt, at = torch.from_numpy(x), torch.from_numpy(a)
    out = dict(sd_np(model))
    out.update(x=x, a=a, adj=adj)

    # ---- eval forward (bit-stable): z_mean, q, recon loc
    model.eval()
    with torch.no_grad():
        dist, z, q, km = model(xt, at)
        enc = model.encoder(xt, at)
    out.update(eval_z=z.numpy(), eval_q=q.numpy(), eval_loc=dist.base_dist.base_dist.loc.numpy(),
               eval_kmeans=np.float64(km), eval_enc=enc.numpy())

    # ---- train forward with known noise + every loss term + all grads, both phases
    eps = torch.randn(B, L, generator=torch.Generator().manual_seed(seed + 2))
    eps_mc = torch.randn(32, B, L, generator=torch.Generator().manual_seed(seed + 3))
    out.update(eps=eps.numpy(), eps_mc=eps_mc.numpy())
    tau = torch.softmax(torch.randn(B, K, generator=torch.Generator().manual_seed(seed + 4)) * 2.0, dim=-1)
    out["tau"] = tau.numpy()

    real_randn, real_randn_like = torch.randn, torch.randn_like

    def patched_randn(*size, **kw):
        if len(size) == 3 and tuple(size) == (32, B, L):
            return eps_mc.clone()
        return real_randn(*size, **kw)

    def patched_randn_like(t, **kw):
        if tuple(t.shape) == (B, L):
            return eps.clone()
        return real_randn_like(t, **kw)

    for phase, klw, with_teacher, extra in [
        ("pre", 0.13, False, {}),
        ("main", 0.7, False, {}),
        ("mainT", 0.7, True, {}),
        ("mainX", 0.45, True, dict(repel_weight=0.3, reg_scatter_weight=0.2, temporal_cohesion_weight=0.1,
                                   reg_cat_clusters=0.5, tf_cluster_weight=0.7)),
    ]:
        common, vade, teacher = _cfgs(K, L, **extra)
        if phase == "mainX":
            common.kmeans_loss = 0.5
            teacher.distill_conf_weight = True
        crit = R.L.VadeLoss(common_cfg=common, vade_cfg=vade, teacher_cfg=teacher)
        crit.set_mode("pretrain" if phase == "pre" else "main")
        crit.kl_scheduler = SimpleNamespace(get_weight=lambda k=klw: k, max_weight=1.0, current_iteration=0)
        if with_teacher:
            crit.set_teacher(tau_star=tau, lambda_distill=1.7)
        model.train()
        model.zero_grad(set_to_none=True)
        torch.randn, torch.randn_like = patched_randn, patched_randn_like
        try:
            outputs = model(xt, at, return_gmm_params=True)
            ld = crit(outputs, xt, batch_indices=torch.arange(B) if with_teacher else None)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        ld["total_loss"].backward()
        for k, v in ld.items():
            out[f"{phase}::loss::{k}"] = np.float64(float(v))
        out[f"{phase}::z"] = outputs[1].detach().numpy()
        out[f"{phase}::q"] = outputs[2].detach().numpy()
        out[f"{phase}::loc"] = outputs[0].base_dist.base_dist.loc.detach().numpy()
        for n, p in model.named_parameters():
            if p.grad is not None:
                out[f"{phase}::grad::{n}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"vade_{tag}.npz"), **out)


def gen_train_trace():
    """3 pretrain + 3 main reference optimisation steps (step_vade + clip + Adam), noise injected."""
    nodes, edges = bodypart_graph([""])
    adj = adjacency_from_graph(nodes, edges)
    N, E, T, L, K, B = len(nodes), len(edges), 25, 8, 10, 16
    torch.manual_seed(3)
    model = R.M.VaDEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type="recurrent", kmeans_loss=1.0)
    out = dict(sd_np(model, "sd0::"))
    common, vade, teacher = _cfgs(K, L)
    crit = R.L.VadeLoss(common_cfg=common, vade_cfg=vade, teacher_cfg=teacher)
    real_randn, real_randn_like = torch.randn, torch.randn_like
    step_id = 0
    for phase, nsteps, lr_b, lr_g, klws in [("pre", 3, 1e-3, 0.0, [0.0, 0.05, 0.2]), ("main", 3, 5e-4, 2e-4, [0.1, 0.6, 1.0])]:
        crit.set_mode("pretrain" if phase == "pre" else "main")
        model.set_pretrain_mode(phase == "pre")
        opt = R.L.build_optimizer_vade(model=model, base_lr=lr_b, gmm_lr=lr_g)
        for i in range(nsteps):
            x, a = synth_batch(B, T, N, E, 100 + step_id)
            eps = torch.randn(B, L, generator=torch.Generator().manual_seed(200 + step_id))
            eps_mc = torch.randn(32, B, L, generator=torch.Generator().manual_seed(300 + step_id))
            klw = klws[i]
            crit.kl_scheduler = SimpleNamespace(get_weight=lambda k=klw: k, max_weight=1.0, current_iteration=0)
            torch.randn = lambda *s, **kw: eps_mc.clone() if tuple(s) == (32, B, L) else real_randn(*s, **kw)
            torch.randn_like = lambda t, **kw: eps.clone() if tuple(t.shape) == (B, L) else real_randn_like(t, **kw)
            try:
                model.train()
                res = R.T.step_vade(model, (torch.from_numpy(x), torch.from_numpy(a), torch.arange(B)),
                                    SimpleNamespace(criterion=crit, train=True))
            finally:
                torch.randn, torch.randn_like = real_randn, real_randn_like
            opt.zero_grad(set_to_none=True)
            res.loss.backward()
            torch.nn.utils.clip_grad_value_(model.parameters(), 0.75)
            opt.step()
            out[f"step{step_id}::x"], out[f"step{step_id}::a"] = x, a
            out[f"step{step_id}::eps"], out[f"step{step_id}::eps_mc"] = eps.numpy(), eps_mc.numpy()
            out[f"step{step_id}::klw"] = np.float64(klw)
            out[f"step{step_id}::lr"] = np.array([lr_b, lr_g])
            out[f"step{step_id}::phase"] = np.array(phase)
            for k, v in res.logs.items():
                out[f"step{step_id}::log::{k}"] = np.float64(v)
            out[f"step{step_id}::pnorm"] = np.float64(
                float(torch.sqrt(sum((p.detach() ** 2).sum() for p in model.parameters()))))
            step_id += 1
    out.update(sd_np(model, "sd_final::"))
    np.savez_compressed(os.path.join(HERE, "vade_train_trace.npz"), **out)


def gen_schedules_kmeans():
    out = {}
    for mode in ["linear", "sigmoid", "tf_sigmoid"]:
        m = R.L.Dynamic_weight_manager(7, mode=mode, warmup_epochs=3, max_weight=0.8, at_max_epochs=2,
                                       cooldown_epochs=4, end_weight=0.25)
        ws = []
        for _ in range(80):
            ws.append(m.get_weight())
            m.step()
        out[f"sched_{mode}"] = np.array(ws)
    m = R.L.Dynamic_weight_manager(5, mode="tf_sigmoid", warmup_epochs=0, max_weight=4.0, at_max_epochs=2,
                                   cooldown_epochs=2, end_weight=0.2)
    ws = []
    for _ in range(30):
        ws.append(m.get_weight())
        m.step()
    out["sched_lambda"] = np.array(ws)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(64, 8, generator=g, requires_grad=True)
    with torch.no_grad():
        z[:, 3] = z[:, 2] * 0.5  # rank-deficient-ish
    km = R.L.compute_kmeans_loss_pt(z, 1.3)
    km.backward()
    out["km_z"] = z.detach().numpy()
    out["km_val"] = np.float64(float(km))
    out["km_grad"] = z.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "schedules_kmeans.npz"), **out)


def gen_vqvae(tag, ids, T, L, K, B, seed, kmeans=0.0):
    nodes, edges = bodypart_graph(ids)
    adj = adjacency_from_graph(nodes, edges)
    N, E = len(nodes), len(edges)
    torch.manual_seed(seed)
    model = R.M.VQVAEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type="recurrent", kmeans_loss=kmeans)
    with torch.no_grad():  # spread the codebook so that several codes are populated
        model.vq_layer.codebook.copy_(torch.randn(L, K) * 0.6)
    x, a = synth_batch(B, T, N, E, seed + 1)
    xt, at = torch.from_numpy(x), torch.from_numpy(a)
    out = dict(sd_np(model))
    out.update(x=x, a=a, adj=adj, kmeans=np.float64(kmeans))
    model.train()
    model.zero_grad(set_to_none=True)
    res = R.T.step_vqvae_distill(model, (xt, at, torch.arange(B)), SimpleNamespace(apply_distill=False))
    res.loss.backward()
    enc_rec, rec, quant, soft, ze, vql = model(xt, at, return_losses=True, return_all_outputs=True)
    out.update(quantized=quant.detach().numpy(), soft_counts=soft.detach().numpy(), ze=ze.detach().numpy(),
               loc_q=enc_rec.base_dist.base_dist.loc.detach().numpy(), loc_e=rec.base_dist.base_dist.loc.detach().numpy(),
               idx=model.vq_layer.get_code_indices(ze.detach()).numpy())
    for k, v in res.logs.items():
        out[f"log::{k}"] = np.float64(v)
    for n, p in model.named_parameters():
        if p.grad is not None:
            out[f"grad::{n}"] = p.grad.numpy().copy()
    # 3 optimisation steps with the generic optimiser (Adam + weight decay 1e-4), clip 0.75
    opt = R.L.build_optimizer_generic(model, None, base_lr=1e-3, weight_decay=1e-4)
    for i in range(3):
        xs, as_ = synth_batch(B, T, N, E, seed + 10 + i)
        r = R.T.step_vqvae_distill(model, (torch.from_numpy(xs), torch.from_numpy(as_), torch.arange(B)),
                                   SimpleNamespace(apply_distill=False))
        opt.zero_grad(set_to_none=True)
        r.loss.backward()
        torch.nn.utils.clip_grad_value_(model.parameters(), 0.75)
        opt.step()
        out[f"step{i}::x"], out[f"step{i}::a"] = xs, as_
        for k, v in r.logs.items():
            out[f"step{i}::log::{k}"] = np.float64(v)
    out.update(sd_np(model, "sd_final::"))
    if tag in ("rec28", "c5l8"):
        # distillation head (teacher_model.py:795-808) inside step_vqvae_distill, on the trained weights above
        import deepof.clustering.teacher_model as TM
        torch.manual_seed(seed + 77)
        head = TM.DiscriminativeHead(L, K)
        tau = torch.softmax(torch.randn(B, K) * 1.5, dim=-1)
        ctx = SimpleNamespace(apply_distill=True, distill_head=head, tau_star=tau, distill_sharpen_T=0.5,
                              distill_conf_weight=True, distill_conf_thresh=0.2,
                              lambda_scheduler=SimpleNamespace(get_weight=lambda: 1.3))
        model.zero_grad(set_to_none=True)
        r = R.T.step_vqvae_distill(model, (xt, at, torch.arange(B)), ctx)
        r.loss.backward()
        out["dist::tau"] = tau.numpy()
        out.update(sd_np(head, "dist::head::"))
        for k, v in r.logs.items():
            out[f"dist::log::{k}"] = np.float64(v)
        for n, p in list(model.named_parameters()) + [("distill_head." + n, p) for n, p in head.named_parameters()]:
            if p.grad is not None:
                out[f"dist::grad::{n}"] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"vqvae_{tag}.npz"), **out)


class _Recorder:
    """Records every torch.rand/randint/randn/randperm result produced inside the reference call."""

    def __init__(self):
        self.calls = []
        self._orig = {}

    def __enter__(self):
        for name in ("rand", "randint", "randn", "randperm"):
            self._orig[name] = getattr(torch, name)

            def wrap(*a, _n=name, **k):
                out = self._orig[_n](*a, **k)
                self.calls.append((_n, out.clone()))
                return out

            setattr(torch, name, wrap)
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(torch, name, fn)


def _draws_from_calls(calls, ccfg, T_full, N, centers, branches_a, branches_c, n_trip):
    """Resolve the recorded draws of one _make_augmented_view call (training.py:2373-2403) into the explicit
    augmentation parameters that the oracle / the HIP kernel take (oracle/contrastive.py AugDraws)."""
    from oracle.contrastive import choose_rotations
    it = iter(calls)

    def nxt(kind):
        k, v = next(it)
        assert k == kind, (k, kind)
        return v

    half = T_full // 2
    base = (T_full - half) // 2
    ap = nxt("rand") < ccfg.aug_p_shift
    mag, sgn = nxt("randint"), nxt("randint") * 2 - 1
    start = (base + mag * sgn * ap.long()).clamp(0, T_full - half)
    ap = (nxt("rand") < ccfg.aug_p_rot).float()
    perm = nxt("randperm").tolist()
    chosen = choose_rotations(perm, centers, N, ccfg.aug_n_rot)
    piv, nodes, thetas = [], [], []
    for k in chosen:
        side_a = bool(nxt("rand") < 0.5)
        th = (nxt("rand") * 2.0 - 1.0) * (float(ccfg.aug_max_rot) * np.pi / 180.0) * ap
        piv.append(centers[k]); nodes.append(branches_a[k] if side_a else branches_c[k]); thetas.append(th)
    ap = nxt("rand") < ccfg.aug_p_interp
    ln = nxt("randint")
    t0 = torch.minimum(nxt("randint"), (half - ln - 1).clamp_min(1))
    ln = ln * ap.long()
    ap = (nxt("rand") < ccfg.aug_p_noise).float().view(-1, 1)
    axis = nxt("randint")
    off = ccfg.aug_noise_sigma * nxt("randn") * ap
    ds = ccfg.aug_noise_sigma * nxt("randn") * ap
    noise = torch.stack([off * (axis == 0).float(), off * (axis == 1).float(), ds], -1)
    assert next(it, None) is None
    return dict(start=start.numpy().astype(np.int32), rot_pivot=np.array(piv, dtype=np.int32),
                rot_nodes=[np.array(n, dtype=np.int32) for n in nodes],
                theta=torch.stack(thetas).numpy().astype(np.float32) if thetas else np.zeros((0, len(start)), np.float32),
                interp_t0=t0.numpy().astype(np.int32), interp_len=ln.numpy().astype(np.int32),
                noise=noise.numpy().astype(np.float32))


def gen_contrastive(tag, ids, T_full, L, B, seed, encoder_type="recurrent", cases=None):
    """ContrastivePT (recurrent encoder on the half window) + step_contrastive_distill, RNG draws recorded."""
    from deepof_amd.graph import make_meta_info
    from oracle.contrastive import rotation_triplets
    nodes, edges = bodypart_graph(ids)
    adj = adjacency_from_graph(nodes, edges)
    N, E = len(nodes), len(edges)
    meta = make_meta_info(nodes, edges)
    dev = torch.device("cpu")
    ei_g, ei_l, _ = R.T._build_edge_from_metainfo(meta, dev, N, return_local=True)
    pre = R.T.build_rotation_precomp(ei_l, N, dev)
    trips, ba, bc = rotation_triplets(ei_l.tolist(), N)
    assert [tuple(t) for t in pre.triplets.tolist()] == trips
    for k in range(len(trips)):
        assert sorted(pre.branches_a[k].tolist()) == ba[k] and sorted(pre.branches_c[k].tolist()) == bc[k]
    centers = [t[1] for t in trips]
    x_full, _ = synth_batch(B, T_full, N, E, seed + 1)
    x_full = (x_full * 0.5).astype(np.float32)
    xt = torch.from_numpy(x_full)
    out = dict(x_full=x_full, adj=adj, edge_index=ei_g.numpy().astype(np.int32),
               edge_index_local=ei_l.numpy().astype(np.int32))
    # fixed-embedding loss table (losses.py:35-249), all similarity x loss combinations
    g = torch.Generator().manual_seed(seed)
    zz, zza = torch.randn(B, L, generator=g), torch.randn(B, L, generator=g)
    zz, zza = torch.nn.functional.normalize(zz, dim=1), torch.nn.functional.normalize(0.6 * zz + 0.8 * zza, dim=1)
    out.update(loss_z=zz.numpy(), loss_za=zza.numpy())
    for sim in ("cosine", "dot", "euclidean", "edit"):
        for lf in ("nce", "dcl", "fc", "hard_dcl"):
            a_ = zz.clone().requires_grad_(True); b_ = zza.clone().requires_grad_(True)
            l_, p_, n_ = R.L.select_contrastive_loss_pt(a_, b_, similarity=sim, loss_fn=lf, temperature=0.1, tau=0.1,
                                                        beta=0.1, elimination_topk=0.1)
            ga, gb = torch.autograd.grad(l_, [a_, b_])
            out[f"loss::{sim}::{lf}"] = np.array([float(l_), float(p_), float(n_)])
            out[f"loss_grad::{sim}::{lf}"] = np.stack([ga.numpy(), gb.numpy()])
    for ci, (sim, lf) in enumerate(cases or [("cosine", "nce"), ("euclidean", "hard_dcl"), ("dot", "dcl")]):
        torch.manual_seed(seed + ci)
        model = R.M.ContrastivePT((T_full, N, 3), (T_full, E, 1), adj, latent_dim=L, encoder_type=encoder_type,
                                  similarity_function=sim, loss_function=lf, temperature=0.1, beta=0.1, tau=0.1)
        ccfg = R.U.ContrastiveCfg(aug_n_rot=3, aug_p_rot=0.7, aug_p_noise=0.9, aug_p_interp=0.6)
        ctx = SimpleNamespace(apply_distill=False, edge_index=ei_g, edge_index_local=ei_l, contrastive_cfg=ccfg,
                              rot_precomp=pre)
        opt = R.L.build_optimizer_generic(model, None, base_lr=1e-3, weight_decay=1e-4)  # before any forward (Q11)
        if encoder_type != "recurrent":
            model.eval()
            R.U._materialize_encoder(model, (T_full // 2, N, 3), (T_full // 2, E, 1), torch.device("cpu"))
            sd0 = sd_np(model, f"c{ci}::sd::")
        model.train()
        model.zero_grad(set_to_none=True)
        a_dummy = torch.zeros(B, T_full, E, 1)
        with _Recorder() as rec:
            res = R.T.step_contrastive_distill(model, (xt, a_dummy, torch.arange(B)), ctx)
        res.loss.backward()
        calls = rec.calls
        dr = _draws_from_calls(calls, ccfg, T_full, N, centers, ba, bc, len(trips))
        pfx = f"c{ci}::"
        out[pfx + "sim"], out[pfx + "loss_fn"] = np.array(sim), np.array(lf)
        for k, v in dr.items():
            if k == "rot_nodes":
                mask = np.zeros((len(v), N), dtype=np.int32)
                for r, nn_ in enumerate(v):
                    mask[r, nn_] = 1
                out[pfx + "aug::rot_mask"] = mask
            else:
                out[pfx + "aug::" + k] = v
        # replay the same draws through the reference augmentation to store the views themselves
        seq = [v for _, v in calls]
        it = iter(seq)
        orig = {n: getattr(torch, n) for n in ("rand", "randint", "randn", "randperm")}
        try:
            for n in orig:
                setattr(torch, n, lambda *a, **k: next(it).clone())
            xa, aa = R.T._make_augmented_view(
                xt, R.U.recompute_edges(xt, ei_g), ei_g, pre, min_shift=ccfg.aug_min_shift, max_shift=ccfg.aug_max_shift,
                p_shift=ccfg.aug_p_shift, noise_sigma=ccfg.aug_noise_sigma, p_noise=ccfg.aug_p_noise,
                max_interp=ccfg.aug_max_interp, min_interp=ccfg.aug_min_interp, p_interp=ccfg.aug_p_interp,
                max_rot=ccfg.aug_max_rot, n_rot=ccfg.aug_n_rot, p_rot=ccfg.aug_p_rot)
        finally:
            for n, fn in orig.items():
                setattr(torch, n, fn)
        out[pfx + "x_aug"], out[pfx + "a_aug"] = xa.numpy(), aa.numpy()
        half = T_full // 2
        st = (torch.ones(B) * half // 2).int()
        xc = R.U.slice_time_per_sample(xt, st, half)
        out[pfx + "x"], out[pfx + "a"] = xc.numpy(), R.U.recompute_edges(xc, ei_g).numpy()
        if encoder_type == "recurrent":
            with torch.no_grad():
                out[pfx + "z"] = model(xc, R.U.recompute_edges(xc, ei_g)).numpy()
                out[pfx + "z_aug"] = model(xa, aa).numpy()
            out.update(sd_np(model, pfx + "sd::"))
        else:
            # BatchNorm: the train-mode embeddings depend on (and update) the running buffers -> replay on a copy
            import copy
            twin = copy.deepcopy(model)
            twin.load_state_dict({k[len(pfx) + 4:]: torch.from_numpy(v) for k, v in sd0.items()})
            twin.train()
            with torch.no_grad():
                out[pfx + "z"] = twin(xc, R.U.recompute_edges(xc, ei_g)).numpy()
                out[pfx + "z_aug"] = twin(xa, aa).numpy()
            out.update({k: v for k, v in sd_np(twin, pfx + "sd_after::").items()  # running statistics after
                        if "running_" in k or "num_batches" in k})                    # the two forward passes
            twin.load_state_dict({k[len(pfx) + 4:]: torch.from_numpy(v) for k, v in sd0.items()})
            twin.eval()
            with torch.no_grad():
                out[pfx + "z_eval"] = twin(xc, R.U.recompute_edges(xc, ei_g)).numpy()
            out.update(sd0)
        for k, v in res.logs.items():
            out[pfx + f"log::{k}"] = np.float64(v)
        for n, p_ in model.named_parameters():
            if p_.grad is not None:
                out[pfx + f"grad::{n}"] = p_.grad.numpy().copy()
        if encoder_type == "recurrent" and tag in ("rec28", "c5l8") and ci == 0:
            # same step with the distillation head on (replaying the same draws)
            import deepof.clustering.teacher_model as TM
            torch.manual_seed(seed + 99)
            Kd = 5
            head = TM.DiscriminativeHead(L, Kd)
            tau = torch.softmax(torch.randn(B, Kd) * 1.5, dim=-1)
            ctx2 = SimpleNamespace(apply_distill=True, edge_index=ei_g, edge_index_local=ei_l, contrastive_cfg=ccfg,
                                   rot_precomp=pre, distill_head=head, tau_star=tau, distill_sharpen_T=0.5,
                                   distill_conf_weight=True, distill_conf_thresh=0.2,
                                   lambda_scheduler=SimpleNamespace(get_weight=lambda: 0.9))
            model.zero_grad(set_to_none=True)
            it2 = iter([v for _, v in calls])
            orig2 = {n: getattr(torch, n) for n in ("rand", "randint", "randn", "randperm")}
            try:
                for n in orig2:
                    setattr(torch, n, lambda *a, **k: next(it2).clone())
                r2 = R.T.step_contrastive_distill(model, (xt, a_dummy, torch.arange(B)), ctx2)
            finally:
                for n, fn in orig2.items():
                    setattr(torch, n, fn)
            r2.loss.backward()
            out["dist::tau"] = tau.numpy()
            out.update(sd_np(head, "dist::head::"))
            for k, v in r2.logs.items():
                out[f"dist::log::{k}"] = np.float64(v)
            for n, p_ in list(model.named_parameters()) + [("distill_head." + n, p_) for n, p_ in head.named_parameters()]:
                if p_.grad is not None:
                    out[f"dist::grad::{n}"] = p_.grad.numpy().copy()
        if encoder_type != "recurrent":
            # finish optimiser step 1 on the recorded gradients, then one more full step (draws recorded again)
            torch.nn.utils.clip_grad_value_(model.parameters(), 0.75)
            opt.step()
            opt.zero_grad(set_to_none=True)
            with _Recorder() as rec2:
                res2 = R.T.step_contrastive_distill(model, (xt, a_dummy, torch.arange(B)), ctx)
            res2.loss.backward()
            torch.nn.utils.clip_grad_value_(model.parameters(), 0.75)
            opt.step()
            dr2 = _draws_from_calls(rec2.calls, ccfg, T_full, N, centers, ba, bc, len(trips))
            for k, v in dr2.items():
                if k == "rot_nodes":
                    mask = np.zeros((len(v), N), dtype=np.int32)
                    for r, nn_ in enumerate(v):
                        mask[r, nn_] = 1
                    out[pfx + "aug2::rot_mask"] = mask
                else:
                    out[pfx + "aug2::" + k] = v
            for k, v in res2.logs.items():
                out[pfx + f"log2::{k}"] = np.float64(v)
            out.update(sd_np(model, pfx + "sd_step2::"))
    np.savez_compressed(os.path.join(HERE, f"contrastive_{tag}.npz"), **out)


def gen_bootstrap():
    """Batch starts of the reference loader's block bootstrap (dataset.py:505-559, 585-618) for a few cases."""
    out = {}
    cases = [(3, [700, 130, 420], 64, 250, 1, 0), (2, [300, 90], 100, 250, 2, 1), (4, [64, 200, 63, 500], 32, 100, 3, 2)]
    for ci, (nv, lens, bs, L, world, seed) in enumerate(cases):
        vid = np.concatenate([np.full(n, i, dtype=np.int32) for i, n in enumerate(lens)])
        n = len(vid)
        fake = SimpleNamespace(bootstrap_block_len=L)
        fake._compute_video_ranges = lambda v: R.D._H5BatchIterableDataset._compute_video_ranges(fake, v)
        for epoch in (1, 2):
            starts = np.arange(0, n, bs, dtype=np.int64)
            rng = np.random.default_rng((seed + epoch) % (2 ** 32))
            rng.shuffle(starts)
            if world > 1:
                starts = starts[: (len(starts) // world) * world]
            bs_starts = R.D._H5BatchIterableDataset._block_bootstrap_batch_starts(fake, rng, vid, n, bs, len(starts))
            out[f"c{ci}::e{epoch}"] = bs_starts
        out[f"c{ci}::cfg"] = np.array([bs, L, world, seed], dtype=np.int64)
        out[f"c{ci}::vid"] = vid
    np.savez_compressed(os.path.join(HERE, "bootstrap.npz"), **out)


def gen_turtle():
    """TURTLE teacher (teacher_model.py:43-350): a short fit over a fixed batch list from recorded initial weights,
    final weights, tau* of a prediction pass, and initialize_gmm_from_teacher on a latent view."""
    import deepof.clustering.teacher_model as TM
    torch.manual_seed(7)
    dims, K, B, nb = [8, 32, 32], 6, 96, 3
    g = torch.Generator().manual_seed(11)
    centers = [torch.randn(K, d, generator=g) * 1.5 for d in dims]
    batches = []
    for _ in range(nb):
        lab = torch.randint(0, K, (B,), generator=g)
        batches.append([c[lab] + 0.7 * torch.randn(B, c.shape[1], generator=g) for c in centers])
    teacher = TM.TurtleTeacher(feature_dims=dims, n_components=K, gamma=8.0, alpha_sample_entropy=2.0, inner_lr=0.1,
                               inner_steps=12, head_wd=1e-4, head_temp=0.35, task_temp=0.35, normalize_feats=True,
                               lr_theta=1e-3, device="cpu")
    out = {"dims": np.array(dims), "cfg": np.array([K, B, nb, 12, 7])}
    for k, v in teacher.state_dict().items():
        out["init::" + k] = v.numpy().copy()
    for i, bt in enumerate(batches):
        for v, f in enumerate(bt):
            out[f"batch{i}::{v}"] = f.numpy()
    teacher.fit(batches, outer_steps=7, rho=0.04, verbose=False)
    for k, v in teacher.state_dict().items():
        out["final::" + k] = v.numpy().copy()
    tau = teacher.predict([batches[0], batches[1]])
    out["tau_star"] = tau.numpy()
    z = torch.cat([batches[0][0], batches[1][0]])
    fake = SimpleNamespace(latent_space=SimpleNamespace(gmm_means=torch.nn.Parameter(torch.zeros(K, dims[0])),
                                                        gmm_log_vars=torch.nn.Parameter(torch.zeros(K, dims[0])),
                                                        prior=torch.zeros(K)),
                           eval=lambda: None, parameters=lambda: iter([torch.zeros(1)]))
    TM.initialize_gmm_from_teacher(fake, z, tau, min_var=0.01)
    out["gmm_means"], out["gmm_log_vars"] = fake.latent_space.gmm_means.detach().numpy(), fake.latent_space.gmm_log_vars.detach().numpy()
    out["gmm_prior"] = fake.latent_space.prior.numpy()
    np.savez_compressed(os.path.join(HERE, "turtle.npz"), **out)


def gen_vade_tcn(tag, ids, T, L, K, B, seed):
    """VaDEPT(encoder_type="TCN") -- TCN encoder + GMM latent + TCN decoder (R12): eval forward, and for two phases
    (each restarted from the same initial state, BatchNorm buffers included) the train-mode outputs, loss terms,
    BatchNorm buffers after the pass, and gradients (all of them for "pre"; the latent space only for "mainT")."""
    nodes, edges = bodypart_graph(ids)
    adj = adjacency_from_graph(nodes, edges)
    N, E = len(nodes), len(edges)
    torch.manual_seed(seed)
    model = R.M.VaDEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type="TCN", kmeans_loss=1.0)
    model.eval()
    R.U._materialize_encoder(model, (T, N, 3), (T, E, 1), torch.device("cpu"))
    with torch.no_grad():
        model.latent_space.gmm_means.mul_(3.0)
        for n, b in model.named_buffers():      # non-trivial running statistics for the eval-mode check
            if n.endswith("running_mean"):
                b.normal_(0.0, 0.1)
            elif n.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    x, a = synth_batch(B, T, N, E, seed + 1)
    xt, at = torch.from_numpy(x), torch.from_numpy(a)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    out = dict(sd_np(model))
    out.update(x=x, a=a, adj=adj)
    with torch.no_grad():
        dist, z, q, km = model(xt, at)
        enc = model.encoder(xt, at)
    out.update(eval_z=z.numpy(), eval_q=q.numpy(), eval_loc=dist.base_dist.base_dist.loc.numpy(), eval_enc=enc.numpy())
    eps = torch.randn(B, L, generator=torch.Generator().manual_seed(seed + 2))
    eps_mc = torch.randn(32, B, L, generator=torch.Generator().manual_seed(seed + 3))
    tau = torch.softmax(torch.randn(B, K, generator=torch.Generator().manual_seed(seed + 4)) * 2.0, dim=-1)
    out.update(eps=eps.numpy(), eps_mc=eps_mc.numpy(), tau=tau.numpy())
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def patched_randn(*size, **kw):
        if len(size) == 3 and tuple(size) == (32, B, L):
            return eps_mc.clone()
        return real_randn(*size, **kw)

    def patched_randn_like(t, **kw):
        if tuple(t.shape) == (B, L):
            return eps.clone()
        return real_randn_like(t, **kw)

    for phase, klw, with_teacher in [("pre", 0.13, False), ("mainT", 0.7, True)]:
        model.load_state_dict(sd0)
        common, vade, teacher = _cfgs(K, L)
        crit = R.L.VadeLoss(common_cfg=common, vade_cfg=vade, teacher_cfg=teacher)
        crit.set_mode("pretrain" if phase == "pre" else "main")
        crit.kl_scheduler = SimpleNamespace(get_weight=lambda k=klw: k, max_weight=1.0, current_iteration=0)
        if with_teacher:
            crit.set_teacher(tau_star=tau, lambda_distill=1.7)
        model.train()
        model.zero_grad(set_to_none=True)
        torch.randn, torch.randn_like = patched_randn, patched_randn_like
        try:
            outputs = model(xt, at, return_gmm_params=True)
            ld = crit(outputs, xt, batch_indices=torch.arange(B) if with_teacher else None)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        ld["total_loss"].backward()
        for k, v in ld.items():
            out[f"{phase}::loss::{k}"] = np.float64(float(v))
        if phase == "pre":
            out.update({k: v for k, v in sd_np(model, "pre::sd_after::").items() if "running_" in k})
        # The same step in float64 (the reference's .float() casts redirected to double).  With BatchNorm over a
        # handful of windows this problem amplifies fp32 rounding to ~1e-4 relative in the gradients, so parity is
        # stated against this fp64 evaluation, in units of the reference's OWN fp32 deviation from it ("noise").
        import copy
        m64 = copy.deepcopy(model)
        m64.load_state_dict(sd0)
        m64 = m64.double()
        m64.train()
        m64.zero_grad(set_to_none=True)
        crit64 = R.L.VadeLoss(common_cfg=common, vade_cfg=vade, teacher_cfg=teacher)
        crit64.set_mode("pretrain" if phase == "pre" else "main")
        crit64.kl_scheduler = crit.kl_scheduler
        if with_teacher:
            crit64.set_teacher(tau_star=tau.double(), lambda_distill=1.7)
        orig_float = torch.Tensor.float
        torch.Tensor.float = lambda self: self.double()
        torch.randn = lambda *size, **kw: eps_mc.double() if tuple(size) == (32, B, L) else real_randn(*size, **kw)
        torch.randn_like = lambda t, **kw: eps.double() if tuple(t.shape) == (B, L) else real_randn_like(t, **kw)
        try:
            o64 = m64(xt.double(), at.double(), return_gmm_params=True)
            l64 = crit64(o64, xt.double(), batch_indices=torch.arange(B) if with_teacher else None)
            l64["total_loss"].backward()
        finally:
            torch.Tensor.float = orig_float
            torch.randn, torch.randn_like = real_randn, real_randn_like
        for key, t32, t64 in (("z", outputs[1], o64[1]), ("q", outputs[2], o64[2]),
                              ("loc", outputs[0].base_dist.base_dist.loc, o64[0].base_dist.base_dist.loc)):
            out[f"{phase}::{key}"] = t64.detach().numpy().astype(np.float32)
            out[f"{phase}::noise::{key}"] = np.float64((t32.detach().double() - t64.detach()).abs().max())
        p64 = dict(m64.named_parameters())
        for n, p_ in model.named_parameters():
            if p_.grad is not None and (phase == "pre" or n.startswith("latent_space") or n.startswith("decoder.fc")):
                out[f"{phase}::grad::{n}"] = p64[n].grad.numpy().astype(np.float32)
                out[f"{phase}::gnoise::{n}"] = np.float64((p_.grad.double() - p64[n].grad).abs().max())
    np.savez_compressed(os.path.join(HERE, f"vade_{tag}.npz"), **out)


if __name__ == "__main__":
    gen_scramble()
    gen_graph_ops()
    gen_recurrent_block()
    gen_vade("rec14", [""], 25, 8, 10, 16, 21)
    gen_vade("rec28", ["B", "W"], 12, 6, 5, 6, 31)
    gen_train_trace()
    gen_schedules_kmeans()
    gen_vqvae("rec14", [""], 25, 8, 64, 16, 41)
    gen_vqvae("rec28", ["B", "W"], 12, 6, 20, 6, 51, kmeans=0.5)
    gen_contrastive("rec14", [""], 24, 8, 16, 61)
    gen_contrastive("rec28", ["B", "W"], 25, 6, 7, 71)
    gen_contrastive("tcn14", [""], 24, 8, 6, 81, encoder_type="TCN", cases=[("cosine", "nce")])
    gen_vade_tcn("tcn14", [""], 25, 8, 10, 6, 91)
    gen_bootstrap()
    gen_turtle()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
