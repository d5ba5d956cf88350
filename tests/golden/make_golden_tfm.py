"""Transformer-family golden fixtures, produced by running the REFERENCE in place (build container only;
see make_golden.py): VaDEPT / VQVAEPT / ContrastivePT with encoder_type="transformer"
(/root/reference/deepof/clustering/models_new.py:832-1327).

Dropout is the family's only random element.  While the reference runs in train mode, torch.nn.functional.dropout
and torch.nn.functional.scaled_dot_product_attention are replaced by versions that draw the keep-masks from a
dedicated generator and RECORD them (``drop::NNN`` entries, in the order the forward draws them); the replaced SDPA is
the mathematical definition softmax(q k^T / sqrt(d) + mask) -> dropout -> @ v, checked here against the reference's
own fused SDPA in eval mode (``sdpa_check``).  The masks are what the oracle / the HIP path are run on.

* vade_tfm14.npz         eval forward (also on a window with zeroed frames -> padded keys), two train steps
                         (pretrain objective / main objective with teacher) with every logged term, all gradients and
                         the BatchNorm buffers after the step; the reference's fp32-vs-fp64 deviation per tensor.
* vqvae_tfm14.npz        eval forward, one step_vqvae_distill step (two decoder passes, each with its own masks).
* contrastive_tfm14.npz  train-mode embeddings of two views, NCE/cosine loss (losses.py:131-142) and its gradients.
"""
import copy
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as MG  # noqa: E402
from deepof_amd.graph import adjacency_from_graph, bodypart_graph  # noqa: E402

R = MG.R
torch.set_num_threads(1)
TF = torch.nn.functional


class DropoutRecorder:
    """Context manager: record (default) or replay the dropout keep-masks of a reference forward."""

    def __init__(self, seed=0, replay=None):
        self.gen = torch.Generator().manual_seed(seed)
        self.masks = [] if replay is None else list(replay)
        self.replay = replay is not None
        self.pos = 0

    def _mask(self, shape, p):
        if self.replay:
            m = self.masks[self.pos]
            self.pos += 1
            assert tuple(m.shape) == tuple(shape)
            return m
        m = torch.bernoulli(torch.full(tuple(shape), 1.0 - p), generator=self.gen).to(torch.uint8)
        self.masks.append(m)
        return m

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        return x * (self._mask(x.shape, p).to(x.dtype) / (1.0 - p))

    def sdpa(self, q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None):
        d = q.shape[-1]
        s = q @ k.transpose(-1, -2) * (scale if scale is not None else 1.0 / d ** 0.5)
        if is_causal:
            T = q.shape[-2]
            s = s + torch.full((T, T), float("-inf"), dtype=q.dtype).triu(1)
        if attn_mask is not None:
            s = s + attn_mask.to(q.dtype)
        w = torch.softmax(s, dim=-1)
        if dropout_p > 0.0:
            w = w * (self._mask(w.shape, dropout_p).to(w.dtype) / (1.0 - dropout_p))
        return w @ v

    def __enter__(self):
        self._orig = (TF.dropout, TF.scaled_dot_product_attention)
        TF.dropout, TF.scaled_dot_product_attention = self.dropout, self.sdpa
        return self

    def __exit__(self, *exc):
        TF.dropout, TF.scaled_dot_product_attention = self._orig
        return False


class ReluKinkFlipper:
    """Context manager: every ReLU input element with |x| < delta takes the OTHER branch's derivative (values move by
    less than delta).  The sign of such an element is decided by fp32 rounding, so two correct fp32 implementations
    may disagree on it; the gradient change under this flip (``gkink::<name>`` in the fixtures) is the reference's own
    sensitivity to those undecidable signs and enters the parity bar of the tensors it reaches."""

    def __init__(self, delta=2e-6):
        self.delta = delta
        self.flipped = 0

    def relu(self, x, inplace=False):
        near = (x.detach().abs() < self.delta) & (x.detach() != 0)  # exact zeros: a ReLU applied to a ReLU output
        self.flipped += int(near.sum())
        return torch.where(near, torch.where(x > 0, x * 0.0, x), x.clamp_min(0.0))

    def __enter__(self):
        self._orig = (TF.relu, torch.relu)
        TF.relu = self.relu
        torch.relu = lambda x: self.relu(x)
        return self

    def __exit__(self, *exc):
        TF.relu, torch.relu = self._orig
        return False


def store_kink(out, prefix, model, rerun):
    """gkink::<name> = max |gradient with the near-zero ReLU signs flipped - gradient| per parameter."""
    base = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    with ReluKinkFlipper() as fl:
        rerun()
    for n, p in model.named_parameters():
        if p.grad is not None and n in base:
            out[f"{prefix}gkink::{n}"] = np.float64((p.grad - base[n]).abs().max())
    out[f"{prefix}kink_count"] = np.int64(fl.flipped)
    print(f"{prefix or 'step'}: {fl.flipped} ReLU inputs within 2e-6 of zero")


def trained_like_state(model, seed=5):
    """LayerNorm / BatchNorm scales and shifts and all biases away from their initial 1 / 0 / 0 (any trained state)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n or "head.2" in n or "head.5" in n:
                if n.endswith("weight"):
                    p.copy_(0.7 + 0.6 * torch.rand(p.shape, generator=g))
                else:
                    p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for n, b in model.named_buffers():
            if n.endswith("running_mean"):
                b.normal_(0.0, 0.1, generator=g)
            elif n.endswith("running_var"):
                b.uniform_(0.5, 1.5, generator=g)


def _graph():
    nodes, edges = bodypart_graph([""])
    return adjacency_from_graph(nodes, edges), len(nodes), len(edges)


def _store_masks(out, prefix, masks):
    for i, m in enumerate(masks):
        out[f"{prefix}drop::{i:03d}"] = m.numpy()


def _as64(fn):
    """Run fn with Tensor.float() mapped to .double() (the reference casts to fp32 in places)."""
    orig = torch.Tensor.float
    torch.Tensor.float = lambda self: self.double()
    try:
        return fn()
    finally:
        torch.Tensor.float = orig


def gen_vade_tfm(seed=301, B=16, T=25, L=8, K=10):
    adj, N, E = _graph()
    torch.manual_seed(seed)
    model = R.M.VaDEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type="transformer", kmeans_loss=1.0)
    model.eval()
    R.U._materialize_encoder(model, (T, N, 3), (T, E, 1), torch.device("cpu"))
    with torch.no_grad():
        model.latent_space.gmm_means.mul_(3.0)
    trained_like_state(model)
    x, a = MG.synth_batch(B, T, N, E, seed + 1)
    xt, at = torch.from_numpy(x), torch.from_numpy(a)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    out = dict(MG.sd_np(model))
    out.update(x=x, a=a, adj=adj)
    with torch.no_grad():
        dist, z, q, _km = model(xt, at)
        enc = model.encoder(xt, at)
        with DropoutRecorder() as rec:  # the replaced SDPA against the reference's fused one (eval: no masks drawn)
            dist2, z2, _q2, _ = model(xt, at)
        assert not rec.masks
    loc = dist.base_dist.base_dist.loc
    out["sdpa_check"] = np.array([float((z - z2).abs().max()), float((loc - dist2.base_dist.base_dist.loc).abs().max())])
    print("vade_tfm: fused vs restated SDPA (eval): max |dz|, |dloc| =", out["sdpa_check"])
    assert out["sdpa_check"].max() < 2e-5
    out.update(eval_z=z.numpy(), eval_q=q.numpy(), eval_loc=loc.numpy(), eval_enc=enc.numpy())
    # a window batch with zeroed frames: padded keys in the encoder streams, masked frames in the decoder
    xm, am = x.copy(), a.copy()
    xm[1, 3] = 0.0; am[1, 3] = 0.0
    xm[2, 20:] = 0.0; am[2, 20:] = 0.0
    xm[5, 0] = 0.0
    with torch.no_grad():
        distm, zm, qm, _ = model(torch.from_numpy(xm), torch.from_numpy(am))
    out.update(xm=xm, am=am, evalm_z=zm.numpy(), evalm_q=qm.numpy(), evalm_loc=distm.base_dist.base_dist.loc.numpy())
    eps = torch.randn(B, L, generator=torch.Generator().manual_seed(seed + 2))
    eps_mc = torch.randn(32, B, L, generator=torch.Generator().manual_seed(seed + 3))
    tau = torch.softmax(torch.randn(B, K, generator=torch.Generator().manual_seed(seed + 4)) * 2.0, dim=-1)
    out.update(eps=eps.numpy(), eps_mc=eps_mc.numpy(), tau=tau.numpy())
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def run(m, dtype, phase, klw, with_teacher, rec):
        common, vade, teacher = MG._cfgs(K, L)
        crit = R.L.VadeLoss(common_cfg=common, vade_cfg=vade, teacher_cfg=teacher)
        crit.set_mode("pretrain" if phase == "pre" else "main")
        crit.kl_scheduler = SimpleNamespace(get_weight=lambda k=klw: k, max_weight=1.0, current_iteration=0)
        if with_teacher:
            crit.set_teacher(tau_star=tau.to(dtype), lambda_distill=1.7)
        m.train()
        m.zero_grad(set_to_none=True)
        torch.randn = lambda *s, **kw: eps_mc.to(dtype) if tuple(s) == (32, B, L) else real_randn(*s, **kw)
        torch.randn_like = lambda t, **kw: eps.to(dtype) if tuple(t.shape) == (B, L) else real_randn_like(t, **kw)
        try:
            with rec:
                o = m(xt.to(dtype), at.to(dtype), return_gmm_params=True)
                ld = crit(o, xt.to(dtype), batch_indices=torch.arange(B) if with_teacher else None)
            ld["total_loss"].backward()
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        return o, ld

    for pi, (phase, klw, with_teacher) in enumerate([("pre", 0.13, False), ("mainT", 0.7, True)]):
        model.load_state_dict(sd0)
        rec = DropoutRecorder(seed + 10 + pi)
        o32, l32 = run(model, torch.float32, phase, klw, with_teacher, rec)
        _store_masks(out, f"{phase}::", rec.masks)
        for k, v in l32.items():
            out[f"{phase}::loss::{k}"] = np.float64(float(v))
        if phase == "pre":
            out.update({k: v for k, v in MG.sd_np(model, "pre::sd_after::").items() if "running_" in k or "num_batches" in k})
        m64 = copy.deepcopy(model)
        m64.load_state_dict(sd0)
        m64 = m64.double()
        o64, _l64 = _as64(lambda: run(m64, torch.float64, phase, klw, with_teacher, DropoutRecorder(replay=rec.masks)))
        for key, t32, t64 in (("z", o32[1], o64[1]), ("q", o32[2], o64[2]),
                              ("loc", o32[0].base_dist.base_dist.loc, o64[0].base_dist.base_dist.loc)):
            out[f"{phase}::{key}"] = t32.detach().numpy()
            out[f"{phase}::noise::{key}"] = np.float64((t32.detach().double() - t64.detach()).abs().max())
        p64 = dict(m64.named_parameters())
        for n, p_ in model.named_parameters():
            if p_.grad is not None:
                out[f"{phase}::grad::{n}"] = p_.grad.numpy().copy()
                out[f"{phase}::gnoise::{n}"] = np.float64((p_.grad.double() - p64[n].grad).abs().max())

        def rerun():
            model.load_state_dict(sd0)
            run(model, torch.float32, phase, klw, with_teacher, DropoutRecorder(replay=rec.masks))

        store_kink(out, f"{phase}::", model, rerun)
    np.savez_compressed(os.path.join(HERE, "vade_tfm14.npz"), **out)


def gen_vqvae_tfm(seed=331, B=16, T=25, L=8, K=24):
    adj, N, E = _graph()
    torch.manual_seed(seed)
    model = R.M.VQVAEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type="transformer", kmeans_loss=0.0)
    model.eval()
    R.U._materialize_encoder(model, (T, N, 3), (T, E, 1), torch.device("cpu"))
    with torch.no_grad():
        model.vq_layer.codebook.copy_(torch.randn(L, K) * 0.6)
    trained_like_state(model)
    x, a = MG.synth_batch(B, T, N, E, seed + 1)
    xt, at = torch.from_numpy(x), torch.from_numpy(a)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    out = dict(MG.sd_np(model))
    out.update(x=x, a=a, adj=adj)
    with torch.no_grad():
        enc_rec, rec_, quant, soft, ze, _ = model(xt, at, return_losses=True, return_all_outputs=True)
    out.update(eval_quantized=quant.numpy(), eval_soft_counts=soft.numpy(), eval_ze=ze.numpy(),
               eval_loc_q=enc_rec.base_dist.base_dist.loc.numpy(), eval_loc_e=rec_.base_dist.base_dist.loc.numpy(),
               eval_idx=model.vq_layer.get_code_indices(ze).numpy())
    model.train()
    model.zero_grad(set_to_none=True)
    rec = DropoutRecorder(seed + 10)
    with rec:
        res = R.T.step_vqvae_distill(model, (xt, at, torch.arange(B)), SimpleNamespace(apply_distill=False))
    res.loss.backward()
    _store_masks(out, "", rec.masks)
    for k, v in res.logs.items():
        out[f"log::{k}"] = np.float64(v)
    for n, p in model.named_parameters():
        if p.grad is not None:
            out[f"grad::{n}"] = p.grad.numpy().copy()
    out.update({k: v for k, v in MG.sd_np(model, "sd_after::").items() if "running_" in k or "num_batches" in k})
    m64 = copy.deepcopy(model)
    m64.load_state_dict(sd0)
    m64 = m64.double()
    m64.train()
    m64.zero_grad(set_to_none=True)

    def step64():
        with DropoutRecorder(replay=rec.masks):
            r = R.T.step_vqvae_distill(m64, (xt.double(), at.double(), torch.arange(B)), SimpleNamespace(apply_distill=False))
        r.loss.backward()

    _as64(step64)
    p64 = dict(m64.named_parameters())
    for n, p in model.named_parameters():
        if p.grad is not None and p64[n].grad is not None:
            out[f"gnoise::{n}"] = np.float64((p.grad.double() - p64[n].grad).abs().max())

    def rerun():
        model.load_state_dict(sd0)
        model.train()
        model.zero_grad(set_to_none=True)
        with DropoutRecorder(replay=rec.masks):
            r = R.T.step_vqvae_distill(model, (xt, at, torch.arange(B)), SimpleNamespace(apply_distill=False))
        r.loss.backward()

    store_kink(out, "", model, rerun)
    np.savez_compressed(os.path.join(HERE, "vqvae_tfm14.npz"), **out)


def gen_contrastive_tfm(seed=361, B=12, T_full=50, L=8):
    adj, N, E = _graph()
    T = T_full // 2
    torch.manual_seed(seed)
    model = R.M.ContrastivePT((T_full, N, 3), (T_full, E, 1), adj, latent_dim=L, encoder_type="transformer",
                              similarity_function="cosine", loss_function="nce", temperature=0.1, beta=0.1, tau=0.1)
    model.eval()
    R.U._materialize_encoder(model, (T, N, 3), (T, E, 1), torch.device("cpu"))
    trained_like_state(model)
    x, a = MG.synth_batch(B, T, N, E, seed + 1)
    x2, a2 = MG.synth_batch(B, T, N, E, seed + 2)
    x2 = (0.7 * x + 0.3 * x2).astype(np.float32)
    a2 = (0.7 * a + 0.3 * a2).astype(np.float32)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    out = dict(MG.sd_np(model))
    out.update(x=x, a=a, x_aug=x2, a_aug=a2, adj=adj)
    with torch.no_grad():
        out["eval_z"] = model(torch.from_numpy(x), torch.from_numpy(a)).numpy()

    def step(m, dtype, rec):
        m.train()
        m.zero_grad(set_to_none=True)
        with rec:
            z = m(torch.from_numpy(x).to(dtype), torch.from_numpy(a).to(dtype))
            za = m(torch.from_numpy(x2).to(dtype), torch.from_numpy(a2).to(dtype))
        # training.py:525-546: normalise, then the selected loss
        zn, zan = TF.normalize(z, dim=1), TF.normalize(za, dim=1)
        loss, pos, neg = R.L.select_contrastive_loss_pt(zn, zan, similarity="cosine", loss_fn="nce", temperature=0.1,
                                                        tau=0.1, beta=0.1, elimination_topk=0.1)
        loss.backward()
        return z, za, loss, pos, neg

    rec = DropoutRecorder(seed + 10)
    z, za, loss, pos, neg = step(model, torch.float32, rec)
    _store_masks(out, "", rec.masks)
    out.update(z=z.detach().numpy(), z_aug=za.detach().numpy(), loss=np.array([float(loss), float(pos), float(neg)]))
    for n, p in model.named_parameters():
        if p.grad is not None:
            out[f"grad::{n}"] = p.grad.numpy().copy()
    out.update({k: v for k, v in MG.sd_np(model, "sd_after::").items() if "running_" in k or "num_batches" in k})
    m64 = copy.deepcopy(model)
    m64.load_state_dict(sd0)
    m64 = m64.double()
    _as64(lambda: step(m64, torch.float64, DropoutRecorder(replay=rec.masks)))
    p64 = dict(m64.named_parameters())
    for n, p in model.named_parameters():
        if p.grad is not None:
            out[f"gnoise::{n}"] = np.float64((p.grad.double() - p64[n].grad).abs().max())

    def rerun():
        model.load_state_dict(sd0)
        step(model, torch.float32, DropoutRecorder(replay=rec.masks))

    store_kink(out, "", model, rerun)
    np.savez_compressed(os.path.join(HERE, "contrastive_tfm14.npz"), **out)


if __name__ == "__main__":
    gen_vade_tfm()
    gen_vqvae_tfm()
    gen_contrastive_tfm()
    for f in ("vade_tfm14.npz", "vqvae_tfm14.npz", "contrastive_tfm14.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
