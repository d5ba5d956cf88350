"""Round-6 reference fixtures (run in the build container, where /root/reference is importable):

    python tests/golden/make_golden_r06.py l4 l6       # recurrent family at latent 4 (the API's default latent_dim) and 6 (the tutorial's)
    python tests/golden/make_golden_r06.py lodd        # latent 7, 9, 14 (recurrent family)
    python tests/golden/make_golden_r06.py tcnkinks    # ReLU-kink attribution for the two small contrastive TCN fixtures
    python tests/golden/make_golden_r06.py kinkloc     # where each ReLU-kink candidate of the contrastive TCN fixtures sits

* vade_rec14l4.npz / vqvae_rec14l4.npz / contrastive_rec14l4.npz
      the recurrent encoder / decoder at latent_dim = 4 -- /root/reference/deepof/data.py:3260 (`latent_dim: int = 4`) and the
      size the reference's own regression tests run (tests/regression/test_model_regression.py:130-193); same generators,
      shapes and seeds-by-convention as the latent 5 .. 32 fixtures of make_golden_r03.py.
* tcn_kinks.npz  +=  contrastive_tcn14l16::c0::*  and  contrastive_tcn14::c0::*
      make_golden_r03.kink_attribution for the recorded step of those two fixtures (B = 6, window 24 -> 12): for every
      BatchNorm+ReLU pre-activation of the REFERENCE run within 3e-5 of zero, the reference's own gradient change when that
      element takes the other branch.  Round 6 moved the TCN convolutions to bf16-piece products (fp32-exact operands, another
      summation order): one such element of the latent-16 fixture changes branch on the MI355X, and the test now NAMES it
      (KinkAttribution) instead of holding a flipped branch to the plain bar.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as MG  # noqa: E402
import make_golden_r03 as MG3  # noqa: E402
from deepof_amd.graph import adjacency_from_graph, bodypart_graph, make_meta_info  # noqa: E402

R = MG.R
torch.set_num_threads(1)
KINK_DELTA = 3e-5


def _replay_setup(tag, seed, B, t_full, L):
    """The recorded step of contrastive_<tag>.npz again (the operation order of make_golden.gen_contrastive, case 0): the model in
    its recorded initial state, the recorded RNG draws and a rerun() that replays them.  Asserts that the regenerated
    gradients equal the committed fixture's bit for bit."""
    d = dict(np.load(os.path.join(HERE, f"contrastive_{tag}.npz")))
    nodes, edges = bodypart_graph([""])
    adj = adjacency_from_graph(nodes, edges)
    N, E = len(nodes), len(edges)
    meta = make_meta_info(nodes, edges)
    dev = torch.device("cpu")
    ei_g, ei_l, _ = R.T._build_edge_from_metainfo(meta, dev, N, return_local=True)
    pre = R.T.build_rotation_precomp(ei_l, N, dev)
    xt = torch.from_numpy(d["x_full"])
    torch.manual_seed(seed + 0)
    model = R.M.ContrastivePT((t_full, N, 3), (t_full, E, 1), adj, latent_dim=L, encoder_type="TCN",
                              similarity_function="cosine", loss_function="nce", temperature=0.1, beta=0.1, tau=0.1)
    ccfg = R.U.ContrastiveCfg(aug_n_rot=3, aug_p_rot=0.7, aug_p_noise=0.9, aug_p_interp=0.6)
    ctx = SimpleNamespace(apply_distill=False, edge_index=ei_g, edge_index_local=ei_l, contrastive_cfg=ccfg, rot_precomp=pre)
    R.L.build_optimizer_generic(model, None, base_lr=1e-3, weight_decay=1e-4)
    model.eval()
    R.U._materialize_encoder(model, (t_full // 2, N, 3), (t_full // 2, E, 1), dev)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    for k, v in sd0.items():
        assert np.array_equal(v.numpy(), d["c0::sd::" + k]), ("initial state differs from the committed golden", k)
    a_dummy = torch.zeros(B, t_full, E, 1)
    model.train()
    model.zero_grad(set_to_none=True)
    with MG._Recorder() as rec:
        res = R.T.step_contrastive_distill(model, (xt, a_dummy, torch.arange(B)), ctx)
    res.loss.backward()
    calls = [v for _, v in rec.calls]
    base = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    for n, g in base.items():
        assert np.array_equal(g.numpy(), d["c0::grad::" + n].reshape(g.shape)), ("regenerated step differs from the committed golden", n)

    def rerun():
        model.load_state_dict(sd0)
        model.train()
        model.zero_grad(set_to_none=True)
        it = iter(calls)
        orig = {n: getattr(torch, n) for n in ("rand", "randint", "randn", "randperm")}
        try:
            for n in orig:
                setattr(torch, n, lambda *a, **k: next(it).clone())
            r = R.T.step_contrastive_distill(model, (xt, a_dummy, torch.arange(B)), ctx)
        finally:
            for n, fn in orig.items():
                setattr(torch, n, fn)
        r.loss.backward()
    return model, base, rerun


def gen_small_tcn_kinks():
    keep = dict(np.load(os.path.join(HERE, "tcn_kinks.npz")).items())
    MG3.KINK_DELTA = KINK_DELTA
    for tag, seed, B, t_full, L in (("tcn14l16", 481, 6, 24, 16), ("tcn14", 81, 6, 24, 8)):
        pfx = f"contrastive_{tag}::c0::"
        keep = {k: v for k, v in keep.items() if not k.startswith(pfx)}
        model, base, rerun = _replay_setup(tag, seed, B, t_full, L)
        MG3.kink_attribution(keep, pfx, model, rerun, base, set(base))
        keep[pfx + "delta"] = np.float64(KINK_DELTA)
    np.savez_compressed(os.path.join(HERE, "tcn_kinks.npz"), **keep)


class KinkLocator(MG3.KinkFlipper):
    """The dry pass of KinkFlipper that also records WHERE each candidate sits: the index of the ReLU call within the step and
    the call's tensor shape (the ordinals KinkAttribution names are positions in this sequence)."""

    def __init__(self, delta):
        super().__init__(delta=delta, dry=True)
        self.calls, self.where = [], []

    def relu(self, x, inplace=False):
        k = len(self.calls)
        self.calls.append(tuple(x.shape))
        before = len(self.ident)
        y = super().relu(x, inplace)
        self.where += [k] * (len(self.ident) - before)
        return y


def gen_kink_locations():
    """tcn_kinks.npz += <prefix>loc_call / loc_flat / loc_value (per candidate ORDINAL: ReLU call index, flat index in that
    call's (S, C, T) input, the reference's pre-activation value) and <prefix>loc_shapes (per call, padded to 3 dims) for the
    contrastive TCN fixtures -- what tests/parity_common.py::confirm_flips_on_device needs to look a named flip up in the
    device's own tensors."""
    import make_golden_r04 as MG4
    keep = dict(np.load(os.path.join(HERE, "tcn_kinks.npz")).items())
    jobs = []
    d, model, base, rerun = MG4._c4_replay_setup()
    jobs.append((f"contrastive_{MG4.C4_TAG}::c0::", MG4.C4_DELTA, rerun))
    for tag, seed, B, t_full, L in (("tcn14l16", 481, 6, 24, 16), ("tcn14", 81, 6, 24, 8)):
        model, base, rerun = _replay_setup(tag, seed, B, t_full, L)
        jobs.append((f"contrastive_{tag}::c0::", KINK_DELTA, rerun))
    # VaDE with the TCN encoder / decoder (make_golden_r03.vade_tcn_kinks' replays): the encoder's calls come first
    for fname, tagp in (("vade_tcn14_b64.npz", "vade_tcn14_b64"), ("vade_tcn14_onepass.npz", "vade_tcn14_onepass")):
        d = dict(np.load(os.path.join(HERE, fname)))
        x, a = d["x"], d["a"]
        B, T, N, _ = x.shape
        E = a.shape[2]
        K, L = d["sd::latent_space.gmm_means"].shape
        vmodel, sd0 = MG3._tcn_vade_model(d, T, N, E, L, K)
        xt, at = torch.from_numpy(x), torch.from_numpy(a)
        eps, eps_mc, tau = (torch.from_numpy(d[k]) for k in ("eps", "eps_mc", "tau"))
        for phase, klw, teacher in (("pre", 0.13, False), ("mainT", 0.7, True)):
            if f"{tagp}::{phase}::first_ordinal" not in keep:
                continue

            def vrerun(phase=phase, klw=klw, teacher=teacher, vmodel=vmodel, sd0=sd0, xt=xt, at=at, eps=eps, eps_mc=eps_mc, tau=tau, K=K, L=L):
                vmodel.load_state_dict(sd0)
                MG3._vade_run(vmodel, xt, at, eps, eps_mc, tau, phase, klw, teacher, K, L)
            jobs.append((f"{tagp}::{phase}::", float(keep.get(f"{tagp}::delta", MG3.KINK_DELTA)), vrerun))
    for pfx, delta, rerun in jobs:
        with KinkLocator(delta) as loc:
            rerun()
        n = len(loc.ident)
        assert n == loc.count == len(loc.where)
        assert n == int(keep[pfx + "count"]), ("another candidate sequence than the attribution's", pfx, n, int(keep[pfx + "count"]))
        first = keep[pfx + "first_ordinal"]
        assert first.size == 0 or int(first.max()) < n, (pfx, n, first.max() if first.size else None)
        keep[pfx + "loc_call"] = np.asarray(loc.where, dtype=np.int32)
        keep[pfx + "loc_flat"] = np.asarray([j for j, _ in loc.ident], dtype=np.int64)
        keep[pfx + "loc_value"] = np.asarray([v for _, v in loc.ident], dtype=np.float32)
        shp = np.ones((len(loc.calls), 3), dtype=np.int64)
        for k, sh in enumerate(loc.calls):
            assert len(sh) <= 3
            shp[k, :len(sh)] = sh
        keep[pfx + "loc_shapes"] = shp
        print(pfx, n, "candidates in", len(loc.calls), "ReLU calls")
    np.savez_compressed(os.path.join(HERE, "tcn_kinks.npz"), **keep)


def gen_latent4():
    MG.gen_vade("rec14l4", [""], 25, 4, 10, 12, 1131)
    MG.gen_vqvae("rec14l4", [""], 25, 4, 48, 12, 1141, kmeans=0.5)
    MG.gen_contrastive("rec14l4", [""], 24, 4, 12, 1161)


def gen_latent6():
    """latent_dim = 6, the size the reference's tutorial trains: GRU(12 -> 12) / GRU(24 -> 6) streams"""
    MG.gen_vade("rec14l6", [""], 25, 6, 10, 12, 1231)
    MG.gen_vqvae("rec14l6", [""], 25, 6, 48, 12, 1241, kmeans=0.5)
    MG.gen_contrastive("rec14l6", [""], 24, 6, 12, 1261)


def gen_latent_odd():
    """latent 7, 9, 14 (round 6: sizes between the built ones; recurrent family) -- VaDE only, batch > latent as for the other sizes"""
    MG.gen_vade("rec14l7", [""], 25, 7, 10, 12, 1331)
    MG.gen_vade("rec14l9", [""], 25, 9, 10, 12, 1431)
    MG.gen_vade("rec14l14", [""], 25, 14, 10, 16, 1531)
    MG.gen_contrastive("rec14l7", [""], 24, 7, 12, 1361)
    MG.gen_vqvae("rec14l14", [""], 25, 14, 48, 16, 1541, kmeans=0.5)


if __name__ == "__main__":
    what = sys.argv[1:] or ["l4", "l6", "tcnkinks"]
    if "l4" in what:
        gen_latent4()
    if "l6" in what:
        gen_latent6()
    if "lodd" in what:
        gen_latent_odd()
    if "tcnkinks" in what:
        gen_small_tcn_kinks()
    if "kinkloc" in what:
        gen_kink_locations()
    for f in ("vade_rec14l4.npz", "vqvae_rec14l4.npz", "contrastive_rec14l4.npz", "vade_rec14l6.npz", "vqvae_rec14l6.npz",
              "contrastive_rec14l6.npz", "tcn_kinks.npz"):
        p = os.path.join(HERE, f)
        if os.path.exists(p):
            print(f, os.path.getsize(p))
