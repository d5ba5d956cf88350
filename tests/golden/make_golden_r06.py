"""Round-6 reference fixtures (run in the build container, where /root/reference is importable):

    python tests/golden/make_golden_r06.py l4 l6       # recurrent family at latent 4 (the API's default latent_dim) and 6 (the tutorial's)
    python tests/golden/make_golden_r06.py lodd        # latent 7, 9, 14 (recurrent family)
    python tests/golden/make_golden_r06.py tcnkinks    # ReLU-kink attribution for the two small contrastive TCN fixtures

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
    for f in ("vade_rec14l4.npz", "vqvae_rec14l4.npz", "contrastive_rec14l4.npz", "vade_rec14l6.npz", "vqvae_rec14l6.npz",
              "contrastive_rec14l6.npz", "tcn_kinks.npz"):
        p = os.path.join(HERE, f)
        if os.path.exists(p):
            print(f, os.path.getsize(p))
