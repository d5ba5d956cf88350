"""Kernel-logic checks of the HIP sources under the CPU emulator (no GPU needed) against the oracle
and the reference golden fixtures.  The same comparisons run on the real MI355X in test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

from deepof_amd.engine import VadeEngine
import parity_common as PC
from emu_util import emu_lib
from parity_common import gather_check, load_golden, params_from, run_phase_check, run_trace_check, run_vqvae_check


def test_gather_emu():
    gather_check(emu_lib(), "cpu")


# Emulator runs of a minute or more whose check the GPU suite repeats on the device (the same parity_common function through
# the HIP build): skipped in the default CPU run so that `pytest -m "not gpu"` stays under ten minutes on 8 cores; DOF_EMU_FULL=1
# runs them here (it is how a kernel change is debugged before it goes to the GPU box).
_FULL = pytest.mark.skipif(os.environ.get("DOF_EMU_FULL") != "1",
                           reason="minutes under the emulator; the GPU suite runs the same check (DOF_EMU_FULL=1 to run it here)")


def _full(*values):
    return pytest.param(*values, marks=_FULL)


@pytest.mark.parametrize("tag", ["rec14", "rec28", "c5l8", "rec14l16", _full("rec14l32"), "rec14l4", "rec14l5", "rec14l6", "rec14l7", "rec14l9", "rec14l14", "rec14l10", "rec14l12", "rec14l20", "rec14l24"])
def test_vade_eval_forward_emu(golden_dir, tag):
    d = load_golden(golden_dir, f"vade_{tag}.npz")
    x, a = torch.from_numpy(d["x"]), torch.from_numpy(d["a"])
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = VadeEngine(emu_lib(), "cpu", B, T, d["adj"], L, K)
    eng.load_state_dict(params_from(d))
    out = eng.forward(x, a, None, want_loc=True, want_enc=True)
    np.testing.assert_allclose(out["enc"].numpy(), d["eval_enc"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(out["z"].numpy(), d["eval_z"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(out["q"].numpy(), d["eval_q"], atol=1e-5, rtol=1e-3)
    np.testing.assert_allclose(out["loc"].numpy(), d["eval_loc"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("tag,phase", [("rec14", "pre"), ("rec14", "main"), ("rec14", "mainT"), ("rec14", "mainX"),
                                       ("rec28", "pre"), ("rec28", "mainT"), ("rec28", "mainX"),
                                       ("c5l8", "pre"), ("c5l8", "mainX"),
                                       ("rec14l16", "pre"), ("rec14l16", "mainX"), _full("rec14l32", "pre"),
                                       ("rec14l12", "mainX"), ("rec14l10", "pre"), ("rec14l5", "main"), ("rec14l5", "mainX"), ("rec14l4", "pre"), ("rec14l4", "mainT"), ("rec14l6", "pre"), ("rec14l6", "mainX"), ("rec14l7", "mainX"), ("rec14l9", "pre"), ("rec14l14", "mainX")])   # (latent 32: the other phases / models on the GPU)
def test_vade_loss_grads_emu(golden_dir, tag, phase):
    run_phase_check(emu_lib(), "cpu", golden_dir, tag, phase)


def test_step_begin_noise_emu():
    from parity_common import run_step_begin_check
    run_step_begin_check(emu_lib(), "cpu")


@pytest.mark.parametrize("K", [25, 40])
def test_vade_many_components_emu(K):
    from parity_common import run_vade_rec_vs_oracle
    run_vade_rec_vs_oracle(emu_lib(), "cpu", K)


def test_vade_train_trace_emu(golden_dir):
    run_trace_check(emu_lib(), "cpu", golden_dir)


@pytest.mark.parametrize("tag", ["rec14", "rec28", _full("rec14l16"), _full("rec14l12"), "rec14l4", "rec14l5"])   # (c5l8 / c3k512 / rec14l32: GPU only, minutes each under the emulator)
def test_vqvae_emu(golden_dir, tag):
    run_vqvae_check(emu_lib(), "cpu", golden_dir, tag)


@pytest.mark.parametrize("tag", ["rec14", "rec28"])
def test_contrastive_losses_emu(golden_dir, tag):
    from parity_common import run_contrastive_loss_check
    run_contrastive_loss_check(emu_lib(), "cpu", golden_dir, tag)


@pytest.mark.parametrize("tag", ["rec14", "rec28", _full("c5l8"), _full("rec14l16"), "rec14l12", "rec14l4", "rec14l5"])
def test_contrastive_step_emu(golden_dir, tag):
    from parity_common import run_contrastive_check
    run_contrastive_check(emu_lib(), "cpu", golden_dir, tag)


@pytest.mark.parametrize("fixture", ["contrastive_tcn14.npz", _full("contrastive_tcn14l16.npz")])
def test_contrastive_tcn_emu(golden_dir, fixture):
    from parity_common import run_contrastive_tcn_check
    run_contrastive_tcn_check(emu_lib(), "cpu", golden_dir, fixture)


@pytest.mark.parametrize("fixture", ["vade_tcn14.npz", _full("vade_tcn14w50.npz")])
def test_vade_tcn_emu(golden_dir, fixture):
    from parity_common import run_vade_tcn_check
    run_vade_tcn_check(emu_lib(), "cpu", golden_dir, fixture)


def test_vqvae_tcn_emu(golden_dir):
    from parity_common import run_vqvae_tcn_check
    run_vqvae_tcn_check(emu_lib(), "cpu", golden_dir)


def test_turtle_teacher_emu(golden_dir):
    from parity_common import run_turtle_check
    run_turtle_check(emu_lib(), "cpu", golden_dir)


@_FULL
def test_vade_tcn_window_29_emu():
    """An odd window above 25: the 8-sequence time-resident convolutions and the 2-sequence weight-gradient chunks (round 4)."""
    from parity_common import run_vade_tcn_vs_oracle
    run_vade_tcn_vs_oracle(emu_lib(), "cpu", L=8, T=29)


def test_distillation_head_emu(golden_dir):
    from parity_common import run_distill_head_check
    run_distill_head_check(emu_lib(), "cpu", golden_dir)


@pytest.mark.parametrize("L", [4, 5, 6, 12, 16])   # 16: the decoder's repeated input takes 64-channel rows
def test_vade_tcn_padded_decoder_input_emu(L):
    from parity_common import run_vade_tcn_vs_oracle
    run_vade_tcn_vs_oracle(emu_lib(), "cpu", L=L)


def test_preprocess_tables_emu(golden_dir):
    """N2: device preprocessing (emulated kernels) against the outputs of the reference's scale_table / _pp_* run."""
    PC.run_preprocess_check(emu_lib(), "cpu", golden_dir)


def test_preprocess_minmax_and_filter_emu(golden_dir):
    """N2 remainder: scale="minmax" and filter_low_variance (emulated kernels) against the reference's own outputs."""
    PC.run_preprocess_r03_check(emu_lib(), "cpu", golden_dir)


@pytest.mark.parametrize("modes", [dict(), dict(dist="per_column", speed="per_column", coord="per_column"),
                                   dict(dist=None, speed="groupwise", coord="per_column")])
def test_preprocess_tables_vs_oracle_emu(modes):
    PC.run_preprocess_vs_oracle(emu_lib(), "cpu", **modes)


@pytest.mark.parametrize("modes", [dict(scale="robust"), dict(scale="robust", dist="per_column", speed="per_column", coord="per_column"),
                                   dict(scale="minmax", dist=None, speed="groupwise", coord="per_column")])
def test_preprocess_tables_other_scalers_vs_oracle_emu(modes):
    """scale = "robust" (exact radix selection of medians / quartiles) and "minmax" on ragged random tables, rows sampled."""
    PC.run_preprocess_vs_oracle(emu_lib(), "cpu", samples_max=110, seed=23, **modes)


def test_preprocess_tables_sampled_rows_emu():
    PC.run_preprocess_vs_oracle(emu_lib(), "cpu", samples_max=120, seed=9)


def test_preprocess_tables_long_video_emu():
    """> 16384 rows in one video: the size-factor selection leaves the register-resident path."""
    PC.run_preprocess_vs_oracle(emu_lib(), "cpu", n_videos=2, frames=(16_700, 40), seed=13)


@pytest.mark.parametrize("shape", [
    dict(n_videos=4, frames=(1, 7, 8, 9), parts=3, edge_stride=1, nan_rate=0.0),            # one-frame video, no gap at all
    dict(n_videos=2, frames=(70, 33), parts=3, edge_stride=15),                             # a single edge column (packed 8x)
    dict(n_videos=2, frames=(70, 33), parts=4, edge_stride=3, n_angles=3),                  # angle columns
    dict(n_videos=2, frames=(40, 25), parts=9, edge_stride=2),                              # 153 distances, > 64 output columns
    dict(n_videos=2, frames=(30, 17), parts=14, edge_stride=2, dist="per_column"),          # 462 raw columns, > 128 output columns
])
def test_preprocess_tables_shapes_emu(shape):
    PC.run_preprocess_vs_oracle(emu_lib(), "cpu", seed=17, **shape)


def test_preprocess_constant_columns_emu():
    """Columns that are constant within every video.  StandardScaler's near-constant rule gives the per-video scaler
    scale 1, so the value is (x - mean) = 0 up to the rounding of the mean; the reference then fits its GLOBAL scaler on
    those residues (|y| ~ 4e-16 in one video, 0 in another) and amplifies them to O(1) (sklearn gives 0.866 / -1.155
    for the case below).  That is last-bit noise of numpy's pairwise sum and cannot be reproduced; the device path
    computes the per-video mean of a constant exactly, so such columns come out as exactly 0.  Every other column
    must still match the oracle."""
    from deepof_amd.preprocess import preprocess_tables
    from oracle import preprocess as op
    bps = ["B_Nose", "B_Tail_base", "B_Center", "W_Nose", "W_Tail_base", "W_Center"]
    tabs, cols = PC.synth_raw_tables(2, (60, 45), bps, seed=3, nan_rate=0.0)
    pos = {c: i for i, c in enumerate(cols)}
    const = ["B_Center", ("B_Nose", "B_Center"), ("W_Center", "x")]   # a zero speed, a constant distance, a constant coordinate
    for t in tabs.values():
        t[:, pos[const[0]]] = 0.0
        t[:, pos[const[1]]] = 7.25
        t[:, pos[const[2]]] = -3.5
    node_cols, edge_cols, _ = PC.preprocess_output_columns(cols)
    edge_cols = sorted(set(edge_cols) | {const[1]})
    kw = dict(dist_standardize="per_column", speed_standardize="per_column", coord_standardize="per_column")
    want, _ = op.preprocess(tabs, cols, ["B", "W"], **kw)
    res = preprocess_tables(tabs, cols, ["B", "W"], node_cols, edge_cols, (), device="cpu", lib=emu_lib(), **kw)
    assert bool(torch.isfinite(res.node_table).all()) and bool(torch.isfinite(res.edge_table).all())
    assert float(res.node_table[:, node_cols.index(const[0])].abs().max()) == 0.0      # 0 speed: exact in both
    assert float(res.edge_table[:, edge_cols.index(const[1])].abs().max()) == 0.0
    keep_n = [c for c in node_cols if c not in const]
    keep_e = [c for c in edge_cols if c not in const]
    sel_n = torch.tensor([node_cols.index(c) for c in keep_n])
    sel_e = torch.tensor([edge_cols.index(c) for c in keep_e])
    import copy
    sub = copy.copy(res)
    sub.node_table, sub.edge_table = res.node_table[:, sel_n], res.edge_table[:, sel_e]
    PC._check_tables(sub, want, cols, keep_n, keep_e, [], "columns next to constant ones")


def test_vade_tfm_emu(golden_dir):
    """Transformer family (R17) under the emulator against the reference golden."""
    print(PC.run_vade_tfm_check(emu_lib(), "cpu", golden_dir))


def test_vqvae_tfm_emu(golden_dir):
    print(PC.run_vqvae_tfm_check(emu_lib(), "cpu", golden_dir))


def test_contrastive_tfm_emu(golden_dir):
    print(PC.run_contrastive_tfm_check(emu_lib(), "cpu", golden_dir))


@pytest.mark.parametrize("n_nodes,latent,kind", [(8, 4, "vade"), (11, 6, "vqvae"), (16, 8, "vade"), (22, 8, "vqvae"),
                                                  (11, 16, "vade"), (11, 16, "vqvae"), (14, 16, "vade"), (8, 16, "vqvae"),
                                                  (10, 8, "vade"), (12, 6, "vqvae"), (19, 8, "vade"), (7, 6, "vqvae"),
                                                  (11, 10, "vade"), (11, 12, "vqvae")])
def test_tfm_other_widths_emu(n_nodes, latent, kind):
    """key_dim 24 / 32 / 48 / 64 and decoder widths 16 / 24 / 32 of the transformer family against the oracle."""
    print(PC.run_tfm_widths_vs_oracle(emu_lib(), "cpu", n_nodes, latent, B=4, T=6, kind=kind))


@pytest.mark.parametrize("n_nodes,latent,kind,T", [(8, 8, "vade", 66), (11, 6, "vqvae", 70)])
def test_tfm_long_windows_emu(n_nodes, latent, kind, T):
    """Windows beyond the LDS-resident attention kernels (T > 64): k_tfm_attn_fwd_long / _bwd_long against the oracle."""
    print(PC.run_tfm_widths_vs_oracle(emu_lib(), "cpu", n_nodes, latent, B=2, T=T, kind=kind))


@_FULL
def test_gru16_matrix_pipe_kernels_emu():
    """k_gru16m_fwd / k_gru16m_bwd (the encoder streams' (16, 16) GRU on the matrix pipe, gates recomputed in the
    backward pass) against the reference goldens: a child process with DOF_GRU_MFMA_MIN_S=0, because at the goldens'
    batch sizes the product dispatch picks the lane-per-unit kernels (the switch is read once per process)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DOF_GRU_MFMA_MIN_S="0")
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gru_mfma_probe.py")
    r = subprocess.run([sys.executable, probe, "emu"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "PROBE ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@_FULL
def test_gru_unfused_weight_gradient_emu():
    """DOF_GRU_WGRAD_FUSED=0: the lane-per-unit GRU layers' weight gradients through dG + the generic k_outer jobs (the
    default fuses them into k_gru3_bwd) against the reference goldens, in a child process (the switch is read once)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DOF_GRU_WGRAD_FUSED="0")
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gru_wgrad_probe.py")
    r = subprocess.run([sys.executable, probe, "emu"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "PROBE ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@_FULL
def test_outer_fp32_kernel_emu():
    """DOF_OUTER_B3=0: the weight-gradient jobs on k_outer (fp32 matrix instructions; the default is the bf16-piece kernel
    k_outer_b3) against the reference goldens, in a child process (the switch is read once)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DOF_OUTER_B3="0")
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gru_wgrad_probe.py")
    r = subprocess.run([sys.executable, probe, "emu", "DOF_OUTER_B3"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "PROBE ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
