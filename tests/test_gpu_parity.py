"""Parity of the HIP path on a real MI355X (through the C ABI) against the reference golden fixtures
and the CPU oracle.  Tolerances: fp32 everywhere; 1e-5 abs / 1e-4 rel on activations, 5e-5 / 5e-4 on
gradients (fp32 reduction-order differences between oneDNN/ATen and the SoA kernels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from deepof_amd._lib import load_hip_library
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return load_hip_library()


def test_library_is_the_hip_build(hip):
    import deepof_amd._lib as L
    assert L.LIB_PATH.endswith("libdeepof_hip.so")
    assert hip.dof_abi_version() == 1


def test_gather_gpu(hip):
    from parity_common import gather_check
    gather_check(hip, "cuda")


@pytest.mark.parametrize("tag", ["rec14", "rec28"])
def test_vade_eval_forward_gpu(hip, golden_dir, tag):
    from deepof_amd.engine import create_vade_engine
    from parity_common import load_golden, params_from
    d = load_golden(golden_dir, f"vade_{tag}.npz")
    x, a = torch.from_numpy(d["x"]).cuda(), torch.from_numpy(d["a"]).cuda()
    B, T, N, _ = x.shape
    K, L = d["sd::latent_space.gmm_means"].shape
    eng = create_vade_engine(B, T, d["adj"], L, K)
    eng.load_state_dict(params_from(d))
    out = eng.forward(x, a, None, want_loc=True, want_enc=True)
    np.testing.assert_allclose(out["enc"].cpu().numpy(), d["eval_enc"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(out["z"].cpu().numpy(), d["eval_z"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(out["q"].cpu().numpy(), d["eval_q"], atol=1e-5, rtol=1e-3)
    np.testing.assert_allclose(out["loc"].cpu().numpy(), d["eval_loc"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("tag,phase", [("rec14", "pre"), ("rec14", "main"), ("rec14", "mainT"), ("rec28", "pre"),
                                       ("rec28", "main"), ("rec28", "mainT")])
def test_vade_loss_grads_gpu(hip, golden_dir, tag, phase):
    from parity_common import run_phase_check
    worst = run_phase_check(hip, "cuda", golden_dir, tag, phase)
    print("worst grad errors", worst)


def test_vade_train_trace_gpu(hip, golden_dir):
    from parity_common import run_trace_check
    run_trace_check(hip, "cuda", golden_dir)


def test_full_size_oracle_parity_c2(hip):
    """BASELINE config C2 shapes (B=1024, N=E=14, W=25, K=10, L=8): HIP forward vs the CPU oracle."""
    from deepof_amd.engine import create_vade_engine
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph
    from oracle import vade as OV
    nodes, edges = bodypart_graph([""])
    adj = adjacency_from_graph(nodes, edges)
    B, T, L, K = 1024, 25, 8, 10
    eng = create_vade_engine(B, T, adj, L, K)
    g = torch.Generator().manual_seed(0)
    for n in eng.names:
        shape = eng.layout[n][2]
        scale = 0.3 if len(shape) > 1 else 0.1
        v = torch.randn(shape, generator=g) * scale
        if n.endswith("norm1.weight") or n.endswith("norm2.weight") or n.endswith("norm3.weight"):
            v = 1.0 + v
        eng.view(n).copy_(v)
    x = torch.randn(B, T, len(nodes), 3, generator=g)
    a = torch.randn(B, T, len(edges), 1, generator=g)
    x[5, 3:12] = 0.0  # masked frames: shorter decoder length + zero conv rows in the encoder
    out = eng.forward(x.cuda(), a.cuda(), None, want_loc=True)
    P = eng.state_dict()
    with torch.no_grad():
        ref = OV.vade_forward(P, x, a, training=False)
    np.testing.assert_allclose(out["z"].cpu().numpy(), ref["z"].numpy(), atol=3e-5, rtol=1e-3)
    np.testing.assert_allclose(out["q"].cpu().numpy(), ref["q"].numpy(), atol=3e-5, rtol=2e-3)
    np.testing.assert_allclose(out["loc"].cpu().numpy(), ref["loc"].numpy(), atol=1e-4, rtol=1e-3)
