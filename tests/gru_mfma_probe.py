"""Child process of test_gru16_matrix_pipe_* : runs the recurrent reference goldens with DOF_GRU_MFMA_MIN_S=0 (set by the
parent), i.e. through k_gru16x_fwd / _bwd and k_gru8x_fwd / _bwd (matrix-pipe recurrences, recomputed gates) and -- latent 16 / 32 --
k_grumx_fwd / k_grum_bwd at the goldens' small batch sizes, where the product would pick the lane-per-unit kernels.  argv[1] = "emu" | "gpu"."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
assert os.environ.get("DOF_GRU_MFMA_MIN_S") == "0"
import parity_common as PC  # noqa: E402

if sys.argv[1] == "emu":
    from emu_util import emu_lib
    lib, dev = emu_lib(), "cpu"
    cases = [("rec14", "pre"), ("rec14", "mainX"), ("c5l8", "mainT"), ("rec14l16", "mainT")]
else:
    from deepof_amd._lib import load_hip_library
    lib, dev = load_hip_library(), "cuda"
    cases = [(t, p) for t in ("rec14", "c5l8", "rec14l16", "rec14l32") for p in ("pre", "main", "mainT", "mainX")]
G = os.path.join(HERE, "golden")
for tag, phase in cases:
    print(tag, phase, PC.run_phase_check(lib, dev, G, tag, phase))
PC.run_trace_check(lib, dev, G)                      # 6 optimiser steps
PC.run_vqvae_check(lib, dev, G, "rec14")             # VQ-VAE: two decoder passes share the encoder's kernels
if sys.argv[1] == "gpu":   # the GEMM-shaped recurrence of the wider layers (k_grumx_fwd / k_grum_bwd; latent 16 / 32)
    for tag in ("rec14l16", "rec14l32"):
        PC.run_vqvae_check(lib, dev, G, tag)
        PC.run_contrastive_check(lib, dev, G, tag)
if sys.argv[1] == "gpu":
    PC.run_contrastive_check(lib, dev, G, "rec14")
print("PROBE ok")
