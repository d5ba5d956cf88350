"""Child process of test_gru_unfused_weight_gradient_* : the small-latent recurrent reference goldens with
DOF_GRU_WGRAD_FUSED=0 (set by the parent), i.e. the lane-per-unit GRU backward kernels writing dG and the generic k_outer
jobs reducing it -- the path the product took before round 6 fused the weight gradients into k_gru3_bwd.  argv[1] = "emu" | "gpu"."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
assert os.environ.get("DOF_GRU_WGRAD_FUSED") == "0"
import parity_common as PC  # noqa: E402

if sys.argv[1] == "emu":
    from emu_util import emu_lib
    lib, dev = emu_lib(), "cpu"
    cases = [("rec14l4", "pre"), ("rec14l6", "mainX"), ("rec14l5", "main")]
else:
    from deepof_amd._lib import load_hip_library
    lib, dev = load_hip_library(), "cuda"
    cases = [(t, p) for t in ("rec14l4", "rec14l5", "rec14l6", "rec14l7", "rec14l9", "rec14l10") for p in ("pre", "main", "mainT", "mainX")]
G = os.path.join(HERE, "golden")
for tag, phase in cases:
    print(tag, phase, PC.run_phase_check(lib, dev, G, tag, phase))
PC.run_vqvae_check(lib, dev, G, "rec14l4")
PC.run_contrastive_check(lib, dev, G, "rec14l6")
print("PROBE ok")
