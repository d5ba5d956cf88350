"""Child process of test_gru_unfused_weight_gradient_* and test_outer_fp32_kernel_* : the small-latent recurrent reference
goldens with one of the weight-gradient switches at 0 (set by the parent, named in argv[2]):
  DOF_GRU_WGRAD_FUSED=0  the lane-per-unit GRU backward kernels write dG and the generic k_outer jobs reduce it -- the path the
                         product took before round 6 fused the weight gradients into k_gru3_bwd;
  DOF_OUTER_B3=0         every weight-gradient job on the fp32 matrix instructions (k_outer) instead of the bf16-piece kernel
                         k_outer_b3 (+ the latent-8 and latent-16 goldens, whose dense / decoder jobs are the kernel's).
argv[1] = "emu" | "gpu"."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
SWITCH = sys.argv[2] if len(sys.argv) > 2 else "DOF_GRU_WGRAD_FUSED"
assert os.environ.get(SWITCH) == "0"
import parity_common as PC  # noqa: E402

if sys.argv[1] == "emu":
    from emu_util import emu_lib
    lib, dev = emu_lib(), "cpu"
    cases = [("rec14l4", "pre"), ("rec14l6", "mainX"), ("rec14l5", "main")]
    if SWITCH == "DOF_OUTER_B3":
        cases += [("rec14", "mainT"), ("rec14l16", "pre")]
else:
    from deepof_amd._lib import load_hip_library
    lib, dev = load_hip_library(), "cuda"
    cases = [(t, p) for t in ("rec14l4", "rec14l5", "rec14l6", "rec14l7", "rec14l9", "rec14l10") for p in ("pre", "main", "mainT", "mainX")]
    if SWITCH == "DOF_OUTER_B3":
        cases += [(t, p) for t in ("rec14", "c5l8", "rec14l16", "rec14l32") for p in ("pre", "mainT")]
G = os.path.join(HERE, "golden")
for tag, phase in cases:
    print(tag, phase, PC.run_phase_check(lib, dev, G, tag, phase))
PC.run_vqvae_check(lib, dev, G, "rec14l4")
PC.run_contrastive_check(lib, dev, G, "rec14l6")
print("PROBE ok")
