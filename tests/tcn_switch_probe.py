"""Child process of test_tcn_kernel_switches_gpu: the B = 64 VaDE-TCN reference check (gradients against the reference golden
with the explicit ReLU-flip attribution) under whatever DOF_TCN_* switches the parent set; prints one JSON line."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from deepof_amd._lib import load_hip_library  # noqa: E402
import parity_common as PC  # noqa: E402

fixture = os.environ.get("DOF_PROBE_FIXTURE", "vade_tcn14_b64.npz")
if fixture == "oracle_t29_t30":  # windows of 26 .. 50 steps against the oracle on tie-free draws (run_vade_tcn_vs_oracle)
    for T in (29, 30):
        PC.run_vade_tcn_vs_oracle(load_hip_library(), "cuda", L=8, T=T)
    print("PROBE " + json.dumps({"ok": True}))
    sys.exit(0)
res = PC.run_vade_tcn_b64_check(load_hip_library(), "cuda", os.path.join(HERE, "golden"), fixture=fixture,
                                min_main=200 if "onepass" in fixture else 20)
print("PROBE " + json.dumps({"ok": True, "result": repr(res)[:300]}))
