"""The C-ABI shared library loads and exports every symbol include/deepof_hip.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "deepof_hip.h")
LIB = os.path.join(ROOT, "deepof_amd", "csrc", "libdeepof_hip.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dof_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.dirname(LIB), "-j4"], check=True, stdout=subprocess.DEVNULL)
    import torch  # noqa: F401  (same load order as the product: torch's HIP runtime first)
    return ctypes.CDLL(LIB)


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("dof_window_gather", "dof_window_gather_range", "dof_vade_plan_create", "dof_vade_forward",
                 "dof_vade_loss_grads", "dof_optimizer_step", "dof_vade_workspace_bytes", "dof_last_error_string"):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"libdeepof_hip.so lacks: {missing}"


def test_python_binding_table_matches_header(lib):
    from deepof_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_functions()
    _capi.bind(lib)
    assert lib.dof_abi_version() == _capi.ABI_VERSION


def test_plan_layout_without_gpu(lib):
    """Host-only entry points work without a device: parameter table of the 14-body-part VaDE."""
    import numpy as np
    from deepof_amd import _capi
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph, censnet_operators
    _capi.bind(lib)
    nodes, edges = bodypart_graph([""])
    lap, elap, inc = censnet_operators(adjacency_from_graph(nodes, edges))
    dims = _capi.VadeDims(1024, 25, 14, 14, 8, 10, 32)
    plan = ctypes.c_void_p()
    assert lib.dof_vade_plan_create(ctypes.byref(dims), lap.ctypes.data, elap.ctypes.data, inc.ctypes.data,
                                    ctypes.byref(plan)) == 0
    assert lib.dof_vade_param_total(plan) == 21626        # reference VaDE-recurrent parameter count
    assert lib.dof_vade_param_count(plan) == 87
    assert lib.dof_vade_param_name(plan, 0) == b"encoder.node_recurrent_block.conv1d.weight"
    assert lib.dof_vade_workspace_bytes(plan) > 0
    lib.dof_vade_plan_destroy(plan)
    # transformer family: VaDEPT(encoder_type="transformer") has 102,184 parameters + 4 BatchNorm
    # running buffers (16 + 16 + 8 + 8 floats) in the flat buffer
    assert lib.dof_vade_tfm_plan_create(ctypes.byref(dims), lap.ctypes.data, elap.ctypes.data, inc.ctypes.data,
                                        ctypes.byref(plan)) == 0
    assert lib.dof_vade_param_total(plan) == 102184 + 48
    assert lib.dof_vade_param_name(plan, 0) == b"encoder.node_tf.embed.weight"
    assert lib.dof_tfm_dropout_site_count(plan) == 22
    assert lib.dof_tfm_dropout_site_name(plan, 1) == b"enc.node.l0.attn"
    assert lib.dof_tfm_dropout_site_numel(plan, 1) == 1024 * 14 * 4 * 25 * 25
    lib.dof_vade_plan_destroy(plan)
    bad = _capi.VadeDims(8, 25, 14, 14, 11, 10, 32)
    assert lib.dof_vade_plan_create(ctypes.byref(bad), lap.ctypes.data, elap.ctypes.data, inc.ctypes.data,
                                    ctypes.byref(plan)) == -2
    assert b"latent_dim 11" in lib.dof_last_error_string()
    odd = _capi.VadeDims(8, 25, 14, 14, 7, 10, 32)   # 7, 9, 14 (round 6): the recurrent family only
    assert lib.dof_vade_plan_create(ctypes.byref(odd), lap.ctypes.data, elap.ctypes.data, inc.ctypes.data,
                                    ctypes.byref(plan)) == 0
    lib.dof_vade_plan_destroy(plan)
    assert lib.dof_vade_tcn_plan_create(ctypes.byref(odd), lap.ctypes.data, elap.ctypes.data, inc.ctypes.data,
                                        ctypes.byref(plan)) == -2


def test_product_path_fails_loudly_without_gpu():
    import torch
    from deepof_amd.engine import create_vade_engine
    import numpy as np
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        create_vade_engine(8, 25, np.eye(3, dtype=np.float32), 8, 4)


def test_native_comm_refuses_without_rccl():
    """The emulation build has no RCCL: the communicator entry points fail loudly (no host fallback)."""
    import ctypes
    from emu_util import emu_lib
    lib = emu_lib()
    buf = ctypes.create_string_buffer(128)
    assert lib.dof_comm_unique_id(buf) != 0
    assert b"librccl" in lib.dof_last_error_string()


def test_switch_table_is_complete():
    """deepof_amd/_switches.py lists EVERY environment variable the compiled library reads: the table against the getenv
    sites of the kernel sources (a switch cannot be added without documenting it and naming the test that runs it)."""
    import glob
    import re
    from deepof_amd._switches import HOST_SWITCHES, LIBRARY_SWITCHES
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepof_amd")
    found = set()
    for f in glob.glob(os.path.join(root, "csrc", "*.hip")) + glob.glob(os.path.join(root, "csrc", "*.h")):
        found |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(f).read()))
    assert found == set(LIBRARY_SWITCHES), (sorted(found - set(LIBRARY_SWITCHES)), sorted(set(LIBRARY_SWITCHES) - found))
    for name, (default, other, meaning, probe) in LIBRARY_SWITCHES.items():
        assert default != other and meaning and probe, name
    host = set()
    for f in glob.glob(os.path.join(root, "*.py")):
        if not f.endswith("_switches.py"):
            host |= set(re.findall(r'"(DOF_[A-Z0-9_]+)"', open(f).read()))
    assert host <= set(HOST_SWITCHES) | set(LIBRARY_SWITCHES), sorted(host - set(HOST_SWITCHES) - set(LIBRARY_SWITCHES))
