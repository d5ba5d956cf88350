"""pytest helper: build + load the CPU *emulation* build of the HIP kernels (tests/emu).

Test tool only -- lets `-m "not gpu"` tests execute the real kernel sources (fibers emulate HIP
threads, __syncthreads, wave shuffles and MFMA) in the GPU-less build container.
"""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "emu", "_build", "libdeepof_emu.so")
_lib = None


def emu_lib():
    """DOF_EMU_SO=<path>: load that emulation build instead (tools/emu_asan.sh builds one with AddressSanitizer: LDS and
    register arrays are ordinary arrays there, so a kernel overrunning a fixed-size tile is reported with its line)."""
    global _lib
    if _lib is None:
        from deepof_amd import _capi
        override = os.environ.get("DOF_EMU_SO")
        if override:
            _lib = _capi.bind(ctypes.CDLL(override))
            return _lib
        subprocess.run(["make", "-C", os.path.join(ROOT, "deepof_amd", "csrc"), "emu", "-j4"], check=True,
                       stdout=subprocess.DEVNULL)
        _lib = _capi.bind(ctypes.CDLL(EMU_SO))
    return _lib
