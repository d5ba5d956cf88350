import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

if not os.path.exists("/dev/kfd"):
    # CPU container: one worker process per core (below), so one BLAS / OpenMP thread each -- eight workers with a thread pool
    # per core each only fight for the cores, and the oracle's own fp32 rounding then does not move with the thread count
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """GPU-less container: spread the CPU suite (most of it runs HIP kernels under the SIMT emulator, one fiber at a
    time) over a few worker processes when pytest-xdist is available and the caller did not choose -n / -p no:xdist.
    On a GPU box nothing changes: the device tests stay in one process."""
    if _gpu_present() or not config.pluginmanager.hasplugin("xdist") or "PYTEST_XDIST_WORKER" in os.environ:
        return None
    if getattr(config.option, "numprocesses", None) in (None, 0) and not getattr(config.option, "collectonly", False):
        # build the emulator library once, before the workers race to do it
        subprocess.run(["make", "-C", os.path.join(ROOT, "deepof_amd", "csrc"), "emu", "-j4"], check=False,
                       stdout=subprocess.DEVNULL)
        config.option.numprocesses = min(8, os.cpu_count() or 1)
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
