import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gpu_available():
    """True when a HIP device is visible.  Uses the render core's own rt_init (no torch import needed)."""
    try:
        from raytracer_amd import _ffi as F
        return F.core().rt_init(-1) == 0
    except OSError:
        return False


@pytest.fixture(scope="session")
def need_gpu():
    # GPU tests must FAIL, not skip, when the HIP library is missing on a GPU box: a silent skip would
    # hide a broken build.  Without any device (this container) they are deselected by `-m "not gpu"`.
    from raytracer_amd import _ffi as F
    lib = F.core()
    rc = lib.rt_init(-1)
    assert rc == 0, f"rt_init failed: {lib.rt_last_error()}"
    return lib
