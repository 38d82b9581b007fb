import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import poselib_amd

        return poselib_amd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """The product library on a real device.  On the GPU box a missing library / device is a FAILURE
    (no silent fallback); elsewhere -m gpu tests are simply not selected."""
    import poselib_amd

    assert os.path.exists(poselib_amd.LIB_PATH), "libposelib_amd.so not built (python __graft_entry__.py)"
    assert poselib_amd.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    return poselib_amd
