import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def _has_gpu() -> bool:
    try:
        from cosdata_amd import _lib
        import ctypes
        n = ctypes.c_int32()
        return _lib.lib().cos_device_count(ctypes.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a device must FAIL loudly, not skip: the driver records which .so got loaded.
    pass
