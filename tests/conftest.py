import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (os.path.join(ROOT, "sat-bundleadjust_amd"), ROOT, os.path.dirname(os.path.abspath(__file__))):
    if path not in sys.path:
        sys.path.insert(0, path)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A diverged solve would otherwise sit in scipy-style retry loops until max_nfev: bound every test."""
    if config.pluginmanager.hasplugin("timeout"):
        for item in items:
            if item.get_closest_marker("timeout") is None:
                item.add_marker(pytest.mark.timeout(600))


def _gpu_ok():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must not silently pass without a device: fail loudly instead of skipping."""
    assert _gpu_ok(), "this test needs a HIP device (run it through gpurun with -m gpu)"
    from satba import engine_hip

    engine_hip.load_library()  # raises if libsatba_hip.so was not built
    return True
