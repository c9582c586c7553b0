import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_lib():
    """The HIP library, built in-tree.  GPU tests fail (not skip) if it is missing on a GPU box."""
    from slam3d_gx_amd import capi
    if not _gpu_available():
        pytest.skip("no GPU visible")
    return capi.load_library()
