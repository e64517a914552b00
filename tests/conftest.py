import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hiplib():
    """The in-tree HIP library; GPU tests fail loudly (no skip, no fallback) if it is missing."""
    from ndzip_amd import hip

    return hip.lib()


@pytest.fixture(scope="session")
def cuda_device():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch.device("cuda:0")
