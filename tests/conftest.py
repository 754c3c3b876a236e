import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The ctypes-bound HIP library; GPU tests fail loudly if it is not built/loadable."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked test collected on a machine without a GPU"
    from reftr_amd import hip as H
    H.lib()
    return H
