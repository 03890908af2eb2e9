import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
    """PyTorch wheels carry their own HIP runtime.  When both it and the system runtime libsmashx links against
    live in one process, torch's must initialise first (the other order leaves torch with "No HIP GPUs are
    available").  bench.py does so naturally; the GPU tests that use torch tensors (tile exchange) need this."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield
