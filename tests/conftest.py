import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must fail loudly on a GPU box if the HIP library is unusable, and be deselected elsewhere via -m
    pass


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (there is no CPU fallback)")
    return torch.device("cuda:0")
