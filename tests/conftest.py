import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return REPO


@pytest.fixture(autouse=True)
def _seed_torch():
    """Every test starts from the same torch RNG state (CPU and GPU): tests that draw their jitter with torch.rand would otherwise
    depend on what ran before them (some tolerances are tight enough for a rare unlucky draw to matter)."""
    import torch
    torch.manual_seed(1234)
    yield
