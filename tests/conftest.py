import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "libtike-cufft_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model():
    """The reference's own test fixtures (tests/model/*)."""
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden",
                                     "model_fixtures.npz")))
