import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return os.path.join(GOLDEN, name + ".npz")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def _load(name):
        return np.load(golden_path(name), allow_pickle=False)
    return _load
