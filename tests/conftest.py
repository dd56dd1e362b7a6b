import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "future-object-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture(autouse=True)
def _no_leaked_dropout_base():
    """A captured train-mode step installs a device-side dropout seed base (ops.DROP_BASE) for the process; tests that
    restate the dropout hash from a host seed must not inherit it from an earlier test."""
    yield
    mod = sys.modules.get("future_od.native.ops")
    if mod is not None:
        mod.DROP_BASE = None
