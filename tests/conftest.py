import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(GOLDEN_DIR, "golden.npz")))


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN_DIR, "golden_meta.json")) as f:
        return json.load(f)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture
def deterministic():
    """GPU tests whose bounds should carry no run-to-run slack: the training kernels' float-atomic sections in a fixed workgroup order, tuner off
    (ted_spad_amd.engine.set_deterministic); the test also checks that no workgroup gave up waiting at a gate."""
    from ted_spad_amd import engine as E
    E.set_deterministic(True)
    try:
        yield
        assert E.deterministic_giveups() == 0
    finally:
        E.set_deterministic(False)

