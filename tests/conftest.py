"""pytest configuration: the `gpu` marker and shared fixtures."""

import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Outputs of the reference itself (tests/golden/make_golden.py)."""
    return np.load(ROOT / "tests" / "golden" / "reference_outputs.npz")
