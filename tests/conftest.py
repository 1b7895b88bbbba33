import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _engine_precision(request):
    """Test modules exercise the float32 engine (the headline path, north_star's tolerance) unless they declare
    ``SC_PRECISION``: tests/test_gpu_fp64.py runs with "dtype" -- the package default, where ``dtype=complex128`` (the
    reference's default) selects the float64 engine."""
    from spectral_connectivity_amd import options
    old = options.precision
    options.precision = getattr(request.module, "SC_PRECISION", "float32")
    yield
    options.precision = old
