"""Shared fixtures.  Tests marked `gpu` need an MI355X; everything else runs on CPU."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference header; only exists in the build container."""
    from oracle.pyoracle import Reference, reference_available

    if not reference_available():
        pytest.skip("oracle/_ref not built (reference sources absent on this machine)")
    return Reference()


@pytest.fixture(scope="session")
def literals():
    with open(os.path.join(GOLDEN, "ref_test_literals.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    with np.load(os.path.join(GOLDEN, "ref_outputs.npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def scaled_err(y, y_ref):
    """max|y - y_ref| / max|y_ref|: the parity metric (pointwise relative error is
    meaningless at the zero crossings of a filtered signal)."""
    y, y_ref = np.asarray(y, dtype=np.float64), np.asarray(y_ref, dtype=np.float64)
    if y_ref.size == 0:
        return 0.0
    den = float(np.max(np.abs(y_ref)))
    return float(np.max(np.abs(y - y_ref))) / (den if den > 0 else 1.0)
