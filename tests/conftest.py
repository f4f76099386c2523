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
    # DSPX_TEST_RUN_SHARES=1|2 runs the whole suite with the matrix kernels' runs in fixed shares / drawn from the
    # stream's counters whatever the input's size (the library's test hook dspx_test_run_shares; results must not
    # depend on it -- tests that set the hook themselves restore the default, 0, when they end)
    mode = os.environ.get("DSPX_TEST_RUN_SHARES")
    if mode:
        import dsp_amd

        dsp_amd.lib().dspx_test_run_shares(int(mode))


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


def cwt_row_tolerance(oracle, sig, scale, psi, wx, base=1e-12):
    """Parity bound of one row of the wavelet bank, derived from what the row is: the first difference of the
    convolution c of the signal with the scale's gathered wavelet, times -sqrt(scale) (reference dsp.hpp:1185-1191).
    The kernels match the reference's c to `base` of max|c| (a different order of the same products); the
    difference of two neighbouring values of c then carries that error against a result of size max|c[i+1] - c[i]|.
    So a row is held to base * max|c| / max|diff c| (never less than base), measured on the row's own c."""
    sig, psi, wx = (np.asarray(a, dtype=np.float64) for a in (sig, psi, wx))
    step, rng = wx[1] - wx[0], wx[-1] - wx[0]
    idx = oracle.cwt_scale_indices(float(scale), float(step), float(rng), psi.size, int(np.ceil(scale * rng)) + 1)
    c = np.convolve(sig, psi[idx][::-1])
    d = np.abs(np.diff(c)).max() if c.size > 1 else 0.0
    return base * max(1.0, float(np.abs(c).max()) / d) if d > 0 else base
