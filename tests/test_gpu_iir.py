"""The recursive tail of dsp::lfilter (dsp.hpp:366-368) on the GPU, against the oracle's restatement of
the reference loop.  EXACT mode must be bit-identical; the default block-parallel evaluation rounds
differently (runs joined by matrix powers), so it is held to the scaled-error tolerance."""
import warnings

import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu
TOL = 1e-12


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def dsp():
    import torch

    import dsp_amd

    assert torch.cuda.is_available() and dsp_amd.device_count() >= 1
    return dsp_amd


def designs():
    """(b, a) pairs of orders 1..8 with poles inside the unit circle; a[0] is deliberately not 1 for some:
    the reference never reads it."""
    from scipy import signal

    out = [(np.array([0.25]), np.array([1.0, -0.75]))]                      # one pole
    for order, wn in ((2, 0.2), (3, 0.35), (4, 0.1), (5, 0.45), (6, 0.3), (7, 0.25), (8, 0.4)):
        b, a = signal.butter(order, wn)
        out.append((b, a))
    b, a = signal.cheby1(4, 1.0, 0.3)
    out.append((b, a))
    b, a = signal.butter(2, 0.05)
    a = a.copy()
    a[0] = 123.0                                                           # ignored by the reference
    out.append((b, a))
    out.append((np.array([1.0, 0.5, 0.25]), np.array([1.0, 0.0, 0.81])))   # resonator, poles at +-0.9j
    # numerators of 16 taps (the longest the tail kernels compute themselves), 17 and 200 taps (separate FIR pass,
    # vector and matrix pipe)
    for nb in (16, 17, 200):
        out.append((signal.firwin(nb, 0.3), np.array([1.0, -0.6, 0.2])))
    return out


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 15, 16, 17, 255, 4095, 4096, 4097, 9000, 70001])
def test_exact_mode_is_bit_identical(dsp, oracle, n):
    for b, a in designs():
        x = oracle.fill_splitmix(n, dsp.SEED, 11 * n)
        ref = oracle.lfilter(b, a, x)
        y = dsp.lfilter_iir(b, a, x, exact=True)
        assert np.array_equal(bits(y), bits(ref)), (n, a.size)


@pytest.mark.parametrize("n", [1, 5, 8, 9, 16, 17, 100, 4095, 4096, 4097, 8193, 70001, (1 << 20) + 3])
def test_block_parallel_tail_within_tolerance(dsp, oracle, n):
    for b, a in designs():
        x = oracle.fill_splitmix(n, dsp.SEED, 7 * n)
        ref = oracle.lfilter(b, a, x)
        y = dsp.lfilter_iir(b, a, x)
        assert scaled_err(y, ref) <= TOL, (n, a.size, scaled_err(y, ref))


@pytest.mark.parametrize("n", [100, 70001, (1 << 20) + 3])
def test_block_parallel_tail_is_as_accurate_as_the_reference(dsp, oracle, n):
    """Against the recurrence evaluated in extended precision (scipy's lfilter on long doubles, a[0] set to 1
    as the reference ignores it): the GPU result may not be further from it than twice the reference's own
    distance.  Narrow-band designs (butter(4, 0.1)) are the hard cases: their states cancel heavily under the
    matrix powers, which is why those are applied in double-double."""
    from scipy import signal

    assert np.finfo(np.longdouble).eps < 1e-18
    x = oracle.fill_splitmix(n, dsp.SEED, 7 * n)
    for b, a in designs():
        a1 = a.copy()
        a1[0] = 1.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)   # long doubles: accepted by the installed SciPy
            try:
                truth = signal.lfilter(b.astype(np.longdouble), a1.astype(np.longdouble), x.astype(np.longdouble))
            except (TypeError, ValueError, NotImplementedError):
                pytest.skip("this SciPy does not filter long doubles")
        if truth.dtype != np.longdouble:
            pytest.skip("this SciPy does not filter long doubles")
        scale = float(np.max(np.abs(truth)))

        def dist(v):
            return float(np.max(np.abs(v.astype(np.longdouble) - truth))) / scale

        ref_d, gpu_d = dist(oracle.lfilter(b, a, x)), dist(dsp.lfilter_iir(b, a, x))
        assert gpu_d <= 2.0 * ref_d + 2e-16, (n, a.size, ref_d, gpu_d)


def test_batched_channels_are_independent(dsp, oracle):
    from scipy import signal

    b, a = signal.butter(4, 0.2)
    for nchan, n in ((3, 1000), (70, 4099), (200, 333), (1, 3 * 4096)):
        x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c * n) for c in range(nchan)])
        ref = np.stack([oracle.lfilter(b, a, x[c]) for c in range(nchan)])
        assert np.array_equal(bits(dsp.lfilter_iir(b, a, x, exact=True)), bits(ref))
        assert scaled_err(dsp.lfilter_iir(b, a, x), ref) <= TOL


def test_long_stream_crosses_staged_pieces(dsp, oracle):
    """A channel longer than several staged pieces: the filter state travels from piece to piece on the device."""
    from scipy import signal

    n = 5 * (1 << 21) + 12345
    x = oracle.fill_splitmix(n, dsp.SEED, 0)
    for b, a in (signal.butter(2, 0.1), signal.butter(6, 0.3)):
        ref = oracle.lfilter(b, a, x)
        assert scaled_err(dsp.lfilter_iir(b, a, x), ref) <= TOL
    b, a = signal.butter(2, 0.1)
    m = 2 * (1 << 21) + 77                                                  # exact mode is sequential: keep it short
    assert np.array_equal(bits(dsp.lfilter_iir(b, a, x[:m], exact=True)), bits(oracle.lfilter(b, a, x[:m])))


def test_degenerate_and_rejected_arguments(dsp, oracle):
    x = oracle.fill_splitmix(1000, dsp.SEED, 0)
    b = oracle.firwin(31, [0.3])
    # na <= 1 is the plain filter, whatever a[0] is
    assert np.array_equal(bits(dsp.lfilter_iir(b, [2.0], x, exact=True)), bits(oracle.lfilter(b, [1.0], x)))
    # empty numerator: the FIR part is zero and so is everything after it
    assert not dsp.lfilter_iir(np.zeros(0), [1.0, -0.5], x).any()
    assert dsp.lfilter_iir(b, [1.0, -0.5], x[:0]).size == 0
    with pytest.raises(dsp.DspxError, match="at most 9 denominator"):
        dsp.lfilter_iir(b, np.ones(10), x)
    # all-zero tail coefficients change nothing
    assert np.array_equal(bits(dsp.lfilter_iir(b, [1.0, 0.0, 0.0], x, exact=True)), bits(oracle.lfilter(b, [1.0], x)))


def test_growing_recurrence_runs_sequentially(dsp, oracle):
    """Poles outside the unit circle: the matrix powers of the block-parallel path overflow, so the call runs on
    the sequential kernel and still follows the reference (to rounding: fused multiply-adds)."""
    b, a = np.array([1.0]), np.array([1.0, -1.5])
    x = oracle.fill_splitmix(600, dsp.SEED, 0)
    ref = oracle.lfilter(b, a, x)
    assert np.isfinite(ref).all()
    assert scaled_err(dsp.lfilter_iir(b, a, x), ref) <= TOL
    assert np.array_equal(bits(dsp.lfilter_iir(b, a, x, exact=True)), bits(ref))


def test_many_channels_walk_their_tiles(dsp, oracle):
    """From 1024 channels on, one workgroup walks a whole channel and carries the state itself (one pass).
    Checked against the oracle on sampled channels and against the tile-parallel path (the same channels in
    launches of fewer than 1024) on all of them; with a short numerator (computed inside the tail kernels) and
    a long one (separate FIR pass), with and without incoming state."""
    import torch
    from scipy import signal

    nchan, n = 1100, 9001
    x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c * n) for c in range(nchan)])
    xd = torch.from_numpy(x).cuda()
    for b, a in (signal.butter(2, 0.1), signal.butter(5, 0.3), (signal.firwin(17, 0.3), np.array([1.0, -0.6, 0.2]))):
        bt = torch.from_numpy(np.ascontiguousarray(b)).cuda()
        y = dsp.lfilter_iir_device(bt, a, xd)
        for c in (0, 1, 555, nchan - 1):
            assert scaled_err(y[c].cpu().numpy(), oracle.lfilter(b, a, x[c])) <= TOL, (a.size, c)
        parts = torch.cat([dsp.lfilter_iir_device(bt, a, xd[lo:lo + 500].contiguous()) for lo in range(0, nchan, 500)])
        assert scaled_err(y.cpu().numpy(), parts.cpu().numpy()) <= TOL
        # second half given the first half's input and output history equals the whole
        cut = 4096 + 8
        second = dsp.lfilter_iir_device(bt, a, xd[:, cut:].contiguous(), xhist=xd[:, cut - (b.size - 1):cut].contiguous(),
                                        yhist=y[:, cut - (a.size - 1):cut].contiguous())
        assert scaled_err(second.cpu().numpy(), y[:, cut:].cpu().numpy()) <= TOL


def test_device_entry_with_state(dsp, oracle):
    """dspx_lfilter_f64_dev: a stream cut into two calls, the second given the input history and the previous
    outputs, equals one call."""
    import torch
    from scipy import signal

    b, a = signal.butter(3, 0.2)
    bt = torch.from_numpy(b).cuda()
    n, cut = 50000, 20480
    x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c << 20) for c in range(5)])
    xd = torch.from_numpy(x).cuda()
    for exact in (True, False):
        whole = dsp.lfilter_iir_device(bt, a, xd, exact=exact)
        ref = np.stack([oracle.lfilter(b, a, x[c]) for c in range(5)])
        if exact:
            assert np.array_equal(bits(whole.cpu().numpy()), bits(ref))
        else:
            assert scaled_err(whole.cpu().numpy(), ref) <= TOL
        first = dsp.lfilter_iir_device(bt, a, xd[:, :cut].contiguous(), exact=exact)
        second = dsp.lfilter_iir_device(bt, a, xd[:, cut:].contiguous(), exact=exact,
                                        xhist=xd[:, cut - (b.size - 1):cut].contiguous(),
                                        yhist=first[:, cut - (a.size - 1):].contiguous())
        joined = torch.cat([first, second], dim=1)
        if exact:
            assert torch.equal(joined, whole)
        else:
            assert scaled_err(joined.cpu().numpy(), ref) <= TOL
