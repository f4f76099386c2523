"""Seeded differential fuzzing of the device entry points against the oracle: random channel counts,
lengths around tile and stage boundaries, tap counts around every dispatch threshold, padded and
misaligned strides, optional history, every pipe flag; and the same for the recursive tail."""
import numpy as np
import pytest

from conftest import cwt_row_tolerance, scaled_err

pytestmark = pytest.mark.gpu
TOL = 1e-12


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def env():
    import torch

    import dsp_amd

    assert torch.cuda.is_available() and dsp_amd.device_count() >= 1
    return torch, dsp_amd


def strided(torch, rng, rows, cols, fill=None):
    """A [rows, cols] view into a larger buffer with a random row stride and a random (possibly odd, i.e.
    8-byte-only aligned) start offset; returns (view, host copy)."""
    stride = cols + int(rng.integers(0, 6))
    off = int(rng.integers(0, 4))
    buf = torch.full((off + rows * stride + 8,), 123.0, dtype=torch.float64, device="cuda")
    view = buf[off: off + rows * stride].view(rows, stride)[:, :cols]
    if fill is not None:
        view.copy_(torch.from_numpy(fill))
    return view, buf


def test_fir_device_entry_fuzz(env, oracle):
    torch, dsp = env
    rng = np.random.default_rng(20251111)
    ns = [1, 2, 3, 7, 63, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2040, 2046, 2047, 2048, 2049, 4094, 4095, 4096,
          4097, 6000, 10000, 40001]
    taps_n = [1, 2, 3, 15, 16, 17, 31, 51, 52, 55, 56, 77, 78, 79, 95, 96, 97, 160, 189, 190, 223, 224, 225, 255, 256, 257, 271, 272, 273, 300,
              313, 314, 441, 442, 515, 569, 570, 640, 1100, 2300]
    for case in range(240):
        nchan = int(rng.integers(1, 6))
        n = int(rng.choice(ns))
        ntaps = int(rng.choice(taps_n))
        b = rng.uniform(-1, 1, ntaps)
        x = rng.uniform(-1, 1, (nchan, n))
        use_hist = bool(rng.integers(0, 2)) and ntaps > 1
        hist = rng.uniform(-1, 1, (nchan, ntaps - 1)) if use_hist else None
        exact, pipe = [(True, None), (False, None), (False, True), (False, False)][int(rng.integers(0, 4))]
        xd, _xbuf = strided(torch, rng, nchan, n, x)
        yd, ybuf = strided(torch, rng, nchan, n)
        hd = None
        if use_hist:
            hd, _hbuf = strided(torch, rng, nchan, ntaps - 1, hist)
        taps = torch.from_numpy(b).cuda()
        dsp.fir_device(taps, xd, out=yd, hist=hd, exact=exact, mfma=pipe)
        got = yd.cpu().numpy()
        ref = np.stack([oracle.fir_hist(b, hist[c] if use_hist else None, x[c]) for c in range(nchan)])
        tag = (case, nchan, n, ntaps, use_hist, exact, pipe)
        if exact:
            assert np.array_equal(bits(got), bits(ref)), tag
        else:
            assert scaled_err(got, ref) <= TOL, tag
        # nothing outside the output view was written
        mask = torch.ones_like(ybuf, dtype=torch.bool)
        base = (yd.data_ptr() - ybuf.data_ptr()) // 8
        for c in range(nchan):
            mask[base + c * yd.stride(0): base + c * yd.stride(0) + n] = False
        assert bool((ybuf[mask] == 123.0).all()), tag


def test_recursive_device_entry_fuzz(env, oracle):
    torch, dsp = env
    rng = np.random.default_rng(7)
    ns = [1, 2, 15, 16, 17, 100, 4095, 4096, 4097, 8191, 8193, 12000]
    for case in range(80):
        nchan = int(rng.choice([1, 2, 5, 70]))
        n = int(rng.choice(ns))
        order = int(rng.integers(1, 9))
        # stable denominators: poles drawn inside the unit circle (conjugate pairs, plus a real one if odd)
        poles = []
        for _ in range(order // 2):
            r, th = rng.uniform(0.2, 0.95), rng.uniform(0.1, 3.0)
            poles += [r * np.exp(1j * th), r * np.exp(-1j * th)]
        if order % 2:
            poles.append(rng.uniform(-0.9, 0.9))
        a = np.real(np.poly(poles))
        a[0] = rng.uniform(0.5, 3.0)                      # never read by the reference
        nb = int(rng.choice([1, 2, 5, 16, 17, 40, 130]))
        b = rng.uniform(-1, 1, nb)
        x = rng.uniform(-1, 1, (nchan, n))
        with_state = bool(rng.integers(0, 2))
        exact = bool(rng.integers(0, 2))
        xh = rng.uniform(-1, 1, (nchan, nb - 1)) if with_state and nb > 1 else None
        yh = rng.uniform(-1, 1, (nchan, order)) if with_state else None
        # the oracle has no state arguments: prepend the history and filter from a state that reproduces yh
        ref = np.empty_like(x)
        for c in range(nchan):
            w = oracle.fir_hist(b, xh[c] if xh is not None else None, x[c])
            y = np.empty(n)
            prev = list(yh[c]) if yh is not None else []    # oldest first
            for i in range(n):
                acc = w[i]
                for j in range(1, order + 1):
                    if i - j >= 0:
                        acc = acc - a[j] * y[i - j]
                    elif yh is not None:
                        acc = acc - a[j] * prev[order + (i - j)]
                y[i] = acc
            ref[c] = y
        xd, _ = strided(torch, rng, nchan, n, x)
        yd, _ = strided(torch, rng, nchan, n)
        xhd = strided(torch, rng, nchan, nb - 1, xh)[0] if xh is not None else None
        yhd = strided(torch, rng, nchan, order, yh)[0] if yh is not None else None
        dsp.lfilter_iir_device(torch.from_numpy(b).cuda(), a, xd, out=yd, xhist=xhd, yhist=yhd, exact=exact)
        got = yd.cpu().numpy()
        tag = (case, nchan, n, order, nb, with_state, exact)
        if exact:
            assert np.array_equal(bits(got), bits(ref)), tag
        else:
            assert scaled_err(got, ref) <= 1e-11, tag    # random pole sets: the loop above is itself only ~1e-13 accurate


def test_wavelet_bank_fuzz(env, oracle, golden):
    """Random scale sets (fractional, unsorted, repeated lengths) and signal lengths: every row against the oracle."""
    torch, dsp = env
    rng = np.random.default_rng(99)
    psi, wx = golden["morlet10_psi"], golden["morlet10_x"]
    for case in range(40):
        n = int(rng.choice([2, 3, 17, 500, 2046, 4094, 4095, 7000, 12289]))
        scales = np.round(rng.uniform(0.3, 70.0, int(rng.integers(1, 9))), int(rng.integers(0, 3)))
        scales = scales[scales >= 0.25]
        if scales.size == 0:
            continue
        sig = rng.uniform(-1, 1, n)
        try:
            ref = oracle.cwt(sig, np.sort(scales), psi, wx)
        except (ValueError, RuntimeError):
            continue                                        # a scale the reference rejects for this length
        got = dsp.cwt(sig, scales, psi, wx, mfma=[None, True, False][case % 3])
        assert got.shape == ref.shape
        for r in range(ref.shape[0]):
            # (rows with NaN padding, where the reference has fewer values than samples, compare where both are finite)
            ok = np.isfinite(ref[r])
            assert np.array_equal(ok, np.isfinite(got[r])), (case, n, np.sort(scales)[r])
            tol = cwt_row_tolerance(oracle, sig, np.sort(scales)[r], psi, wx)
            assert scaled_err(got[r][ok], ref[r][ok]) <= tol, (case, n, np.sort(scales)[r], tol)
