"""Pins oracle/cwt_oracle.c against the reference's test_integrate_wavelet / test_cwt
literals (dsp.hpp:2096-2783), the stored reference outputs, and the live reference."""
import numpy as np
import pytest

from oracle.pyoracle import SEED


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_integrate_wavelet_literals(oracle, literals):
    t = literals["integrate_wavelet"]
    psi, x = oracle.integrate_wavelet(t["precision"])
    assert np.max(np.abs(psi - np.array(t["expected_psi"]))) < t["tol_psi"]  # dsp.hpp:2234-2237
    assert np.max(np.abs(x - np.array(t["expected_x"]))) < t["tol_x"]        # dsp.hpp:2238-2241


def test_cwt_literals(oracle, literals):
    t = literals["cwt"]
    psi, x = oracle.integrate_wavelet(t["precision"])
    r = oracle.cwt(t["data"], t["scales"], psi, x)
    assert r.shape == (30, 64)
    assert np.max(np.abs(r.ravel() - np.array(t["expected"]))) < t["tol"]    # dsp.hpp:2766-2771


def test_morlet_and_seeded_cwt_bit_exact(oracle, golden):
    psi, x = oracle.integrate_wavelet(10)
    assert np.array_equal(bits(psi), bits(golden["morlet10_psi"]))
    assert np.array_equal(bits(x), bits(golden["morlet10_x"]))
    sig = oracle.fill_splitmix(2048, SEED, 777)
    r = oracle.cwt(sig, golden["cwt_scales"], psi, x)
    assert np.array_equal(bits(r), bits(golden["cwt_seeded"]))


def test_cwt_kernel_lengths(oracle, golden):
    """With the 1024-point Morlet table the kernel of integer scale s has 16 s + 1 taps."""
    wx = golden["morlet10_x"]
    step, rng = wx[1] - wx[0], wx[-1] - wx[0]
    total = 0
    for s in range(1, 66):
        n = len(oracle.cwt_scale_indices(float(s), step, rng, 1024, int(np.ceil(65 * rng)) + 1))
        assert n == 16 * s + 1
        total += n
    assert total == 34385


def test_cwt_errors(oracle, golden):
    psi, x = golden["morlet10_psi"], golden["morlet10_x"]
    with pytest.raises(ValueError):
        oracle.cwt(np.ones(16), [0.0625], psi, x)  # kernel of 2 taps: "scale too small" (dsp.hpp:1195-1197)
    with pytest.raises(ValueError):
        oracle.cwt(np.ones(16), [0.05], psi, x)  # kernel of 1 tap: the reference asserts (dsp.hpp:1193)
    with pytest.raises(ValueError):
        oracle.cwt(np.ones(16), [1.0], psi[:2], x[:2])  # table too short (dsp.hpp:1047-1049)


def test_cwt_fuzz_vs_reference(oracle, reference):
    rng = np.random.default_rng(3)
    psi, x = reference.integrate_wavelet(10)
    for _ in range(6):
        n = int(rng.integers(5, 400))
        sc = rng.uniform(0.5, 12.0, size=int(rng.integers(1, 6)))
        sig = rng.standard_normal(n)
        assert np.array_equal(bits(oracle.cwt(sig, sc, psi, x)), bits(reference.cwt(sig, sc, psi, x)))
    p8, x8 = reference.integrate_wavelet(8)
    o8, ox8 = oracle.integrate_wavelet(8)
    assert np.array_equal(bits(p8), bits(o8)) and np.array_equal(bits(x8), bits(ox8))
