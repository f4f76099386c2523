"""dsp::integrate_wavelet / dsp::cwt on the GPU FIR kernel (bank mode): host tables without a GPU,
the transform itself against the reference's test_cwt literals (dsp.hpp:2259-2783), the stored
reference outputs and the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import scaled_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-12


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def dsp():
    from dsp_amd import build

    build.build()
    import dsp_amd

    return dsp_amd


@pytest.fixture(scope="module")
def dropin(dsp):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)
    return os.path.join(ROOT, "tests", "cpp", "dropin_check")


# ---- host side, no GPU ----

def test_integrate_wavelet_bit_identical(dsp, golden, literals, oracle):
    psi, x = dsp.integrate_wavelet(10)
    assert np.array_equal(bits(psi), bits(golden["morlet10_psi"])) and np.array_equal(bits(x), bits(golden["morlet10_x"]))
    t = literals["integrate_wavelet"]
    p8, x8 = dsp.integrate_wavelet(t["precision"])
    assert np.max(np.abs(p8 - np.array(t["expected_psi"]))) < t["tol_psi"]   # dsp.hpp:2234-2237
    assert np.max(np.abs(x8 - np.array(t["expected_x"]))) < t["tol_x"]
    for prec, lo, hi in ((3, -8.0, 8.0), (6, -4.0, 2.5), (12, -8.0, 8.0)):
        a, b = dsp.integrate_wavelet(prec, lo, hi)
        c, d = oracle.integrate_wavelet(prec, lo, hi)
        assert np.array_equal(bits(a), bits(c)) and np.array_equal(bits(b), bits(d))


def test_integrate_wavelet_errors(dsp):
    L = dsp.lib()
    buf = np.empty(16)
    assert L.dspx_integrate_wavelet_morlet_f64(2, -8.0, 8.0, buf.ctypes.data, buf.ctypes.data) == -1
    assert b"percision must be greater than 2" in L.dspx_last_error()     # dsp.hpp:938
    assert L.dspx_integrate_wavelet_morlet_f64(4, 1.0, 1.0, buf.ctypes.data, buf.ctypes.data) == -1
    assert b"lower must be less than upper" in L.dspx_last_error()        # dsp.hpp:945


def test_cpp_integrate_wavelet(dropin, golden):
    r = subprocess.run([dropin, "wavelet", "10"], capture_output=True, text=True, check=True)
    toks = r.stdout.split()
    psi = np.array([float.fromhex(v) for v in toks[2:2 + 1024]])
    x = np.array([float.fromhex(v) for v in toks[2 + 1024 + 2:]])
    assert np.array_equal(bits(psi), bits(golden["morlet10_psi"])) and np.array_equal(bits(x), bits(golden["morlet10_x"]))


def test_scale_tables_and_plan(dsp, oracle, golden):
    wx = golden["morlet10_x"]
    scales = np.arange(1, 66, dtype=np.float64)
    dev, ntaps = dsp.cwt_plan(scales[::-1], 1024, wx, 8)           # any order in, sorted rows out
    assert ntaps == [16 * s + 1 for s in range(1, 66)] and sum(ntaps) == 34385
    step, rng = wx[1] - wx[0], wx[-1] - wx[0]
    for s in (0.7, 1.0, 2.5, 64.9):
        _, nt = dsp.cwt_plan([s], 1024, wx, 1)
        assert nt[0] == len(oracle.cwt_scale_indices(s, step, rng, 1024, int(np.ceil(s * rng)) + 1))
    load = [sum(n for d, n in zip(dev, ntaps) if d == g) for g in range(8)]
    assert sorted(set(dev)) == list(range(8)) and max(load) - min(load) <= max(ntaps)   # balanced by taps
    dev1, _ = dsp.cwt_plan(scales, 1024, wx, 1)
    assert set(dev1) == {0}


def test_cwt_argument_errors_without_gpu(dsp, golden):
    L = dsp.lib()
    psi, wx = golden["morlet10_psi"], golden["morlet10_x"]
    sig, out = np.ones(16), np.empty(64)

    def call(signal, n, scales, psi_, wx_):
        sc = np.asarray(scales, dtype=np.float64)
        return L.dspx_cwt_f64(signal.ctypes.data, n, sc.ctypes.data if sc.size else None, sc.size, psi_.ctypes.data,
                              psi_.size, wx_.ctypes.data, wx_.size, out.ctypes.data, 0)

    assert call(sig, 16, [1.0], psi[:2], wx[:2]) == -1 and b"wavelet_psi must be greater than 2" in L.dspx_last_error()
    assert call(sig, 16, [1.0], psi[:8], wx[:16]) == -1 and b"less than or equal to wavelet_psi" in L.dspx_last_error()
    assert call(sig, 16, [1.0], psi, np.zeros(1024)) == -1 and b"step of wavelet_x" in L.dspx_last_error()
    assert call(sig, 16, [], psi, wx) == -5 and b"ctx.scales must be greater than 0" in L.dspx_last_error()
    assert call(sig, 0, [1.0], psi, wx) == -1 and b"size of signal must be greater than 0" in L.dspx_last_error()
    assert call(sig, 16, [0.0625], psi, wx) == -5 and b"too small" in L.dspx_last_error()       # dsp.hpp:1195-1197
    assert call(sig, 16, [1e-9], psi, wx) == -5                                                  # single-tap table
    assert call(sig, 16, [1.0], psi, wx) in (-2, 0)    # valid arguments: only the missing GPU can stop it


# ---- GPU ----

@pytest.mark.gpu
def test_reference_test_cwt_literals(dsp, literals):
    t = literals["cwt"]
    psi, x = dsp.integrate_wavelet(t["precision"])
    r = dsp.cwt(t["data"], t["scales"], psi, x)
    assert r.shape == (30, 64)
    assert np.max(np.abs(r.ravel() - np.array(t["expected"]))) < t["tol"]    # dsp.hpp:2766-2771, 1e-14 abs
    r = dsp.cwt(t["data"], t["scales"], psi, x, exact=True)
    assert np.max(np.abs(r.ravel() - np.array(t["expected"]))) < t["tol"]


@pytest.mark.gpu
def test_cpp_cwt_matches_reference_literals(dropin, literals):
    t = literals["cwt"]
    inp = "\n".join(float(v).hex() for v in t["data"])
    r = subprocess.run([dropin, "cwt", "30"], input=inp, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    vals = np.array([float.fromhex(v) for v in r.stdout.split()[2:]])
    assert vals.size == 30 * 64 and np.max(np.abs(vals - np.array(t["expected"]))) < t["tol"]


@pytest.mark.gpu
def test_seeded_cwt_vs_reference_output(dsp, oracle, golden):
    sig = oracle.fill_splitmix(2048, dsp.SEED, 777)
    r = dsp.cwt(sig, golden["cwt_scales"], golden["morlet10_psi"], golden["morlet10_x"])
    for row, ref in zip(r, golden["cwt_seeded"]):
        assert scaled_err(row, ref) <= TOL
    r1 = dsp.cwt(sig, golden["cwt_scales"], golden["morlet10_psi"], golden["morlet10_x"], ngpus=1)
    assert np.array_equal(bits(r1), bits(r))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 7, 100, 1021, 1022, 1023, 1024, 2039, 2040, 2041, 2043, 2044, 2045, 2046, 2047, 2048,
                               4081, 4093, 4094, 4095, 8189, 10000])
def test_cwt_lengths_vs_oracle(dsp, oracle, golden, n):
    """Signals shorter than the kernels, and lengths around the strides of bank mode (2040 outputs per workgroup on the
    vector pipe; 1022 per wave on the matrix pipe, 2046 and 4094 per workgroup in its tiled form)."""
    psi, wx = golden["morlet10_psi"], golden["morlet10_x"]
    sig = oracle.fill_splitmix(n, dsp.SEED, n)
    scales = [0.7, 1.0, 3.3, 16.0, 65.0]
    ref = oracle.cwt(sig, scales, psi, wx)
    got = dsp.cwt(sig, scales[::-1], psi, wx)          # unsorted in, rows by ascending scale out
    assert got.shape == ref.shape and np.all(np.isfinite(got))
    for a, b in zip(got, ref):
        assert scaled_err(a, b) <= TOL, n
    gx = dsp.cwt(sig, scales, psi, wx, exact=True)
    for a, b in zip(gx, ref):
        assert scaled_err(a, b) <= TOL
    gt = dsp.cwt(sig, scales, psi, wx, tiled=True)      # the workgroup-tiled form of the matrix-pipe kernel
    for a, b in zip(gt, ref):
        assert scaled_err(a, b) <= TOL


@pytest.mark.gpu
def test_device_bank_65_scales(dsp, oracle, golden):
    """The benchmark shape in small: 65 integer scales, device-resident signal and result."""
    import torch

    psi, wx = golden["morlet10_psi"], golden["morlet10_x"]
    n = 1 << 16
    scales = np.arange(1, 66, dtype=np.float64)
    bank = dsp.CwtBank(scales, psi, wx)
    assert bank.rows == 65 and bank.total_taps == 34385
    sig = torch.empty(n, dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(sig, dsp.SEED, 5)
    out = bank.apply(sig)
    host = dsp.cwt(sig.cpu().numpy(), scales, psi, wx)
    assert np.array_equal(bits(out.cpu().numpy()), bits(host))
    xs = oracle.fill_splitmix(n, dsp.SEED, 5)
    for s in (1, 17, 65):       # full rows against the oracle
        ref = oracle.cwt(xs, [float(s)], psi, wx)[0]
        assert scaled_err(out[s - 1].cpu().numpy(), ref) <= TOL
    # padded rows (row stride > n) and an 8-byte-aligned signal take the scalar-access variant
    big = torch.full((65, n + 6), -3.0, dtype=torch.float64, device="cuda")
    bank.apply(sig, out=big[:, :n])
    assert torch.equal(big[:, :n], out) and bool((big[:, n:] == -3.0).all())
    sig2 = torch.empty(n + 1, dtype=torch.float64, device="cuda")
    sig2[1:] = sig
    assert torch.equal(bank.apply(sig2[1:]), out)
    bank.close()


@pytest.mark.gpu
def test_cwt_host_path_in_several_passes(dsp, oracle, golden, monkeypatch):
    """When the rows of a transform do not fit in device memory together they are transformed in passes;
    DSPX_CWT_ROWS_PER_PASS forces that on a small case (here 7 scales, 3 rows per pass): same bits as one pass."""
    psi, wx = golden["morlet10_psi"], golden["morlet10_x"]
    sig = oracle.fill_splitmix(6000, dsp.SEED, 3)
    scales = [1.0, 2.5, 7.0, 12.0, 20.0, 33.0, 64.0]
    one = dsp.cwt(sig, scales, psi, wx)
    monkeypatch.setenv("DSPX_CWT_ROWS_PER_PASS", "3")
    several = dsp.cwt(sig, scales, psi, wx)
    assert np.array_equal(one.view(np.uint64), several.view(np.uint64))
    ref = oracle.cwt(sig, scales, psi, wx)
    for a, b in zip(several, ref):
        assert scaled_err(a, b) <= TOL
