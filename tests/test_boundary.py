"""The drop-in boundary without a GPU: libdspx.so loads, exports every symbol that
include/dspx.h declares, host tap design is bit-identical to the reference, and argument
errors surface the way the reference's do."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def dsp():
    from dsp_amd import build

    build.build()  # no-op when the in-tree library is current
    import dsp_amd

    return dsp_amd


@pytest.fixture(scope="module")
def dropin(dsp):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)
    return os.path.join(ROOT, "tests", "cpp", "dropin_check")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dspx.h")).read()
    return sorted(set(re.findall(r"DSPX_API\s+[\w\s\*]+?\b(dspx_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(dsp):
    names = declared_symbols()
    assert len(names) >= 12
    L = C.CDLL(dsp._lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dspx.h but not exported"
    assert sorted(dsp._lib.SIGNATURES) == names  # the ctypes table covers the whole header
    assert dsp.lib().dspx_version() >= 100
    assert dsp.lib().dspx_status_string(-1) == b"invalid argument"


def test_no_silent_cpu_path(dsp):
    """Without a device the filter entry points fail loudly; nothing is computed on the CPU."""
    if dsp.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(dsp.DspxError) as e:
        dsp.lfilter(np.ones(3), np.ones(10))
    assert e.value.code == -2


def test_c_abi_header_is_plain_c():
    """include/dspx.h is the boundary a C, cgo or JNI caller binds: it must compile as C11 and as C++17 on its own."""
    hdr = os.path.join(ROOT, "include", "dspx.h")
    for cmd in (["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr],
                ["g++", "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr


def test_pipe_thresholds_mirror_the_library(dsp):
    """The constants the benchmark labels its kernels by are the ones the library selects pipes with."""
    src = open(os.path.join(ROOT, "dsp_amd", "csrc", "dspx_common.h")).read()
    lo = int(re.search(r"kMatrixPipeMinTaps\s*=\s*(\d+)", src).group(1))
    assert dsp.MATRIX_PIPE_MIN_TAPS == lo


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dsp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp")):
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), f
    assert "oracle" not in open(os.path.join(ROOT, "include", "dsp.hpp")).read().lower()


@pytest.mark.parametrize("key,args", [
    ("taps_100_0p5", (100, [0.5], True, True)),
    ("taps_101_0p25", (101, [0.25], True, True)),
    ("taps_255_0p25", (255, [0.25], True, True)),
    ("taps_4095_0p1", (4095, [0.1], True, True)),
    ("taps_bp_65", (65, [0.2, 0.6], False, True)),
    ("taps_hp_33", (33, [0.3], False, True)),
    ("taps_noscale_40", (40, [0.4], True, False)),
    ("taps_multiband_51", (51, [0.1, 0.3, 0.5, 0.7], True, True)),
])
def test_firwin_bit_identical_to_reference(dsp, golden, key, args):
    assert np.array_equal(bits(dsp.firwin(*args)), bits(golden[key]))


def test_design_literals_of_the_reference_tests(dsp, literals):
    L = dsp.lib()
    t = literals["sinc"]
    for x, e in zip(t["x"], t["expected"]):
        assert abs(L.dspx_sinc_f64(x) - e) < t["tol"]
    t = literals["hamming_window"]
    w = np.empty(100)
    assert L.dspx_hamming_window_f64(100, 0.54, 1, w.ctypes.data) == 0
    assert np.max(np.abs(w - np.array(t["expected"]))) < t["tol"]
    t = literals["firwin"]
    assert np.max(np.abs(dsp.firwin(100, 0.5) - np.array(t["expected"]))) < t["tol"]


def test_design_fuzz_vs_oracle(dsp, oracle):
    rng = np.random.default_rng(11)
    for _ in range(50):
        nt = int(rng.integers(2, 400))
        c = float(rng.uniform(0.01, 0.99))
        assert np.array_equal(bits(dsp.firwin(nt, c)), bits(oracle.firwin(nt, [c])))
    w = np.empty(17)
    assert dsp.lib().dspx_hamming_window_f64(17, 0.54, 0, w.ctypes.data) == 0
    assert np.array_equal(bits(w), bits(oracle.hamming_window(17, 0.54, False)))


def test_design_errors_match_reference_messages(dsp):
    L = dsp.lib()
    out = np.empty(128)
    c = np.array([0.5])
    assert L.dspx_firwin_f64(100, c.ctypes.data, 0, 1, 1, out.ctypes.data) == -1
    assert b"At least one cutoffs frequency" in L.dspx_last_error()          # dsp.hpp:202
    assert L.dspx_firwin_f64(100, c.ctypes.data, 1, 0, 1, out.ctypes.data) == -1
    assert b"must be odd when pass_nyquist" in L.dspx_last_error()           # dsp.hpp:211
    assert L.dspx_firwin_f64(1, c.ctypes.data, 1, 1, 1, out.ctypes.data) == -1
    assert b"width must be greater than or equal to 2" in L.dspx_last_error()  # dsp.hpp:78


def test_cpp_header_design_and_errors(dropin, golden):
    r = subprocess.run([dropin, "api", "nogpu"], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([dropin, "design", "255", "0.25"], capture_output=True, text=True, check=True)
    vals = np.array([float.fromhex(v) for v in r.stdout.split()[2:]])
    assert np.array_equal(bits(vals), bits(golden["taps_255_0p25"]))


# ---- with a GPU: the C++ header end to end ----

@pytest.mark.gpu
def test_cpp_readme_example(dropin, golden):
    r = subprocess.run([dropin, "readme"], capture_output=True, text=True, check=True)
    toks = r.stdout.split()
    out, i = {}, 0
    while i < len(toks):
        name, cnt = toks[i], int(toks[i + 1])
        out[name] = np.array([float.fromhex(v) for v in toks[i + 2:i + 2 + cnt]])
        i += 2 + cnt
    assert np.array_equal(bits(out["taps"]), bits(golden["taps_100_0p5"]))
    assert np.array_equal(bits(out["x"]), bits(golden["readme_x"]))  # same libm on this machine
    den = np.max(np.abs(golden["readme_y"]))
    assert np.max(np.abs(out["y"] - golden["readme_y"])) / den <= 1e-12
    assert out["y"][0] == 0.0


@pytest.mark.gpu
def test_cpp_reference_test_lfilter(dropin, literals):
    t = literals["lfilter"]
    inp = "\n".join(float(v).hex() for v in t["data"])
    r = subprocess.run([dropin, "filter", "100", "0.5"], input=inp, capture_output=True, text=True, check=True)
    toks = r.stdout.split()
    assert toks[0] == "y" and toks[1] == "100" and toks[102] == "yf" and toks[103] == "100"
    y = np.array([float.fromhex(v) for v in toks[2:102]])
    assert np.max(np.abs(y - np.array(t["expected"]))) < t["tol"]  # dsp.hpp:1583-1588
    yf = np.array([float.fromhex(v) for v in toks[104:204]])       # lfilter<float>, dsp.hpp:1590-1597
    assert np.max(np.abs(yf - np.array(t["expected"]))) < t["tol_f32"]


@pytest.mark.gpu
def test_cpp_api_behaviour(dropin):
    r = subprocess.run([dropin, "api"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
