"""Pins oracle/fir_oracle.c: (1) against the literals of the reference's own
known-answer tests, at the reference's tolerances; (2) bit-for-bit against outputs
of the reference header compiled in the build container (tests/golden/ref_outputs.npz);
(3) when oracle/_ref is present, bit-for-bit on fresh random cases."""
import numpy as np
import pytest

from oracle.pyoracle import SEED


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_sinc_literals(oracle, literals):
    t = literals["sinc"]
    for x, e in zip(t["x"], t["expected"]):
        assert abs(oracle.sinc(x) - e) < t["tol"]  # dsp.hpp:1415-1419


def test_hamming_literals(oracle, literals):
    t = literals["hamming_window"]
    w = oracle.hamming_window(len(t["expected"]))
    assert np.max(np.abs(w - np.array(t["expected"]))) < t["tol"]  # dsp.hpp:1458-1463


def test_firwin_literals(oracle, literals):
    t = literals["firwin"]
    h = oracle.firwin(t["numtaps"], t["cutoff"])
    assert np.max(np.abs(h - np.array(t["expected"]))) < t["tol"]  # dsp.hpp:1506-1511


def test_lfilter_literals(oracle, literals):
    t = literals["lfilter"]
    w = oracle.firwin(t["numtaps"], t["cutoff"])
    y = oracle.lfilter(w, [1.0], t["data"])
    assert np.max(np.abs(y - np.array(t["expected"]))) < t["tol"]  # dsp.hpp:1583-1588
    y32 = oracle.lfilter_f32(w, t["data"])
    assert np.max(np.abs(y32.astype(np.float64) - np.array(t["expected"]))) < t["tol_f32"]
    # what the reference's test really instantiates: lfilter<float> over its double taps and data (dsp.hpp:1590-1597)
    y32 = oracle.lfilter_f32_from_f64(w, t["data"])
    assert np.max(np.abs(y32.astype(np.float64) - np.array(t["expected"]))) < t["tol_f32"]


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
def test_firwin_bit_exact_vs_reference_outputs(oracle, golden, key, args):
    h = oracle.firwin(*args)
    assert np.array_equal(bits(h), bits(golden[key]))


def test_hamming_periodic_bit_exact(oracle, golden):
    assert np.array_equal(bits(oracle.hamming_window(16, 0.54, False)), bits(golden["hamming_periodic_16"]))


def test_firwin_error_cases(oracle):
    with pytest.raises(ValueError):
        oracle.firwin(100, [])  # dsp.hpp:200-202
    with pytest.raises(ValueError):
        oracle.firwin(100, [0.5], pass_zero=False)  # even numtaps with pass_nyquist, dsp.hpp:209-211
    with pytest.raises(ValueError):
        oracle.firwin(1, [0.5])  # hamming_window(1) throws, dsp.hpp:76-78


def test_readme_bit_exact(oracle, golden):
    y = oracle.lfilter(golden["taps_100_0p5"], [1.0], golden["readme_x"])
    assert np.array_equal(bits(y), bits(golden["readme_y"]))
    assert y[0] == 0.0 and abs(y[1023] - (-0.38376489195896102)) < 1e-16


@pytest.mark.parametrize("key", ["taps_101_0p25", "taps_255_0p25", "taps_4095_0p1"])
def test_seeded_windows_bit_exact(oracle, golden, key):
    x = oracle.fill_splitmix(1 << 14, SEED, 0)
    y = oracle.lfilter(golden[key], [1.0], x)
    assert np.array_equal(bits(y), bits(golden["seeded_y_" + key]))


def test_ragged_short_and_iir_bit_exact(oracle, golden):
    x = oracle.fill_splitmix(1000, SEED, 12345)
    assert np.array_equal(bits(oracle.lfilter(golden["taps_255_0p25"], [1.0], x[:777])), bits(golden["ragged_y_255"]))
    assert np.array_equal(bits(oracle.lfilter(golden["taps_4095_0p1"], [1.0], x)), bits(golden["short_y_4095"]))
    assert np.array_equal(bits(oracle.lfilter(golden["taps_100_0p5"][:7], golden["iir_a"], x)), bits(golden["iir_y"]))
    x = oracle.fill_splitmix(9000, SEED, 777)
    for k in ("iir4", "iir2"):
        assert np.array_equal(bits(oracle.lfilter(golden[k + "_b"], golden[k + "_a"], x)), bits(golden[k + "_y"]))


def test_edge_cases(oracle):
    b = oracle.firwin(9, [0.3])
    assert oracle.lfilter(b, [1.0], []).size == 0          # empty x -> empty y
    assert np.all(oracle.lfilter([], [1.0], [1.0, 2.0]) == 0.0)   # empty numerators -> zeros
    x = oracle.fill_splitmix(50)
    # a[0] is never read (dsp.hpp:366): any scalar denominator gives the same bits
    assert np.array_equal(bits(oracle.lfilter(b, [2.0], x)), bits(oracle.lfilter(b, [1.0], x)))


def test_hist_variant_matches_split_stream(oracle):
    """Filtering [x0 | x1] in one go == filtering x1 with the tail of x0 as history."""
    b = oracle.firwin(31, [0.2])
    x = oracle.fill_splitmix(500, SEED, 99)
    full = oracle.lfilter(b, [1.0], x)
    cut = 200
    tail = oracle.fir_hist(b, x[cut - 30:cut], x[cut:])
    assert np.array_equal(bits(tail), bits(full[cut:]))
    assert np.array_equal(bits(oracle.fir_hist(b, None, x)), bits(full))


def test_splitmix_range_and_determinism(oracle):
    x = oracle.fill_splitmix(4096, SEED, 0)
    assert x.min() >= -1.0 and x.max() < 1.0 and abs(x.mean()) < 0.05
    assert np.array_equal(oracle.fill_splitmix(100, SEED, 1000), oracle.fill_splitmix(1100, SEED, 0)[1000:])
    # pure-Python restatement of the generator
    M = (1 << 64) - 1
    for i in (0, 1, 4095):
        z = (SEED + (i + 1) * 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        z ^= z >> 31
        assert x[i] == (z >> 11) * 2.0 ** -52 - 1.0


def test_checksum_is_partition_independent(oracle):
    y = oracle.fill_splitmix(1000, 7, 0)
    whole = oracle.checksum(y, 0)
    parts = (oracle.checksum(y[:333], 0) + oracle.checksum(y[333:], 333)) & (2**64 - 1)
    assert whole == parts


# ---- live comparison with the reference header (build container only) ----

def test_reference_own_tests_pass(reference):
    assert reference.run_hot_path_tests() == 0


def test_fuzz_bit_exact_vs_reference(oracle, reference):
    rng = np.random.default_rng(1)
    for _ in range(40):
        nt = int(rng.integers(2, 300))
        n = int(rng.integers(0, 900))
        c = float(rng.uniform(0.02, 0.98))
        h_ref = reference.firwin(nt, [c])
        assert np.array_equal(bits(oracle.firwin(nt, [c])), bits(h_ref))
        x = rng.standard_normal(n)
        assert np.array_equal(bits(oracle.lfilter(h_ref, [1.0], x)), bits(reference.lfilter_scalar_den(h_ref, 1.0, x)))
    for nt in (3, 5, 21, 77):
        cs = sorted(rng.uniform(0.05, 0.95, size=2).tolist())
        assert np.array_equal(bits(oracle.firwin(nt, cs, pass_zero=False)), bits(reference.firwin(nt, cs, pass_zero=False)))
    for w in (2, 3, 64, 255):
        for sym in (True, False):
            assert np.array_equal(bits(oracle.hamming_window(w, 0.54, sym)), bits(reference.hamming_window(w, 0.54, sym)))
    xs = rng.uniform(-12, 12, 200).tolist() + [0.0, 1e-21, -1e-21, 1e-20]
    for x in xs:
        assert oracle.sinc(x) == reference.sinc(x)
    b32 = rng.standard_normal(40).astype(np.float32)
    x32 = rng.standard_normal(333).astype(np.float32)
    assert np.array_equal(oracle.lfilter_f32(b32, x32).view(np.uint32), reference.lfilter_f32(b32, x32).view(np.uint32))
    # lfilter<float> over double containers (the form the reference's own test uses): double products, float sums
    b64, x64 = rng.standard_normal(40), rng.standard_normal(333)
    assert np.array_equal(oracle.lfilter_f32_from_f64(b64, x64).view(np.uint32),
                          reference.lfilter_f32_from_f64(b64, x64).view(np.uint32))


# ---- narrow element types: lfilter<double> over float containers, lfilter<float> with a recursive part ----

def _narrow_inputs(oracle, golden):
    from oracle.pyoracle import SEED

    x64 = oracle.fill_splitmix(3000, SEED, 4242)
    return x64.astype(np.float32), x64


def test_narrow_types_bit_exact_vs_reference_outputs(oracle, golden):
    """Stored outputs of the reference's lfilter<double>(vector<float>...) and lfilter<float>(b, a, x) (dsp.hpp:356-369):
    products in the type the containers' elements promote to, sums and the tail in the types the reference uses."""
    x32, x64 = _narrow_inputs(oracle, golden)
    got = oracle.lfilter_f64_from_f32(golden["taps_101_0p25"].astype(np.float32), x32)
    assert np.array_equal(bits(got), bits(golden["f64_from_f32_y"]))
    got = oracle.lfilter_f32_iir(golden["iir4_b"].astype(np.float32), golden["iir4_a"].astype(np.float32), x32)
    assert np.array_equal(got.view(np.uint32), golden["f32_iir_y"].view(np.uint32))
    got = oracle.lfilter_f32_from_f64_iir(golden["iir4_b"], golden["iir4_a"], x64)
    assert np.array_equal(got.view(np.uint32), golden["f32_from_f64_iir_y"].view(np.uint32))
    # the float products really differ from double ones: the same data through lfilter<double> over doubles is elsewhere
    wide = oracle.lfilter(golden["taps_101_0p25"].astype(np.float32).astype(np.float64), [1.0], x32.astype(np.float64))
    assert not np.array_equal(bits(wide), bits(golden["f64_from_f32_y"]))


def test_narrow_types_fuzz_vs_reference(oracle, reference):
    rng = np.random.default_rng(5)
    for _ in range(30):
        nb, na, n = int(rng.integers(1, 40)), int(rng.integers(1, 7)), int(rng.integers(1, 600))
        b, x = rng.uniform(-1, 1, nb), rng.uniform(-1, 1, n)
        a = np.concatenate([[rng.uniform(0.5, 2)], rng.uniform(-0.3, 0.3, na - 1)])
        b32, a32, x32 = b.astype(np.float32), a.astype(np.float32), x.astype(np.float32)
        assert np.array_equal(bits(oracle.lfilter_f64_from_f32(b32, x32)), bits(reference.lfilter_f64_from_f32(b32, x32)))
        assert np.array_equal(oracle.lfilter_f32_iir(b32, a32, x32).view(np.uint32),
                              reference.lfilter_f32_iir(b32, a32, x32).view(np.uint32))
        assert np.array_equal(oracle.lfilter_f32_from_f64_iir(b, a, x).view(np.uint32),
                              reference.lfilter_f32_from_f64_iir(b, a, x).view(np.uint32))
