#!/usr/bin/env python3
"""Generate tests/golden/ from the reference.  Runs ONLY in the build container
(needs /root/reference and oracle/_ref/libdsp_ref.so); the outputs are committed.

Two kinds of fixture are written:

  ref_test_literals.json   the numeric literals (inputs and expected outputs) that
                           the reference's own embedded known-answer tests hold
                           (dsp.hpp:1311-1599 sinc / hamming / firwin / lfilter,
                           dsp.hpp:2096-2783 integrate_wavelet / cwt).  Data only.
  ref_outputs.npz          outputs of the reference header itself, compiled here with
                           g++ -std=c++20 -O2 -ffp-contract=off, for the benchmark
                           tap designs and for seeded inputs (raw float64, bit exact).
"""
from __future__ import annotations

import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.pyoracle import SEED, Oracle, Reference  # noqa: E402

HEADER = "/root/reference/dsp/include/dsp.hpp"
NUM = r"[-+]?\d+\.\d+e[-+]\d+"


def _function_body(text: str, name: str) -> str:
    start = text.index(f"void {name}() {{")
    nxt = text.find("\nvoid test_", start + 1)
    return text[start: nxt if nxt > 0 else len(text)]


def _vector(body: str, name: str) -> list[float]:
    m = re.search(r"std::vector<double>\s+" + name + r"\s*=\s*\{(.*?)\};", body, re.S)
    return [float(v) for v in re.findall(NUM, m.group(1))]


def literals() -> dict:
    text = open(HEADER).read()
    out: dict = {}
    b = _function_body(text, "test_sinc")
    m = re.search(r"sinc_values\s*=\s*\{(.*?)\n\s*\};", b, re.S)
    pairs = re.findall(r"\{\s*(" + NUM + r"),\s*(" + NUM + r")\}", m.group(1))
    out["sinc"] = {"x": [float(p[0]) for p in pairs], "expected": [float(p[1]) for p in pairs], "tol": 1e-16}
    b = _function_body(text, "test_hamming_window")
    out["hamming_window"] = {"expected": _vector(b, "expected"), "tol": 1e-14}
    b = _function_body(text, "test_firwin")
    out["firwin"] = {"numtaps": 100, "cutoff": 0.5, "expected": _vector(b, "expected"), "tol": 1e-17}
    b = _function_body(text, "test_lfilter")
    out["lfilter"] = {"numtaps": 100, "cutoff": 0.5, "data": _vector(b, "data"), "expected": _vector(b, "expected"),
                      "tol": 1e-15, "tol_f32": 1e-6}
    b = _function_body(text, "test_integrate_wavelet")
    out["integrate_wavelet"] = {"precision": 8, "expected_psi": _vector(b, "expected_psi"),
                                "expected_x": _vector(b, "expected_x"), "tol_psi": 1e-15, "tol_x": 1e-14}
    b = _function_body(text, "test_cwt")
    out["cwt"] = {"precision": 10, "scales": list(range(1, 31)), "data": _vector(b, "data"),
                  "expected": _vector(b, "expected"), "tol": 1e-14}
    assert len(out["sinc"]["x"]) == 100 and len(out["hamming_window"]["expected"]) == 100
    assert len(out["firwin"]["expected"]) == 100
    assert len(out["lfilter"]["data"]) == 100 and len(out["lfilter"]["expected"]) == 100
    assert len(out["integrate_wavelet"]["expected_psi"]) == 256
    assert len(out["cwt"]["data"]) == 64 and len(out["cwt"]["expected"]) == 30 * 64
    return out


def outputs() -> dict:
    ref, orc = Reference(), Oracle()
    assert ref.run_hot_path_tests() == 0
    z: dict = {}
    designs = {"taps_100_0p5": (100, 0.5), "taps_101_0p25": (101, 0.25), "taps_255_0p25": (255, 0.25),
               "taps_4095_0p1": (4095, 0.1)}
    for k, (n, c) in designs.items():
        z[k] = ref.firwin_scalar(n, c)
    # other firwin argument shapes: band-pass / high-pass / unscaled / multi-band quirk
    z["taps_bp_65"] = ref.firwin(65, [0.2, 0.6], pass_zero=False)
    z["taps_hp_33"] = ref.firwin(33, [0.3], pass_zero=False)
    z["taps_noscale_40"] = ref.firwin(40, [0.4], scale=False)
    z["taps_multiband_51"] = ref.firwin(51, [0.1, 0.3, 0.5, 0.7])
    z["hamming_periodic_16"] = ref.hamming_window(16, 0.54, False)
    # README example: x[n] = sin(n), n < 1024 (README.md:13-27); store x too (libm may differ elsewhere)
    x = np.sin(np.arange(1024, dtype=np.float64))
    z["readme_x"] = x
    z["readme_y"] = ref.lfilter_scalar_den(z["taps_100_0p5"], 1.0, x)
    # seeded windows: inputs are regenerated from the integer generator, outputs stored
    for k, nwin in (("taps_101_0p25", 1 << 14), ("taps_255_0p25", 1 << 14), ("taps_4095_0p1", 1 << 14)):
        xs = orc.fill_splitmix(nwin, SEED, 0)
        z["seeded_y_" + k] = ref.lfilter_scalar_den(z[k], 1.0, xs)
    # a ragged length, n < ntaps (all warm-up), and the IIR branch (not on the GPU path; pins the oracle)
    xs = orc.fill_splitmix(1000, SEED, 12345)
    z["ragged_y_255"] = ref.lfilter_scalar_den(z["taps_255_0p25"], 1.0, xs[:777])
    z["short_y_4095"] = ref.lfilter_scalar_den(z["taps_4095_0p1"], 1.0, xs)
    z["iir_a"] = np.array([1.0, -0.5, 0.25])
    z["iir_y"] = ref.lfilter(z["taps_100_0p5"][:7], z["iir_a"], xs)
    # the recursive branch over more than two tiles of the GPU kernels: a 4-pole low-pass whose a[0] is not 1
    # (the reference never reads it, dsp.hpp:366), and a resonator
    xs = orc.fill_splitmix(9000, SEED, 777)
    z["iir4_b"] = np.array([0.0048, 0.0193, 0.0289, 0.0193, 0.0048])
    z["iir4_a"] = np.array([3.0, -2.3695, 2.3140, -1.0547, 0.1874])
    z["iir4_y"] = ref.lfilter(z["iir4_b"], z["iir4_a"], xs)
    z["iir2_b"] = np.array([1.0, 0.5, 0.25])
    z["iir2_a"] = np.array([1.0, 0.0, 0.81])
    z["iir2_y"] = ref.lfilter(z["iir2_b"], z["iir2_a"], xs)
    # narrow element types: lfilter<double> over float containers, and lfilter<float> with a recursive part
    xs32 = orc.fill_splitmix(3000, SEED, 4242).astype(np.float32)
    z["f64_from_f32_y"] = ref.lfilter_f64_from_f32(z["taps_101_0p25"].astype(np.float32), xs32)
    z["f32_iir_y"] = ref.lfilter_f32_iir(z["iir4_b"].astype(np.float32), z["iir4_a"].astype(np.float32), xs32)
    z["f32_from_f64_iir_y"] = ref.lfilter_f32_from_f64_iir(z["iir4_b"], z["iir4_a"], orc.fill_splitmix(3000, SEED, 4242))
    # wavelet bank
    psi, wx = ref.integrate_wavelet(10)
    z["morlet10_psi"], z["morlet10_x"] = psi, wx
    sig = orc.fill_splitmix(2048, SEED, 777)
    z["cwt_scales"] = np.array([1.0, 2.0, 5.0, 17.0, 65.0])
    z["cwt_seeded"] = ref.cwt(sig, z["cwt_scales"], psi, wx)
    return z


def main() -> None:
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    with open(os.path.join(gold, "ref_test_literals.json"), "w") as f:
        json.dump(literals(), f, indent=0)
    np.savez_compressed(os.path.join(gold, "ref_outputs.npz"), **outputs())
    print("wrote", os.listdir(gold))


if __name__ == "__main__":
    main()
