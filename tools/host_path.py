#!/usr/bin/env python3
"""End-to-end timing of the host-pointer entry points (dspx_fir_f64 / _batched): pageable host
buffers in, pageable host buffers out, PCIe both ways (development aid; bench.py's value is the
device-resident rate).   python tools/host_path.py [log2n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import dsp_amd



def _t(f):
    t0 = time.perf_counter()
    f()
    return time.perf_counter() - t0


taps = dsp_amd.firwin(255, 0.25)
rng = np.random.default_rng(1)
for lg in [int(a) for a in sys.argv[1:]] or [20, 24, 27]:
    n = 1 << lg
    x = rng.uniform(-1, 1, n)
    dsp_amd.lfilter(taps, x[: min(n, 1 << 16)])  # context, streams, first-touch
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        y = dsp_amd.lfilter(taps, x)
        best = min(best, time.perf_counter() - t0)
    reuse = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        dsp_amd.lfilter(taps, x, out=y)
        reuse = min(reuse, time.perf_counter() - t0)
    print(f"lfilter 255 taps, n=2^{lg}: fresh output {best * 1e3:9.2f} ms {n / best / 1e9:6.3f} GS/s | reused output "
          f"{reuse * 1e3:9.2f} ms {n / reuse / 1e9:6.3f} GS/s ({16 * n / reuse / 1e9:5.1f} GB/s both directions)", flush=True)

# a recursive filter (biquad) from host memory: pieces of a channel are issued in order, state carried on the device
a2 = np.array([1.0, -1.1429805025399011, 0.4128015980961886])
b2 = np.array([0.06745527, 0.13491055, 0.06745527])
for lg in (20, 24, 27):
    x = rng.uniform(-1, 1, 1 << lg)
    dsp_amd.lfilter_iir(b2, a2, x[:4096])
    best = min(_t(lambda: dsp_amd.lfilter_iir(b2, a2, x)) for _ in range(3))
    print(f"lfilter biquad, n=2^{lg}: {best * 1e3:9.2f} ms  {x.size / best / 1e9:6.3f} GS/s (fresh output)", flush=True)

# the wavelet bank from host memory: 65 scales, result 65 x n doubles brought back through pinned staging
psi, wx = dsp_amd.integrate_wavelet(10)
scales = np.arange(1, 66, dtype=np.float64)
for lg in (16, 20, 22):
    sig = rng.uniform(-1, 1, 1 << lg)
    dsp_amd.cwt(sig[:4096], scales, psi, wx)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = dsp_amd.cwt(sig, scales, psi, wx)
        best = min(best, time.perf_counter() - t0)
    print(f"cwt 65 scales, n=2^{lg}: {best * 1e3:9.2f} ms  ({out.nbytes / best / 1e9:5.1f} GB/s of result)", flush=True)
