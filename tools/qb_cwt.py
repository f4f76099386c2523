#!/usr/bin/env python3
"""Device-resident timing of the wavelet bank by classes of rows (scales lo..hi of the 65-scale Morlet bank over 2^22
samples), default and workgroup-tiled form of the matrix-pipe kernel:  python tools/qb_cwt.py [lo-hi ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dsp_amd

psi, wx = dsp_amd.integrate_wavelet(10)
n = 1 << 22
sig = torch.empty(n, dtype=torch.float64, device="cuda")
dsp_amd.fill_splitmix(sig, dsp_amd.SEED, 0)
for spec in sys.argv[1:] or ["1-65", "4-17", "18-35", "36-53", "54-65"]:
    lo, hi = (int(v) for v in spec.split("-"))
    scales = np.arange(lo, hi + 1, dtype=np.float64)
    bank = dsp_amd.CwtBank(scales, psi, wx)
    out = torch.empty((bank.rows, n), dtype=torch.float64, device="cuda")
    line = f"scales {spec:>6}: {bank.total_taps:6d} taps"
    for tiled in (False, True, False, True):
        for _ in range(3):
            bank.apply(sig, out=out, tiled=tiled)
        torch.cuda.synchronize()
        reps = max(5, int(0.4 / (2.0 * bank.total_taps * n / 60e12)))
        t0 = time.perf_counter()
        for _ in range(reps):
            bank.apply(sig, out=out, tiled=tiled)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        line += f"   {'tiled' if tiled else 'wave '} {dt * 1e3:7.3f} ms {2.0 * bank.total_taps * n / dt / 78.6e12 * 100:5.1f} %"
    print(line, flush=True)
    bank.close()
