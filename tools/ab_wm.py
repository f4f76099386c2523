#!/usr/bin/env python3
"""Development check of the several-windows kernel (570 taps and more, wavelet banks): fixed shares of chunks per wave
(the library's test hook) against the default, alternating, with a bit-for-bit comparison of the results first.
   python tools/ab_wm.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dsp_amd

hook = dsp_amd.lib().dspx_test_run_shares
psi, wx = dsp_amd.integrate_wavelet(10)
n = 1 << 22
sig = torch.empty(n, dtype=torch.float64, device="cuda")
dsp_amd.fill_splitmix(sig, dsp_amd.SEED, 0)


def timed(fn, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    reps = max(5, int(0.5 / (flops / 60e12)))
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


work = []
for spec in ("1-65", "18-35", "36-65"):
    lo, hi = (int(v) for v in spec.split("-"))
    bank = dsp_amd.CwtBank(np.arange(lo, hi + 1, dtype=np.float64), psi, wx)
    out = torch.empty((bank.rows, n), dtype=torch.float64, device="cuda")
    work.append((f"bank {spec}", (lambda b=bank, o=out: b.apply(sig, out=o)), out, 2.0 * bank.total_taps * n))
for nchan, ln, nt in ((2048, 20, 1023), (2048, 20, 700), (1, 26, 4095), (512, 20, 2047)):
    taps = torch.from_numpy(dsp_amd.firwin(nt, 0.25)).cuda()
    x = torch.empty((nchan, 1 << ln), dtype=torch.float64, device="cuda")
    dsp_amd.fill_splitmix(x)
    y = torch.empty_like(x)
    work.append((f"{nchan} x 2^{ln}, {nt} taps", (lambda t=taps, xx=x, yy=y: dsp_amd.fir_device(t, xx, out=yy)), y, 2.0 * nt * nchan * (1 << ln)))
for name, fn, out, flops in work:
    hook(1)
    fn()
    want = dsp_amd.checksum(out.reshape(-1) if out.dim() == 1 else out)
    hook(0)
    same = []
    for _ in range(3):
        out.zero_()
        fn()
        same.append(dsp_amd.checksum(out) == want)
    line = f"{name:24s} bits equal {all(same)}"
    for mode in (1, 0, 1, 0):
        hook(mode)
        dt = timed(fn, flops)
        line += f"   {'fixed' if mode else 'drawn'} {dt * 1e3:8.3f} ms {flops / dt / 78.6e12 * 100:5.1f} %"
    hook(0)
    print(line, flush=True)
