#!/usr/bin/env python3
"""Development check: the wave-autonomous matrix kernel against the workgroup-tiled one (DSPX_FIR_TILED), bit for bit,
and against the oracle, over shapes that exercise run starts, channel ends, history and non-finite data."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dsp_amd
from oracle.pyoracle import Oracle

orc = Oracle()
if os.environ.get("M4W_SHARES"):   # 1: fixed shares of runs per wave, 2: runs drawn from the stream's counter at any size
    dsp_amd.lib().dspx_test_run_shares(int(os.environ["M4W_SHARES"]))
bad = 0
rng = np.random.default_rng(7)
cases = [(1, 512, 101), (1, 513, 101), (3, 5000, 255), (2, 100, 63), (1, 1 << 16, 281), (5, 70001, 57), (64, 1 << 14, 128),
         (1, 1 << 20, 101), (7, 33333, 200), (1, 1, 56), (2, 511, 255), (300, 2048, 96), (1, (1 << 18) + 3, 77),
         (40, 500_003, 101), (3, (1 << 23) + 5, 255), (1, (1 << 26) + 1000, 63), (130, 1 << 18, 160),
         (1, 513, 282), (3, 5000, 300), (2, 100, 512), (5, 70_001, 700), (1, 1 << 20, 1023), (1, (1 << 22) + 77, 4095),
         (40, 500_003, 320), (2, (1 << 23) + 5, 569), (7, 2049, 4095), (1, 300, 4095), (3, 70_001, 570), (64, 1 << 17, 400),
         (1, (1 << 24) + 3, 283), (3, 3_000_001, 600), (5000, 1000, 570)]
for minrun in ("default",):
    for nchan, n, nt in cases:
        taps_h = dsp_amd.firwin(nt, 0.25)
        taps = torch.from_numpy(taps_h).cuda()
        xh = rng.uniform(-1, 1, (nchan, n))
        x = torch.from_numpy(xh).cuda()
        for with_hist in (False, True):
            hist = None
            if with_hist:
                hh = rng.uniform(-1, 1, (nchan, nt - 1))
                hist = torch.from_numpy(hh).cuda()
            y0 = dsp_amd.fir_device(taps, x, hist=hist, mfma=True, tiled=True).cpu().numpy()
            y1 = dsp_amd.fir_device(taps, x, hist=hist, mfma=True).cpu().numpy()
            same = np.array_equal(y0.view(np.uint64), y1.view(np.uint64))
            err = 0.0
            if nchan * n <= 1 << 18:
                for c in range(min(nchan, 3)):
                    xin = np.concatenate([hh[c], xh[c]]) if with_hist else xh[c]
                    ref = orc.lfilter(taps_h, [1.0], xin)[-n:] if with_hist else orc.lfilter(taps_h, [1.0], xin)
                    err = max(err, float(np.max(np.abs(ref - y1[c])) / max(np.max(np.abs(ref)), 1e-300)))
            ok = same and err < 1e-12
            bad += not ok
            print(f"minrun={minrun} nchan={nchan} n={n} taps={nt} hist={with_hist}: bits_equal={same} err_vs_oracle={err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
# non-finite samples and taps
for nt in (101, 255):
    taps_h = dsp_amd.firwin(nt, 0.25)
    xh = rng.uniform(-1, 1, (2, 20000))
    xh[0, 3000] = np.inf
    xh[1, 12345] = np.nan
    xh[1, 0] = -np.inf
    th = taps_h.copy()
    for tcase in ("finite", "inf_tap"):
        if tcase == "inf_tap":
            th[5] = np.inf
        taps = torch.from_numpy(th).cuda()
        x = torch.from_numpy(xh).cuda()
        y0 = dsp_amd.fir_device(taps, x, mfma=True, tiled=True).cpu().numpy()
        y1 = dsp_amd.fir_device(taps, x, mfma=True).cpu().numpy()
        # a wave that meets a non-finite value sums its own chunk again, taps ascending: the forms differ in how many
        # outputs that is (512 or 1024), so near such values they agree to rounding, not to the bit
        same = np.array_equal(y0.view(np.uint64), y1.view(np.uint64))
        refs = [orc.lfilter(th, [1.0], xh[c]) for c in range(2)]
        nanmap = all(np.array_equal(np.isfinite(refs[c]), np.isfinite(y1[c])) for c in range(2))
        close = all(not np.isfinite(refs[c]).any() or np.max(np.abs(np.where(np.isfinite(refs[c]), y1[c] - refs[c], 0.0))) <= 1e-12 * np.max(np.abs(refs[c][np.isfinite(refs[c])])) for c in range(2))
        bad += not (nanmap and close)
        print(f"non-finite taps={nt} {tcase}: bits_equal={same} finite_map_equals_oracle={nanmap} finite_values_within_1e-12={close}", flush=True)
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
