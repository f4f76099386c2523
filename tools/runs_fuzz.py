#!/usr/bin/env python3
"""Development check: random shapes through the wave-autonomous matrix kernels with their runs drawn from the stream's
counters (forced on for every size, the library's test hook) against fixed shares, bit for bit, on several streams
with launches in flight on all of them.   python tools/runs_fuzz.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dsp_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
hook = dsp_amd.lib().dspx_test_run_shares
streams = [torch.cuda.Stream() for _ in range(6)]
bad = 0
pending = []   # (drawn result, fixed result, description), checked in batches so that launches overlap
for case in range(cases):
    nt = int(rng.choice([52, 57, 63, 77, 78, 101, 128, 189, 190, 255, 313, 314, 400, 441, 442, 569, 570, 700, 1023]))
    kind = rng.integers(4) if nt < 570 else 4
    if kind == 0:
        nchan, n = 1, int(rng.integers(1, 1 << 23))
    elif kind == 1:
        nchan, n = int(rng.integers(2, 16)), int(rng.integers(1, 1 << 20))
    elif kind == 2:
        nchan, n = int(rng.integers(16, 400)), int(rng.integers(1, 1 << 16))
    elif kind == 3:
        nchan, n = int(rng.integers(400, 5000)), int(rng.integers(1, 3000))
    else:   # the several-windows kernel draws from four chunks per wave on: 2^23 samples and more
        nchan = int(rng.choice([1, 3, 20, 300]))
        n = int(rng.integers(1 << 23, 1 << 24)) // nchan
    n += n & 1   # (16-byte aligned rows: the wave-autonomous kernels)
    taps = torch.from_numpy(dsp_amd.firwin(nt, 0.3)).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    dsp_amd.fill_splitmix(x, offset=case << 24)
    hist = None
    if rng.integers(2):
        full = torch.empty((nchan, nt - 1 + ((nt - 1) & 1)), dtype=torch.float64, device="cuda")
        dsp_amd.fill_splitmix(full, offset=(case << 24) + (1 << 23))
        hist = full[:, :nt - 1]
    torch.cuda.synchronize()
    hook(1)
    fixed = dsp_amd.fir_device(taps, x, hist=hist, mfma=True)
    hook(2)
    st = streams[case % len(streams)]
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        drawn = dsp_amd.fir_device(taps, x, hist=hist, mfma=True)
        drawn2 = dsp_amd.fir_device(taps, x, hist=hist, mfma=True)
    # (the inputs stay referenced until the batch is checked: the allocator does not know that another stream reads them)
    pending.append((drawn, drawn2, fixed, f"case {case}: nchan={nchan} n={n} taps={nt} hist={hist is not None}", (x, hist, taps)))
    if len(pending) == 12 or case == cases - 1:
        torch.cuda.synchronize()
        for d1, d2, f, what, _inputs in pending:
            ok = torch.equal(d1.view(torch.int64), f.view(torch.int64)) and torch.equal(d2.view(torch.int64), f.view(torch.int64))
            bad += not ok
            if not ok:
                print("MISMATCH", what, flush=True)
        pending = []
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
hook(0)
sys.exit(1 if bad else 0)
