#!/usr/bin/env python3
"""Device-resident timing of lfilter with a recursive part (development aid).
   python tools/qb_iir.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scipy import signal

import dsp_amd

for nchan, n, order, exact in ((4096, 1 << 20, 2, False), (4096, 1 << 20, 4, False), (4096, 1 << 20, 8, False),
                               (1, 1 << 28, 2, False), (1, 1 << 28, 6, False), (4096, 1 << 20, 2, True), (1, 1 << 24, 2, True)):
    b, a = signal.butter(order, 0.2)
    bt = torch.from_numpy(b).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp_amd.fill_splitmix(x)
    for _ in range(2):
        dsp_amd.lfilter_iir_device(bt, a, x, out=y, exact=exact)
    torch.cuda.synchronize()
    reps = 3 if exact else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dsp_amd.lfilter_iir_device(bt, a, x, out=y, exact=exact)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{nchan} x 2^{n.bit_length() - 1}, butter({order}), {'exact (sequential)' if exact else 'block-parallel'}: "
          f"{ms:9.3f} ms  {nchan * n / ms / 1e6:8.2f} GS/s  ({16 * nchan * n / ms / 1e9:5.2f} TB/s in algorithmic bytes, 16 per sample)",
          flush=True)
    del x, y
