#!/usr/bin/env python3
"""Quick device-resident timing of the FIR kernel over several workloads (development aid).
   python tools/qb.py [name ...]     names: b255 l101 l4095 b255s b31 b63 b1023 ..., or c<channels>n<samples>t<taps>
   (c64n2e16t255 = 64 channels x 2^16 samples, 255 taps)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dsp_amd

W = {"b255": (4096, 1 << 20, 255), "l101": (1, 1 << 28, 101), "l4095": (1, 1 << 26, 4095), "b255s": (256, 1 << 20, 255), "b251": (1024, 1 << 20, 251), "b315": (1024, 1 << 20, 315), "b187": (1024, 1 << 20, 187), "b123": (1024, 1 << 20, 123), "b59": (1024, 1 << 20, 59),
     "b31": (1024, 1 << 20, 31), "b63": (1024, 1 << 20, 63), "b1023": (256, 1 << 20, 1023), "b101": (2048, 1 << 20, 101), "b256": (4096, 1 << 20, 256), "b240": (4096, 1 << 20, 240), "b512": (2048, 1 << 20, 512),
     "b96": (2048, 1 << 20, 96), "b128": (2048, 1 << 20, 128), "b160": (2048, 1 << 20, 160), "b192": (2048, 1 << 20, 192),
     "b144": (2048, 1 << 20, 144), "b300": (2048, 1 << 20, 300), "b112": (2048, 1 << 20, 112), "b80": (2048, 1 << 20, 80), "b224": (2048, 1 << 20, 224), "b208": (2048, 1 << 20, 208), "b48": (2048, 1 << 20, 48), "b2": (4096, 1 << 20, 2), "b8": (4096, 1 << 20, 8), "b16": (4096, 1 << 20, 16), "b24": (4096, 1 << 20, 24)}
MFMA = {"1": True, "0": False}.get(os.environ.get("QB_MFMA"))
TILED = os.environ.get("QB_TILED") == "1"   # the workgroup-tiled form of the matrix kernel (DSPX_FIR_TILED)
if os.environ.get("QB_SHARES"):   # 1: fixed shares of runs per wave, 2: runs drawn from the stream's counter at any size
    dsp_amd.lib().dspx_test_run_shares(int(os.environ["QB_SHARES"]))
names = sys.argv[1:] or ["b255", "l101", "l4095"]
def shape_of(nm):
    """A table name, or c<channels>n<samples>t<taps> (n may be 2^k written as 2e<k>)."""
    if nm in W:
        return W[nm]
    import re
    m = re.fullmatch(r"c(\d+)n(2e)?(\d+)t(\d+)", nm)
    if not m:
        raise SystemExit(f"unknown workload {nm}")
    n = (1 << int(m.group(3))) if m.group(2) else int(m.group(3))
    return int(m.group(1)), n, int(m.group(4))


for nm in names:
    nchan, n, nt = shape_of(nm)
    taps = torch.from_numpy(dsp_amd.firwin(nt, 0.25)).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp_amd.fill_splitmix(x)
    for _ in range(3):
        dsp_amd.fir_device(taps, x, out=y, mfma=MFMA, tiled=TILED)
    torch.cuda.synchronize()
    reps = max(5, int(0.6 / (nchan * n * nt * 2 / 50e12 + 1e-4)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dsp_amd.fir_device(taps, x, out=y, mfma=MFMA, tiled=TILED)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gs = nchan * n / ms / 1e6
    print(f"{nm:14s} taps={nt:5d} {ms:9.3f} ms  {gs:8.2f} GS/s  {gs * 2 * nt / 1e3:6.2f} TFLOP/s ({gs * 2 * nt / 786:5.1f}% fp64)  "
          f"{gs * 16 / 1e3:5.2f} TB/s ({gs * 16 / 80:5.1f}% hbm)", flush=True)
    del x, y
