#!/usr/bin/env python3
"""Tap-count sweep of the FIR kernels on one box, one process: vector pipe, matrix pipe (the default form: wave-autonomous
at every tap count) and the matrix pipe's workgroup-tiled form, device-resident, 2048 x 2^20 samples unless the row names a
benchmark configuration.  Writes JSON (the crossover between the pipes, kMatrixPipeMinTaps, is read off this table).
   python tools/tap_sweep.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dsp_amd

ROWS = [(nt, 2048, 1 << 20, None) for nt in (16, 24, 31, 40, 48, 52, 56, 60, 63, 72, 80, 96, 112, 128, 160, 192, 224, 255, 281, 282,
                                            320, 512, 1023)]
ROWS += [(101, 1, 1 << 28, "configs[1]"), (255, 4096, 1 << 20, "configs[2]"), (4095, 1, 1 << 26, "configs[3]")]
ROWS.sort(key=lambda r: (r[0], r[1]))
PEAK_TF, PEAK_GBS = 78.6, 8000.0


def timed(taps, x, y, **kw):
    for _ in range(3):
        dsp_amd.fir_device(taps, x, out=y, **kw)
    torch.cuda.synchronize()
    reps = max(5, int(0.5 / (x.numel() * taps.numel() * 2 / 50e12 + 1e-4)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dsp_amd.fir_device(taps, x, out=y, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, reps


out = {"tool": "tools/tap_sweep.py", "device": torch.cuda.get_device_name(0), "started": time.strftime("%Y-%m-%d %H:%M:%S"),
       "units": "GS/s of output; frac_fp64 = 2*taps*GS/s / 78.6 TFLOP/s; frac_hbm = 16 B * GS/s / 8 TB/s",
       "default_matrix_pipe_from_taps": dsp_amd.MATRIX_PIPE_MIN_TAPS, "rows": []}
for nt, nchan, n, label in ROWS:
    taps = torch.from_numpy(dsp_amd.firwin(nt, 0.25)).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp_amd.fill_splitmix(x)
    row = {"taps": nt, "nchan": nchan, "n": n, "config": label}
    for name, kw in (("vector", dict(mfma=False)), ("matrix", dict(mfma=True)), ("matrix_tiled", dict(mfma=True, tiled=True))):
        ms, reps = timed(taps, x, y, **kw)
        gs = nchan * n / ms / 1e6
        row[name] = {"ms": round(ms, 4), "launches": reps, "GS/s": round(gs, 2), "frac_fp64": round(gs * 2 * nt / 1e3 / PEAK_TF, 4),
                     "frac_hbm": round(gs * 16 / PEAK_GBS, 4)}
    row["fastest"] = max(("vector", "matrix", "matrix_tiled"), key=lambda k: row[k]["GS/s"])
    row["default_picks"] = "matrix" if nt >= dsp_amd.MATRIX_PIPE_MIN_TAPS else "vector"
    out["rows"].append(row)
    print(f"{nt:5d} taps {nchan:5d} x {n:10d}: vector {row['vector']['GS/s']:8.2f}  matrix {row['matrix']['GS/s']:8.2f}  "
          f"tiled {row['matrix_tiled']['GS/s']:8.2f} GS/s   fastest {row['fastest']}", flush=True)
    del x, y
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
