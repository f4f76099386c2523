#!/usr/bin/env python3
"""Development aid: per-wave phase times of the wave-autonomous matrix kernel, from a library built with
-DDSPX_M4W_TIMING (tools/build_variant.sh timing -DDSPX_M4W_TIMING; DSPX_LIB=... python tools/m4w_timing.py l101)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dsp_amd
from dsp_amd._lib import lib

if os.environ.get("M4W_SHARES"):   # 1: fixed shares of runs per wave, 2: runs drawn from the stream's counter at any size
    lib().dspx_test_run_shares(int(os.environ["M4W_SHARES"]))
W = {"b160": (2048, 1 << 20, 160), "b96": (2048, 1 << 20, 96), "b255": (4096, 1 << 20, 255), "l101": (1, 1 << 28, 101), "b128": (2048, 1 << 20, 128), "b63": (1024, 1 << 20, 63)}
for nm in sys.argv[1:] or ["l101"]:
    nchan, n, nt = W[nm]
    taps = torch.from_numpy(dsp_amd.firwin(nt, 0.25)).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp_amd.fill_splitmix(x)
    for _ in range(max(5, int(0.8 / (nchan * n * nt * 2 / 50e12 + 1e-4)))):   # long enough for the clock to settle
        dsp_amd.fir_device(taps, x, out=y, mfma=True)
    torch.cuda.synchronize()
    buf = np.zeros(16 * 8192, dtype=np.uint64)
    fn = lib().dspx_debug_m4w_timing
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_size_t]
    assert fn(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(-1, 16)
    t = t[t[:, 3] > 0]
    # Waves of the last launch only: a launch with fewer waves leaves the higher slots of an earlier one untouched.  The
    # cycle counters of different parts of the chip are not aligned with each other, so "recent" is judged per CU.
    hw = t[:, 6].astype(np.int64)
    where = ((((hw >> 32) & 15) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)
    newest = np.zeros(where.max() + 1, dtype=np.int64)
    np.maximum.at(newest, where, t[:, 4].astype(np.int64))
    t = t[t[:, 4].astype(np.int64) >= newest[where] - 1_000_000]
    prod, probe, rest, nch, t0, t1, hw, real, t64 = [t[:, i].astype(np.float64) for i in range(9)]
    hwi = t[:, 6].astype(np.int64) & 0xffffffff
    simd = (hwi >> 4) & 3
    cu = (hwi >> 8) & 15
    sh = (hwi >> 12) & 1
    se = (hwi >> 13) & 7
    xc = (t[:, 6].astype(np.int64) >> 32) & 15
    steps = (nt + 2) // 4 + 1
    rb = 16 if 21 <= steps <= 143 else 8   # matrix instructions per step: 1024-output chunks from 78 to 569 taps
    print(f"== {nm}: {len(t)} waves, {nch.mean():.1f} chunks of {64 * rb} outputs per wave, {steps} steps = {steps * rb * 16} pipe cycles per chunk")
    print(f"   per chunk (mean over waves, s_memtime ticks): products {np.mean(prod / nch):8.0f}  probe {np.mean(probe / nch):7.0f}  "
          f"rest {np.mean(rest / nch):7.0f}   wave lifetime {np.mean(t1 - t0):.0f}, span {(t1.max() - t0.min()):.0f}")
    print(f"   clock: {np.mean((t1 - t0) / real) * 0.1:.3f} GHz (s_memtime ticks per 10 ns of s_memrealtime), kernel {real.max() * 1e-5:.3f} ms")
    # rank of a wave among the waves that share its SIMD, by start time
    key = ((xc * 8 + se) * 2 + sh) * 16 + cu
    key = key * 4 + simd
    order = np.lexsort((t0, key))
    ranks = np.zeros(len(t), dtype=int)
    prev, r = None, 0
    for i in order:
        r = r + 1 if key[i] == prev else 0
        prev = key[i]
        ranks[i] = r
    for r in range(ranks.max() + 1):
        m = ranks == r
        print(f"   wave #{r} of its SIMD ({m.sum():5d} waves): products {np.mean(prod[m] / nch[m]):8.0f}  probe {np.mean(probe[m] / nch[m]):7.0f}  "
              f"rest {np.mean(rest[m] / nch[m]):7.0f}  lifetime {np.mean((t1 - t0)[m]):9.0f}  first 64 chunks {np.mean((t64 - t0)[m & (t64 > 0)]) / 64:8.0f} per chunk")
    share = sum(steps * rb * 16 / (np.mean((t64 - t0)[(ranks == r) & (t64 > 0)]) / 64) for r in range(ranks.max() + 1))
    print(f"   matrix pipe share while all waves run (first 64 chunks): {share:.3f}")
    # per XCD: the XCDs do not run equally fast (span = first wave start to last wave end on that XCD)
    for g in np.unique(xc):
        m = xc == g
        print(f"   XCD {int(g)}: {m.sum():4d} waves  chunks per wave {nch[m].mean():7.1f}  per chunk {np.mean((t1 - t0)[m] / nch[m]):8.0f}  "
              f"span {(t1[m].max() - t0[m].min()):10.0f}  clock {np.mean((t1 - t0)[m] / real[m]) * 0.1:.3f} GHz")
    print(f"   SIMDs seen: {len(np.unique(key))}, waves per SIMD: {np.bincount(np.unique(key, return_inverse=True)[1]).mean():.2f}")
