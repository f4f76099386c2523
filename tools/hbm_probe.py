#!/usr/bin/env python3
"""Attainable HBM bandwidth of the box (calibration, not part of the library): a device-to-device copy
of FP64 buffers counts 16 bytes per element, exactly like the FIR kernel's algorithmic bytes per output."""
import torch

for gib in (1, 8, 32):
    n = gib * (1 << 30) // 8
    x = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(3, 64 // gib)
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"copy {gib:3d} GiB -> {gib:3d} GiB: {ms:8.3f} ms  {16 * n / ms / 1e9:6.2f} TB/s  ({n / ms / 1e6:6.1f} G elements/s)", flush=True)
    del x, y
