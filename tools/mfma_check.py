"""Quick check of the matrix-pipe FIR kernel against the oracle over tap counts and lengths around its
tile and stage boundaries (development aid; the test-suite covers the same ground).  Run from the repository root."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, dsp_amd
from oracle.pyoracle import Oracle
o = Oracle()
worst = 0
for nt in (1, 3, 16, 31, 32, 100, 255, 256, 257, 513, 1041):
    rng = np.random.default_rng(nt)
    b = rng.standard_normal(nt)
    for n in (1, 63, 4095, 4096, 4097, 10000):
        x = rng.standard_normal(n)
        ref = o.lfilter(b, [1.0], x)
        y = dsp_amd.lfilter(b, x, mfma=True)
        err = np.max(np.abs(y - ref)) / max(np.max(np.abs(ref)), 1e-300)
        worst = max(worst, err)
        if err > 1e-12: print("FAIL", nt, n, err)
print("worst scaled err", worst)
