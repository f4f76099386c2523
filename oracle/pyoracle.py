"""ctypes bindings for the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this module.  The shipped package (dsp_amd/) never does.

`Oracle` wraps oracle/liboracle.so (the plain-C restatement, fir_oracle.c and
cwt_oracle.c).  `Reference` wraps oracle/_ref/libdsp_ref.so, the unmodified
reference header compiled in the build container; it is optional and is absent
wherever /root/reference was not available at build time.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_szp = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")

SEED = 0x44535046495236  # "DSPFIR6": the seed every synthetic workload uses


def build_oracle(force: bool = False) -> str:
    """Compile liboracle.so (and _ref when the reference header is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("fir_oracle.c", "cwt_oracle.c")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return so


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    def __init__(self) -> None:
        L = C.CDLL(build_oracle())
        L.oracle_sinc.restype = C.c_double
        L.oracle_sinc.argtypes = [C.c_double]
        L.oracle_hamming_window.restype = C.c_int
        L.oracle_hamming_window.argtypes = [C.c_size_t, C.c_double, C.c_int, _f64p]
        L.oracle_firwin.restype = C.c_int
        L.oracle_firwin.argtypes = [C.c_size_t, _f64p, C.c_size_t, C.c_int, C.c_int, _f64p]
        L.oracle_lfilter_f64.restype = None
        L.oracle_lfilter_f64.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f64p, C.c_size_t, _f64p]
        L.oracle_fir_f64_hist.restype = None
        L.oracle_fir_f64_hist.argtypes = [_f64p, C.c_size_t, C.c_void_p, _f64p, C.c_size_t, _f64p]
        L.oracle_lfilter_f32.restype = None
        L.oracle_lfilter_f32.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _f32p]
        L.oracle_lfilter_f32_from_f64.restype = None
        L.oracle_lfilter_f32_from_f64.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f32p]
        L.oracle_fill_splitmix_f64.restype = None
        L.oracle_lfilter_f64_from_f32.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _f64p]
        L.oracle_lfilter_f32_iir.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _f32p, C.c_size_t, _f32p]
        L.oracle_lfilter_f32_from_f64_iir.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f64p, C.c_size_t, _f32p]
        L.oracle_fill_splitmix_f64.argtypes = [_f64p, C.c_size_t, C.c_uint64, C.c_uint64]
        L.oracle_checksum_f64.restype = C.c_uint64
        L.oracle_checksum_f64.argtypes = [_f64p, C.c_size_t, C.c_uint64]
        L.oracle_integrate_wavelet_morlet.restype = C.c_int
        L.oracle_integrate_wavelet_morlet.argtypes = [C.c_size_t, C.c_double, C.c_double, _f64p, _f64p]
        L.oracle_cwt.restype = C.c_int
        L.oracle_cwt.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f64p, C.c_size_t, _f64p, C.c_size_t, _f64p]
        L.oracle_cwt_scale_indices.restype = C.c_size_t
        L.oracle_cwt_scale_indices.argtypes = [C.c_double, C.c_double, C.c_double, C.c_size_t, C.c_size_t, _szp]
        self.L = L

    def sinc(self, x: float) -> float:
        return self.L.oracle_sinc(float(x))

    def hamming_window(self, width: int, alpha: float = 0.54, sym: bool = True) -> np.ndarray:
        w = np.empty(max(width, 1), dtype=np.float64)
        rc = self.L.oracle_hamming_window(width, alpha, int(sym), w)
        if rc != 0:
            raise ValueError("hamming_window: width must be >= 2")
        return w[:width]

    def firwin(self, numtaps: int, cutoffs, pass_zero: bool = True, scale: bool = True) -> np.ndarray:
        c = _f64(np.atleast_1d(cutoffs))
        h = np.empty(max(numtaps, 1), dtype=np.float64)
        rc = self.L.oracle_firwin(numtaps, c if c.size else np.zeros(1), c.size, int(pass_zero), int(scale), h)
        if rc != 0:
            raise ValueError(f"firwin: invalid arguments (code {rc})")
        return h[:numtaps]

    def lfilter(self, b, a, x) -> np.ndarray:
        b, x = _f64(b), _f64(x)
        a = _f64(np.atleast_1d(a))
        y = np.empty(x.size, dtype=np.float64)
        self.L.oracle_lfilter_f64(b if b.size else np.zeros(1), b.size, a, a.size, x if x.size else np.zeros(1),
                                  x.size, y if y.size else np.zeros(1))
        return y

    def fir_hist(self, b, hist, x) -> np.ndarray:
        b, x = _f64(b), _f64(x)
        y = np.empty(x.size, dtype=np.float64)
        if hist is None:
            hp = None
        else:
            hist = _f64(hist)
            assert hist.size == max(b.size, 1) - 1
            hp = hist.ctypes.data_as(C.c_void_p) if hist.size else None
        self.L.oracle_fir_f64_hist(b if b.size else np.zeros(1), b.size, hp, x if x.size else np.zeros(1), x.size,
                                   y if y.size else np.zeros(1))
        return y

    def lfilter_f32(self, b, x) -> np.ndarray:
        b = np.ascontiguousarray(b, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty(x.size, dtype=np.float32)
        self.L.oracle_lfilter_f32(b if b.size else np.zeros(1, np.float32), b.size,
                                  x if x.size else np.zeros(1, np.float32), x.size,
                                  y if y.size else np.zeros(1, np.float32))
        return y

    def lfilter_f32_from_f64(self, b, x) -> np.ndarray:
        """lfilter<float> over double containers: double products, rounded to float, float sums."""
        b, x = _f64(b), _f64(x)
        y = np.empty(x.size, dtype=np.float32)
        self.L.oracle_lfilter_f32_from_f64(b if b.size else np.zeros(1), b.size, x if x.size else np.zeros(1), x.size,
                                           y if y.size else np.zeros(1, np.float32))
        return y

    def lfilter_f64_from_f32(self, b, x) -> np.ndarray:
        """lfilter<double> over float containers: float products, widened, double sums."""
        b = np.ascontiguousarray(b, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty(x.size, dtype=np.float64)
        self.L.oracle_lfilter_f64_from_f32(b if b.size else np.zeros(1, np.float32), b.size,
                                           x if x.size else np.zeros(1, np.float32), x.size, y if y.size else np.zeros(1))
        return y

    def lfilter_f32_iir(self, b, a, x) -> np.ndarray:
        """lfilter<float> with a recursive part over float containers (tail in float)."""
        b, a, x = (np.ascontiguousarray(v, dtype=np.float32) for v in (b, a, x))
        y = np.empty(x.size, dtype=np.float32)
        one = np.zeros(1, np.float32)
        self.L.oracle_lfilter_f32_iir(b if b.size else one, b.size, a if a.size else one, a.size, x if x.size else one, x.size,
                                      y if y.size else one)
        return y

    def lfilter_f32_from_f64_iir(self, b, a, x) -> np.ndarray:
        """lfilter<float> with a recursive part over double containers (products and tail in double, float results)."""
        b, a, x = _f64(b), _f64(a), _f64(x)
        y = np.empty(x.size, dtype=np.float32)
        one = np.zeros(1)
        self.L.oracle_lfilter_f32_from_f64_iir(b if b.size else one, b.size, a if a.size else one, a.size,
                                               x if x.size else one, x.size, y if y.size else np.zeros(1, np.float32))
        return y

    def fill_splitmix(self, n: int, seed: int = SEED, offset: int = 0) -> np.ndarray:
        x = np.empty(n, dtype=np.float64)
        if n:
            self.L.oracle_fill_splitmix_f64(x, n, seed & (2**64 - 1), offset & (2**64 - 1))
        return x

    def checksum(self, y, offset: int = 0) -> int:
        y = _f64(y)
        return int(self.L.oracle_checksum_f64(y if y.size else np.zeros(1), y.size, offset))

    def integrate_wavelet(self, precision: int = 10, lower: float = -8.0, upper: float = 8.0):
        n = 1 << precision
        psi, x = np.empty(n), np.empty(n)
        rc = self.L.oracle_integrate_wavelet_morlet(precision, lower, upper, psi, x)
        if rc != 0:
            raise ValueError("integrate_wavelet: invalid arguments")
        return psi, x

    def cwt(self, signal, scales, psi, wx) -> np.ndarray:
        signal, scales, psi, wx = _f64(signal), _f64(scales), _f64(psi), _f64(wx)
        out = np.empty(scales.size * signal.size, dtype=np.float64)
        rc = self.L.oracle_cwt(signal, signal.size, scales, scales.size, psi, psi.size, wx, wx.size, out)
        if rc != 0:
            raise ValueError(f"cwt: error code {rc}")
        return out.reshape(scales.size, signal.size)

    def cwt_scale_indices(self, scale, x_step, x_range, npsi, cap) -> np.ndarray:
        idx = np.empty(max(cap, 1), dtype=np.uint64)
        n = self.L.oracle_cwt_scale_indices(scale, x_step, x_range, npsi, cap, idx)
        return idx[:n].astype(np.int64)


def reference_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libdsp_ref.so"))


class Reference:
    """The unmodified reference header (build container only)."""

    def __init__(self) -> None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libdsp_ref.so"))
        L.ref_sinc.restype = C.c_double
        L.ref_sinc.argtypes = [C.c_double]
        L.ref_hamming_window.argtypes = [C.c_size_t, C.c_double, C.c_int, _f64p]
        L.ref_firwin.argtypes = [C.c_size_t, _f64p, C.c_size_t, C.c_int, C.c_int, _f64p]
        L.ref_firwin_scalar.argtypes = [C.c_size_t, C.c_double, _f64p]
        L.ref_lfilter.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f64p, C.c_size_t, _f64p]
        L.ref_lfilter_scalar_den.argtypes = [_f64p, C.c_size_t, C.c_double, _f64p, C.c_size_t, _f64p]
        L.ref_lfilter_f32.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _f32p]
        L.ref_lfilter_f32_from_f64.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f32p]
        L.ref_lfilter_f64_from_f32.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _f64p]
        L.ref_lfilter_f32_iir.argtypes = [_f32p, C.c_size_t, _f32p, C.c_size_t, _f32p, C.c_size_t, _f32p]
        L.ref_lfilter_f32_from_f64_iir.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f64p, C.c_size_t, _f32p]
        L.ref_integrate_wavelet.argtypes = [C.c_size_t, _f64p, _f64p]
        L.ref_cwt.argtypes = [_f64p, C.c_size_t, _f64p, C.c_size_t, _f64p, _f64p, C.c_size_t, _f64p]
        self.L = L

    def sinc(self, x):
        return self.L.ref_sinc(float(x))

    def hamming_window(self, width, alpha=0.54, sym=True):
        w = np.empty(width + 1, dtype=np.float64)
        rc = self.L.ref_hamming_window(width, alpha, int(sym), w)
        if rc != 0:
            raise ValueError(f"reference threw (code {rc})")
        return w[:width]

    def firwin(self, numtaps, cutoffs, pass_zero=True, scale=True):
        c = _f64(np.atleast_1d(cutoffs))
        h = np.empty(max(numtaps, 1), dtype=np.float64)
        rc = self.L.ref_firwin(numtaps, c if c.size else np.zeros(1), c.size, int(pass_zero), int(scale), h)
        if rc != 0:
            raise ValueError(f"reference threw (code {rc})")
        return h[:numtaps]

    def firwin_scalar(self, numtaps, cutoff):
        h = np.empty(max(numtaps, 1), dtype=np.float64)
        rc = self.L.ref_firwin_scalar(numtaps, float(cutoff), h)
        if rc != 0:
            raise ValueError(f"reference threw (code {rc})")
        return h[:numtaps]

    def lfilter(self, b, a, x):
        b, x = _f64(b), _f64(x)
        a = _f64(np.atleast_1d(a))
        y = np.empty(max(x.size, 1), dtype=np.float64)
        rc = self.L.ref_lfilter(b if b.size else np.zeros(1), b.size, a, a.size, x if x.size else np.zeros(1), x.size, y)
        assert rc == 0
        return y[: x.size]

    def lfilter_scalar_den(self, b, den, x):
        b, x = _f64(b), _f64(x)
        y = np.empty(max(x.size, 1), dtype=np.float64)
        rc = self.L.ref_lfilter_scalar_den(b if b.size else np.zeros(1), b.size, float(den),
                                           x if x.size else np.zeros(1), x.size, y)
        assert rc == 0
        return y[: x.size]

    def lfilter_f32(self, b, x):
        b = np.ascontiguousarray(b, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty(max(x.size, 1), dtype=np.float32)
        rc = self.L.ref_lfilter_f32(b, b.size, x, x.size, y)
        assert rc == 0
        return y[: x.size]

    def lfilter_f32_from_f64(self, b, x):
        b, x = _f64(b), _f64(x)
        y = np.empty(max(x.size, 1), dtype=np.float32)
        rc = self.L.ref_lfilter_f32_from_f64(b if b.size else np.zeros(1), b.size, x if x.size else np.zeros(1), x.size, y)
        assert rc == 0
        return y[: x.size]

    def lfilter_f64_from_f32(self, b, x):
        b = np.ascontiguousarray(b, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty(max(x.size, 1), dtype=np.float64)
        assert self.L.ref_lfilter_f64_from_f32(b, b.size, x, x.size, y) == 0
        return y[: x.size]

    def lfilter_f32_iir(self, b, a, x):
        b, a, x = (np.ascontiguousarray(v, dtype=np.float32) for v in (b, a, x))
        y = np.empty(max(x.size, 1), dtype=np.float32)
        assert self.L.ref_lfilter_f32_iir(b, b.size, a, a.size, x, x.size, y) == 0
        return y[: x.size]

    def lfilter_f32_from_f64_iir(self, b, a, x):
        b, a, x = _f64(b), _f64(a), _f64(x)
        y = np.empty(max(x.size, 1), dtype=np.float32)
        assert self.L.ref_lfilter_f32_from_f64_iir(b, b.size, a, a.size, x, x.size, y) == 0
        return y[: x.size]

    def integrate_wavelet(self, precision=10):
        n = 1 << precision
        psi, x = np.empty(n), np.empty(n)
        rc = self.L.ref_integrate_wavelet(precision, psi, x)
        if rc != 0:
            raise ValueError(f"reference threw (code {rc})")
        return psi, x

    def cwt(self, signal, scales, psi, wx):
        signal, scales, psi, wx = _f64(signal), _f64(scales), _f64(psi), _f64(wx)
        out = np.empty(scales.size * signal.size, dtype=np.float64)
        rc = self.L.ref_cwt(signal, signal.size, scales, scales.size, psi, wx, psi.size, out)
        if rc != 0:
            raise ValueError(f"reference threw (code {rc})")
        return out.reshape(scales.size, signal.size)

    def run_hot_path_tests(self):
        return self.L.ref_run_hot_path_tests()
