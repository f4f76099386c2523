"""ctypes loader for dsp_amd/lib/libdspx.so (the HIP kernels and their C ABI).

There is no fallback: if the library is missing or cannot be loaded, `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSPX_LIB selects another build of the same library (kernel experiments); it is never a fallback
LIB_PATH = os.environ.get("DSPX_LIB") or os.path.join(_HERE, "lib", "libdspx.so")

DSPX_OK = 0
DSPX_FIR_FMA = 0
DSPX_FIR_EXACT = 1
DSPX_FIR_MFMA = 2
DSPX_FIR_VECTOR = 4
DSPX_FIR_TILED = 8

class Piece(C.Structure):
    _fields_ = [("device", C.c_int), ("chan", C.c_size_t), ("nchan", C.c_size_t), ("start", C.c_size_t),
                ("len", C.c_size_t)]


# name -> (restype, argtypes); one entry per declaration in include/dspx.h
SIGNATURES = {
    "dspx_version": (C.c_int, []),
    "dspx_device_count": (C.c_int, []),
    "dspx_release_staging": (C.c_int, []),
    "dspx_last_error": (C.c_char_p, []),
    "dspx_status_string": (C.c_char_p, [C.c_int]),
    "dspx_firwin_f64": (C.c_int, [C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "dspx_hamming_window_f64": (C.c_int, [C.c_size_t, C.c_double, C.c_int, C.c_void_p]),
    "dspx_sinc_f64": (C.c_double, [C.c_double]),
    "dspx_fir_f64": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint]),
    "dspx_fir_f64_batched": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                       C.c_size_t, C.c_uint]),
    "dspx_fir_f64_multi": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                                     C.c_size_t, C.c_int, C.c_uint]),
    "dspx_fir_f64_stream": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_size_t, C.c_size_t, C.c_uint]),
    "dspx_lfilter_f64": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                   C.c_size_t, C.c_size_t, C.c_uint]),
    "dspx_lfilter_f64_stream": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint]),
    "dspx_lfilter_f64_multi": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_uint]),
    "dspx_lfilter_workspace_bytes": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_size_t]),
    "dspx_lfilter_f64_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]),
    "dspx_lfilter_f64_dev_noalloc": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p]),
    "dspx_fir_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dspx_fir_f32_from_f64": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dspx_fir_f64_from_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dspx_fir_f64_from_f32_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "dspx_lfilter_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_size_t]),
    "dspx_lfilter_f32_from_f64": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_size_t]),
    "dspx_fir_f32_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "dspx_fir_f32_from_f64_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "dspx_fir_plan": (C.c_long, [C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]),
    "dspx_fir_f64_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint, C.c_void_p]),
    "dspx_integrate_wavelet_morlet_f64": (C.c_int, [C.c_size_t, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "dspx_cwt_f64": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                               C.c_size_t, C.c_void_p, C.c_uint]),
    "dspx_cwt_f64_multi": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_int, C.c_uint]),
    "dspx_cwt_plan": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                C.c_void_p]),
    "dspx_cwt_bank_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p]),
    "dspx_cwt_bank_rows": (C.c_size_t, [C.c_void_p]),
    "dspx_cwt_bank_total_taps": (C.c_size_t, [C.c_void_p]),
    "dspx_cwt_bank_apply_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint,
                                          C.c_void_p]),
    "dspx_cwt_bank_destroy": (None, [C.c_void_p]),
    "dspx_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dspx_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dspx_comm_ranks": (C.c_int, [C.c_void_p]),
    "dspx_allgather_f64_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dspx_comm_destroy": (None, [C.c_void_p]),
    "dspx_fill_splitmix_f64_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p]),
    "dspx_checksum_f64_dev": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]),
}

_lib = None


class DspxError(RuntimeError):
    def __init__(self, code: int, detail: str):
        super().__init__(f"dspx error {code}: {detail}")
        self.code = code


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m dsp_amd.build` (needs hipcc). "
                "dsp_amd has no CPU fallback.")
        # torch wheels bundle their own libamdhip64.so.7; a process can hold only one HIP
        # runtime, and the dynamic loader shares it only if torch's copy is loaded first.
        if os.environ.get("DSPX_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != DSPX_OK:
        raise DspxError(rc, (lib().dspx_last_error() or b"").decode(errors="replace"))
