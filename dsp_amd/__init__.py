"""dsp_amd: MI355X (gfx950) FIR engine behind dsp::firwin -> dsp::lfilter(b, 1.0, x).

Layout:
  csrc/        HIP kernels (fir_kernels.hip, iir_kernels.hip) and the extern "C" ABI (dspx_capi.hip, dspx_cwt.hip,
               dspx_comm.hip)
  lib/         the built libdspx.so (in-tree, not tracked by git)
  build.py     hipcc recipe
  _lib.py      ctypes loader (raises if the library is missing; no CPU fallback)
  api.py       numpy / torch marshalling for tests and the benchmark
The C++ drop-in header is include/dsp.hpp; the C ABI is include/dspx.h.
"""
from ._lib import DspxError, lib  # noqa: F401
from .api import (MATRIX_PIPE_MIN_TAPS, SEED, Comm, CwtBank, allgather_channels, channel_range, checksum, cwt, cwt_plan, integrate_wavelet, device_count, fill_splitmix, fir_device, fir_f32_device, fir_plan, firwin, lfilter, lfilter_f32, lfilter_f64_from_f32,  # noqa: F401
                  lfilter_iir, lfilter_iir_device, lfilter_iir_stream, lfilter_multi, lfilter_stream, release_staging)
