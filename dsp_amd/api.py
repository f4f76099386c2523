"""Python-side callers of the C ABI, for tests, the smoke check and the benchmark.

The product's host language is C++ (include/dsp.hpp mirrors the reference's
dsp::firwin / dsp::lfilter); these helpers only marshal numpy arrays and torch
device tensors into the same extern "C" entry points.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import DSPX_FIR_EXACT, DSPX_FIR_FMA, DSPX_FIR_MFMA, DSPX_FIR_TILED, DSPX_FIR_VECTOR, check, lib

SEED = 0x44535046495236
# Pipe selection of the default flags (mirrors kMatrixPipeMinTaps in csrc/dspx_common.h):
# filters of at least MATRIX_PIPE_MIN_TAPS taps run on the FP64 matrix pipe.
MATRIX_PIPE_MIN_TAPS = 52


def device_count() -> int:
    return int(lib().dspx_device_count())


def release_staging() -> int:
    """Frees the idle staging sets of the host-pointer entry points (dspx_release_staging)."""
    return int(lib().dspx_release_staging())


def _flags(exact: bool, mfma=None, tiled: bool = False) -> int:
    """exact -> DSPX_FIR_EXACT; mfma: None = library's choice, True / False = force the matrix / vector pipe;
    tiled -> DSPX_FIR_TILED (the workgroup-tiled form of the matrix-pipe kernel, for comparison measurements)."""
    f = DSPX_FIR_EXACT if exact else DSPX_FIR_FMA
    if mfma is True:
        f |= DSPX_FIR_MFMA
    elif mfma is False:
        f |= DSPX_FIR_VECTOR
    if tiled:
        f |= DSPX_FIR_TILED
    return f


def _np_f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def firwin(numtaps: int, cutoff, pass_zero: bool = True, scale: bool = True) -> np.ndarray:
    """dsp::firwin through the library (host arithmetic, see include/dsp.hpp)."""
    c = _np_f64(np.atleast_1d(cutoff))
    out = np.empty(max(int(numtaps), 1), dtype=np.float64)
    check(lib().dspx_firwin_f64(int(numtaps), c.ctypes.data, c.size, int(pass_zero), int(scale), out.ctypes.data))
    return out[: int(numtaps)]


def lfilter(b, x, exact: bool = False, mfma=None, out=None) -> np.ndarray:
    """dsp::lfilter(b, 1.0, x) for host arrays: 1-D x, or 2-D (channels, samples).  `out`, if given, is a
    C-contiguous float64 array of x's shape that receives the result (as lfilter_ctx_t::result is reused)."""
    b, x = _np_f64(b), _np_f64(x)
    if out is None:
        y = np.empty_like(x)
    else:
        if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous and out.shape == x.shape):
            raise ValueError("out must be a C-contiguous float64 array of x's shape")
        y = out
    if x.ndim == 1:
        check(lib().dspx_fir_f64(b.ctypes.data, b.size, x.ctypes.data, y.ctypes.data, x.size, _flags(exact, mfma)))
    elif x.ndim == 2:
        check(lib().dspx_fir_f64_batched(b.ctypes.data, b.size, x.ctypes.data, y.ctypes.data, x.shape[0], x.shape[1],
                                         x.shape[1], _flags(exact, mfma)))
    else:
        raise ValueError("x must be 1-D or 2-D")
    return y


def lfilter_f32(b, x, a=None) -> np.ndarray:
    """dsp::lfilter<float>(b, a, x) for 1-D host arrays (a = None: no recursive part).  float32 taps and samples:
    products in float (dspx_fir_f32 / dspx_lfilter_f32); anything else is taken as double: products in double, rounded
    to float, float sums (dspx_fir_f32_from_f64 / dspx_lfilter_f32_from_f64, what the reference computes for containers
    of double).  The tail is evaluated in float for float32 denominators and in double otherwise, as the reference's
    promotion rules have it."""
    b, x = np.asarray(b), np.asarray(x)
    narrow = b.dtype == np.float32 and x.dtype == np.float32
    if narrow:
        b, x = np.ascontiguousarray(b), np.ascontiguousarray(x)
    else:
        b, x = _np_f64(b), _np_f64(x)
    y = np.empty(x.size, dtype=np.float32)
    if a is None or np.size(a) <= 1:
        fn = lib().dspx_fir_f32 if narrow else lib().dspx_fir_f32_from_f64
        check(fn(b.ctypes.data, b.size, x.ctypes.data, y.ctypes.data, x.size))
        return y
    a = np.asarray(a)
    in_double = 0 if a.dtype == np.float32 else 1
    a64 = _np_f64(a)
    fn = lib().dspx_lfilter_f32 if narrow else lib().dspx_lfilter_f32_from_f64
    check(fn(b.ctypes.data, b.size, a64.ctypes.data, a64.size, in_double, x.ctypes.data, y.ctypes.data, x.size))
    return y


def lfilter_f64_from_f32(b, x) -> np.ndarray:
    """dsp::lfilter<double>(b, 1.0, x) over float32 taps and samples: float products, double sums (dspx_fir_f64_from_f32)."""
    b = np.ascontiguousarray(b, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(x.size, dtype=np.float64)
    check(lib().dspx_fir_f64_from_f32(b.ctypes.data, b.size, x.ctypes.data, y.ctypes.data, x.size))
    return y


def fir_f32_device(taps, x, out=None, hist=None):
    """dspx_fir_f32_dev / dspx_fir_f32_from_f64_dev on torch CUDA tensors (float32 or float64 taps and x, float32 out)."""
    import torch

    assert x.is_cuda and taps.is_cuda and taps.dtype == x.dtype and x.dtype in (torch.float32, torch.float64)
    x2 = x.reshape(1, -1) if x.dim() == 1 else x
    assert x2.dim() == 2 and x2.stride(1) == 1
    if out is None:
        out = torch.empty(x2.shape, dtype=torch.float32, device=x.device)
    o2 = out.reshape(1, -1) if out.dim() == 1 else out
    assert o2.shape == x2.shape and o2.stride(1) == 1 and o2.dtype == torch.float32
    hptr, hstride = None, 0
    if hist is not None:
        h2 = hist.reshape(1, -1) if hist.dim() == 1 else hist
        assert h2.dtype == x.dtype and h2.shape == (x2.shape[0], max(taps.numel() - 1, 0)) and (h2.stride(1) == 1 or h2.shape[1] <= 1)
        hptr, hstride = h2.data_ptr(), h2.stride(0)
    fn = lib().dspx_fir_f32_dev if x.dtype == torch.float32 else lib().dspx_fir_f32_from_f64_dev
    check(fn(taps.data_ptr(), taps.numel(), x2.data_ptr(), x2.stride(0), hptr, hstride, o2.data_ptr(), o2.stride(0),
             x2.shape[0], x2.shape[1], _stream_ptr()))
    return out.reshape(x.shape) if out.dim() != x.dim() else out


def lfilter_iir(b, a, x, exact: bool = False, mfma=None, ngpus=None) -> np.ndarray:
    """dsp::lfilter(b, a, x) with a recursive part (dsp.hpp:342-371) for host arrays: 1-D x, or 2-D
    (channels, samples).  a[0] is never read.  ngpus: shard the channels over that many devices (0 = all)."""
    b, a, x = _np_f64(b), _np_f64(np.atleast_1d(a)), _np_f64(x)
    x2 = x.reshape(1, -1) if x.ndim == 1 else x
    if x2.ndim != 2:
        raise ValueError("x must be 1-D or 2-D")
    y = np.empty_like(x2)
    if ngpus is None:
        check(lib().dspx_lfilter_f64(b.ctypes.data, b.size, a.ctypes.data, a.size, x2.ctypes.data, y.ctypes.data,
                                     x2.shape[0], x2.shape[1], x2.shape[1], _flags(exact, mfma)))
    else:
        check(lib().dspx_lfilter_f64_multi(b.ctypes.data, b.size, a.ctypes.data, a.size, x2.ctypes.data, y.ctypes.data,
                                           x2.shape[0], x2.shape[1], x2.shape[1], int(ngpus), _flags(exact, mfma)))
    return y.reshape(x.shape)


def lfilter_iir_stream(b, a, xhist, yhist, x, exact: bool = False) -> np.ndarray:
    """One chunk of a recursive filter's stream: `xhist` (nb-1 per channel) and `yhist` (na-1 per channel), oldest
    first, carry the state and are updated in place (dspx_lfilter_f64_stream)."""
    b, a, x = _np_f64(b), _np_f64(np.atleast_1d(a)), _np_f64(x)
    x2 = x.reshape(1, -1) if x.ndim == 1 else x
    for h, w in ((xhist, max(b.size - 1, 0)), (yhist, max(a.size - 1, 0))):
        assert isinstance(h, np.ndarray) and h.dtype == np.float64 and h.flags.c_contiguous and h.size == x2.shape[0] * w
    y = np.empty_like(x2)
    check(lib().dspx_lfilter_f64_stream(b.ctypes.data, b.size, a.ctypes.data, a.size, xhist.ctypes.data, yhist.ctypes.data,
                                        x2.ctypes.data, y.ctypes.data, x2.shape[0], x2.shape[1], x2.shape[1], _flags(exact)))
    return y.reshape(x.shape)


def lfilter_iir_device(b, a, x, out=None, xhist=None, yhist=None, exact: bool = False, mfma=None, noalloc: bool = False):
    """dspx_lfilter_f64_dev (noalloc: dspx_lfilter_f64_dev_noalloc) on torch CUDA tensors: `b` (device taps), `a` (host coefficients), x/out
    [nchan, n] or [n]; xhist / yhist: the samples / outputs that precede the call, oldest first."""
    import torch

    a = _np_f64(np.atleast_1d(a))
    x2 = x.reshape(1, -1) if x.dim() == 1 else x
    assert x2.is_cuda and x2.dtype == torch.float64 and x2.stride(1) == 1 and b.is_cuda and b.dtype == torch.float64
    nchan, n = x2.shape
    if out is None:
        if noalloc:
            raise ValueError("noalloc=True needs `out`: allocating the result here would defeat the call's purpose")
        out = torch.empty_like(x)
    o2 = out.reshape(1, -1) if out.dim() == 1 else out
    assert o2.shape == x2.shape and o2.stride(1) == 1 and o2.dtype == torch.float64

    def hist_args(h, width):
        if h is None:
            return None, 0
        h2 = h.reshape(1, -1) if h.dim() == 1 else h
        assert h2.is_cuda and h2.dtype == torch.float64 and h2.shape == (nchan, width) and h2.stride(1) == 1
        return h2.data_ptr(), h2.stride(0)

    xh, xhs = hist_args(xhist, max(b.numel() - 1, 0))
    yh, yhs = hist_args(yhist, max(a.size - 1, 0))
    if noalloc:   # the library's static scratch instead of a workspace
        check(lib().dspx_lfilter_f64_dev_noalloc(b.data_ptr(), b.numel(), a.ctypes.data, a.size, x2.data_ptr(), x2.stride(0),
                                                 xh, xhs, yh, yhs, o2.data_ptr(), o2.stride(0), nchan, n,
                                                 _flags(exact, mfma), _stream_ptr()))
        return out
    need = int(lib().dspx_lfilter_workspace_bytes(a.size, nchan, n))
    work = torch.empty(max(need, 1), dtype=torch.uint8, device=x.device)
    check(lib().dspx_lfilter_f64_dev(b.data_ptr(), b.numel(), a.ctypes.data, a.size, x2.data_ptr(), x2.stride(0), xh, xhs,
                                     yh, yhs, o2.data_ptr(), o2.stride(0), nchan, n, work.data_ptr(), need,
                                     _flags(exact, mfma), _stream_ptr()))
    return out


def lfilter_stream(b, hist, x, exact: bool = False) -> np.ndarray:
    """One chunk of a stream: `hist` (ntaps-1 samples per channel, updated in place) carries the state."""
    b, x = _np_f64(b), _np_f64(x)
    assert isinstance(hist, np.ndarray) and hist.dtype == np.float64 and hist.flags.c_contiguous
    x2 = x.reshape(1, -1) if x.ndim == 1 else x
    assert hist.size == x2.shape[0] * max(b.size - 1, 0)
    y = np.empty_like(x2)
    check(lib().dspx_fir_f64_stream(b.ctypes.data, b.size, hist.ctypes.data, x2.ctypes.data, y.ctypes.data,
                                    x2.shape[0], x2.shape[1], x2.shape[1], _flags(exact)))
    return y.reshape(x.shape)


def lfilter_multi(b, x, ngpus: int = 0, exact: bool = False) -> np.ndarray:
    b, x = _np_f64(b), _np_f64(x)
    x2 = x.reshape(1, -1) if x.ndim == 1 else x
    y = np.empty_like(x2)
    check(lib().dspx_fir_f64_multi(b.ctypes.data, b.size, x2.ctypes.data, y.ctypes.data, x2.shape[0], x2.shape[1],
                                   x2.shape[1], int(ngpus), _flags(exact)))
    return y.reshape(x.shape)


def integrate_wavelet(precision: int = 10, lower: float = -8.0, upper: float = 8.0):
    """dsp::integrate_wavelet(MORLET, precision, lower, upper) -> (psi, x), host arithmetic."""
    n = 1 << precision if 0 <= precision < 40 else 1
    psi, x = np.empty(n), np.empty(n)
    check(lib().dspx_integrate_wavelet_morlet_f64(precision, lower, upper, psi.ctypes.data, x.ctypes.data))
    return psi, x


def cwt(signal, scales, psi, wx, exact: bool = False, ngpus=None, mfma=None, tiled: bool = False) -> np.ndarray:
    """dsp::cwt(signal, scales, psi, x) for host arrays -> (len(scales), len(signal)), rows by ascending scale.
    tiled: the workgroup-tiled form of the matrix-pipe kernel (for comparison measurements; results differ in rounding)."""
    signal, scales, psi, wx = _np_f64(signal), _np_f64(np.atleast_1d(scales)), _np_f64(psi), _np_f64(wx)
    out = np.empty((scales.size, signal.size), dtype=np.float64)
    if ngpus is None:
        rc = lib().dspx_cwt_f64(signal.ctypes.data, signal.size, scales.ctypes.data, scales.size, psi.ctypes.data,
                                psi.size, wx.ctypes.data, wx.size, out.ctypes.data, _flags(exact, mfma, tiled))
    else:
        rc = lib().dspx_cwt_f64_multi(signal.ctypes.data, signal.size, scales.ctypes.data, scales.size,
                                      psi.ctypes.data, psi.size, wx.ctypes.data, wx.size, out.ctypes.data, int(ngpus),
                                      _flags(exact, mfma, tiled))
    check(rc)
    return out


def cwt_plan(scales, npsi: int, wx, ngpus: int):
    """(device, ntaps) of every sorted scale, as dspx_cwt_f64_multi would assign them."""
    scales, wx = _np_f64(np.atleast_1d(scales)), _np_f64(wx)
    dev = np.zeros(scales.size, dtype=np.int32)
    nt = np.zeros(scales.size, dtype=np.uint64)
    check(lib().dspx_cwt_plan(scales.ctypes.data, scales.size, npsi, wx.ctypes.data, wx.size, ngpus, dev.ctypes.data,
                              nt.ctypes.data))
    return dev.tolist(), nt.astype(np.int64).tolist()


class CwtBank:
    """Per-scale tables of dsp::cwt kept on the current device (dspx_cwt_bank_*)."""

    def __init__(self, scales, psi, wx):
        scales, psi, wx = _np_f64(np.atleast_1d(scales)), _np_f64(psi), _np_f64(wx)
        h = C.c_void_p()
        check(lib().dspx_cwt_bank_create(scales.ctypes.data, scales.size, psi.ctypes.data, psi.size, wx.ctypes.data,
                                         wx.size, C.byref(h)))
        self._h = h
        self.rows = int(lib().dspx_cwt_bank_rows(h))
        self.total_taps = int(lib().dspx_cwt_bank_total_taps(h))

    def apply(self, signal, out=None, exact: bool = False, mfma=None, tiled: bool = False):
        import torch

        assert signal.is_cuda and signal.dtype == torch.float64 and signal.dim() == 1 and signal.is_contiguous()
        if out is None:
            out = torch.empty((self.rows, signal.numel()), dtype=torch.float64, device=signal.device)
        assert out.shape == (self.rows, signal.numel()) and out.stride(1) == 1
        check(lib().dspx_cwt_bank_apply_dev(self._h, signal.data_ptr(), signal.numel(), out.data_ptr(), out.stride(0),
                                            _flags(exact, mfma, tiled), _stream_ptr()))
        return out

    def close(self):
        if self._h:
            lib().dspx_cwt_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fir_plan(nchan: int, n: int, ngpus: int):
    """The multi-GPU sharding plan as a list of (device, chan, nchan, start, len) tuples."""
    total = lib().dspx_fir_plan(nchan, n, ngpus, None, 0)
    if total < 0:
        check(int(total))
    arr = (_lib.Piece * max(int(total), 1))()
    lib().dspx_fir_plan(nchan, n, ngpus, C.cast(arr, C.c_void_p), total)
    return [(p.device, p.chan, p.nchan, p.start, p.len) for p in arr[:total]]


# ---- torch device tensors ----------------------------------------------------------

def _stream_ptr():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def fir_device(taps, x, out=None, hist=None, exact: bool = False, mfma=None, tiled: bool = False):
    """Filters a float64 CUDA tensor x of shape (n,) or (channels, n) with device taps.
    `hist`, if given, has shape (..., ntaps-1): the samples preceding x along the last axis."""
    import torch

    assert x.is_cuda and x.dtype == torch.float64 and taps.is_cuda and taps.dtype == torch.float64
    x2 = x.reshape(1, -1) if x.dim() == 1 else x
    assert x2.dim() == 2 and x2.stride(1) == 1
    if out is None:
        out = torch.empty_like(x2, memory_format=torch.contiguous_format)
    o2 = out.reshape(1, -1) if out.dim() == 1 else out
    assert o2.shape == x2.shape and o2.stride(1) == 1
    hptr, hstride = None, 0
    if hist is not None:
        h2 = hist.reshape(1, -1) if hist.dim() == 1 else hist
        assert h2.is_cuda and h2.dtype == torch.float64 and h2.shape == (x2.shape[0], max(taps.numel() - 1, 0))
        assert h2.stride(1) == 1 or h2.shape[1] <= 1
        hptr, hstride = h2.data_ptr(), h2.stride(0)
    check(lib().dspx_fir_f64_dev(taps.data_ptr(), taps.numel(), x2.data_ptr(), x2.stride(0), hptr, hstride,
                                 o2.data_ptr(), o2.stride(0), x2.shape[0], x2.shape[1], _flags(exact, mfma, tiled),
                                 _stream_ptr()))
    return out.reshape(x.shape) if out.dim() != x.dim() else out


def fill_splitmix(t, seed: int = SEED, offset: int = 0):
    """Fills a contiguous float64 CUDA tensor with the deterministic [-1, 1) sequence."""
    import torch

    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    check(lib().dspx_fill_splitmix_f64_dev(t.data_ptr(), t.numel(), seed & (2**64 - 1), offset & (2**64 - 1),
                                           _stream_ptr()))
    return t


def checksum(t, offset: int = 0) -> int:
    import torch

    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    out = torch.zeros(1, dtype=torch.int64, device=t.device)
    check(lib().dspx_checksum_f64_dev(t.data_ptr(), t.numel(), offset & (2**64 - 1), out.data_ptr(), _stream_ptr()))
    return int(out.item()) & (2**64 - 1)


def channel_range(nchan_total: int, rank: int, world: int):
    """Contiguous channel range [lo, hi) of `rank` when `nchan_total` channels are sharded over `world` devices."""
    return nchan_total * rank // world, nchan_total * (rank + 1) // world


def allgather_channels(y_local, nchan_total: int, group=None):
    """Optional exchange step after a channel-sharded filter: every rank receives the shards of all ranks
    (one process per GPU; RCCL over xGMI when the process group's backend is "nccl").  `y_local` holds this
    rank's channels `channel_range(nchan_total, rank, world)` as rows; returns the [nchan_total, n] result.
    The filter itself needs no collective; this only serves callers that want the whole result on every GPU."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = channel_range(nchan_total, rank, world)
    if y_local.dim() != 2 or y_local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} must hold channels [{lo}, {hi}) as rows, got shape {tuple(y_local.shape)}")
    y_local = y_local.contiguous()
    n = y_local.shape[1]
    full = torch.empty((nchan_total, n), dtype=y_local.dtype, device=y_local.device)
    if nchan_total % world == 0:
        dist.all_gather_into_tensor(full, y_local, group=group)   # equal shards land in rank order = channel order
        return full
    # uneven shards: pad every shard to the largest, gather, then drop the padding rows
    most = max(channel_range(nchan_total, r, world)[1] - channel_range(nchan_total, r, world)[0] for r in range(world))
    send = torch.zeros((most, n), dtype=y_local.dtype, device=y_local.device)
    send[: hi - lo] = y_local
    recv = torch.empty((world * most, n), dtype=y_local.dtype, device=y_local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    for r in range(world):
        rlo, rhi = channel_range(nchan_total, r, world)
        full[rlo:rhi] = recv[r * most: r * most + (rhi - rlo)]
    return full


class Comm:
    """RCCL communicator of the optional exchange step (dspx_comm_*): one process per GPU.  `Comm.unique_id()` on
    rank 0, hand the bytes to every rank, then `Comm(id_bytes, nranks, rank)` on all ranks with their device current."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(Comm.ID_BYTES)
        check(lib().dspx_comm_unique_id(buf))
        return buf.raw

    def __init__(self, id_bytes: bytes, nranks: int, rank: int):
        assert len(id_bytes) == Comm.ID_BYTES
        h = C.c_void_p()
        check(lib().dspx_comm_create(id_bytes, int(nranks), int(rank), C.byref(h)))
        self._h, self.nranks, self.rank = h, int(nranks), int(rank)

    def allgather(self, y_local, out=None):
        """Every rank's contiguous float64 CUDA tensor `y_local` (equal shapes) -> [nranks * rows, ...] in rank order."""
        import torch

        assert y_local.is_cuda and y_local.dtype == torch.float64 and y_local.is_contiguous()
        if out is None:
            out = torch.empty((self.nranks * y_local.shape[0],) + tuple(y_local.shape[1:]), dtype=torch.float64,
                              device=y_local.device)
        assert out.is_contiguous() and out.numel() == self.nranks * y_local.numel() and out.dtype == torch.float64
        check(lib().dspx_allgather_f64_dev(self._h, y_local.data_ptr(), out.data_ptr(), y_local.numel(), _stream_ptr()))
        return out

    def close(self):
        if self._h:
            lib().dspx_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
