"""The cut of an input into runs of chunks for the wave-autonomous FIR kernels (plan_runs / run_span in
dsp_amd/csrc/fir_kernels.hip), checked on the CPU through the library's test hook dspx_test_run_plan (exported, not
part of include/dspx.h): whatever the shape and whether the runs are dealt in fixed shares or drawn from a counter,
the runs of a plan must cover every chunk of every channel exactly once and never cross a channel."""
import ctypes

import numpy as np
import pytest

import dsp_amd


def _plan(nchan, n, max_waves, chunk, drawn):
    fn = dsp_amd.lib().dspx_test_run_plan
    fn.restype = ctypes.c_longlong
    fn.argtypes = [ctypes.c_uint, ctypes.c_longlong, ctypes.c_uint, ctypes.c_uint, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    cap = 1 << 22
    spans = np.zeros((cap, 3), dtype=np.uint32)
    lead = ctypes.c_uint(0)
    got = fn(nchan, n, max_waves, chunk, drawn, spans.ctypes.data, cap, ctypes.addressof(lead))
    assert got >= 0, (nchan, n, max_waves, chunk, drawn)
    return spans[:got], int(lead.value)


def _check(nchan, n, max_waves, chunk, drawn):
    spans, lead = _plan(nchan, n, max_waves, chunk, drawn)
    cpc = -(-n // chunk)
    chan, c0, c1 = (spans[:, i].astype(np.int64) for i in range(3))
    assert np.all(chan < nchan) and np.all(c0 < c1) and np.all(c1 <= cpc), (nchan, n, max_waves, chunk, drawn)
    covered = np.zeros(nchan * cpc + 1, dtype=np.int64)   # difference array over (chan, chunk)
    np.add.at(covered, chan * cpc + c0, 1)
    np.add.at(covered, chan * cpc + c1, -1)
    assert np.all(np.cumsum(covered)[:-1] == 1), (nchan, n, max_waves, chunk, drawn)
    return spans, lead


SHAPES = [(1, 1 << 28), (1, (1 << 28) - 37), (2, 1 << 26), (3, 10_000_001), (15, 1 << 23), (16, 1 << 23), (17, 999_999),
          (4096, 1 << 20), (4096, 1000), (1000, 65_537), (1 << 20, 3), (5, 511), (1, 1), (64, 1 << 22), (7, 1 << 25)]


@pytest.mark.parametrize("chunk,max_waves", [(512, 4096), (1024, 2048), (512, 1216)])
@pytest.mark.parametrize("drawn", [0, 1])
def test_runs_cover_every_chunk_once(chunk, max_waves, drawn):
    for nchan, n in SHAPES:
        _check(nchan, n, max_waves, chunk, drawn)


def test_drawn_runs_end_with_short_runs():
    """Large inputs: long runs first, then runs about an eighth as long over the last quarter, so that the waves end
    within a few chunks of each other."""
    for nchan, n, chunk, waves in [(1, 1 << 28, 512, 4096), (4096, 1 << 20, 1024, 2048), (8, 1 << 25, 512, 4096)]:
        spans, lead = _check(nchan, n, waves, chunk, 1)
        length = (spans[:, 2] - spans[:, 1]).astype(np.int64)
        assert 0 < lead < len(spans)
        long_, short = length[:lead], length[lead:]
        assert short.max() * 4 <= long_.max() and short.max() >= 4
        total = length.sum()
        assert 0.15 * total <= short.sum() <= 0.35 * total
        assert len(short) >= 4 * waves            # several short runs per wave
        assert 3 * waves <= lead <= 12 * waves    # a handful of long ones


def test_fixed_shares_are_the_default_plan_for_small_inputs():
    """Inputs of fewer than 100 chunks per wave keep the fixed shares even when a counter is available."""
    for nchan, n in [(16, 1 << 20), (1, 1 << 24), (4096, 1000), (1, 1 << 27), (64, 1 << 21)]:
        a, _ = _plan(nchan, n, 4096, 512, 0)
        b, lead = _plan(nchan, n, 4096, 512, 1)
        assert np.array_equal(a, b) and lead == len(b)


def test_chunk_pool_of_the_several_windows_kernel():
    """The several-windows kernel (runs of one chunk): the first seven eighths of every channel are dealt in stretches,
    the last eighth of every channel is the pool the waves draw from.  Every chunk exactly once; small inputs keep
    the plain plan."""
    for nchan, n, chunk, waves in [(62, 1 << 22, 1022, 2048), (2048, 1 << 20, 1024, 2048), (1, 1 << 26, 1024, 2048),
                                   (3, 99_999_999, 1024, 2048), (512, 1 << 20, 1024, 2048)]:
        spans, lead = _check(nchan, n, waves, chunk, 2)
        cpc = -(-n // chunk)
        assert np.all(spans[:, 2] - spans[:, 1] == 1)
        assert lead == nchan * (cpc * 7 // 8)
        assert np.all(spans[:lead, 1] < cpc * 7 // 8) and np.all(spans[lead:, 1] >= cpc * 7 // 8)
    for nchan, n in [(4, 1 << 20), (1, 1 << 22), (3, 9_999_999)]:   # fewer than four pool chunks per wave: no pool
        spans, lead = _check(nchan, n, 2048, 1024, 2)
        assert lead == len(spans)
