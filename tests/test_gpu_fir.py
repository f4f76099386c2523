"""GPU parity tests: every call goes through the C ABI in libdspx.so (ctypes), and is
checked against the CPU oracle, the committed reference outputs and the literals of the
reference's own tests.

Parity bar (north_star): max|y - y_ref| / max|y_ref| <= 1e-12 for the default FMA mode
(the only difference from the reference is the rounding of fused multiply-adds and, on the
matrix pipe, the order of four neighbouring taps); bit-for-bit for DSPX_FIR_EXACT.
`mfma=None` lets the library pick the pipe, True / False force the matrix / vector pipe.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu

TOL = 1e-12


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def dsp():
    import dsp_amd

    assert dsp_amd.device_count() >= 1, "no MI355X visible"
    return dsp_amd


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available()
    return torch


# ---- the reference's own known-answer tests, through the GPU path ----

def test_reference_test_lfilter_literals(dsp, literals):
    t = literals["lfilter"]
    w = dsp.firwin(t["numtaps"], t["cutoff"])
    assert np.max(np.abs(w - np.array(literals["firwin"]["expected"]))) < literals["firwin"]["tol"]
    y = dsp.lfilter(w, t["data"])
    assert np.max(np.abs(y - np.array(t["expected"]))) < t["tol"]          # dsp.hpp:1583-1588, 1e-15 abs
    y = dsp.lfilter(w, t["data"], exact=True)
    assert np.max(np.abs(y - np.array(t["expected"]))) < t["tol"]


def test_readme_example(dsp, golden):
    w = dsp.firwin(100, 0.5)
    assert np.array_equal(bits(w), bits(golden["taps_100_0p5"]))
    y = dsp.lfilter(w, golden["readme_x"])
    assert scaled_err(y, golden["readme_y"]) <= TOL
    assert np.array_equal(bits(dsp.lfilter(w, golden["readme_x"], exact=True)), bits(golden["readme_y"]))


@pytest.mark.parametrize("key", ["taps_101_0p25", "taps_255_0p25", "taps_4095_0p1"])
def test_seeded_windows_vs_reference_outputs(dsp, oracle, golden, key):
    x = oracle.fill_splitmix(1 << 14, dsp.SEED, 0)
    ref = golden["seeded_y_" + key]
    assert scaled_err(dsp.lfilter(golden[key], x), ref) <= TOL
    assert np.array_equal(bits(dsp.lfilter(golden[key], x, exact=True)), bits(ref))


def test_ragged_and_short_vs_reference_outputs(dsp, oracle, golden):
    x = oracle.fill_splitmix(1000, dsp.SEED, 12345)
    assert np.array_equal(bits(dsp.lfilter(golden["taps_255_0p25"], x[:777], exact=True)), bits(golden["ragged_y_255"]))
    assert np.array_equal(bits(dsp.lfilter(golden["taps_4095_0p1"], x, exact=True)), bits(golden["short_y_4095"]))
    assert scaled_err(dsp.lfilter(golden["taps_4095_0p1"], x), golden["short_y_4095"]) <= TOL


# ---- tap-count and length sweeps against the oracle ----

# (25 / 26: fewer than / exactly one full group of matrix-pipe steps; 281 / 282: the longest filter one window
# serves / the shortest that takes two)
@pytest.mark.parametrize("ntaps", [1, 2, 7, 8, 15, 16, 17, 25, 26, 31, 32, 33, 57, 100, 255, 256, 257, 281, 282, 283, 300, 511, 512, 513,
                                   1041])
def test_tap_count_sweep(dsp, oracle, ntaps):
    rng = np.random.default_rng(ntaps)
    b = rng.standard_normal(ntaps)
    for n in (1, 2, 63, 2047, 2048, 2049, 5000):
        x = rng.standard_normal(n)
        ref = oracle.lfilter(b, [1.0], x)
        assert np.array_equal(bits(dsp.lfilter(b, x, exact=True)), bits(ref)), (ntaps, n)
        for pipe in (None, True, False):
            assert scaled_err(dsp.lfilter(b, x, mfma=pipe), ref) <= TOL, (ntaps, n, pipe)


def test_empty_inputs(dsp):
    b = np.ones(5)
    assert dsp.lfilter(b, np.zeros(0)).size == 0                       # empty x -> empty y (no error)
    y = dsp.lfilter(np.zeros(0), np.arange(10.0))
    assert y.shape == (10,) and np.all(y == 0.0)                       # no numerators -> zeros
    assert dsp.lfilter(b, np.zeros((0, 8))).shape == (0, 8)
    assert dsp.lfilter(b, np.zeros((3, 0))).shape == (3, 0)


def test_invalid_arguments(dsp):
    L = dsp.lib()
    x = np.zeros(16)
    b = np.ones(3)
    assert L.dspx_fir_f64(b.ctypes.data, 3, x.ctypes.data, x.ctypes.data, 16, 0) == -1       # x and y overlap
    assert L.dspx_fir_f64(None, 3, x.ctypes.data, x.ctypes.data, 16, 0) == -1                # NULL taps
    y = np.zeros(16)
    assert L.dspx_fir_f64(b.ctypes.data, 3, x.ctypes.data, y.ctypes.data, 16, 0x80) == -1    # unknown flag
    assert L.dspx_fir_f64(b.ctypes.data, 3, x.ctypes.data, y.ctypes.data, 16, 2 | 4) == -1   # both pipes forced
    assert L.dspx_fir_f64_batched(b.ctypes.data, 3, x.ctypes.data, y.ctypes.data, 2, 8, 4, 0) == -1  # stride < n
    assert b"stride" in L.dspx_last_error()


def test_unaligned_host_and_device_pointers(dsp, oracle, torch):
    """8-byte (not 16-byte) aligned buffers take the scalar load/store variant."""
    b = oracle.firwin(33, [0.3])
    x = oracle.fill_splitmix(6001, 5, 0)
    ref = oracle.lfilter(b, [1.0], x)
    taps = torch.from_numpy(b).cuda()
    big = torch.zeros(6002 * 2, dtype=torch.float64, device="cuda")
    xin = big[1:6002]
    xin.copy_(torch.from_numpy(x))
    out = big[6002 + 1: 6002 + 6002]
    assert xin.data_ptr() % 16 == 8 and out.data_ptr() % 16 == 8
    dsp.fir_device(taps, xin, out=out, exact=True)
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    for pipe in (True, False):
        dsp.fir_device(taps, xin, out=out, mfma=pipe)
        assert scaled_err(out.cpu().numpy(), ref) <= TOL


def test_batched_channels_with_stride(dsp, oracle, torch):
    nchan, n, stride = 5, 3001, 3008
    b = oracle.firwin(101, [0.25])
    x = np.zeros((nchan, stride))
    for c in range(nchan):
        x[c, :n] = oracle.fill_splitmix(n, dsp.SEED, c * n)
    x[:, n:] = np.nan  # padding between channels must never be read into a result
    y = np.full((nchan, stride), -7.0)
    L = dsp.lib()
    dsp._lib.check(L.dspx_fir_f64_batched(b.ctypes.data, b.size, x.ctypes.data, y.ctypes.data, nchan, n, stride, 1))
    for c in range(nchan):
        assert np.array_equal(bits(y[c, :n]), bits(oracle.lfilter(b, [1.0], x[c, :n])))
    assert np.all(y[:, n:] == -7.0)  # and never written
    # the same through device pointers, odd channel stride (scalar variant)
    xd = torch.from_numpy(np.ascontiguousarray(x[:, :3003])).cuda()
    yd = torch.full((nchan, 3003), -7.0, dtype=torch.float64, device="cuda")
    dsp.fir_device(torch.from_numpy(b).cuda(), xd[:, :n], out=yd[:, :n], exact=True)
    yh = yd.cpu().numpy()
    for c in range(nchan):
        assert np.array_equal(bits(yh[c, :n]), bits(oracle.lfilter(b, [1.0], x[c, :n])))
    assert np.all(yh[:, n:] == -7.0)


@pytest.mark.parametrize("pipe", [False, True])
@pytest.mark.parametrize("ntaps", [2, 64, 255, 700])
def test_history_makes_segments_seamless(dsp, oracle, torch, ntaps, pipe):
    """Filtering a stream in segments, each given the previous ntaps-1 samples as history,
    reproduces the one-shot result bit for bit (this is what segment sharding relies on).
    On the vector pipe any cut works; the matrix pipe groups taps in fours by the distance of
    an output from the start of its call, so there the cuts are multiples of 4 (as the host
    sharder's are)."""
    n = 10000
    b = oracle.firwin(ntaps, [0.2]) if ntaps > 2 else np.array([0.75, -0.25])
    x = oracle.fill_splitmix(n, dsp.SEED, 31)
    taps = torch.from_numpy(b).cuda()
    xd = torch.from_numpy(x).cuda()
    whole = dsp.fir_device(taps, xd, mfma=pipe).cpu().numpy()
    assert scaled_err(whole, oracle.lfilter(b, [1.0], x)) <= TOL
    cuts = [0, 4, 4 * (ntaps // 8) + 4, 2048, 4100, 7776, n] if pipe else [0, 1, ntaps // 2 + 1, 2048, 4097, 7777, n]
    out = torch.empty_like(xd)
    for a, e in zip(cuts[:-1], cuts[1:]):
        hist = None
        if a > 0:
            h = np.zeros(ntaps - 1)
            lead = min(ntaps - 1, a)
            h[ntaps - 1 - lead:] = x[a - lead:a]
            hist = torch.from_numpy(h).cuda()
        dsp.fir_device(taps, xd[a:e], out=out[a:e], hist=hist, mfma=pipe)
    assert np.array_equal(bits(out.cpu().numpy()), bits(whole))


def test_host_path_piece_boundaries(dsp, oracle):
    """One channel much longer than a staging piece (2^21 samples): the pieces are stitched with history
    and flow through the threaded pipeline."""
    n = (1 << 24) + 4321
    b = oracle.firwin(31, [0.4])
    x = oracle.fill_splitmix(n, dsp.SEED, 0)
    ref = oracle.lfilter(b, [1.0], x)
    y = dsp.lfilter(b, x, exact=True)
    assert np.array_equal(bits(y), bits(ref))
    y = dsp.lfilter_multi(b, x, ngpus=1)
    assert scaled_err(y, ref) <= TOL


def test_host_path_pipeline_shapes(dsp, oracle):
    """The staged pipeline over the piece shapes it distinguishes: one or two pieces (run in the calling
    thread), a few pieces (one thread each way), many pieces (two threads each way), many short channels per
    piece, a long filter whose history is a large part of every row, and two host threads calling at once.
    The staging sets survive between calls and can be released."""
    import threading

    b31, b4095 = oracle.firwin(31, [0.4]), oracle.firwin(4095, [0.1])
    cases = [(1, 1000, b31), (1, (1 << 21) + 5, b31), (1, 5 * (1 << 21) + 3, b31), (1, 9 * (1 << 21) + 1, b31),
             (3000, 1031, b31), (37, (1 << 17) + 1, b31), (3, (1 << 22) + 7, b4095)]
    for nchan, n, b in cases:
        x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c * n) for c in range(nchan)])
        y = dsp.lfilter(b, x, exact=True)
        for c in sorted({0, nchan // 2, nchan - 1}):
            assert np.array_equal(bits(y[c]), bits(oracle.lfilter(b, [1.0], x[c]))), (nchan, n, c)
        if nchan > 8:   # every channel, cheaply: a channel's output depends on its own input only
            rows = dsp.lfilter(b, x[::-1].copy(), exact=True)[::-1]
            assert np.array_equal(bits(rows), bits(y))
    # streaming state through the same pipeline, chunks longer than a piece
    n = 3 * (1 << 21) + 11
    x = oracle.fill_splitmix(n, dsp.SEED, 0)
    hist = np.zeros(30)
    ys = np.concatenate([dsp.lfilter_stream(b31, hist, x[:n // 2], exact=True),
                         dsp.lfilter_stream(b31, hist, x[n // 2:], exact=True)])
    assert np.array_equal(bits(ys), bits(oracle.lfilter(b31, [1.0], x)))
    # two callers at once on one device: each gets a staging set of its own
    xs = [oracle.fill_splitmix(4 * (1 << 21) + k, dsp.SEED, k << 30) for k in range(2)]
    out = [None, None]

    def call(k):
        out[k] = dsp.lfilter(b31, xs[k], exact=True)

    th = [threading.Thread(target=call, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(2):
        assert np.array_equal(bits(out[k]), bits(oracle.lfilter(b31, [1.0], xs[k])))
    assert 1 <= dsp.release_staging() <= 2
    assert dsp.release_staging() == 0
    assert np.array_equal(bits(dsp.lfilter(b31, xs[0][:5000], exact=True)), bits(out[0][:5000]))


def test_more_workgroups_than_one_launch_holds(dsp, torch):
    """2^24 + 5 one-sample channels need more workgroups than one launch can have (a launch of 2^32 threads or
    more is dropped by the runtime without an error): the call is split, on either pipe."""
    nchan = (1 << 24) + 5
    x = torch.empty((nchan, 1), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(x, dsp.SEED, 0)
    taps = torch.tensor([0.5, 0.25, 0.125], dtype=torch.float64, device="cuda")
    for pipe in (False, True):
        y = torch.full_like(x, 7.0)
        dsp.fir_device(taps, x, out=y, mfma=pipe)
        assert torch.equal(y, 0.5 * x), pipe
    y = dsp.lfilter_iir_device(taps, [1.0, -0.5], x)       # the recursive tail splits the same way
    assert torch.equal(y, 0.5 * x)


def test_splitmix_fill_and_checksum_match_oracle(dsp, oracle, torch):
    n = 100003
    t = torch.empty(n, dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(t, dsp.SEED, 1 << 40)
    host = oracle.fill_splitmix(n, dsp.SEED, 1 << 40)
    assert np.array_equal(bits(t.cpu().numpy()), bits(host))
    assert dsp.checksum(t, 77) == oracle.checksum(host, 77)
    assert dsp.checksum(t[:0].contiguous(), 0) == 0


def test_nan_inf_samples_stay_local(dsp, oracle):
    """A non-finite sample poisons exactly the outputs the reference poisons."""
    b = oracle.firwin(255, [0.25])
    x = oracle.fill_splitmix(5000, 9, 0)
    x[1234] = np.inf
    x[4000] = np.nan
    ref = oracle.lfilter(b, [1.0], x)
    for exact, pipe in ((True, None), (False, False), (False, True), (False, None)):
        y = dsp.lfilter(b, x, exact=exact, mfma=pipe)
        assert np.array_equal(np.isfinite(y), np.isfinite(ref)), (exact, pipe)
        ok = np.isfinite(ref)
        assert scaled_err(y[ok], ref[ok]) <= TOL
    # the same with the bad samples far apart in a long stream and many channels (most tiles stay on the fast path)
    xl = np.stack([oracle.fill_splitmix(30000, 9, c) for c in range(3)])
    xl[0, 17] = np.nan
    xl[1, 29999] = -np.inf
    xl[2, 12345] = np.inf
    refl = np.stack([oracle.lfilter(b, [1.0], xl[c]) for c in range(3)])
    for pipe in (False, True):
        yl = dsp.lfilter(b, xl, mfma=pipe)
        assert np.array_equal(np.isfinite(yl), np.isfinite(refl)), pipe
        ok = np.isfinite(refl)
        assert scaled_err(yl[ok], refl[ok]) <= TOL


@pytest.mark.parametrize("ntaps,where", [(31, 5), (31, 30), (255, 100), (255, 254), (255, 0)])
def test_non_finite_taps_follow_the_reference(dsp, oracle, ntaps, where):
    """An Inf or NaN tap: the reference's truncated sums at the start of a stream never multiply a sample that does
    not exist (dsp.hpp:359-364), so the first `where` outputs stay finite there -- and here, on both pipes and in
    exact mode (bit for bit); the same at the start of every channel.  (A history buffer holds real samples, zeros
    included: the streaming entry points multiply them like any other sample.)"""
    for bad in (np.inf, np.nan):
        b = oracle.firwin(ntaps, [0.25]).copy()
        b[where] = bad
        x = np.stack([oracle.fill_splitmix(9000, 9, c) for c in range(2)])
        ref = np.stack([oracle.lfilter(b, [1.0], x[c]) for c in range(2)])
        assert np.isfinite(ref[:, :where]).all() and not np.isfinite(ref[:, where:]).any()
        for exact, pipe in ((True, None), (False, False), (False, True), (False, None)):
            y = dsp.lfilter(b, x, exact=exact, mfma=pipe)
            assert np.array_equal(np.isfinite(y), np.isfinite(ref)), (bad, exact, pipe)
            if where:
                assert scaled_err(y[:, :where], ref[:, :where]) <= TOL
            if exact:
                assert np.array_equal(bits(y), bits(ref))


@pytest.mark.parametrize("ntaps", [1, 2, 33, 255])
def test_streaming_chunks_reproduce_one_call(dsp, oracle, ntaps):
    """dspx_fir_f64_stream: history carried between chunks (also chunks shorter than the filter)."""
    rng = np.random.default_rng(ntaps)
    b = rng.standard_normal(ntaps)
    nchan, n = 3, 7000
    x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c * n) for c in range(nchan)])
    ref = np.stack([oracle.lfilter(b, [1.0], x[c]) for c in range(nchan)])
    hist = np.zeros((nchan, max(ntaps - 1, 0)))
    out = []
    cuts = [0, 1, 2, 10, 11, 300, 2348, 2349, 6999, n]
    for a, e in zip(cuts[:-1], cuts[1:]):
        out.append(dsp.lfilter_stream(b, hist, np.ascontiguousarray(x[:, a:e]), exact=True))
    assert np.array_equal(bits(np.concatenate(out, axis=1)), bits(ref))
    if ntaps > 1:
        assert np.array_equal(bits(hist), bits(x[:, n - (ntaps - 1):]))


@pytest.mark.parametrize("ntaps", [1, 7, 100, 255, 256, 257, 700])
def test_single_precision_is_bit_identical(dsp, oracle, ntaps):
    """dsp::lfilter<float> (dspx_fir_f32, dspx_fir_f32_from_f64): float sums of products formed in float (containers
    of float) or in double (containers of double), bit for bit the reference loop; ragged lengths, several channels
    and a history through the device entry points."""
    import torch

    rng = np.random.default_rng(ntaps)
    b = rng.standard_normal(ntaps)
    for n in (1, 2, 255, 2047, 2048, 2049, 5000):
        x = rng.standard_normal(n)
        want = oracle.lfilter_f32_from_f64(b, x)
        assert np.array_equal(dsp.lfilter_f32(b, x).view(np.uint32), want.view(np.uint32)), n
        b32, x32 = b.astype(np.float32), x.astype(np.float32)
        want = oracle.lfilter_f32(b32, x32)
        assert np.array_equal(dsp.lfilter_f32(b32, x32).view(np.uint32), want.view(np.uint32)), n
    # device entry points: 3 channels, and a second half that continues the first through a history
    n, cut = 6001, 2500
    x = rng.standard_normal((3, n))
    for dt, orc in ((torch.float64, oracle.lfilter_f32_from_f64), (torch.float32, oracle.lfilter_f32)):
        xh = x.astype(np.float32) if dt == torch.float32 else x
        bh = b.astype(np.float32) if dt == torch.float32 else b
        want = np.stack([orc(bh, xh[c]) for c in range(3)])
        xd, bd = torch.from_numpy(xh).cuda(), torch.from_numpy(bh).cuda()
        got = dsp.fir_f32_device(bd, xd)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
        if ntaps > 1 and cut >= ntaps - 1:
            second = dsp.fir_f32_device(bd, xd[:, cut:].contiguous(), hist=xd[:, cut - (ntaps - 1):cut].contiguous())
            assert np.array_equal(second.cpu().numpy().view(np.uint32), want[:, cut:].view(np.uint32))


def test_single_precision_reference_literals(dsp, oracle, literals):
    """The reference's test_lfilter in its single-precision form (dsp.hpp:1590-1597): lfilter<float> over the double
    taps and data, within its 1e-6; an Inf tap does not reach the truncated sums at the start."""
    t = literals["lfilter"]
    w = dsp.firwin(t["numtaps"], t["cutoff"])
    y = dsp.lfilter_f32(w, np.array(t["data"]))
    assert np.max(np.abs(y.astype(np.float64) - np.array(t["expected"]))) < t["tol_f32"]
    w2 = w.copy()
    w2[40] = np.inf
    x = oracle.fill_splitmix(3000, dsp.SEED, 0)
    ref = oracle.lfilter_f32_from_f64(w2, x)
    got = dsp.lfilter_f32(w2, x)
    assert np.isfinite(ref[:40]).all() and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


# ---- the two forms of the matrix-pipe kernel ----

@pytest.mark.parametrize("nchan,n,ntaps", [(1, 513, 101), (3, 5000, 255), (2, 100, 63), (5, 70_001, 57), (300, 2048, 96),
                                           (40, 500_003, 101), (3, (1 << 23) + 5, 255), (1, (1 << 25) + 1000, 63),
                                           (130, 1 << 18, 160), (1, 1 << 20, 281),
                                           # one window with the 576-slot halo (282 .. 569 taps)
                                           (1, 513, 282), (3, 5000, 300), (40, 500_003, 320), (2, (1 << 23) + 5, 569),
                                           # several windows per chunk (570 taps and more)
                                           (3, 70_001, 570), (5, 70_001, 700), (1, 1 << 20, 1023), (7, 2049, 4095),
                                           (1, 300, 4095), (1, (1 << 22) + 77, 4095),
                                           # ... over several stretches of the chip's resident waves, channels ending inside them
                                           (3, 3_000_001, 600), (5000, 1000, 570)])
def test_wave_autonomous_form_equals_workgroup_tiled_form(dsp, oracle, torch, nchan, n, ntaps):
    """Over aligned buffers the matrix pipe runs its wave-autonomous kernels (every wave walks a run of 512-output
    chunks with an LDS image of its own: one window per chunk and the halo carried in LDS up to 569 taps, several
    windows per chunk beyond); DSPX_FIR_TILED forces the workgroup-tiled kernel.  Same products in the same order: the
    same bits -- with and without history, for runs of one chunk (small inputs) and of several (the large ones here),
    channels that end inside a chunk, and against the oracle on windows."""
    taps_h = dsp.firwin(ntaps, 0.25)
    taps = torch.from_numpy(taps_h).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(x, dsp.SEED, 11)
    hist = torch.empty((nchan, ntaps - 1), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(hist, dsp.SEED, 1 << 40)
    for h in (None, hist):
        wave = dsp.fir_device(taps, x, hist=h, mfma=True)
        tiled = dsp.fir_device(taps, x, hist=h, mfma=True, tiled=True)
        assert torch.equal(wave.view(torch.int64), tiled.view(torch.int64))
        # windows against the oracle: the start, the end, and a stretch in the middle of the first and last channel
        for c in {0, nchan - 1}:
            xc = x[c].cpu().numpy()
            lead = h[c].cpu().numpy() if h is not None else np.zeros(0)
            for lo in {0, max(0, n // 2 - 3000), max(0, n - 6000)}:
                hi = min(n, lo + 6000)
                back = min(lo, ntaps - 1)
                seg = np.concatenate([lead, xc[:hi]]) if lo < ntaps else xc[lo - back:hi]
                ref = oracle.lfilter(taps_h, [1.0], seg)
                ref = ref[len(lead) + lo:] if lo < ntaps else ref[back:]
                if lo >= ntaps or h is not None or lo == 0:
                    got = wave[c, lo:hi].cpu().numpy()
                    assert scaled_err(got, ref[:hi - lo]) <= TOL, (c, lo)


# ---- narrow element types: lfilter<double> over float containers, lfilter<float> with a recursive part ----

def test_narrow_types_match_reference_outputs(dsp, oracle, golden):
    """The reference forms numerators[j] * x[i-j] in the type the containers' elements promote to and only then converts
    to T (dsp.hpp:363); its tail y_i -= denominators[j] * y[i-j] runs in the type denominators[j] * T promotes to
    (dsp.hpp:367).  Stored outputs of the reference itself, bit for bit, then random cases against the oracle."""
    x64 = oracle.fill_splitmix(3000, dsp.SEED, 4242)
    x32 = x64.astype(np.float32)
    got = dsp.lfilter_f64_from_f32(golden["taps_101_0p25"].astype(np.float32), x32)
    assert np.array_equal(bits(got), bits(golden["f64_from_f32_y"]))
    b32, a32 = golden["iir4_b"].astype(np.float32), golden["iir4_a"].astype(np.float32)
    assert np.array_equal(dsp.lfilter_f32(b32, x32, a=a32).view(np.uint32), golden["f32_iir_y"].view(np.uint32))
    assert np.array_equal(dsp.lfilter_f32(golden["iir4_b"], x64, a=golden["iir4_a"]).view(np.uint32),
                          golden["f32_from_f64_iir_y"].view(np.uint32))
    rng = np.random.default_rng(17)
    for nb, na, n in ((1, 2, 1), (3, 1, 5000), (40, 9, 2049), (300, 3, 70_001), (7, 5, 4096)):
        b, x = rng.uniform(-1, 1, nb), rng.uniform(-1, 1, n)
        a = np.concatenate([[rng.uniform(0.5, 2)], rng.uniform(-0.2, 0.2, na - 1)])
        bf, af, xf = b.astype(np.float32), a.astype(np.float32), x.astype(np.float32)
        assert np.array_equal(bits(dsp.lfilter_f64_from_f32(bf, xf)), bits(oracle.lfilter_f64_from_f32(bf, xf))), (nb, n)
        assert np.array_equal(dsp.lfilter_f32(bf, xf, a=af).view(np.uint32), oracle.lfilter_f32_iir(bf, af, xf).view(np.uint32))
        assert np.array_equal(dsp.lfilter_f32(b, x, a=a).view(np.uint32), oracle.lfilter_f32_from_f64_iir(b, a, x).view(np.uint32))
    with pytest.raises(dsp.DspxError, match="at most"):
        dsp.lfilter_f32(np.ones(2, np.float32), np.ones(10, np.float32), a=np.ones(11, np.float32))


def test_cpp_header_narrow_instantiations(dsp, oracle, golden):
    """The same through include/dsp.hpp, called as a user of the reference would: lfilter<double>(vector<float>, ...),
    lfilter<float>(b, a, x) over float and over double containers, and lfilter<double> over float taps and samples with
    double denominators (float products, double sums, double tail)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "tests", "cpp")], check=True, capture_output=True)
    b, a = golden["iir4_b"], golden["iir4_a"]
    x64 = oracle.fill_splitmix(3000, dsp.SEED, 4242)
    bf, af, xf = b.astype(np.float32), a.astype(np.float32), x64.astype(np.float32)
    # the program reads the samples as doubles and narrows them itself: hand it the float values
    inp = "\n".join(float(v).hex() for v in xf.astype(np.float64))
    args = [float(v).hex() for v in bf.astype(np.float64)] + ["/"] + [float(v).hex() for v in af.astype(np.float64)]
    r = subprocess.run([os.path.join(root, "tests", "cpp", "dropin_check"), "narrow", *args], input=inp, capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    toks, out, i = r.stdout.split(), {}, 0
    while i < len(toks):
        name, cnt = toks[i], int(toks[i + 1])
        out[name] = np.array([float.fromhex(v) for v in toks[i + 2:i + 2 + cnt]])
        i += 2 + cnt
    assert np.array_equal(bits(out["f64_from_f32"]), bits(oracle.lfilter_f64_from_f32(bf, xf)))
    assert np.array_equal(out["f32_iir"].astype(np.float32).view(np.uint32), oracle.lfilter_f32_iir(bf, af, xf).view(np.uint32))
    wide_b, wide_a, wide_x = bf.astype(np.float64), af.astype(np.float64), xf.astype(np.float64)
    assert np.array_equal(out["f32_from_f64_iir"].astype(np.float32).view(np.uint32),
                          oracle.lfilter_f32_from_f64_iir(wide_b, wide_a, wide_x).view(np.uint32))
    # float products, double sums, double tail: the FIR part of the oracle's float-product form, then its double tail
    part = oracle.lfilter_f64_from_f32(bf, xf)
    want = oracle.lfilter([1.0], wide_a, part)
    assert np.array_equal(bits(out["f64_from_f32_iir"]), bits(want))
