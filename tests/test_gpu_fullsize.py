"""Parity at the benchmark's full sizes (BASELINE.json configs 2-4), where the CPU oracle
cannot filter everything: sampled windows against the oracle, plus size-independent
properties that must hold bit for bit -- sharding invariance (a checksum of the whole output
is the same however the work is cut into channel ranges or history-carrying segments, which
is what 1/2/4/8-GPU runs rely on), shift invariance across tile boundaries, and the unit DC
gain of a scaled low-pass design."""
import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu
TOL = 1e-12


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def env():
    import torch

    import dsp_amd

    assert torch.cuda.is_available() and dsp_amd.device_count() >= 1
    return torch, dsp_amd


def check_window(dsp, oracle, taps_h, y_row, chan_offset, start, win=4096):
    """Compares y_row[start:start+win] with the oracle run on regenerated input."""
    ntaps = taps_h.size
    lead = min(start, ntaps - 1)
    xs = oracle.fill_splitmix(lead + win, dsp.SEED, chan_offset + start - lead)
    hist = None
    if start > 0:
        hist = np.zeros(ntaps - 1)
        hist[ntaps - 1 - lead:] = xs[:lead]
    ref = oracle.fir_hist(taps_h, hist, xs[lead:])
    got = y_row[start:start + win].cpu().numpy()
    assert scaled_err(got, ref) <= TOL, (chan_offset, start, scaled_err(got, ref))


def test_batched_255_taps_4096x2e20(env, oracle, golden):
    torch, dsp = env
    nchan, n = 4096, 1 << 20
    taps_h = dsp.firwin(255, 0.25)
    assert np.array_equal(bits(taps_h), bits(golden["taps_255_0p25"]))
    taps = torch.from_numpy(taps_h).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp.fill_splitmix(x, dsp.SEED, 0)
    dsp.fir_device(taps, x, out=y)
    # sampled windows: start / interior tile boundary / end of first, middle and last channels;
    # channel c+1 must start from zero state, not from channel c's tail
    for c in (0, 1, 2047, 4095):
        for start in (0, 2048 - 100, (1 << 19) + 5, n - 4096):
            check_window(dsp, oracle, taps_h, y[c], c * n, start)
    # the first 2^14 outputs of channel 0 are the committed reference output itself
    assert scaled_err(y[0, :1 << 14].cpu().numpy(), golden["seeded_y_taps_255_0p25"]) <= TOL
    whole = dsp.checksum(y, 0)
    # channel sharding as 8 GPUs would do it: 8 separate launches over contiguous channel ranges
    y.zero_()
    for g in range(8):
        lo, hi = nchan * g // 8, nchan * (g + 1) // 8
        dsp.fir_device(taps, x[lo:hi], out=y[lo:hi])
    assert dsp.checksum(y, 0) == whole
    # checksum of checksums: per-range sums add up to the whole
    parts = sum(dsp.checksum(y[nchan * g // 8: nchan * (g + 1) // 8], (nchan * g // 8) * n) for g in range(8))
    assert parts & (2**64 - 1) == whole
    # exact mode agrees with FMA mode to rounding on a few channels, and is itself bit-exact vs the oracle
    ye = dsp.fir_device(taps, x[:2], exact=True)
    assert scaled_err(y[:2].cpu().numpy(), ye.cpu().numpy()) <= TOL
    xs = oracle.fill_splitmix(1 << 16, dsp.SEED, 0)
    assert np.array_equal(bits(ye[0, :1 << 16].cpu().numpy()), bits(oracle.lfilter(taps_h, [1.0], xs)))


@pytest.mark.parametrize("numtaps,cutoff,log2n,key", [(101, 0.25, 28, "taps_101_0p25"), (4095, 0.1, 26, "taps_4095_0p1")])
def test_long_stream_segment_sharding(env, oracle, golden, numtaps, cutoff, log2n, key):
    torch, dsp = env
    n = 1 << log2n
    taps_h = dsp.firwin(numtaps, cutoff)
    assert np.array_equal(bits(taps_h), bits(golden[key]))
    taps = torch.from_numpy(taps_h).cuda()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp.fill_splitmix(x, dsp.SEED, 0)
    dsp.fir_device(taps, x, out=y)
    assert scaled_err(y[:1 << 14].cpu().numpy(), golden["seeded_y_" + key]) <= TOL
    for start in (0, 2048 - 7, n // 8 - 2000, n // 2 + 12345, n - 4096):
        check_window(dsp, oracle, taps_h, y, 0, start)
    whole = dsp.checksum(y, 0)
    # segment sharding as 8 GPUs would do it: each segment sees the ntaps-1 samples before it
    y2 = torch.zeros_like(x)
    for g in range(8):
        lo, hi = n * g // 8, n * (g + 1) // 8
        hist = x[lo - (numtaps - 1): lo] if g else None
        dsp.fir_device(taps, x[lo:hi], out=y2[lo:hi], hist=hist)
    assert dsp.checksum(y2, 0) == whole
    # uneven cuts that do not fall on tile boundaries: multiples of 4 for the default (matrix) pipe,
    # which groups taps in fours by an output's distance from the start of its call; any cut on the vector pipe
    for pipe, cuts in ((None, [0, 4, 4 * (numtaps // 8), 5000, 4 * (n // 12) + 4, n - 4, n]),
                       (False, [0, 1, numtaps // 2, 5000, n // 3 + 1, n - 3, n])):
        ref_sum = whole if pipe is None else dsp.checksum(dsp.fir_device(taps, x, mfma=False), 0)
        y2.zero_()
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if lo == 0:
                hist = None
            else:
                lead = min(lo, numtaps - 1)
                hist = torch.zeros(numtaps - 1, dtype=torch.float64, device="cuda")
                hist[numtaps - 1 - lead:] = x[lo - lead: lo]
            dsp.fir_device(taps, x[lo:hi], out=y2[lo:hi], hist=hist, mfma=pipe)
        assert dsp.checksum(y2, 0) == ref_sum, pipe


def test_stream_longer_than_2e32_samples(env, oracle):
    """Sample indices beyond 32 bits: one stream of 2^32 + 5 samples (32 GiB in, 32 GiB out), windows around
    2^31, 2^32 and the ragged end against the oracle, on both pipes."""
    torch, dsp = env
    n = (1 << 32) + 5
    taps_h = dsp.firwin(31, 0.3)
    taps = torch.from_numpy(taps_h).cuda()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp.fill_splitmix(x, dsp.SEED, 0)
    for pipe in (None, True):
        y.zero_()
        dsp.fir_device(taps, x, out=y, mfma=pipe)
        for start in (0, (1 << 31) - 2048, (1 << 32) - 4096, n - 4096):
            check_window(dsp, oracle, taps_h, y, 0, start)
    del x, y


def test_shift_invariance_across_tiles(env):
    """Delaying the input by k samples delays the output by k samples, bit for bit, for delays
    that move every sample to a different place in its tile."""
    torch, dsp = env
    n = 1 << 22
    taps = torch.from_numpy(dsp.firwin(255, 0.25)).cuda()
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(x, dsp.SEED, 42)
    for pipe, delays in ((False, (1, 7, 1000, 2047, 2049)), (True, (4, 1000, 2044, 4100))):
        y = dsp.fir_device(taps, x, mfma=pipe)
        for k in delays:  # the matrix pipe is invariant under delays that are multiples of 4
            xs = torch.zeros(n + k, dtype=torch.float64, device="cuda")
            xs[k:] = x
            ys = dsp.fir_device(taps, xs, mfma=pipe)
            assert torch.equal(ys[k:], y) and bool((ys[:k] == 0).all()), (pipe, k)


def test_unit_dc_gain_everywhere(env):
    """firwin(..., scale=True) low-pass taps sum to one: a constant input settles to the same
    constant.  On the vector pipe every steady-state sample of every channel has the same bits;
    on the matrix pipe the bits depend only on the sample's position modulo 16 (the row of the
    16x16 product it falls in)."""
    torch, dsp = env
    taps_h = dsp.firwin(255, 0.25)
    taps = torch.from_numpy(taps_h).cuda()
    x = torch.ones((64, 1 << 20), dtype=torch.float64, device="cuda")
    yv = dsp.fir_device(taps, x, mfma=False)
    steady = yv[:, 254:]
    v = float(steady[0, 0])
    assert abs(v - 1.0) < 1e-14
    assert bool((steady == v).all())
    # on the vector pipe the warm-up samples are the running sums of the taps in ascending order
    # (an FMA with 1.0 is exact)
    assert np.array_equal(bits(yv[3, :254].cpu().numpy()), bits(np.cumsum(taps_h)[:254]))
    ym = dsp.fir_device(taps, x, mfma=True)
    assert float((ym[:, 254:] - 1.0).abs().max()) < 1e-14
    body = ym[:, 256:256 + 16 * 65000].reshape(64, -1, 16)
    assert bool((body == body[0, 0]).all())
    assert scaled_err(ym[3, :254].cpu().numpy(), np.cumsum(taps_h)[:254]) <= TOL


def test_cwt_bank_65_scales_2e22(env, oracle, golden):
    """BASELINE config 5 at full size: windows of several rows against the oracle run on a
    sub-signal (interior coefficients of a sub-signal's transform equal those of the whole
    signal's: the same products are added in the same order), and row-sharding invariance."""
    torch, dsp = env
    psi, wx = golden["morlet10_psi"], golden["morlet10_x"]
    n = 1 << 22
    scales = np.arange(1, 66, dtype=np.float64)
    bank = dsp.CwtBank(scales, psi, wx)
    sig = torch.empty(n, dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(sig, dsp.SEED, 0)
    out = bank.apply(sig)
    assert bool(torch.isfinite(out).all())
    margin, win = 1100, 3000          # margin > the longest kernel (1041 taps)
    for a in (0, 2040 * 7 - 1500, n // 2 + 777, n - win - 2 * margin):
        xs = oracle.fill_splitmix(win + 2 * margin, dsp.SEED, a)
        for s in (1, 2, 16, 33, 65):
            ref = oracle.cwt(xs, [float(s)], psi, wx)[0]
            lo = 0 if a == 0 else margin          # the left edge of the signal is a real edge
            got = out[s - 1, a + lo: a + margin + win].cpu().numpy()
            assert scaled_err(got, ref[lo: margin + win]) <= TOL, (a, s)
    # the right edge of the signal
    xs = oracle.fill_splitmix(win + margin, dsp.SEED, n - win - margin)
    for s in (1, 65):
        ref = oracle.cwt(xs, [float(s)], psi, wx)[0]
        assert scaled_err(out[s - 1, n - win:].cpu().numpy(), ref[margin:]) <= TOL
    whole = dsp.checksum(out, 0)
    # scale sharding as 8 GPUs would do it: each device's rows in a bank of their own
    dev, _ = dsp.cwt_plan(scales, 1024, wx, 8)
    out2 = torch.zeros_like(out)
    for g in range(8):
        rows = [r for r in range(65) if dev[r] == g]
        sub = dsp.CwtBank(scales[rows], psi, wx)
        part = sub.apply(sig)
        out2[rows] = part
        sub.close()
    assert dsp.checksum(out2, 0) == whole
    bank.close()
