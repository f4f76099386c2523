"""GPU tests of how the wave-autonomous FIR kernels (one window per chunk, several windows per chunk, wavelet bank)
hand out their runs of chunks (dsp_amd/csrc/fir_kernels.hip, FirRuns): in fixed shares per wave, or drawn from the
counters of the launch's stream.  Which wave computes a chunk
never changes the chunk's arithmetic, so the two must agree bit for bit, launch after launch (the counters are left at
zero by the kernel itself), on several streams at once, and on more streams than a device has sets of counters (the
later ones keep the fixed shares until a set is given back).  Calls go through the C ABI; dspx_test_run_shares is the library's test hook (exported,
not part of include/dspx.h): 1 forces fixed shares, 2 draws whatever the input's size, 0 is the default (drawn from
128 chunks per wave on)."""
import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu


def bits(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.fixture(scope="module")
def env():
    import torch

    import dsp_amd

    assert torch.cuda.is_available() and dsp_amd.device_count() >= 1
    hook = dsp_amd.lib().dspx_test_run_shares
    yield dsp_amd, torch, hook
    hook(0)


SHAPES = [(1, 300_000, 101), (3, 70_001, 255), (40, 50_003, 63), (1, (1 << 21) + 77, 200), (300, 2048, 96),
          (17, 33_333, 320), (2, 1 << 20, 569), (5, 1000, 57), (1, 513, 101), (64, 1 << 14, 128)]


def test_drawn_runs_equal_fixed_shares_bit_for_bit(env, oracle):
    """Small and medium inputs with the draw forced on (hook 2): every run start, channel end and history case of the
    kernels, twice in a row on the same stream (the second launch finds the counters the first one left)."""
    dsp, torch, hook = env
    rng = np.random.default_rng(11)
    for nchan, n, nt in SHAPES:
        taps_h = dsp.firwin(nt, 0.25)
        taps = torch.from_numpy(taps_h).cuda()
        xh = rng.uniform(-1, 1, (nchan, n))
        hh = rng.uniform(-1, 1, (nchan, nt - 1))
        x, hist = torch.from_numpy(xh).cuda(), torch.from_numpy(hh).cuda()
        for h in (hist, None):
            hook(1)
            fixed = dsp.fir_device(taps, x, hist=h, mfma=True)
            hook(2)
            for _ in range(2):
                drawn = dsp.fir_device(taps, x, hist=h, mfma=True)
                assert np.array_equal(bits(fixed), bits(drawn)), (nchan, n, nt, h is not None)
        hook(0)
        if nchan * n <= 300_000:
            ref = oracle.lfilter(taps_h, [1.0], xh[0])
            assert scaled_err(drawn[0].cpu().numpy(), ref) <= 1e-12   # (the last round ran without history)


@pytest.mark.parametrize("nchan,log2n,nt", [(256, 20, 255), (1, 28, 101), (8, 25, 63), (2048, 17, 128)])
def test_default_draw_on_large_inputs(env, nchan, log2n, nt):
    """Inputs large enough for the default to draw (128 chunks per wave or more): three launches in a row against the
    fixed shares, bit for bit -- long runs, the short runs of the last quarter, one counter or eight."""
    dsp, torch, hook = env
    n = 1 << log2n
    taps = torch.from_numpy(dsp.firwin(nt, 0.25)).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(x)
    hook(1)
    want = dsp.checksum(dsp.fir_device(taps, x))
    hook(0)
    y = torch.empty_like(x)
    for _ in range(3):
        y.zero_()
        dsp.fir_device(taps, x, out=y)
        assert dsp.checksum(y) == want


@pytest.mark.parametrize("nchan,log2n,nt", [(64, 20, 700), (1, 26, 1023), (512, 17, 2047), (16, 22, 570)])
def test_several_windows_kernel_draws_its_chunks(env, nchan, log2n, nt):
    """Filters of 570 to 3072 taps (several windows per chunk): the chunks after a wave's first two are drawn.  Three
    launches in a row against fixed shares, bit for bit."""
    dsp, torch, hook = env
    taps = torch.from_numpy(dsp.firwin(nt, 0.2)).cuda()
    x = torch.empty((nchan, 1 << log2n), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(x)
    hook(1)
    want = dsp.checksum(dsp.fir_device(taps, x))
    y = torch.empty_like(x)
    for mode in (0, 2):
        hook(mode)
        for _ in range(3):
            y.zero_()
            dsp.fir_device(taps, x, out=y)
            assert dsp.checksum(y) == want
    hook(0)


def test_wavelet_bank_draws_its_chunks(env):
    """The wavelet bank on the several-windows kernel: a row's bits depend on how the row is cut into chunks, never on
    which wave computes a chunk -- drawn chunks against fixed shares, bit for bit, apply after apply."""
    dsp, torch, hook = env
    psi, wx = dsp.integrate_wavelet(10)
    n = 1 << 20
    buf = torch.empty(n + 2, dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(buf)
    for scales, sig in ((np.arange(1, 66, dtype=np.float64), buf[:n]), (np.array([3.5, 9.0, 20.25, 41.0, 64.0]), buf[:n]),
                        (np.arange(20, 66, dtype=np.float64), buf[1:n + 1])):   # (the last signal is only 8-byte aligned)
        bank = dsp.CwtBank(scales, psi, wx)
        out = torch.empty((bank.rows, n), dtype=torch.float64, device="cuda")
        hook(1)
        bank.apply(sig, out=out)
        want = out.clone()
        for mode in (0, 2):
            hook(mode)
            for _ in range(2):
                out.zero_()
                bank.apply(sig, out=out)
                both_nan = torch.isnan(out) & torch.isnan(want)
                assert bool(((out.view(torch.int64) == want.view(torch.int64)) | both_nan).all())
        hook(0)
        bank.close()


def test_streams_have_counters_of_their_own(env):
    """Launches that overlap on four streams, several rounds: each stream draws from its own counters."""
    dsp, torch, hook = env
    nt = 101
    taps = torch.from_numpy(dsp.firwin(nt, 0.25)).cuda()
    xs = [torch.empty((2, 1 << 24), dtype=torch.float64, device="cuda") for _ in range(4)]
    for i, x in enumerate(xs):
        dsp.fill_splitmix(x, offset=i << 26)
    hook(1)
    want = [dsp.checksum(dsp.fir_device(taps, x)) for x in xs]
    hook(0)
    streams = [torch.cuda.Stream() for _ in xs]
    ys = [torch.empty_like(x) for x in xs]
    torch.cuda.synchronize()
    for _ in range(3):
        for st, x, y in zip(streams, xs, ys):
            with torch.cuda.stream(st):
                dsp.fir_device(taps, x, out=y)
        torch.cuda.synchronize()
        assert [dsp.checksum(y) for y in ys] == want
        for y in ys:
            y.zero_()
        torch.cuda.synchronize()


def test_more_streams_than_counters(env):
    """A device has 64 sets of counters and a stream keeps its set: streams beyond that use the fixed shares until a
    set is given back (the library does that for the streams it creates itself; here the test hook does).  Same
    results either way."""
    import ctypes as C

    dsp, torch, hook = env
    forget = dsp.lib().dspx_test_forget_stream
    forget.restype, forget.argtypes = None, [C.c_void_p]
    taps = torch.from_numpy(dsp.firwin(63, 0.25)).cuda()
    x = torch.empty((4, 1 << 16), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(x)
    hook(1)
    want = bits(dsp.fir_device(taps, x, mfma=True))
    hook(2)
    streams = [torch.cuda.Stream() for _ in range(80)]
    try:
        for round_ in range(2):
            ys = []
            for st in streams:
                with torch.cuda.stream(st):
                    ys.append(dsp.fir_device(taps, x, mfma=True))
                if round_ == 0:
                    st.synchronize()
            torch.cuda.synchronize()
            for y in ys:
                assert np.array_equal(bits(y), want)
    finally:
        torch.cuda.synchronize()
        for st in streams:
            forget(st.cuda_stream)
        hook(0)
