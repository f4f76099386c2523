"""dspx_lfilter_f64_dev_noalloc: the allocation-free variant of dspx_lfilter_f64_dev (its tail scratch is a device
array inside the library instead of a caller workspace).  Same kernels, so the results must be the same bits as
dspx_lfilter_f64_dev's on the same inputs; the second call of every pair is smaller than the first, so stale
contents of the shared scratch (tile states and matrix powers of the larger call) would show."""
import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu


def bits(t):
    return np.ascontiguousarray(t.cpu().numpy(), dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def dsp():
    import torch

    import dsp_amd

    assert torch.cuda.is_available() and dsp_amd.device_count() >= 1
    return dsp_amd


def run_both(dsp, b, a, x, **kw):
    import torch

    bt = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).cuda()
    want = dsp.lfilter_iir_device(bt, a, x, **kw)
    got = torch.full_like(x, float("nan"))
    dsp.lfilter_iir_device(bt, a, x, out=got, noalloc=True, **kw)
    torch.cuda.synchronize()
    return got, want


def test_two_calls_in_a_row_second_smaller(dsp, oracle):
    """Tile-parallel form (few channels: tile states in the scratch), then the same filter over a shorter input,
    then another order over a still shorter one; each against dspx_lfilter_f64_dev bit for bit and against the
    oracle to the parity bar."""
    import torch
    from scipy import signal

    calls = [(signal.butter(4, 0.2), 3, 70001), (signal.butter(4, 0.2), 2, 9000), (signal.butter(2, 0.1), 1, 4097),
             ((signal.firwin(40, 0.3), np.array([1.0, -0.6, 0.2])), 2, 5000)]   # 40 taps: separate FIR pass first
    for (b, a), nchan, n in calls:
        xh = np.stack([oracle.fill_splitmix(n, dsp.SEED, c * n) for c in range(nchan)])
        x = torch.from_numpy(xh).cuda()
        got, want = run_both(dsp, b, a, x)
        assert np.array_equal(bits(got), bits(want)), (nchan, n, len(a))
        ref = np.stack([oracle.lfilter(b, a, xh[c]) for c in range(nchan)])
        assert scaled_err(got.cpu().numpy(), ref) <= 1e-12


def test_exact_mode_and_histories(dsp, oracle):
    """Sequential (bit-exact) tail and incoming input / output histories go through the same entry point."""
    import torch
    from scipy import signal

    b, a = signal.butter(3, 0.2)
    n, cut = 30000, 12288
    xh = np.stack([oracle.fill_splitmix(n, dsp.SEED, c << 20) for c in range(4)])
    x = torch.from_numpy(xh).cuda()
    got, want = run_both(dsp, b, a, x, exact=True)
    ref = np.stack([oracle.lfilter(b, a, xh[c]) for c in range(4)])
    assert np.array_equal(bits(got), bits(want)) and np.array_equal(bits(got), ref.view(np.uint64))
    whole, _ = run_both(dsp, b, a, x)
    second, second_want = run_both(dsp, b, a, x[:, cut:].contiguous(), xhist=x[:, cut - (b.size - 1):cut].contiguous(),
                                   yhist=whole[:, cut - (a.size - 1):cut].contiguous())
    assert np.array_equal(bits(second), bits(second_want))
    assert scaled_err(second.cpu().numpy(), ref[:, cut:]) <= 1e-12


def test_many_channels_need_no_tile_states(dsp, oracle):
    """From 1024 channels on a workgroup walks its channel: only the matrix powers live in the scratch, so sizes far
    beyond 16 MiB of tile states are accepted; then a smaller, tile-parallel call right after it."""
    import torch
    from scipy import signal

    b, a = signal.butter(2, 0.1)
    x = torch.empty((1500, 20000), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(x, dsp.SEED, 0)
    got, want = run_both(dsp, b, a, x)
    assert np.array_equal(bits(got), bits(want))
    got, want = run_both(dsp, b, a, x[:5, :8191].contiguous())
    assert np.array_equal(bits(got), bits(want))


def test_sizes_beyond_the_scratch_are_refused(dsp):
    """8 poles, 900 channels of 2^21 samples: 900 * 512 tiles * 8 states = 29 MB of tile states, more than the 16 MiB
    scratch: refused with a clear message before anything is launched (the output stays untouched)."""
    import torch
    from scipy import signal

    b, a = signal.butter(8, 0.4)
    bt = torch.from_numpy(b).cuda()
    nchan, n = 900, 1 << 21
    need = int(dsp.lib().dspx_lfilter_workspace_bytes(a.size, nchan, n))
    assert need > 16 << 20
    x = torch.zeros((nchan, 1), dtype=torch.float64, device="cuda").expand(nchan, n)   # never read: the call is refused
    out = torch.zeros((4,), dtype=torch.float64, device="cuda")
    rc = dsp.lib().dspx_lfilter_f64_dev_noalloc(bt.data_ptr(), bt.numel(), a.ctypes.data, a.size, x.data_ptr(), n, None, 0,
                                                None, 0, out.data_ptr(), n, nchan, n, 0, None)
    assert rc == -1 and b"static scratch" in dsp.lib().dspx_last_error()
    assert not out.any()


def test_out_is_required(dsp):
    import torch

    x = torch.zeros(100, dtype=torch.float64, device="cuda")
    with pytest.raises(ValueError, match="needs `out`"):
        dsp.lfilter_iir_device(torch.ones(2, dtype=torch.float64, device="cuda"), np.array([1.0, -0.5]), x, noalloc=True)


CAPTURE_CASES = [("tile-parallel tail, fused numerator", 3, 70001, 4, 0), ("walk form (1024 channels or more)", 1500, 20000, 2, 0),
                 ("FIR pass first (40 taps), then the tail", 2, 50000, 2, 40), ("sequential exact tail", 4, 30000, 3, -1)]


@pytest.mark.parametrize("what,nchan,n,order,ntaps", CAPTURE_CASES, ids=[c[0] for c in CAPTURE_CASES])
def test_runs_under_stream_capture_and_allocates_nothing(dsp, oracle, what, nchan, n, order, ntaps):
    """The contract itself: after one warm-up call on the device (which looks the scratch's address up) the entry point
    is recorded inside a HIP stream capture -- where hipMalloc, hipFree and every synchronising call are errors that
    invalidate the capture -- and the graph is replayed twice on fresh inputs.  Each replay must give the bits of
    dspx_lfilter_f64_dev on the same input; and a plain call leaves the device's free memory where it was."""
    import torch
    from scipy import signal

    exact = ntaps < 0
    b, a = signal.butter(order, 0.2)
    if ntaps > 0:
        b = signal.firwin(ntaps, 0.3)
    bt = torch.from_numpy(np.ascontiguousarray(b, dtype=np.float64)).cuda()
    x = torch.empty((nchan, n), dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    dsp.fill_splitmix(x, dsp.SEED, 1)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        dsp.lfilter_iir_device(bt, a, x, out=y, noalloc=True, exact=exact)   # warm-up: resolves the scratch on this device
        side.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        dsp.lfilter_iir_device(bt, a, x, out=y, noalloc=True, exact=exact)
        side.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before, "the call changed the device's free memory"
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            dsp.lfilter_iir_device(bt, a, x, out=y, noalloc=True, exact=exact)
    for seed_offset in (1 << 30, 1 << 31):
        dsp.fill_splitmix(x, dsp.SEED, seed_offset)
        y.fill_(float("nan"))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        want = dsp.lfilter_iir_device(bt, a, x, exact=exact)
        assert np.array_equal(bits(y), bits(want)), what
    xh = x[0].cpu().numpy()
    assert scaled_err(y[0].cpu().numpy(), oracle.lfilter(b, a, xh)) <= 1e-12
