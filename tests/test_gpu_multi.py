"""The multi-device host sharder (dspx_fir_f64_multi, dspx_lfilter_f64_multi, dspx_cwt_f64_multi) with more than
one logical device: a thread, a staging engine and a plan per device, all running at once.  On a one-GPU machine
the library's test hook dspx_test_alias_devices (not part of include/dspx.h) maps the logical devices onto the GPU
that is there (the code path is the same, only hipSetDevice's argument differs); the same checks run un-aliased
when the machine has two GPUs or more.
Results must be the same bits for every device count, and must meet the parity bar against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import cwt_row_tolerance, scaled_err

pytestmark = pytest.mark.gpu
TOL = 1e-12
COUNTS = (2, 3, 8)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.fixture(scope="module")
def dsp():
    import torch

    import dsp_amd

    assert torch.cuda.is_available() and dsp_amd.device_count() >= 1
    return dsp_amd


@pytest.fixture(params=["aliased", "physical"])
def counts(request, dsp):
    """Device counts to run with; switches the library's aliasing test hook on and off around the test."""
    hook = dsp.lib().dspx_test_alias_devices   # a test hook: exported, but not declared in include/dspx.h
    hook.restype, hook.argtypes = C.c_int, [C.c_int]
    if request.param == "aliased":
        old = hook(1)
        try:
            yield COUNTS
        finally:
            hook(old)
    else:
        have = dsp.device_count()
        if have < 2:
            pytest.skip("needs two GPUs or more")
        old = hook(0)
        try:
            yield tuple(g for g in COUNTS if g <= have)
        finally:
            hook(old)


def lfilter_multi(dsp, b, x, ngpus, exact=False, mfma=None):
    from dsp_amd.api import _flags, _np_f64
    from dsp_amd._lib import check, lib

    b, x = _np_f64(b), _np_f64(x)
    x2 = x.reshape(1, -1) if x.ndim == 1 else x
    y = np.full_like(x2, np.nan)
    check(lib().dspx_fir_f64_multi(b.ctypes.data, b.size, x2.ctypes.data, y.ctypes.data, x2.shape[0], x2.shape[1],
                                   x2.shape[1], int(ngpus), _flags(exact, mfma)))
    return y.reshape(x.shape)


@pytest.mark.parametrize("ntaps", [31, 255])
def test_fir_channel_ranges(dsp, oracle, counts, ntaps):
    """nchan >= G: contiguous channel ranges per device."""
    taps = dsp.firwin(ntaps, 0.25)
    nchan, n = 21, 200_003
    x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c * n) for c in range(nchan)])
    ref = np.stack([oracle.lfilter(taps, [1.0], x[c]) for c in range(nchan)])
    one = lfilter_multi(dsp, taps, x, 1)
    assert scaled_err(one, ref) <= TOL
    for g in counts:
        assert np.array_equal(bits(lfilter_multi(dsp, taps, x, g)), bits(one)), g
        assert np.array_equal(bits(lfilter_multi(dsp, taps, x, g, exact=True)), bits(ref)), g


@pytest.mark.parametrize("ntaps,nchan,n", [(101, 1, 3_000_001), (255, 2, 1_500_007), (4095, 1, 700_001)])
def test_fir_segments_with_history(dsp, oracle, counts, ntaps, nchan, n):
    """nchan < G: every device filters a segment of every channel, with the ntaps-1 samples before it as history."""
    taps = dsp.firwin(ntaps, 0.2)
    x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c * n) for c in range(nchan)])
    one = lfilter_multi(dsp, taps, x, 1)
    ref = np.stack([oracle.lfilter(taps, [1.0], x[c]) for c in range(nchan)])
    assert scaled_err(one, ref) <= TOL
    for g in counts:
        if g <= nchan:
            continue
        assert np.array_equal(bits(lfilter_multi(dsp, taps, x, g)), bits(one)), g
        assert np.array_equal(bits(lfilter_multi(dsp, taps, x, g, exact=True)), bits(ref)), g


def test_recursive_filter_shards_whole_channels(dsp, oracle, counts):
    """A recursive filter's state is serial along a channel: with fewer channels than devices only nchan devices
    work, each on whole channels whose state travels from staged piece to staged piece."""
    from scipy import signal

    for (b, a), nchan, n in ((signal.butter(4, 0.2), 3, 150_000), (signal.butter(2, 0.1), 11, 70_001),
                             (signal.butter(6, 0.3), 2, 400_003)):
        x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c << 24) for c in range(nchan)])
        ref = np.stack([oracle.lfilter(b, a, x[c]) for c in range(nchan)])
        one = dsp.lfilter_iir(b, a, x, ngpus=1)
        assert scaled_err(one, ref) <= TOL
        for g in counts:
            assert np.array_equal(bits(dsp.lfilter_iir(b, a, x, ngpus=g)), bits(one)), (g, nchan)
            assert np.array_equal(bits(dsp.lfilter_iir(b, a, x, ngpus=g, exact=True)), bits(ref)), (g, nchan)


def test_wavelet_bank_shards_scales(dsp, oracle, counts):
    psi, wx = dsp.integrate_wavelet(10)
    scales = np.arange(1, 31, dtype=np.float64)
    sig = oracle.fill_splitmix(20_011, dsp.SEED, 3)
    ref = oracle.cwt(sig, scales, psi, wx)
    one = dsp.cwt(sig, scales, psi, wx, ngpus=1)
    for r in range(scales.size):
        assert scaled_err(one[r], ref[r]) <= cwt_row_tolerance(oracle, sig, scales[r], psi, wx), r
    for g in counts:
        assert np.array_equal(bits(dsp.cwt(sig, scales, psi, wx, ngpus=g)), bits(one)), g


def test_device_count_beyond_the_machine_is_refused(dsp):
    os.environ["DSPX_ALIAS_DEVICES"] = "1"   # no environment variable can widen what ngpus means
    try:
        with pytest.raises(dsp.DspxError, match="exceeds visible devices"):
            lfilter_multi(dsp, np.ones(3), np.ones(100), dsp.device_count() + 1)
    finally:
        os.environ.pop("DSPX_ALIAS_DEVICES", None)


def test_recursive_channels_of_several_pieces_do_not_mix_states(dsp, oracle):
    """Several channels that each span several staged pieces, on one device: a piece starts from the final state
    of the piece before it in ITS channel (each staging slot keeps its own piece's state), whatever the streams'
    relative timing.  Exact mode is bit-identical to the reference loop, the default mode within tolerance."""
    from scipy import signal

    for (b, a), nchan, n in ((signal.butter(4, 0.2), 3, 150_000), (signal.butter(2, 0.1), 4, 140_001),
                             (signal.butter(3, 0.15), 12, 70_000)):
        x = np.stack([oracle.fill_splitmix(n, dsp.SEED, c << 24) for c in range(nchan)])
        ref = np.stack([oracle.lfilter(b, a, x[c]) for c in range(nchan)])
        for _ in range(3):
            assert np.array_equal(bits(dsp.lfilter_iir(b, a, x, exact=True)), bits(ref))
            assert scaled_err(dsp.lfilter_iir(b, a, x), ref) <= TOL


@pytest.mark.parametrize("order,wn", [(2, 0.2), (6, 0.3)])
def test_tile_scan_in_several_passes(dsp, oracle, order, wn):
    """More than 4096 tiles in one channel of one device call (n > 2^24): the tile-state scan runs in passes that
    carry the state from one pass to the next.  With and without incoming output history."""
    import torch
    from scipy import signal

    b, a = signal.butter(order, wn)
    n = (1 << 24) + 3 * 4096 + 5
    xh = oracle.fill_splitmix(n, dsp.SEED, 77)
    ref = oracle.lfilter(b, a, xh)
    x = torch.from_numpy(xh).cuda()
    bt = torch.from_numpy(b).cuda()
    y = dsp.lfilter_iir_device(bt, a, x)
    assert scaled_err(y.cpu().numpy(), ref) <= TOL
    cut = 4099
    second = dsp.lfilter_iir_device(bt, a, x[cut:].contiguous(), xhist=x[cut - (b.size - 1):cut].contiguous(),
                                    yhist=y[cut - (a.size - 1):cut].contiguous())
    assert scaled_err(second.cpu().numpy(), ref[cut:]) <= TOL


def test_bench_two_ranks_without_a_launcher():
    """`python bench.py --gpus 2` started plainly: the script spawns its ranks itself.  Both ranks share the one GPU
    of the test machine and meet over gloo (RCCL refuses two ranks on one device); on a machine with two GPUs the
    same command runs with RCCL and one GPU per rank."""
    import json
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() < 2:
        env.update(DSPX_BENCH_BACKEND="gloo", DSPX_BENCH_ONE_DEVICE="1")
    def run(*extra):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", *extra],
                           env=env, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return lines[0]

    # one rank: the checksum of the whole output, which no sharding may change
    single = {w: run("--gpus", "1", "--workload", w, "--no-cpu-baseline") for w in ("small255", "long4095")}
    # two ranks, defaults: the named workload sharded (channel ranges), a weak-scaling loop beside it
    out = run("--gpus", "2", "--workload", "small255")
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["sharding"] == "channels" and out["config"]["nchan_per_gpu"] == 32
    assert out["output_checksum"] == single["small255"]["output_checksum"]
    assert out["weak"]["value"] > 0 and out["weak"]["nchan_per_gpu"] == 64
    assert out["allgather"].get("matches_per_rank_checksums") is True, out["allgather"]
    # a single stream is cut into segments with history: same checksum again
    seg = run("--gpus", "2", "--workload", "long4095", "--no-weak")
    assert seg["config"]["sharding"] == "segments with history" and seg["scaling"] == "strong"
    assert seg["output_checksum"] == single["long4095"]["output_checksum"]
    # weak scaling on request: every rank its own batch
    weak = run("--gpus", "2", "--workload", "small255", "--scaling", "weak")
    assert weak["scaling"] == "weak" and weak["config"]["nchan_per_gpu"] == 64 and "weak" not in weak
    assert weak["allgather"].get("matches_per_rank_checksums") is True, weak["allgather"]


def _visible_gpus() -> int:
    import torch

    return torch.cuda.device_count()   # (counting devices does not initialise the GPU)


@pytest.mark.skipif(_visible_gpus() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")
def test_rccl_allgather_two_ranks(tmp_path):
    """dspx_comm_create / dspx_allgather_f64_dev with two ranks, one fresh process and one GPU each (started before
    anything in them has touched a GPU; the id travels through a file): every rank receives both shards in rank
    order, the result equals a torch.distributed all-gather over RCCL, and a call made with another device current is
    refused."""
    import json
    import socket
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(here, "_rccl_pair_worker.py"), str(tmp_path / "id"),
                                       str(tmp_path / f"result{r}.json")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, p in enumerate(procs):
        assert p.returncode == 0, outs[r][-3000:]
        res = json.loads((tmp_path / f"result{r}.json").read_text())
        assert res == {"rank": r, "ranks": 2, "gathered_equals_expected": True, "equals_torch_gather": True,
                       "wrong_device_refused": True}


def test_rccl_allgather_entry_point_single_rank(dsp):
    """dspx_comm_* / dspx_allgather_f64_dev: a communicator of one rank (all a one-GPU machine can hold; RCCL refuses
    two ranks on one device) gathers a shard onto itself.  The N-rank exchange runs in bench.py --gpus N."""
    import torch

    ident = dsp.Comm.unique_id()
    assert len(ident) == 128
    comm = dsp.Comm(ident, 1, 0)
    assert dsp.lib().dspx_comm_ranks(comm._h) == 1
    y = torch.empty((7, 3001), dtype=torch.float64, device="cuda")
    dsp.fill_splitmix(y, dsp.SEED, 5)
    full = comm.allgather(y)
    torch.cuda.synchronize()
    assert full.shape == y.shape and torch.equal(full, y)
    comm.close()
    with pytest.raises(dsp.DspxError):
        dsp.Comm(ident, 2, 5)
