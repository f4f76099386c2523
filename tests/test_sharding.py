"""The multi-GPU path on CPU: properties of the sharding plan the host sharder executes
(dspx_fir_plan, no GPU needed) and a world_size-2 gloo run in which each rank plays one
device of that plan."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dsp():
    from dsp_amd import build

    build.build()
    import dsp_amd

    return dsp_amd


@pytest.mark.parametrize("nchan,n,g", [(4096, 1 << 20, 8), (4096, 1 << 20, 3), (1, 1 << 26, 8), (3, 1000, 8),
                                       (7, 5, 2), (1, 1, 4), (2, (1 << 24) + 17, 2), (100, 1 << 22, 5)])
def test_plan_partitions_the_work(dsp, nchan, n, g):
    pieces = dsp.fir_plan(nchan, n, g)
    cover = {}
    per_dev = [0] * g
    for dev, c, cnt, s0, ln in pieces:
        assert 0 <= dev < g and cnt >= 1 and ln >= 1 and c + cnt <= nchan and s0 + ln <= n
        assert cnt * ln <= max(1 << 24, ln)            # a staged piece stays bounded
        per_dev[dev] += cnt * ln
        for ch in range(c, c + cnt):
            cover.setdefault(ch, []).append((s0, s0 + ln))
    assert sorted(cover) == list(range(nchan))
    for ch, spans in cover.items():                    # every channel exactly covered, no overlap
        spans.sort()
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    if nchan * n >= g * 1000:                          # balanced to within one channel / a few samples per device
        unit = n if nchan >= g else 8 * nchan
        assert max(per_dev) - min(per_dev) <= unit


def test_plan_device_ranges_are_contiguous(dsp):
    pieces = dsp.fir_plan(10, 100, 4)
    assert [p[0] for p in pieces] == sorted(p[0] for p in pieces)
    chans = [[c for p in pieces if p[0] == d for c in range(p[1], p[1] + p[2])] for d in range(4)]
    assert sum(chans, []) == list(range(10))
    segs = dsp.fir_plan(1, 1000, 4)                    # fewer channels than devices: segments
    assert [(p[3], p[4]) for p in segs] == [(0, 248), (248, 252), (500, 248), (748, 252)]   # starts are multiples of 4
    assert all(p[3] % 4 == 0 for p in dsp.fir_plan(2, (1 << 25) + 123, 7))
    assert dsp.lib().dspx_fir_plan(1, 10, 0, None, 0) == -1


def test_bench_shard_arithmetic():
    sys.path.insert(0, ROOT)
    import bench

    total = 0
    for r in range(8):
        nch, c0, units = bench.shard("batched255", r, 8, "strong")
        assert (nch, c0, units) == (512, 512 * r, 4096 << 20)
        total += nch
    assert total == 4096
    assert bench.shard("batched255", 3, 4, "weak") == (4096, 3 * 4096, 4 * (4096 << 20))
    assert bench.shard("batched255", 0, 1, "weak") == bench.shard("batched255", 0, 1, "strong")
    # a single stream over several GPUs: contiguous segments, cuts on multiples of 4
    for n, world in ((1 << 28, 8), (1 << 26, 3), (1001, 4)):
        spans = [bench.segment(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] and a[1] % 4 == 0 for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 8


def test_world_size_2_gloo(dsp):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_shard_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("REPORT ")][-1]
    rep = json.loads(line[len("REPORT "):])
    for name in ("channels", "segments"):
        assert rep[name]["bit_identical"], rep
        assert rep[name]["covered"] == rep[name]["total"]
    assert rep["strong_spans"] == [[0, 2048], [2048, 4096]] and rep["strong_units"] == 4096 << 20
    assert rep["weak"] == [[4096, 0, 2 * (4096 << 20)], [4096, 4096, 2 * (4096 << 20)]]
    assert rep["max_over_ranks"] == 2.0
    assert rep["allgather_6"] and rep["allgather_5"]


def run_bench(extra_env, *argv, timeout=300):
    import json
    import subprocess
    import sys

    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it: the parent spawns N ranks (before touching any GPU),
    exactly one JSON line comes back, and ranks_seen is counted by a reduction over the process group."""
    r, lines = run_bench({"DSPX_BENCH_DRY": "1"}, "--gpus", "2", "--workload", "small255", "--steps", "3")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2 and lines[0]["steps"] == 3
    r, lines = run_bench({"DSPX_BENCH_DRY": "1"}, "--gpus", "4")
    assert r.returncode == 0 and lines[0]["ranks_seen"] == 4
    # one process when N = 1: no spawning, the line is the process's own
    r, lines = run_bench({"DSPX_BENCH_DRY": "1"}, "--gpus", "1")
    assert r.returncode == 0 and lines[0]["n_gpus"] == 1 and lines[0]["ranks_seen"] == 1


def test_bench_forwards_a_failing_rank():
    """A rank that dies takes the run down with a non-zero exit code instead of leaving the others in a barrier."""
    r, lines = run_bench({"DSPX_BENCH_DRY": "fail1", "DSPX_BENCH_FAIL_GRACE": "2"}, "--gpus", "2", timeout=120)
    assert r.returncode != 0 and not lines
    assert "a rank failed" in r.stderr


def test_bench_parent_does_not_import_torch_before_spawning():
    """The spawning parent must never initialise a GPU: importing bench and parsing arguments pulls in neither torch
    nor the HIP library."""
    import subprocess
    import sys

    code = ("import sys; sys.argv=['bench.py','--gpus','2']; import bench; bench.parse_args(); "
            "assert 'torch' not in sys.modules and 'dsp_amd' not in sys.modules, sorted(sys.modules)")
    subprocess.run([sys.executable, "-c", code], cwd=ROOT, check=True, timeout=60)
