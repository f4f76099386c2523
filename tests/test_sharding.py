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
