"""Worker for tests/test_sharding.py: launched by torch.distributed.run with world_size 2 on
CPU (gloo).  Each rank plays one device of the multi-GPU sharding plan: it filters the
pieces the plan assigns to it (the oracle stands in for the GPU kernel, with the same
history rule), the ranks combine their outputs, and rank 0 compares the result with a
one-shot filter of the whole input.  No data-path collective exists in the product; the
all_reduce here only gathers results for checking."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
import dsp_amd  # noqa: E402
from oracle.pyoracle import SEED, Oracle  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc = Oracle()
    report = {}
    for name, (nchan, n, ntaps) in {"channels": (5, 3000, 33), "segments": (1, 20001, 64)}.items():
        b = orc.firwin(ntaps, [0.3])
        x = np.stack([orc.fill_splitmix(n, SEED, c * n) for c in range(nchan)])
        y = np.zeros_like(x)
        mine = 0
        for dev, c, cnt, s0, ln in dsp_amd.fir_plan(nchan, n, world):
            if dev != rank:
                continue
            for ch in range(c, c + cnt):
                hist = None
                if s0 > 0:
                    hist = np.zeros(ntaps - 1)
                    lead = min(ntaps - 1, s0)
                    hist[ntaps - 1 - lead:] = x[ch, s0 - lead:s0]
                y[ch, s0:s0 + ln] = orc.fir_hist(b, hist, x[ch, s0:s0 + ln])
                mine += ln
        t = torch.from_numpy(y)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # shards are disjoint, the rest is zero
        cnt_t = torch.tensor([mine], dtype=torch.int64)
        dist.all_reduce(cnt_t)
        if rank == 0:
            ref = np.stack([orc.lfilter(b, [1.0], x[c]) for c in range(nchan)])
            report[name] = {"bit_identical": bool(np.array_equal(t.numpy().view(np.uint64), ref.view(np.uint64))),
                            "covered": int(cnt_t.item()), "total": nchan * n}
    # the optional exchange step: all-gather of channel shards, even (6 channels) and uneven (5 channels)
    for nchan in (6, 5):
        n, ntaps = 777, 21
        b = orc.firwin(ntaps, [0.3])
        ref = np.stack([orc.lfilter(b, [1.0], orc.fill_splitmix(n, SEED, c * n)) for c in range(nchan)])
        lo, hi = dsp_amd.channel_range(nchan, rank, world)
        full = dsp_amd.allgather_channels(torch.from_numpy(ref[lo:hi].copy()), nchan)
        ok = torch.tensor([int(np.array_equal(full.numpy().view(np.uint64), ref.view(np.uint64)))])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)   # every rank must hold the whole result
        report[f"allgather_{nchan}"] = bool(ok.item())
    # the benchmark's rank -> shard arithmetic and its max-over-ranks reduction
    nch, c0, units = bench.shard("batched255", rank, world, "strong")
    span = torch.tensor([c0, c0 + nch], dtype=torch.int64)
    spans = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(spans, span)
    tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        report["strong_spans"] = [s.tolist() for s in spans]
        report["strong_units"] = units
        report["weak"] = [list(bench.shard("batched255", r, world, "weak")) for r in range(world)]
        report["max_over_ranks"] = float(tmax.item())
        print("REPORT " + json.dumps(report), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
