#!/usr/bin/env python3
"""Benchmark of the FIR hot path: dsp::firwin taps -> dsp::lfilter(b, 1.0, x) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the device-resident filter (dspx_fir_f64_dev) over one batch of
synthetic input that already sits in HBM.  The default workload is the configuration the
headline metric is quoted on: 4096 channels x 2^20 samples, 255-tap firwin(255, 0.25).
With N > 1 there is one process (rank) per GPU: either started by torch.distributed.run, or, when
no launcher set WORLD_SIZE, by this script itself, which then only spawns the N ranks, forwards rank
0's JSON line and returns the worst exit code (it never touches a GPU).  By default the named workload
is sharded over the ranks (strong scaling: contiguous channel ranges, or, for a single stream, contiguous
segments with ntaps-1 samples of history; channels and segments are independent, so there is no
data-path collective) and `output_checksum` -- the position-dependent checksum of the whole output, summed
over the ranks -- must be the same for every N.  A short loop in which every rank filters a whole batch of
its own is reported beside it as `weak` (`--scaling weak` makes that the headline instead).  After the
timed region the ranks all-gather a small slice of their results over RCCL (reported apart as `allgather`,
never part of `value`).

Rank 0 prints ONE JSON line with the metric, the roofline of the FIR kernel (per-launch HIP
events on the launch stream) and, at N = 1, the CPU baseline (the reference's own single-threaded
loop when oracle/_ref holds its compiled header, else the oracle's restatement of it, timed on a bounded
sample of the same workload).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (nchan, n, numtaps, cutoff, description)
    "batched255": (4096, 1 << 20, 255, 0.25,
                   "batched multichannel 4096 x 2^20 samples, 255-tap firwin(255, 0.25) (BASELINE configs[2])"),
    "long101": (1, 1 << 28, 101, 0.25, "single stream 2^28 samples, 101-tap firwin(101, 0.25) (BASELINE configs[1])"),
    "long4095": (1, 1 << 26, 4095, 0.1, "single stream 2^26 samples, 4095-tap firwin(4095, 0.1) (BASELINE configs[3])"),
    "batched31": (4096, 1 << 20, 31, 0.25,
                  "batched multichannel 4096 x 2^20 samples, 31-tap firwin(31, 0.25) (the HBM-bound regime of the metric)"),
    "small255": (64, 1 << 20, 255, 0.25, "reduced batch 64 x 2^20 samples, 255 taps (development only)"),
}
# dsp::cwt bank (BASELINE configs[4]): Morlet table of 2^10 points, integer scales 1..65, 2^22 samples.
CWT = {"nscales": 65, "n": 1 << 22, "precision": 10,
       "desc": "dsp::cwt Morlet bank, 65 integer scales over 2^22 samples (BASELINE configs[4])"}
PEAK_FP64_TFLOPS = 78.6   # MI355X FP64: 256 CU x 64 FMA/clk x 2 x 2.4 GHz (vector; the FP64 MFMA peak is the same)
PEAK_HBM_GBS = 8000.0     # MI355X HBM3E


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="batched255", choices=sorted(WORKLOADS) + ["cwt65", "iir_biquad"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1 only.  strong (the default): the workload the metric names is sharded over the ranks "
                         "(channel ranges, or segments with history for a single stream), and a short weak-scaling loop "
                         "is reported beside it; weak: every rank filters a whole batch of its own")
    ap.add_argument("--no-weak", action="store_true", help="with strong scaling: skip the weak-scaling sub-record")
    ap.add_argument("--exact", action="store_true", help="DSPX_FIR_EXACT arithmetic (separate mul/add)")
    ap.add_argument("--vector", action="store_true", help="force the vector-pipe kernel (DSPX_FIR_VECTOR)")
    ap.add_argument("--tiled", action="store_true",
                    help="matrix pipe: the workgroup-tiled form of the kernel (DSPX_FIR_TILED), for A/B against the default")
    ap.add_argument("--allgather", action="store_true",
                    help="with --scaling strong and N > 1: also time an all-gather of the WHOLE result (reported apart); "
                         "without it a 64-channel slice per rank is gathered")
    ap.add_argument("--no-allgather", action="store_true", help="skip the all-gather sub-record")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=1 << 27, help="samples of the workload the CPU baseline filters")
    return ap.parse_args()


def shard(workload: str, rank: int, world: int, scaling: str):
    """(channels of this rank, global index of its first channel, outputs per step over all ranks)."""
    nchan_total, n, _, _, _ = WORKLOADS[workload]
    if scaling == "strong":
        lo, hi = nchan_total * rank // world, nchan_total * (rank + 1) // world   # == dsp_amd.channel_range
        return hi - lo, lo, nchan_total * n
    return nchan_total, rank * nchan_total, nchan_total * world * n


def segment(n: int, rank: int, world: int):
    """Samples [lo, hi) of a single stream that `rank` filters when the stream is sharded over `world` GPUs:
    cuts on multiples of 4, so that every output's arithmetic is what one launch would do (the matrix pipe
    groups taps in fours by an output's distance from the start of its call)."""
    def cut(g):
        return n if g == world else (n * g // world) & ~3
    return cut(rank), cut(rank + 1)


def cpu_baseline(taps: np.ndarray, n: int, nsamples: int, y_gpu) -> dict:
    """Times the CPU loop (single thread) on whole channels of the workload, and, since the CPU result
    is at hand, checks the GPU output of the same channels against it.  The loop is the reference's own
    dsp::lfilter when oracle/_ref/libdsp_ref.so (the unmodified header compiled by oracle/Makefile) is
    in the tree, else oracle/fir_oracle.c, the restatement that is bit-identical to it.  A second pass
    runs the same channels on all host cores of this process's share (one channel per thread at a time)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle.pyoracle import SEED, Oracle, Reference, reference_available

    orc = Oracle()
    cpu, kind, what = orc, "port", "oracle/fir_oracle.c gcc -O2 -ffp-contract=off (bit-identical to the reference loop)"
    if reference_available():
        try:
            cpu, kind = Reference(), "reference"
            what = "dsp::lfilter of the unmodified reference header (oracle/_ref, g++ -std=c++20 -O2 -ffp-contract=off)"
        except OSError:
            pass
    per = min(n, nsamples)
    nchan = max(1, min(nsamples // per, y_gpu.shape[0]))
    x = [orc.fill_splitmix(per, SEED, c * n) for c in range(nchan)]
    ref = []
    t0 = time.perf_counter()
    for c in range(nchan):
        ref.append(cpu.lfilter(taps, [1.0], x[c]))
    dt = time.perf_counter() - t0
    worst = 0.0
    for c in range(nchan):
        got = y_gpu[c, :per].cpu().numpy()
        worst = max(worst, float(np.max(np.abs(got - ref[c])) / max(float(np.max(np.abs(ref[c]))), 1e-300)))
    res = {"value": nchan * per / dt / 1e9, "unit": "GS/s", "cores": 1, "kind": kind,
           "sample": f"{nchan} channel(s) x {per} samples of the same workload, {taps.size} taps, {dt:.1f} s, {what}",
           "gpu_vs_cpu_max_scaled_err": worst}
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16")), nchan)
    if cores > 1:
        # the reference is single-threaded; independent channels are what a caller would spread over cores
        with ThreadPoolExecutor(cores) as pool:
            t0 = time.perf_counter()
            list(pool.map(lambda c: cpu.lfilter(taps, [1.0], x[c]), range(nchan)))
            dt_all = time.perf_counter() - t0
        res["all_cores"] = {"value": nchan * per / dt_all / 1e9, "unit": "GS/s", "cores": cores,
                            "sample": f"the same {nchan} channels, one channel per thread at a time, {dt_all:.1f} s"}
    return res


def run_cwt(args, rank, world, dev, dist, red_dev, ranks_seen=1):
    """The wavelet-bank workload: every rank transforms its own 2^22-sample signal (weak scaling)."""
    import torch

    import dsp_amd

    n, nscales = CWT["n"], CWT["nscales"]
    psi, wx = dsp_amd.integrate_wavelet(CWT["precision"])
    scales = np.arange(1, nscales + 1, dtype=np.float64)
    bank = dsp_amd.CwtBank(scales, psi, wx)
    sig = torch.empty(n, dtype=torch.float64, device=dev)
    out = torch.empty((nscales, n), dtype=torch.float64, device=dev)
    dsp_amd.fill_splitmix(sig, dsp_amd.SEED, rank * n)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        bank.apply(sig, out=out, exact=args.exact, mfma=False if args.vector else None, tiled=args.tiled)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        bank.apply(sig, out=out, exact=args.exact, mfma=False if args.vector else None, tiled=args.tiled)
        ev[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]))
    t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank != 0:
        return
    dt_max, kern_ms = float(t[0]), float(t[1])
    traffic, traffic_src = load_traffic("cwt65")
    if args.tiled or args.vector or args.exact:
        traffic, traffic_src = None, "the profiled figure belongs to the default kernels"
    flops = 2.0 * bank.total_taps * n            # 2 * 34385 per input sample
    bytes_alg = (8.0 + 8.0 * nscales) * n        # read once, one output row per scale
    sec = kern_ms / 1e3
    res = {
        "metric": "dsp::cwt input GS/s (65-scale Morlet bank)", "value": world * n * args.steps / dt_max / 1e9,
        "unit": "GS/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic (splitmix64 uniform [-1,1), generated on device, resident in HBM)",
        "config": {"workload": CWT["desc"], "samples": n, "scales": nscales, "total_taps": bank.total_taps,
                   "arithmetic": "exact (mul+add)" if args.exact else "fma"},
        "roofline": {"bound": "mfma", "achieved": flops / sec / 1e12, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                     "frac": flops / sec / 1e12 / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "fir_f64_kernel (bank mode, vector pipe)" if args.vector or args.exact else
                               ("fir_f64_mfma4_kernel" if args.tiled else "fir_f64_mfma4wm_kernel (wave-autonomous form)") +
                               " + fir_f64_kernel (bank mode; rows of >= 52 taps on the matrix pipe)",
                     "kernel_ms": kern_ms,
                     "note": "FP64 FMA; algorithmic flops = 2 * sum of per-scale tap counts per input sample"},
        "roofline_other": {"bound": "hbm", "achieved": bytes_alg / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": bytes_alg / sec / 1e9 / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src},
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle.pyoracle import SEED, Oracle, Reference, reference_available

        orc = Oracle()
        cpu, kind, what = orc, "port", "oracle/cwt_oracle.c gcc -O2 -ffp-contract=off (bit-identical to the reference loop)"
        if reference_available():
            try:
                cpu, kind = Reference(), "reference"
                what = "dsp::cwt of the unmodified reference header (oracle/_ref, g++ -std=c++20 -O2 -ffp-contract=off)"
            except OSError:
                pass
        m = 1 << 14                               # bounded sample: the first 2^14 samples, all 65 scales
        xs = orc.fill_splitmix(m, SEED, 0)
        t0 = time.perf_counter()
        ref = cpu.cwt(xs, scales, psi, wx)
        cdt = time.perf_counter() - t0
        # the transform of a prefix differs from the prefix of the transform near its right edge only
        keep = m - 1100
        got = out[:, :keep].cpu().numpy()
        err = max(float(np.max(np.abs(got[r] - ref[r, :keep])) / np.max(np.abs(ref[r, :keep]))) for r in range(nscales))
        res["cpu_baseline"] = {"value": m / cdt / 1e9, "unit": "GS/s", "cores": 1, "kind": kind,
                               "sample": f"first {m} samples, all {nscales} scales, {cdt:.1f} s, {what}",
                               "gpu_vs_cpu_max_scaled_err": err}
    print(json.dumps(res), flush=True)


def butter2(wn: float):
    """Second-order Butterworth low-pass (bilinear transform), cutoff wn with Nyquist = 1."""
    k = np.tan(np.pi * wn / 2.0)
    norm = 1.0 / (1.0 + np.sqrt(2.0) * k + k * k)
    b0 = k * k * norm
    return np.array([b0, 2.0 * b0, b0]), np.array([1.0, 2.0 * (k * k - 1.0) * norm, (1.0 - np.sqrt(2.0) * k + k * k) * norm])


def run_iir(args, rank, world, dev, dist, red_dev, ranks_seen=1):
    """lfilter with a recursive part (reference dsp.hpp:366-368): 4096 channels x 2^20 samples through a biquad,
    every rank its own batch (weak scaling)."""
    import torch

    import dsp_amd

    nchan, n = 4096, 1 << 20
    b, a = butter2(0.2)
    bt = torch.from_numpy(b).to(dev)
    x = torch.empty((nchan, n), dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    dsp_amd.fill_splitmix(x, dsp_amd.SEED, rank * nchan * n)

    def step():
        dsp_amd.lfilter_iir_device(bt, a, x, out=y, exact=args.exact)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]))
    t = torch.tensor([dt, kern_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank != 0:
        return
    dt_max, kern_ms = float(t[0]), float(t[1])
    sec = kern_ms / 1e3
    bytes_alg = 16.0 * nchan * n
    traffic, traffic_src = load_traffic("iir_biquad")
    res = {
        "metric": "FP64 lfilter output GS/s (3-tap numerator, 2 poles)", "value": world * nchan * n * args.steps / dt_max / 1e9,
        "unit": "GS/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic (splitmix64 uniform [-1,1), generated on device, resident in HBM)",
        "config": {"workload": "lfilter(b, a, x) with a recursive part: 4096 x 2^20 samples, Butterworth biquad (cutoff 0.2)",
                   "nchan_per_gpu": nchan, "samples_per_channel": n, "nb": 3, "na": 3,
                   "arithmetic": "exact (sequential, mul+sub)" if args.exact else "fma, block-parallel"},
        "roofline": {"bound": "hbm", "achieved": bytes_alg / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": bytes_alg / sec / 1e9 / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "iir_seq_kernel after fir_f64_kernel" if args.exact else
                               "iir_walk_kernel (one workgroup per channel, FIR part computed in the same pass)",
                     "kernel_ms": kern_ms,
                     "note": "algorithmic bytes = 16 per output (8 read + 8 written)"},
    }
    if world == 1 and not args.no_cpu_baseline:
        from oracle.pyoracle import SEED, Oracle, Reference, reference_available

        orc = Oracle()
        cpu, kind, what = orc, "port", "oracle/fir_oracle.c gcc -O2 -ffp-contract=off (bit-identical to the reference loop)"
        if reference_available():
            try:
                cpu, kind = Reference(), "reference"
                what = "dsp::lfilter of the unmodified reference header (oracle/_ref, g++ -std=c++20 -O2 -ffp-contract=off)"
            except OSError:
                pass
        m = 64                                   # bounded sample: the first 64 channels
        xs = [orc.fill_splitmix(n, SEED, c * n) for c in range(m)]
        t0 = time.perf_counter()
        ref = [cpu.lfilter(b, a, xs[c]) for c in range(m)]
        cdt = time.perf_counter() - t0
        got = y[:m].cpu().numpy()
        err = max(float(np.max(np.abs(got[c] - ref[c])) / np.max(np.abs(ref[c]))) for c in range(m))
        res["cpu_baseline"] = {"value": m * n / cdt / 1e9, "unit": "GS/s", "cores": 1, "kind": kind,
                               "sample": f"{m} channels x {n} samples of the same workload, {cdt:.1f} s, {what}",
                               "gpu_vs_cpu_max_scaled_err": err}
    print(json.dumps(res), flush=True)


def kernel_source_hash(workload: str) -> str:
    """SHA-256 of the source file that holds the workload's dominant kernel."""
    name = "iir_kernels.hip" if workload == "iir_biquad" else "fir_kernels.hip"
    with open(os.path.join(ROOT, "dsp_amd", "csrc", name), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def load_traffic(workload: str):
    """(HBM bytes per launch, where the figure comes from).  The bytes are PMC counters (FETCH_SIZE, WRITE_SIZE)
    of a profiled run of this same command, kept in profiles/pmc_traffic.json -- they are not measured in this
    run.  The entry records the hash of the kernel source it was profiled with; when the source has changed
    since, the figure is stale and None is reported instead."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(workload)
    except (OSError, ValueError):
        entry = None
    if not entry:
        return None, "no PMC profile of this workload under profiles/"
    want = entry.get("kernel_source_sha256")
    if want != kernel_source_hash(workload):
        return None, f"stale: {entry.get('source')} was taken with another version of the kernel source"
    return entry.get("hbm_bytes_per_launch"), f"{entry.get('source')}; profiled run, kernel source sha256 {want[:12]}"


def launch_ranks(args) -> int:
    """--gpus N > 1 without a launcher: start N fresh processes of this script, one per GPU, with the
    torch.distributed environment set (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  This process
    has not touched a GPU (no torch import, no HIP call) and never does; rank 0 prints the JSON line straight to
    the inherited stdout.  Returns the worst exit code; when a rank fails, the others are ended after a short grace
    period (they would wait for it in a barrier for ever)."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    worst, failed_at = 0, None
    limit = time.monotonic() + float(os.environ.get("DSPX_BENCH_TIMEOUT", "1500"))
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0:
                worst = rc if worst == 0 else worst
                failed_at = failed_at or time.monotonic()
        overdue = time.monotonic() > limit
        if live and (overdue or (failed_at and time.monotonic() - failed_at > float(os.environ.get("DSPX_BENCH_FAIL_GRACE", "20")))):
            for p in live:          # exactly the processes started above
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            worst = worst or 124
            live = []
    if worst:
        print(f"bench.py: a rank failed (exit code {worst})", file=sys.stderr, flush=True)
    return worst if 0 < worst < 256 else (1 if worst else 0)


def dry_run(args, rank: int, world: int) -> None:
    """DSPX_BENCH_DRY=1: the ranks only meet (gloo, no GPU, no filter) and rank 0 prints who took part.  What the
    CPU tests use to check the launcher: environment, rendezvous, forwarding of the line and of exit codes.
    DSPX_BENCH_DRY=fail<r> makes rank r exit with code 3 before the rendezvous."""
    mode = os.environ["DSPX_BENCH_DRY"]
    if mode == f"fail{rank}":
        raise SystemExit(3)
    seen = 1
    if world > 1:
        import torch
        import torch.distributed as dist

        dist.init_process_group(backend="gloo")
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one)
        seen = int(one[0])
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": seen, "steps": args.steps,
                          "warmup": args.warmup, "workload": args.workload}), flush=True)


def small_allgather(dsp_amd, dist, torch, y, rank, world, fence, on_gpu: bool):
    """The optional exchange step of the path on a bounded slice: every rank contributes its first `rows` result
    channels and receives those of all ranks (all_gather_into_tensor over the process group: RCCL over xGMI when
    on_gpu, else the rehearsal backend on host copies).  Checked against per-rank checksums gathered beside it.
    Returns the sub-record (identical on every rank)."""
    rows = min(64, y.shape[0])
    n = y.shape[1]
    mine_dev = y[:rows].contiguous()
    try:
        own = dsp_amd.checksum(mine_dev, 0)
        mine = mine_dev if on_gpu else mine_dev.cpu()
        sums = torch.zeros(world, dtype=torch.int64, device=mine.device)
        sums[rank] = own - (1 << 64) if own >= (1 << 63) else own
        dist.all_reduce(sums)                                   # sum of one-hot vectors = every rank's checksum
        full = dsp_amd.allgather_channels(mine, rows * world)   # warm-up: sets up the communicator's channels
        fence()
        t0 = time.perf_counter()
        full = dsp_amd.allgather_channels(mine, rows * world)
        fence()
        ms = (time.perf_counter() - t0) * 1e3
        full = full.to(y.device)
        ok = all((dsp_amd.checksum(full[r * rows:(r + 1) * rows].contiguous(), 0) - int(sums[r])) % (1 << 64) == 0
                 for r in range(world))
        t = torch.tensor([ms, 0.0 if ok else 1.0], dtype=torch.float64, device=mine.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        recv = 8.0 * rows * n * (world - 1)
        c_abi = None
        if on_gpu:   # the same exchange through the library's own entry point (dspx_allgather_f64_dev over its RCCL communicator)
            # Every step that can fail on one rank alone is followed by an agreement (all-reduce MIN of an ok flag), so
            # that the ranks skip the rest TOGETHER: a rank that left early would leave the others in a collective.
            def agreed(ok: bool) -> bool:
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=mine.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return bool(int(flag[0]))

            ident, err = None, None
            try:
                if rank == 0:
                    ident = dsp_amd.Comm.unique_id()
                else:
                    dsp_amd.lib()   # the library (and its RCCL binding) must at least load here too
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            if not agreed(err is None):
                c_abi = {"error": err or "another rank could not reach the library's RCCL binding", "verified": False}
            else:
                box = [ident]
                dist.broadcast_object_list(box, src=0)
                comm = None
                try:
                    comm = dsp_amd.Comm(box[0], world, rank)
                except Exception as e:  # noqa: BLE001
                    err = f"{type(e).__name__}: {e}"
                if not agreed(comm is not None):
                    if comm is not None:
                        comm.close()
                    c_abi = {"error": err or "another rank could not create its communicator", "verified": False}
                else:
                    full2 = comm.allgather(mine_dev)          # warm-up
                    fence()
                    t0 = time.perf_counter()
                    full2 = comm.allgather(mine_dev, out=full2)
                    fence()
                    ms2 = (time.perf_counter() - t0) * 1e3
                    same = bool(torch.equal(full2, full))
                    comm.close()
                    t2 = torch.tensor([ms2, 0.0 if same else 1.0], dtype=torch.float64, device=mine.device)
                    dist.all_reduce(t2, op=dist.ReduceOp.MAX)
                    c_abi = {"ms": float(t2[0]), "GB/s_per_gpu": recv / float(t2[0]) / 1e6,
                             "equals_torch_gather": float(t2[1]) == 0.0, "verified": float(t2[1]) == 0.0}
        return {"ms": float(t[0]), "rows_per_rank": rows, "bytes_received_per_gpu": recv, "dspx_allgather_f64_dev": c_abi,
                "GB/s_per_gpu": recv / float(t[0]) / 1e6, "matches_per_rank_checksums": float(t[1]) == 0.0,
                "transport": "RCCL (nccl backend), device buffers" if on_gpu else "rehearsal backend, host copies",
                "note": "all_gather_into_tensor of a slice of the result shards over the process group, outside the "
                        "timed region; not part of value"}
    except Exception as e:  # noqa: BLE001 -- the optional exchange must never take the metric down with it
        return {"error": f"{type(e).__name__}: {e}"}


def main() -> None:
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))   # this process only spawns the ranks; it never touches a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if os.environ.get("DSPX_BENCH_DRY"):
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist

    import dsp_amd

    # Rehearsal on a one-GPU machine: DSPX_BENCH_BACKEND=gloo DSPX_BENCH_ONE_DEVICE=1 puts every rank on GPU 0
    # and runs the barrier and the reductions over gloo (RCCL refuses two ranks on one device).
    backend = os.environ.get("DSPX_BENCH_BACKEND", "nccl")
    if os.environ.get("DSPX_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")   # where the timing reductions live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    # how many ranks really take part: a sum over the process group (RCCL on GPUs), not the environment
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(one)
        ranks_seen = int(one[0])

    if args.workload in ("cwt65", "iir_biquad"):
        (run_cwt if args.workload == "cwt65" else run_iir)(args, rank, world, dev, dist, red_dev, ranks_seen)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # N > 1 measures the workload the metric names, sharded over the ranks, unless --scaling weak asks otherwise
    args.scaling = args.scaling or ("strong" if world > 1 else "weak")
    _, n, numtaps, cutoff, desc = WORKLOADS[args.workload]
    taps_h = dsp_amd.firwin(numtaps, cutoff)
    taps = torch.from_numpy(taps_h).to(dev)
    hist = buf = None
    segments = args.scaling == "strong" and world > 1 and WORKLOADS[args.workload][0] < world
    if segments:
        # fewer streams than GPUs: every rank filters a contiguous segment of every stream, with the
        # ntaps-1 samples before it as history (generated locally, as a host sharder would slice them)
        nchan, chan0 = WORKLOADS[args.workload][0], 0
        units_per_step_all = nchan * n
        lo, hi = segment(n, rank, world)
        lead = min(lo, numtaps - 1)
        buf = torch.empty((nchan, lead + hi - lo), dtype=torch.float64, device=dev)
        for c in range(nchan):
            dsp_amd.fill_splitmix(buf[c], dsp_amd.SEED, c * n + lo - lead)
        x = buf[:, lead:]
        if lo > 0:
            hist = torch.zeros((nchan, numtaps - 1), dtype=torch.float64, device=dev)
            hist[:, numtaps - 1 - lead:] = buf[:, :lead]
        n_local = hi - lo
    else:
        nchan, chan0, units_per_step_all = shard(args.workload, rank, world, args.scaling)
        if nchan == 0:
            raise SystemExit(f"rank {rank}: no channels to filter")
        x = torch.empty((nchan, n), dtype=torch.float64, device=dev)
        dsp_amd.fill_splitmix(x, dsp_amd.SEED, chan0 * n)
        n_local = n
    y = torch.empty((nchan, n_local), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def step():
        dsp_amd.fir_device(taps, x, out=y, hist=hist, exact=args.exact, mfma=False if args.vector else None, tiled=args.tiled)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]

    # optional exchange step, timed apart from the filter: every rank receives all result shards over RCCL
    gather_ms = 0.0
    gather_small = None
    if world > 1 and not args.no_allgather and not (args.allgather and args.scaling == "strong"):
        gather_small = small_allgather(dsp_amd, dist, torch, y, rank, world, fence, backend == "nccl")
    if args.allgather and world > 1 and args.scaling == "strong" and backend == "nccl":
        nchan_total = WORKLOADS[args.workload][0]
        full = dsp_amd.allgather_channels(y, nchan_total)   # warm-up (also sets up the communicator's buffers)
        del full
        fence()
        t0 = time.perf_counter()
        full = dsp_amd.allgather_channels(y, nchan_total)
        fence()
        gather_ms = (time.perf_counter() - t0) * 1e3
        del full

    # The position-dependent checksum of this rank's outputs at their place in the whole output, summed over the
    # ranks modulo 2^64: a sharding never changes an output's arithmetic, so with strong scaling the sum is the
    # same for every N (32-bit halves are summed apart so that no signed 64-bit sum can overflow).
    if segments:
        own = sum(dsp_amd.checksum(y[c], c * n + lo) for c in range(nchan)) % (1 << 64)
    else:
        own = dsp_amd.checksum(y, chan0 * n)
    halves = torch.tensor([own & 0xffffffff, own >> 32], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(halves)
    output_checksum = (int(halves[0]) + (int(halves[1]) << 32)) % (1 << 64)

    t = torch.tensor([dt, float(np.mean(kern_ms)), gather_ms], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max, kern_ms_max, gather_ms = float(t[0]), float(t[1]), float(t[2])

    # Weak scaling beside the strong headline: every rank filters a whole batch of its own for a few steps.
    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_weak:
        nchan_total = WORKLOADS[args.workload][0]
        del x, y, hist, buf
        torch.cuda.empty_cache()
        xw = torch.empty((nchan_total, n), dtype=torch.float64, device=dev)
        yw = torch.empty_like(xw)
        dsp_amd.fill_splitmix(xw, dsp_amd.SEED, rank * nchan_total * n)
        wsteps = max(1, min(args.steps, 5))
        dsp_amd.fir_device(taps, xw, out=yw, exact=args.exact, mfma=False if args.vector else None)   # warm-up
        fence()
        t0 = time.perf_counter()
        for _ in range(wsteps):
            dsp_amd.fir_device(taps, xw, out=yw, exact=args.exact, mfma=False if args.vector else None)
        fence()
        tw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        weak = {"value": world * nchan_total * n * wsteps / float(tw[0]) / 1e9, "unit": "GS/s", "steps": wsteps, "warmup": 1,
                "ms_per_step": float(tw[0]) / wsteps * 1e3, "nchan_per_gpu": nchan_total,
                "note": "every rank filters a whole batch of the named shape (its own data); same timing rules as value"}
        del xw, yw

    if rank == 0:
        matrix = not args.exact and not args.vector and numtaps >= dsp_amd.MATRIX_PIPE_MIN_TAPS
        steps_of_four = (numtaps + 2) // 4 + 1
        kernel_name = ("fir_f64_kernel (v_fma_f64)" if not matrix else
                       "fir_f64_mfma4_kernel (v_mfma_f64_4x4x4_4b_f64, workgroup-tiled form)" if args.tiled else
                       "fir_f64_mfma4w_kernel (v_mfma_f64_4x4x4_4b_f64, wave-autonomous form, one window per chunk, runs of chunks drawn from the stream's device counters)"
                       if steps_of_four <= 143 else
                       "fir_f64_mfma4wm_kernel (v_mfma_f64_4x4x4_4b_f64, wave-autonomous form, several windows per chunk)")
        outputs_per_launch = nchan * n_local
        traffic, traffic_src = (load_traffic(args.workload) if (n_local == n and nchan == WORKLOADS[args.workload][0])
                                else (None, "this rank's shard is not the profiled launch"))
        if args.tiled or args.vector or args.exact:
            traffic, traffic_src = None, "the profiled launch is the default kernel of this workload"
        flops = 2.0 * numtaps * outputs_per_launch
        bytes_alg = 16.0 * outputs_per_launch
        sec = kern_ms_max / 1e3
        fp64 = {"bound": "mfma", "achieved": flops / sec / 1e12, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / sec / 1e12 / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": kernel_name, "kernel_ms": kern_ms_max,
                "note": "MI355X FP64 peak: matrix pipe = vector pipe = 78.6 TFLOP/s, so one number bounds both kernels. "
                        "algorithmic flops = 2*ntaps per output (the Toeplitz-block kernel's multiplies by zero taps "
                        "are not counted)"}
        hbm = {"bound": "hbm", "achieved": bytes_alg / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": bytes_alg / sec / 1e9 / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
               "kernel": kernel_name, "kernel_ms": kern_ms_max,
               "note": "algorithmic bytes = 16 per output (8 read + 8 written)"}
        # the binding roof is the one that allows fewer samples per second
        fp64_roof_gs = PEAK_FP64_TFLOPS * 1e12 / (2.0 * numtaps) / 1e9
        hbm_roof_gs = PEAK_HBM_GBS / 16.0
        primary, secondary = (fp64, hbm) if fp64_roof_gs <= hbm_roof_gs else (hbm, fp64)
        out = {
            "metric": "FP64 FIR output GS/s (255 taps)" if numtaps == 255 else f"FP64 FIR output GS/s ({numtaps} taps)",
            "value": units_per_step_all * args.steps / dt_max / 1e9,
            "unit": "GS/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (splitmix64 uniform [-1,1), generated on device, resident in HBM)",
            "config": {"workload": desc, "nchan_per_gpu": nchan, "samples_per_channel": n, "ntaps": numtaps,
                       "arithmetic": "exact (mul+add)" if args.exact else ("fma, matrix pipe" if matrix else "fma, vector pipe"),
                       "kernel_form": "tiled (forced)" if (matrix and args.tiled) else "default",
                       "sharding": "none" if world == 1 else ("segments with history" if segments else "channels" if args.scaling == "strong"
                                                             else "independent batch per GPU")},
            "roofline": primary,
            "roofline_other": secondary,
            "output_checksum": f"0x{output_checksum:016x}",
        }
        if args.scaling == "strong" or world == 1:
            out["output_checksum_note"] = ("checksum of the whole output of the named workload (each rank's shard at its "
                                           "global position, summed over ranks mod 2^64): identical for every --gpus N")
        if weak is not None:
            out["weak"] = weak
        if gather_small is not None:
            out["allgather"] = gather_small
        if gather_ms > 0.0:
            total_bytes = 8.0 * WORKLOADS[args.workload][0] * n
            out["allgather"] = {"ms": gather_ms, "bytes_received_per_gpu": total_bytes * (world - 1) / world,
                                "GB/s_per_gpu": total_bytes * (world - 1) / world / gather_ms / 1e6,
                                "note": "optional all_gather_into_tensor of the result shards (RCCL); not part of value"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(taps_h, n, args.cpu_samples, y)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
