"""Builds dsp_amd/lib/libdspx.so (HIP kernels + C ABI) for gfx950 with hipcc.

The library is built in-tree so that it travels with the repository snapshot to the
GPU machine; nothing is compiled at import time.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libdspx.so")
SOURCES = ["fir_kernels.hip", "iir_kernels.hip", "dspx_capi.hip", "dspx_cwt.hip", "dspx_comm.hip"]
HEADERS = [os.path.join(CSRC, "fir_kernels.h"), os.path.join(CSRC, "iir_kernels.h"), os.path.join(CSRC, "dspx_common.h"), os.path.join(ROOT, "include", "dspx.h"),
           os.path.join(ROOT, "include", "dsp.hpp")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fPIC", "-shared",
         "-fvisibility=hidden", "-mllvm", "-amdgpu-mfma-vgpr-form",
         # the FIR kernels draw their next run with an atomic whose result is read a whole run later: the pass that
         # combines a wave's atomics would read it (and wait for it) at once
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built on this machine")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h for h in HEADERS if os.path.exists(h)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc(), *FLAGS, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libdspx.so")
    if verbose and r.stderr:
        sys.stderr.write(r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
