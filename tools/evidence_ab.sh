#!/bin/bash
# The same-box A/B records and the tap sweep of profiles/, taken in one go on one box (on the GPU box):
# tools/evidence_ab.sh [max ms per headline step]  -- gives up at once on a box slower than that, so that the records
# come from a box of the same class as the profiles beside them.
MAXMS=${1:-1000}
mkdir -p gpurun_out/ab
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/ab/bench.json 2> gpurun_out/ab/bench.err || exit 1
MS=$(python -c "import json;print(json.loads(open('gpurun_out/ab/bench.json').read().strip().splitlines()[-1])['ms_per_step'])")
echo "headline step: $MS ms"
python -c "import sys; sys.exit(0 if float('$MS') <= float('$MAXMS') else 3)" || { echo "slower than $MAXMS ms: not taken"; exit 3; }
timeout -k 10 600 python tools/ab_wm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab/r03_ab_chunks.txt || exit 1
timeout -k 10 300 python tools/qb_cwt.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab/r03_cwt_rows.txt || exit 1
bash tools/ab_runs.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/ab/r03_ab_runs.txt || exit 1
bash tools/ab_runs.sh "c1n209715200t63 c1n262144000t101 c1n262144000t160" 2>&1 | grep -v amdgpu.ids > gpurun_out/ab/r03_ab_threshold.txt || exit 1
timeout -k 10 600 python tools/tap_sweep.py gpurun_out/ab/r03_tap_sweep.json > gpurun_out/ab/tap_sweep.log 2>&1 || exit 1
tail -3 gpurun_out/ab/tap_sweep.log
