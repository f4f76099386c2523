#!/bin/bash
# quick FETCH_SIZE of a qb workload: fetchq.sh <workload>
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/q3; mkdir -p gpurun_out/q3
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/q3/f -- python3 tools/qb.py $1 > gpurun_out/q3/f.log 2>&1 && python tools/pmc_quick.py gpurun_out/q3 fir_f64 | grep -E "launches|FETCH"
