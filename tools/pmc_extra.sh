#!/bin/bash
# Extra counter passes (issue activity per instruction type, queue levels, FIFO-full events) around one benchmark
# workload: tools/pmc_extra.sh <tag> [bench args]; output under gpurun_out/pmcx_<tag>/, read with tools/pmc_quick.py.
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
pass() {
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline ${EXTRA:-} > $OUT/$name.log 2>&1 || { echo "pmc pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
}
EXTRA="$*"
pass a1 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES &&
pass a2 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR &&
pass a3 SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU &&
echo "ok: $OUT"
