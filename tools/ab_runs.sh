#!/bin/bash
# Same-box A/B of how the wave-autonomous matrix kernels hand out their runs: fixed shares per wave (QB_SHARES=1, the
# library's test hook) against the default (drawn from the stream's counters for large inputs), two alternating rounds,
# one library: tools/ab_runs.sh [workloads]
W=${1:-"b255 l101 b63 b96 b128 b160 b300 b512 c8n2e25t101 c1n2e28t63"}
for round in 1 2; do
  echo "== fixed shares, round $round"; QB_MFMA=1 QB_SHARES=1 timeout -k 10 200 python tools/qb.py $W || exit 1
  echo "== default (drawn runs), round $round"; QB_MFMA=1 timeout -k 10 200 python tools/qb.py $W || exit 1
done
