#!/bin/bash
# Same-box A/B of the two forms of the matrix-pipe kernel (DSPX_FIR_TILED against the default), two alternating rounds,
# one library: tools/ab_forms.sh [workloads]
W=${1:-"b255 l101 b128 b63 b96 b160 l4095"}
for round in 1 2; do
  echo "== workgroup-tiled form (DSPX_FIR_TILED), round $round"; QB_MFMA=1 QB_TILED=1 timeout -k 10 200 python tools/qb.py $W || exit 1
  echo "== default form, round $round"; QB_MFMA=1 timeout -k 10 200 python tools/qb.py $W || exit 1
done
