#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_old_new.sh <variant library> [workloads]
# (the variant, e.g. dsp_amd/lib/libdspx_old.so from tools/build_variant.sh, against the in-tree library, two alternating rounds)
V=$1; shift
W=${*:-"l101 b255 b63 b96 b128 b160 b192 b300 b512"}
for round in 1 2; do
  echo "== $V, round $round"; DSPX_LIB=$PWD/$V QB_MFMA=1 timeout -k 10 200 python tools/qb.py $W || exit 1
  echo "== in-tree library, round $round"; QB_MFMA=1 timeout -k 10 200 python tools/qb.py $W || exit 1
done
