#!/bin/bash
# One box, one process chain: every tap count of the design table on both pipes.
set -e
W="b31 b48 b63 b96 l101 b128 b160 b255 b512 l4095"
echo "== vector pipe"; QB_MFMA=0 timeout -k 10 300 python tools/qb.py $W
echo "== matrix pipe"; QB_MFMA=1 timeout -k 10 300 python tools/qb.py $W
