#!/bin/bash
# Samples board power and clocks (rocm-smi) while a workload runs: tools/power_probe.sh <qb workload>
# Development aid for the question "is the clock under this kernel set by the power limit".
W=$1
( for i in 1 2 3 4 5 6; do timeout -k 5 100 python tools/qb.py $W $W $W > /dev/null 2>&1; done ) &
PID=$!
sleep 12
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|fclk" | sed -e 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 1
done
wait $PID
