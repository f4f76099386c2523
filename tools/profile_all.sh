#!/bin/bash
# Runs tools/profile.sh for every benchmark workload of the FIR path (on the GPU box); tag prefix as argument.
set -e
P=${1:-r03}
bash tools/profile.sh ${P}
bash tools/profile.sh ${P}_101 --workload long101 --steps 1000 --warmup 100
bash tools/profile.sh ${P}_4095 --workload long4095 --steps 100 --warmup 10
bash tools/profile.sh ${P}_cwt65 --workload cwt65
bash tools/profile.sh ${P}_31 --workload batched31
bash tools/profile.sh ${P}_iir --workload iir_biquad
