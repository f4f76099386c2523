#!/bin/bash
# Builds a variant of the library for same-box A/B runs: tools/build_variant.sh <name> [extra hipcc flags]
# -> dsp_amd/lib/libdspx_<name>.so (select it with DSPX_LIB=$PWD/dsp_amd/lib/libdspx_<name>.so)
set -e
NAME=$1; shift
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -fPIC -shared -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -Wall -Wextra -Wno-unused-parameter -I include -I dsp_amd/csrc "$@" \
  dsp_amd/csrc/fir_kernels.hip dsp_amd/csrc/iir_kernels.hip dsp_amd/csrc/dspx_capi.hip dsp_amd/csrc/dspx_cwt.hip dsp_amd/csrc/dspx_comm.hip \
  -o dsp_amd/lib/libdspx_$NAME.so
echo dsp_amd/lib/libdspx_$NAME.so
