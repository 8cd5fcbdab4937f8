#!/bin/bash
# Experiment build of the HIP library into build/exp/lib_<name>.so (the production library is untouched); extra
# arguments go to hipcc (-DSFM_...=...).  Compare builds on ONE box with tools/ab_bench.sh.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/exp
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -std=c++17 \
  -save-temps=obj "$@" -c sfm_amd/csrc/fusion_capi.hip -o build/exp/fusion_capi_$name.o
[ -f build/cloud_sort.o ] || hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c sfm_amd/csrc/cloud_sort.hip -o build/cloud_sort.o
hipcc --offload-arch=gfx950 -fPIC -shared build/exp/fusion_capi_$name.o build/cloud_sort.o -o build/exp/lib_$name.so


python3 tools/kernel_regs.py build/exp/fusion_capi-hip-amdgcn-amd-amdhsa-gfx950.s
