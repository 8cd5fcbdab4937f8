#!/bin/bash
# Diagnostic build with in-kernel phase timestamps (fuse_chain_kernel_stamps / accept_chain_kernel_stamps) into
# build/stamps/libsfm_fusion_stamps.so; the production library is untouched.  Used by tools/chain_stamps.py.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/stamps
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -std=c++17 \
  -DSFM_STAMPS=1 "$@" -c sfm_amd/csrc/fusion_capi.hip -o build/stamps/fusion_capi.o
[ -f build/cloud_sort.o ] || hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c sfm_amd/csrc/cloud_sort.hip -o build/cloud_sort.o
hipcc --offload-arch=gfx950 -fPIC -shared build/stamps/fusion_capi.o build/cloud_sort.o -o build/stamps/libsfm_fusion_stamps.so
