#!/bin/bash
# Builds libsfm_fusion.so (gfx950 only) in-tree.  -ffp-contract=off: bit-exact IEEE double
# evaluation order on device and host (see fusion_kernels.h).
set -e
cd "$(dirname "$0")"
mkdir -p sfm_amd/lib build
# -amdgpu-atomic-optimizer-strategy=None: that pass rewrites a one-lane atomic into atomic + readfirstlane and so
# forces a wait for the atomic's round trip right behind it; the kernels aggregate their atomics themselves
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -std=c++17 -Wall \
  -save-temps=obj -c sfm_amd/csrc/fusion_capi.hip -o build/fusion_capi.o "$@"
# the output cloud's sort (rocPRIM) in its own translation unit; rebuilt only when it changes
if [ ! -f build/cloud_sort.o ] || [ sfm_amd/csrc/cloud_sort.hip -nt build/cloud_sort.o ]; then
  hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c sfm_amd/csrc/cloud_sort.hip -o build/cloud_sort.o
fi
hipcc --offload-arch=gfx950 -fPIC -shared build/fusion_capi.o build/cloud_sort.o -o build/libsfm_fusion.so
# C++ facade example (host-only code over the C ABI); run by tests/test_gpu_cpp_facade.py
g++ -std=c++17 -O2 -Wall -pthread sfm_amd/cpp/facade_example.cpp -Lbuild -lsfm_fusion \
  -Wl,-rpath,'$ORIGIN/../lib' -o sfm_amd/cpp/facade_example
g++ -std=c++17 -O2 -Wall -pthread sfm_amd/cpp/replay_reconstruct.cpp -Lbuild -lsfm_fusion \
  -Wl,-rpath,'$ORIGIN/../lib' -o sfm_amd/cpp/replay_reconstruct
# multi-GPU C++ host (RCCL); run at one rank by tests/test_gpu_replay.py
g++ -std=c++17 -O2 -Wall -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include sfm_amd/cpp/dist_reconstruct.cpp -Lbuild -lsfm_fusion \
  -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib -o sfm_amd/cpp/dist_reconstruct
# host-only formatter probe used by tests/test_writer_parity.py
g++ -std=c++17 -O2 -Wall -pthread tests/cpp/writer_probe.cpp -o build/writer_probe
g++ -std=c++17 -O2 -Wall tests/cpp/recording_probe.cpp -lz -o build/recording_probe
# GPU unit test of the shared-reciprocal division pair; run by tests/test_gpu_div_pair.py
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 tests/hip/div_probe.hip -o build/div_probe
# fuse_chain_kernel must leave VGPRs for a co-resident table-kernel wave (see the kernel's comment)
for nv in $(grep -E "fuse_chain_kernel.*\.num_vgpr," build/fusion_capi-hip-amdgcn-amd-amdhsa-gfx950.s | sed 's/.*num_vgpr, *//'); do
  if [ "$nv" -gt 120 ]; then echo "build.sh: fuse_chain_kernel uses $nv VGPRs (> 120)" >&2; exit 1; fi
done
# ... and the kernels that run beside it on stream B must fit into what it leaves: 512 - 4 * 120 = 32 VGPRs per SIMD
# (a table kernel with 43 VGPRs waited for whole chain launches to end: 8.1 -> 11 ms per C2 run)
for k in table_group_kernel finalize_kernel; do
  nv=$(grep -E "sfm[0-9]+${k}E.*\.num_vgpr," build/fusion_capi-hip-amdgcn-amd-amdhsa-gfx950.s | head -1 | sed 's/.*num_vgpr, *//')
  if [ -n "$nv" ] && [ "$nv" -gt 32 ]; then echo "build.sh: $k uses $nv VGPRs (> 32)" >&2; exit 1; fi
done
cp build/libsfm_fusion.so sfm_amd/lib/libsfm_fusion.so
# the same library with operand checks inside the shared-reciprocal divisions (tests/test_gpu_div_checked.py);
# its kernels are slower and are never loaded by the product
mkdir -p build/checked
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None -fPIC -std=c++17 \
  -DSFM_DIV_CHECKED=1 -c sfm_amd/csrc/fusion_capi.hip -o build/checked/fusion_capi.o
hipcc --offload-arch=gfx950 -fPIC -shared build/checked/fusion_capi.o build/cloud_sort.o -o sfm_amd/lib/libsfm_fusion_checked.so
