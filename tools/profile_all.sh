#!/bin/bash
# Every profile of a round in ONE gpurun call (each rocprofv3 pass is its own process, --pmc passes apart from the
# kernel trace, as MI355X_MICROARCH.md prescribes):   tools/profile_all.sh <tag>
#   default bench (fuse_chain_kernel):  <tag>_{kt,fetch,write,valu}, micro-benchmarks, a --two-phase bench line
#   bench --two-phase (verdict_kernel + accept_chain_kernel):  <tag>tp_{kt,fetch,write,valu}
# Then on the CPU side: tools/profile_summary.py <tag>  and  tools/profile_summary.py <tag>tp --two-phase
tag=$1
[ -x build/barrier_bench ] && timeout -k 10 120 ./build/barrier_bench > gpurun_out/${tag}_barrier_bench.txt 2>&1
[ -x build/valu_bench ] && timeout -k 10 120 ./build/valu_bench > gpurun_out/${tag}_valu_bench.txt 2>&1
timeout -k 10 200 python3 bench.py --two-phase --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_two_phase.json 2>/dev/null || exit 1
for p in kt fetch write valu; do tools/profile_pass.sh $tag $p || exit 1; done
cp gpurun_out/${tag}_two_phase.json gpurun_out/${tag}tp_two_phase.json
for p in kt fetch write valu; do BENCH_ARGS=--two-phase tools/profile_pass.sh ${tag}tp $p || exit 1; done
