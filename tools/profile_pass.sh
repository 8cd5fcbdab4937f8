#!/bin/bash
# One rocprofv3 pass over the default bench command on the GPU box (run each pass in its own gpurun call).
#   tools/profile_pass.sh <tag> kt            kernel trace + stats
#   tools/profile_pass.sh <tag> fetch|write   one --pmc pass (FETCH_SIZE / WRITE_SIZE), as MI355X_MICROARCH.md prescribes
#   tools/profile_pass.sh <tag> valu          one --pmc pass of SQ instruction counters (VALU-issue floor of the chain kernel)
# BENCH_ARGS="--two-phase [--workload ...]" profiles that bench command instead (verdict_kernel + accept_chain_kernel).
# Output: gpurun_out/<tag>_<pass>/ ; then run tools/profile_summary.py <tag> [--two-phase] [--workload W] (CPU side is fine).
tag=$1; pass=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
case $pass in
  kt)    opts="--kernel-trace --stats"; steps="--steps 3 --warmup 1" ;;
  fetch) opts="--pmc FETCH_SIZE"; steps="--steps 1 --warmup 0" ;;
  write) opts="--pmc WRITE_SIZE"; steps="--steps 1 --warmup 0" ;;
  valu)  opts="--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; steps="--steps 1 --warmup 0" ;;
  *) echo "unknown pass $pass"; exit 2 ;;
esac
timeout -s KILL 200 rocprofv3 $opts --output-format csv -d gpurun_out/${tag}_${pass} -- python3 bench.py $steps --no-cpu-baseline $BENCH_ARGS > gpurun_out/${tag}_${pass}.log 2>&1
rc=$?
grep -c . gpurun_out/${tag}_${pass}/*/*.csv 2>/dev/null | tail -4
echo "rocprofv3 exit code $rc"
exit $rc
