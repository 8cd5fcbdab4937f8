#!/bin/bash
# Profiles the default bench command on the GPU box: kernel-trace stats, then two separate PMC
# passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/<tag>_*.
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/${tag}_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/${tag}_write.log 2>&1
python3 tools/profile_summary.py $tag
