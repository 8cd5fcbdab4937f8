#!/bin/bash
# A/B of library builds on ONE box (box-to-box differences are larger than most single changes): runs the default bench
# workload with each build/exp/lib_<name>.so given, twice, interleaved.   usage: [AB_ARGS="--two-phase"] tools/ab_bench.sh base t1 ...
set -e
for round in 1 2; do
for v in "$@"; do
SFM_FUSION_LIB=$PWD/build/exp/lib_$v.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); tp=d.get('two_phase'); print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), d['output_points'], d['chain'], ('phase_a_ms %.3f phase_b_ms %.3f' % (tp['phase_a_ms'], tp['phase_b_ms'])) if tp else '')"
done
done
