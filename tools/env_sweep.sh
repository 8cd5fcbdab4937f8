#!/bin/bash
# Sweep of one environment knob of the library over the default bench workload on ONE box.
#   usage: [AB_ARGS="--two-phase"] tools/env_sweep.sh VAR v1 v2 ...      (two interleaved rounds)
var=$1; shift
for round in 1 2; do
for v in "$@"; do
env $var=$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); tp=d.get('two_phase'); print('$var=$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), d['output_points'], d['chain'], ('phase_a_ms %.3f phase_b_ms %.3f' % (tp['phase_a_ms'], tp['phase_b_ms'])) if tp else '')"
done
done
