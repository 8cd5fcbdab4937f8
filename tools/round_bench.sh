#!/bin/bash
# The bench lines a round commits under profiles/ (one gpurun call):   tools/round_bench.sh <tag>
tag=$1
B="timeout -k 10 300 python3 bench.py"
$B > gpurun_out/${tag}_bench.json 2>/dev/null || exit 1
$B --two-phase --no-cpu-baseline > gpurun_out/${tag}_c2_two_phase_bench.json 2>/dev/null || exit 1
$B --split-run --no-cpu-baseline > gpurun_out/${tag}_c2_split_run_bench.json 2>/dev/null || exit 1
$B --two-phase --split-run --no-cpu-baseline > gpurun_out/${tag}_c2_two_phase_split_run_bench.json 2>/dev/null || exit 1
W="--workload c3_2000kf_1280x720_1cm --steps 5 --warmup 1 --no-cpu-baseline"
$B $W > gpurun_out/${tag}_c3_full_bench.json 2>/dev/null || exit 1
$B $W --two-phase > gpurun_out/${tag}_c3_two_phase_bench.json 2>/dev/null || exit 1
$B $W --split-run > gpurun_out/${tag}_c3_split_run_bench.json 2>/dev/null || exit 1
$B $W --two-phase --split-run > gpurun_out/${tag}_c3_two_phase_split_run_bench.json 2>/dev/null || exit 1
