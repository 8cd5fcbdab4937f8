set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -2
for round in 1 2; do
for v in build/exp/lib_head.so build/exp/lib_both.so; do
SFM_FUSION_LIB=$PWD/$v timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v fused', d['ms_per_step'], d['roofline']['avg_launch_us'], d['output_points'])"
SFM_FUSION_LIB=$PWD/$v timeout -k 10 120 python bench.py --two-phase --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v two-phase A/B', d['two_phase']['phase_a_ms'], d['two_phase']['phase_b_ms'])"
done
done
