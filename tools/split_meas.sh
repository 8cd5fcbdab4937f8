# diagnostic: sfm_fusion_run_chain + sfm_fusion_finish (bench.py --split-run) against the plain run, and the grid of the
# voxel filter when it runs behind the chain (SFM_TABLE_WGS_ALONE).  Run on the GPU box: bash tools/split_meas.sh [wgs...]
set -e
B="python bench.py --no-cpu-baseline"
pick='import json,sys; d=json.loads(sys.stdin.readlines()[-1]); s=d.get("split_run") or {}; print("ms_per_step %.3f masks_final %.3f chain_device %.3f points %d" % (d["ms_per_step"], s.get("masks_final_host_ms",0), s.get("chain_device_ms", d.get("two_phase",{}).get("phase_b_ms",0) or 0), d["output_points"]))'
for W in ${@:-8 16 24 32}; do
for a in "" "--two-phase"; do
echo -n "SFM_TABLE_WGS_ALONE=$W c2 $a: "; SFM_TABLE_WGS_ALONE=$W timeout -k 10 120 $B --steps 20 --warmup 3 --split-run $a 2>/dev/null | python -c "$pick"
done; done
for a in "" "--two-phase"; do
echo -n "plain run c2 $a: "; timeout -k 10 120 $B --steps 20 --warmup 3 $a 2>/dev/null | python -c "$pick"
done
