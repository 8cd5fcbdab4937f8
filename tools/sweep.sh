#!/bin/bash
# Runs bench.py on the default workload under a list of env settings (one per argument, "A=1,B=2"),
# one line per setting into gpurun_out/sweep.log: setting, Gpx/s, ms/step, observe avg us, scan avg us.
out=gpurun_out/sweep.log
mkdir -p gpurun_out
: > $out
for cfg in "$@"; do
  envs=$(echo "$cfg" | tr ',' ' ')
  line=$(env $envs timeout -k 10 200 python3 -u bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1) || { echo "$cfg FAILED" >> $out; exit 1; }
  echo "$line" | python3 -c "
import sys,json
j=json.loads(sys.stdin.read())
print('$cfg', round(j['value']/1e9,3), round(j['ms_per_step'],3), round(j['roofline']['avg_launch_us'],2), round((j['roofline']['scan_kernel'] or {'avg_launch_us':0})['avg_launch_us'],2))" >> $out
done
cat $out
