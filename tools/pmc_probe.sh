#!/bin/bash
# One rocprofv3 --pmc pass with the counters given, over a bench command:  tools/pmc_probe.sh <tag> "<bench args>" CTR...
tag=$1; args=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/${tag} -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $args > gpurun_out/${tag}.log 2>&1
rc=$?
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"gpurun_out/{tag}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "sfm::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in agg.items():
    print(k, {c: (round(v[0] / v[1], 1), v[1]) for c, v in d.items()})
PY
echo "rocprofv3 exit code $rc"; tail -3 gpurun_out/${tag}.log
