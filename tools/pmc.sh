#!/bin/bash
# Diagnostic: one rocprofv3 --pmc pass over a short bench run; prints per-kernel counter sums.
# usage: tools/pmc.sh <tag> <counter> [<counter> ...]   (at most 8 SQ_ counters per pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 1 --warmup 0 --frames 160 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % tag, recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-30:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "sfm" in k:
        print(k, {c: (v, n[(k, c)]) for c, v in acc[k].items()})
PY
