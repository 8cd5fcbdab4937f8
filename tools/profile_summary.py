"""Summarises gpurun_out/<tag>_{kt,fetch,write} into profiles/<tag>_summary.json and
profiles/traffic_latest.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
out = {"tag": tag, "workload": "c2_500kf_640x480_1cm", "kernels": {}}
f = glob.glob(f"gpurun_out/{tag}_kt/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "sfm::" in r["Name"] or "rocclr" in r["Name"]:
        out["kernels"][r["Name"].split("(")[0]] = dict(calls=int(r["Calls"]), total_ms=int(r["TotalDurationNs"]) / 1e6,
                                                       avg_us=float(r["AverageNs"]) / 1e3, min_us=int(r["MinNs"]) / 1e3,
                                                       max_us=int(r["MaxNs"]) / 1e3)
pmc = {}
for name in ("fetch", "write"):
    f = glob.glob(f"gpurun_out/{tag}_{name}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "sfm::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
    pmc[name] = {k: dict(sum_kb=v[0], launches=v[1], kb_per_launch=v[0] / v[1]) for k, v in agg.items()}
out["pmc"] = pmc
# the dominant kernel: the persistent chain kernel, or the observe kernel of the fallback pair
k = "sfm::fuse_chain_kernel" if "sfm::fuse_chain_kernel" in pmc["fetch"] else "sfm::observe_frame_kernel"
short = k.split("::")[1]
fetch_kb = pmc["fetch"][k]["kb_per_launch"]
write_kb = pmc["write"][k]["kb_per_launch"]
# MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies
# 128-B requests at 64 B => doubled.  (Calibrated in the guide for wide streaming reads; this
# kernel's 2-byte-per-lane reads and gathers are outside that calibration.)
hbm = (2.0 * fetch_kb + write_kb) * 1024.0
out["dominant_kernel"] = short
out[short + "_hbm_bytes_per_launch"] = hbm
out[short + "_fetch_raw_bytes_per_launch"] = fetch_kb * 1024.0
out[short + "_write_bytes_per_launch"] = write_kb * 1024.0
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
json.dump({"workload": out["workload"], "tag": tag, "hbm_bytes_per_launch": hbm,
           "kernel": short,
           "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch of the dominant kernel, separate --pmc passes"},
          open("profiles/traffic_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
