"""Summarises gpurun_out/<tag>_{kt,fetch,write} into profiles/<tag>_summary.json and
profiles/traffic_latest.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
two_phase = "--two-phase" in sys.argv[2:]
workload = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "c2_500kf_640x480_1cm"
out = {"tag": tag, "workload": workload, "bench_args": "--two-phase" if two_phase else "", "kernels": {}}
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
if two_phase:
    k = "sfm::accept_chain_kernel"
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
if two_phase:  # phase A as well: one verdict_kernel launch per 32 keyframes
    for kk in pmc["fetch"]:
        if "verdict_kernel" in kk:
            out["verdict_kernel_hbm_bytes_per_launch"] = (2.0 * pmc["fetch"][kk]["kb_per_launch"] + pmc["write"][kk]["kb_per_launch"]) * 1024.0
            out["verdict_kernel_fetch_raw_bytes_per_launch"] = pmc["fetch"][kk]["kb_per_launch"] * 1024.0
            out["verdict_kernel_write_bytes_per_launch"] = pmc["write"][kk]["kb_per_launch"] * 1024.0
# optional: SQ instruction counters of the same command (tools/profile_pass.sh <tag> valu) and the micro-benchmarks
# (gpurun_out/<tag>_valu_bench.txt, <tag>_barrier_bench.txt) -> profiles/bounds_latest.json, read by bench.py for
# roofline.fp64 and roofline.chain
vf = glob.glob(f"gpurun_out/{tag}_valu/*/*counter_collection.csv")
if vf:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for r in csv.DictReader(open(vf[0])):
        if "sfm::" in r["Kernel_Name"]:
            kname = r["Kernel_Name"].split("(")[0]
            agg[kname][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVES":
                launches[kname] += 1
    out["sq"] = {kname: dict(launches=launches[kname], **{c: v / max(1, launches[kname]) for c, v in cs.items()})
                 for kname, cs in agg.items()}
    bounds = {"workload": out["workload"], "tag": tag, "kernel": short}
    if two_phase:
        bounds = None
if vf and bounds is not None:
    sq = out["sq"].get(k)
    if sq:
        # CDNA4: a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles, FP64 fma/mul/add included
        # (tools/valu_bench.hip); 256 CUs x 4 SIMDs; 2.4 GHz peak engine clock
        bounds["valu_insts_per_launch"] = sq.get("SQ_INSTS_VALU", 0.0)
        bounds["valu_issue_floor_us"] = sq.get("SQ_INSTS_VALU", 0.0) * 4.0 / (1024 * 2.4e9) * 1e6
        bounds["sq_per_launch"] = sq
    import re
    try:
        txt = open(f"gpurun_out/{tag}_barrier_bench.txt").read()
        rows = re.findall(r"fenced (\d) traffic (\d) levels (\d) threads\s+(\d+) G\s+(\d+): rc 0 abort 0\s+([0-9.]+) us per barrier", txt)
        pick = [float(r[5]) for r in rows if r[0] == "0" and r[1] == "0" and r[2] == "2" and r[4] == "256"]
        allr = [float(r[5]) for r in rows if r[0] == "0" and r[4] == "256"]
        bounds["barrier_us"] = min(pick) if pick else (min(allr) if allr else None)
        bounds["barrier_note"] = ("tools/barrier_bench.hip: two-level counters + flag, relaxed agent-scope atomics, 256 "
                                  "co-resident workgroups, nothing between the barriers (the latency of the barrier alone)")
    except OSError:
        pass
    try:
        # measured cost of a plain wave64 VALU instruction (loop overhead included): v_add_f64 row of tools/valu_bench.hip
        txt = open(f"gpurun_out/{tag}_valu_bench.txt").read()
        m = re.search(r"v_add_f64\s+[0-9.]+ ms\s+([0-9.]+) ns per step", txt)
        if m and sq:
            bounds["valu_ns_per_inst_measured"] = float(m.group(1))
            bounds["valu_issue_floor_us_measured_rate"] = sq.get("SQ_INSTS_VALU", 0.0) * float(m.group(1)) / 1024 * 1e-3
    except OSError:
        pass
    try:
        # the chain with its arithmetic removed: accept_chain_kernel alone (bench.py --two-phase), per keyframe
        tp = json.loads([l for l in open(f"gpurun_out/{tag}_two_phase.json") if l.startswith("{")][-1])
        bounds["accept_link_us"] = tp["two_phase"]["phase_b_ms"] * 1e3 / tp["config"]["keyframes"]
        bounds["phase_a_us_per_keyframe"] = tp["two_phase"]["phase_a_ms"] * 1e3 / tp["config"]["keyframes"]
    except (OSError, KeyError, IndexError):
        pass
    json.dump(bounds, open("profiles/bounds_latest.json", "w"), indent=1)
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)
json.dump({"workload": out["workload"], "tag": tag, "hbm_bytes_per_launch": hbm,
           "kernel": short,
           "note": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per launch of the dominant kernel, separate --pmc passes"},
          open("profiles/traffic_two_phase.json" if two_phase else "profiles/traffic_latest.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
