#!/usr/bin/env python3
"""Diagnostic: thread-0 phase timestamps of every workgroup of ONE fuse_chain_kernel launch (keyframe
group SFM_STAMP_GROUP) from the -DSFM_STAMPS=1 build.  Run on the GPU box: python3 tools/chain_stamps.py [group]"""
import os, subprocess, sys
import numpy as np
group = sys.argv[1] if len(sys.argv) > 1 else "9"
extra = sys.argv[2:]  # passed on to bench.py, e.g. --two-phase --workload c3_2000kf_1280x720_1cm
out = "gpurun_out/cstamps_%s.bin" % group
os.makedirs("gpurun_out", exist_ok=True)
env = dict(os.environ, SFM_FUSION_LIB=os.path.abspath(os.environ.get("SFM_STAMPS_LIB", "build/stamps/libsfm_fusion_stamps.so")),
           SFM_STAMP_GROUP=group, SFM_STAMP_OUT=out)
subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"] + extra, env=env,
               check=True, stdout=subprocess.DEVNULL)
raw = np.fromfile(out, dtype=np.uint64).astype(np.int64)
SLOTS = 48   # kStampSlots: 16 of thread 0, then 16 + 16 per wave
G = raw.size // (32 * SLOTS)
s = raw.reshape(32, G, SLOTS)
t0 = s[0, :, 0].min()
names = ["frame top", "barrier passed", "A done (survivors in LDS)", "B done (first round)", "C done (last round)",
         "rounds done", "arrived", None, "C: decided", "C: masks issued", "C: sums done", "C: key done", "C: record issued"]
print("workgroups", G)
for f in (1, 2, 8, 16, 30):
    fr = s[f]
    base = fr[:, 0].min()
    print("keyframe +%d: survivors per workgroup min %d mean %.0f max %d; frame span %.2f us" % (
        f, fr[:, 7].min(), fr[:, 7].mean(), fr[:, 7].max(), (s[f + 1][:, 0].min() - base) / 100.0 if f + 1 < 32 else -1))
    for i, n in enumerate(names):
        if n is None: continue
        col = (fr[:, i] - base) / 100.0
        col = col[fr[:, i] > 0]
        if len(col):
            print("   %-28s min %6.2f  p50 %6.2f  max %6.2f" % (n, col.min(), np.percentile(col, 50), col.max()))
# per wave (fuse_chain_kernel only): when wave w left step B's first round, and when it was ready for the NEXT keyframe
# (fusion and preparation done), relative to its workgroup's pass of the barrier (thread 0, slot 1)
if s[:, :, 16:48].any() and "--two-phase" in extra:
    # accept_chain_kernel: data waves 1..15, relative to the moment their workgroup passed the barrier
    for f in (2, 8, 16):
        ref = s[f][:, 1:2]
        back = (s[f][:, 17:32] - ref) / 100.0
        acked = (s[f][:, 33:48] - ref) / 100.0
        print("keyframe +%d, data waves 1..15, median over workgroups (us after the workgroup passed the barrier):" % f)
        print("   mask bytes back    " + " ".join("%5.2f" % np.median(back[:, w]) for w in range(15)))
        print("   marks acknowledged " + " ".join("%5.2f" % np.median(acked[:, w]) for w in range(15)))
        print("   slowest wave of a workgroup: masks back p50 %.2f max %.2f; marks acknowledged p50 %.2f max %.2f" % (
            np.median(back.max(axis=1)), back.max(), np.median(acked.max(axis=1)), acked.max()))
elif s[:, :, 16:48].any():
    for f in (2, 8, 16):
        ref = s[f][:, 1:2]
        bdone = (s[f][:, 32:48] - ref) / 100.0
        ready = (s[f + 1][:, 16:32] - ref) / 100.0
        print("keyframe +%d, per wave 0..15, median over workgroups (us after the workgroup passed the barrier):" % f)
        print("   step B done        " + " ".join("%5.1f" % np.median(bdone[:, w]) for w in range(16)))
        print("   ready for the next " + " ".join("%5.1f" % np.median(ready[:, w]) for w in range(16)))
        late = ready.argmax(axis=1)
        print("   the workgroup's last wave to be ready is wave: " + " ".join("%d:%d" % (w, (late == w).sum()) for w in range(16)))
        nxt = (s[f + 1][:, 1] - s[f][:, 1]) / 100.0
        print("   workgroup's own period %.2f us median; last wave ready %.2f us, next barrier passed %.2f us after this one (medians)" % (
            np.median(nxt), np.median(ready.max(axis=1)), np.median(nxt)))
per_frame = np.diff(s[:, :, 1].min(axis=1)) / 100.0
print("per-keyframe period (first workgroup past the barrier): mean %.2f us  min %.2f  max %.2f" % (
    per_frame[1:].mean(), per_frame[1:].min(), per_frame[1:].max()))

