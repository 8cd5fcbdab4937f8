#!/usr/bin/env python3
"""Diagnostic: per-wave phase timestamps of ONE observe_frame_kernel launch (frame SFM_STAMP_FRAME)
from the -DSFM_STAMPS=1 build (build/stamps/libsfm_fusion_stamps.so, made by hand:
hipcc ... -DSFM_STAMPS=1).  Run on the GPU box:  python3 tools/stamps.py [frame]"""
import os, subprocess, sys
import numpy as np
frame = sys.argv[1] if len(sys.argv) > 1 else "300"
out = "gpurun_out/stamps_%s.bin" % frame
os.makedirs("gpurun_out", exist_ok=True)
env = dict(os.environ, SFM_FUSION_LIB=os.path.abspath("build/stamps/libsfm_fusion_stamps.so"),
           SFM_STAMP_FRAME=frame, SFM_STAMP_OUT=out)
subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env,
               check=True, stdout=subprocess.DEVNULL)
s = np.fromfile(out, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
live = s[:, 0] > 0
s = s[live]
t0 = s[:, 0].min()
us = (s - t0) / 100.0  # 100 MHz
worked = s[:, 6] > 0
print("waves stamped", len(s), "with a chunk", int(worked.sum()))
names = ["entry", "counts read", "chunk loaded", "observed", "barrier passed", "masks scattered", "chunk done"]
for i, n in enumerate(names):
    col = us[worked, i] if i >= 2 else us[:, i]
    print("%-16s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % (
        n, col.min(), np.percentile(col, 10), np.percentile(col, 50), np.percentile(col, 90), col.max()))
d = np.diff(us[worked][:, :7], axis=1)
for i in range(6):
    print("%-16s -> %-16s p50 %6.2f  p90 %6.2f  max %6.2f" % (names[i], names[i + 1], np.percentile(d[:, i], 50),
                                                        np.percentile(d[:, i], 90), d[:, i].max()))
wg = us.reshape(-1, 4, 8) if len(us) % 4 == 0 else None
if wg is not None:
    entry = wg[:, 0, 0]
    print("workgroups entering later than 5 us:", int((entry > 5).sum()), "of", len(entry))
print("span first entry -> last done: %.2f us" % us[:, 6].max())
