#!/usr/bin/env python3
"""End-to-end time of the reference's `reconstruct` tail on the C2 workload: a keyframe replay dump
on disk -> sfm_amd/cpp/replay_reconstruct (mmap, upload, fuse, read back, write the text cloud or a
binary PLY).  Diagnostic for DESIGN.md section 7; bench.py stays the measurement."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sfm_amd import fusion, replay, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_kf, W, H = 500, 640, 480
seq = synth.make_sequence(n_kf, W, H, K=(525.0, 525.0, 319.5, 239.5), device=torch.device("cuda", 0), seed=20240315,
                          components=1, radius_m=0.8)
m = fusion.Map()
for i in range(n_kf):
    m.keyframes.append(fusion.KeyFrame(img=seq.bgr[i].cpu().numpy(), depth_img=seq.depth[i].cpu().numpy().view(np.uint16),
                                       K=seq.K[i], T_wk=seq.T_wk[i], T_kw=seq.T_kw[i]))
for i in range(n_kf):
    for j in seq.nbrs[i]:
        if j > i:
            m.insert_edge(i, j, 150)
tmp = os.environ.get("TMPDIR", "/tmp")
dump = os.path.join(tmp, "c2.sfmr")
t0 = time.perf_counter()
replay.write_replay(dump, m)
print("dump written: %.1f MB in %.2f s" % (os.path.getsize(dump) / 1e6, time.perf_counter() - t0), flush=True)
exe = os.path.join(ROOT, "sfm_amd", "cpp", "replay_reconstruct")
for mode in (["--voxel-size", "10", "--t-dist", "500", "--timing"], ["--voxel-size", "10", "--t-dist", "500", "--timing"],
             ["--ply", "--voxel-size", "10", "--t-dist", "500", "--timing"]):
    out = os.path.join(tmp, "c2_cloud.out")
    t0 = time.perf_counter()
    r = subprocess.run([exe, dump, out] + mode, capture_output=True, text=True, timeout=300)
    dt = time.perf_counter() - t0
    print(" ".join(mode), "| rc", r.returncode, "| process %.3f s | output %.1f MB" % (dt, os.path.getsize(out) / 1e6))
    print("   ", r.stdout.strip().replace("\n", "\n    "), r.stderr.strip()[-300:], flush=True)
os.remove(dump)
