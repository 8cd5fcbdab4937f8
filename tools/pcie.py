#!/usr/bin/env python3
"""PCIe-inclusive time of the C2 workload through the host-buffer boundary:
add_keyframe from host numpy images (H2D inside), run, read_points (D2H + sort by key).
Never the bench value: bench.py times HBM-resident inputs."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sfm_amd import fusion, synth

n_kf, W, H = 500, 640, 480
dev = torch.device("cuda", 0)
seq = synth.make_sequence(n_kf, W, H, K=(525.0, 525.0, 319.5, 239.5), device=dev, seed=20240315, components=1,
                          radius_m=0.8)
depth = [seq.depth[i].cpu().numpy().view(np.uint16) for i in range(n_kf)]
bgr = [seq.bgr[i].cpu().numpy() for i in range(n_kf)]
torch.cuda.synchronize()
res = []
for rep in range(3):
    t0 = time.perf_counter()
    ctx = fusion.FusionContext(fusion.default_params(voxel_size=10.0, T_dist=500.0), 0)
    t1 = time.perf_counter()
    for i in range(n_kf):
        ctx.add_keyframe(depth[i], bgr[i], seq.K[i], seq.T_wk[i], seq.T_kw[i])
    for i in range(n_kf):
        ctx.set_neighbors(i, seq.nbrs[i])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ctx.run()
    t3 = time.perf_counter()
    xyz, col, key, tag = ctx.read_points()
    t4 = time.perf_counter()
    ctx.close()
    res.append({"create_ms": (t1 - t0) * 1e3, "upload_ms": (t2 - t1) * 1e3, "run_ms": (t3 - t2) * 1e3,
                "read_points_ms": (t4 - t3) * 1e3, "points": int(key.shape[0]),
                "px_per_s_upload_run": n_kf * W * H / (t3 - t1),
                "px_per_s_upload_run_read": n_kf * W * H / (t4 - t1)})
print(json.dumps({"workload": "c2_500kf_640x480_1cm", "input_bytes": n_kf * W * H * 5, "runs": res}))
