"""How much re-observation work a one-keyframe-lag pipeline of the keyframe chain would do:
pixels of keyframe f that are still unmasked when the chain has finished keyframe f-1-lag, relative to those unmasked
when f's own turn comes (lag 0 = what fuse_chain_kernel processes today).  CPU only (the oracle's block-wise variant
with mask output), C2's first 40 keyframes.  Writes profiles/r02_lag_superset.json."""
import json, os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
from sfm_amd import synth
n, W, H = 40, 640, 480
seq = synth.make_sequence(500, W, H, K=(525.0, 525.0, 319.5, 239.5), device="cpu", seed=20240315, components=1, radius_m=0.8, only=(0, n))
frames = [dict(depth=seq.depth_u16(i), bgr=seq.bgr_u8(i), K=seq.K[i], T_wk=seq.T_wk[i], T_kw=seq.T_kw[i], nbrs=list(seq.nbrs[i])) for i in range(n)]
for f in frames: f["nbrs"] = [j for j in f["nbrs"] if j < n]
p = oracle.default_params(voxel_size=10.0, T_dist=500.0)
t = time.time()
masks_at = {}
for k in range(24, 37):
    c, m = oracle.fuse(frames, p, n_threads=8, active=(0, k), want_masks=True)
    masks_at[k] = m
print("time", time.time() - t)
rows = []
for f in range(28, 36):
    d = frames[f]["depth"] > 0
    live = {lag: int(((masks_at[f - lag][f] == 0) & d).sum()) for lag in range(0, 5)}
    rows.append({"keyframe": f, "live_px_by_lag": live, "ratio_by_lag": [round(live[l] / live[0], 3) for l in range(5)]})
    print(f, live, rows[-1]["ratio_by_lag"])
mean = [round(float(np.mean([r["ratio_by_lag"][l] for r in rows])), 3) for l in range(5)]
out = {"workload": "c2_500kf_640x480_1cm, keyframes 0..39, temporal +-4 neighbours", "mean_ratio_by_lag": mean, "rows": rows,
       "note": "lag L = verdicts of keyframe f computed against the masks as they are after keyframe f-1-L"}
with open(os.path.join(ROOT, "profiles", "r02_lag_superset.json"), "w") as fp:
    json.dump(out, fp, indent=1)
print("mean ratio by lag", mean)
