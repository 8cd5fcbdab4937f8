"""Shared helpers for the parity tests: run the HIP path through the C ABI and compare it
with the CPU oracle (bit-exact for keys/tags/colours/counts AND coordinates)."""
import os

import numpy as np

from oracle import oracle
from sfm_amd import fusion

PARAM_KEYS = ("base_line", "T_cov", "voxel_size", "T_dist", "min_dist_filter", "dir_dot",
              "depth_div", "close_frac", "S")


def both_params(**over):
    o = {k: v for k, v in over.items() if k in PARAM_KEYS}
    return oracle.default_params(**o), fusion.default_params(**over)


def run_gpu(frames, gparams, device=0):
    ctx = fusion.FusionContext(gparams, device)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f.get("nbrs", []))
    ctx.run()
    xyz, bgr, key, tag = ctx.read_points()
    st = ctx.stats()
    fb, fa = ctx.frame_counts()
    ctx.close()
    want_path = os.environ.get("SFM_FUSE_CHAIN")
    if want_path is not None and len(frames):
        assert st["fused_chain"] == int(want_path != "0"), "the other keyframe-chain path ran"
        many = any(len(f.get("nbrs", [])) > fusion.SFM_MAX_NEIGHBORS for f in frames)   # those always run two-phase
        if st["fused_chain"] and not many:
            assert st["precomputed"] == int(os.environ.get("SFM_PRECOMPUTE", "0") == "1"), "one-phase / two-phase chain mix-up"
    return dict(xyz=xyz, bgr=bgr, key=key, tag=tag, stats=st, frame_before=fb, frame_after=fa)


def assert_same_cloud(g, c, exact_xyz=True):
    """g: run_gpu() result, c: oracle.Cloud."""
    assert g["key"].shape == c.key.shape, (g["key"].shape, c.key.shape)
    assert np.array_equal(g["key"], c.key), "voxel key sets differ"
    assert np.array_equal(g["tag"], c.tag), "winning (frame,pixel) tags differ"
    assert np.array_equal(g["bgr"], c.bgr), "colours differ"
    if exact_xyz:
        # bit-exact, NaNs included (the kernels emit x86's indefinite NaN like the reference)
        assert np.array_equal(g["xyz"].view(np.uint64), c.xyz.view(np.uint64)), "fused coordinates differ"
    else:
        np.testing.assert_allclose(g["xyz"], c.xyz, rtol=1e-5, atol=0, equal_nan=True)
    assert np.array_equal(g["frame_before"], c.frame_before), "per-frame 'before' counts differ"
    assert np.array_equal(g["frame_after"], c.frame_after), "per-frame 'after' counts differ"
    st = g["stats"]
    assert st["global_before"] == c.global_before
    assert st["global_after"] == c.global_after
    for k in ("n_px", "n_gather", "n_acc", "n_hit", "n_vox"):
        assert st[k] == c.counters[k], (k, st[k], c.counters[k])


# ---- golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py) ----------
import glob
import os

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    n = int(z["n_frames"])
    frames = [dict(depth=z[f"depth_{i}"], bgr=z[f"bgr_{i}"], K=z[f"K_{i}"], T_wk=z[f"T_wk_{i}"],
                   T_kw=z[f"T_kw_{i}"], nbrs=z[f"nbrs_{i}"].tolist()) for i in range(n)]
    params = {}
    for k, v in zip(z["param_names"].tolist(), z["param_values"].tolist()):
        params[k] = tuple(z["S"].tolist()) if k == "S" else float(v)
    cnt = dict(zip(("n_px", "n_gather", "n_acc", "n_hit", "n_vox"), (int(x) for x in z["counters"])))
    want = oracle.Cloud(z["xyz"], z["bgr"], z["key"], z["tag"], z["frame_before"], z["frame_after"],
                        int(z["global_before"]), int(z["global_after"]), cnt)
    return frames, params, want


def assert_clouds_equal(a, b):
    """two oracle.Cloud objects, bit-exact"""
    assert np.array_equal(a.key, b.key) and np.array_equal(a.tag, b.tag) and np.array_equal(a.bgr, b.bgr)
    assert np.array_equal(np.nan_to_num(a.xyz, nan=1.25e300), np.nan_to_num(b.xyz, nan=1.25e300))
    assert np.array_equal(a.frame_before, b.frame_before) and np.array_equal(a.frame_after, b.frame_after)
    assert (a.global_before, a.global_after) == (b.global_before, b.global_after)
    for k in ("n_px", "n_gather", "n_acc", "n_hit", "n_vox"):
        assert a.counters[k] == b.counters[k], k
