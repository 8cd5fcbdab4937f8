"""Full BASELINE sizes (640x480 keyframes), where the oracle is too slow to be the checker
for everything: a 24-keyframe prefix is compared with the oracle bit for bit, and a longer
run is checked through size-independent properties of the fused cloud."""
import numpy as np
import pytest

from oracle import oracle
from sfm_amd import fusion, synth
from helpers import assert_same_cloud, both_params, run_gpu

pytestmark = pytest.mark.gpu


def _ctx_from_device_sequence(seq, gp, n):
    import torch
    ctx = fusion.FusionContext(gp)
    H, W = seq.depth.shape[1:]
    for i in range(n):
        ctx.add_keyframe_device(seq.depth[i].data_ptr(), seq.bgr[i].data_ptr(), W, H, seq.K[i], seq.T_wk[i], seq.T_kw[i])
    torch.cuda.synchronize()
    for i in range(n):
        ctx.set_neighbors(i, [j for j in seq.nbrs[i] if j < n])
    return ctx


def test_c2_prefix_against_oracle():
    seq = synth.make_sequence(24, 640, 480, device="cuda")
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    want = oracle.fuse(frames, op, n_threads=8)
    assert want.key.shape[0] > 100000
    assert_same_cloud(run_gpu(frames, gp), want)


def test_c3_prefix_against_oracle():
    """1280x720 (C3's keyframe size): more pixels per workgroup than one pass of the persistent kernel
    takes, and first keyframes whose share is all live (the pass is redone in parts)."""
    seq = synth.make_sequence(24, 1280, 720, K=(913.848, 913.602, 642.941, 371.196), device="cuda")
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    want = oracle.fuse(frames, op, n_threads=8)
    assert want.key.shape[0] > 200000
    assert_same_cloud(run_gpu(frames, gp), want)


def test_c2_properties_at_scale():
    n = 120
    seq = synth.make_sequence(n, 640, 480, device="cuda")
    _, gp = both_params(voxel_size=10.0, T_dist=500.0)
    ctx = _ctx_from_device_sequence(seq, gp, n)
    ctx.run()
    xyz, bgr, key, tag = ctx.read_points()
    st = ctx.stats()
    fb, fa = ctx.frame_counts()
    N = key.shape[0]
    assert N == st["n_vox"] == st["global_after"] == st["table_used"] > 300000
    # sorted, unique voxel keys
    o = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
    assert np.array_equal(o, np.arange(N))
    assert np.unique(key, axis=0).shape[0] == N
    # every point lies in the voxel its key names: key == trunc(xyz / voxel)  (signed view)
    assert np.array_equal(key.view(np.int64), np.trunc(xyz / 10.0).astype(np.int64))
    # tags point at real pixels; the colour is that pixel's colour
    f_idx = (tag >> np.uint64(32)).astype(np.int64)
    pix = (tag & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert f_idx.max() < n and pix.max() < 640 * 480
    sample = np.random.default_rng(0).choice(N, size=2000, replace=False)
    host_bgr = {}
    for s in sample:
        f = int(f_idx[s])
        if f not in host_bgr:
            host_bgr[f] = seq.bgr_u8(f).reshape(-1, 3)
        assert host_bgr[f][pix[s]].tolist() == bgr[s].tolist()
    # the reference's four log counts are consistent with each other
    assert st["n_acc"] == int(fb.sum()) and st["global_before"] == int(fa.sum())
    assert (fa <= fb).all() and st["global_after"] <= st["global_before"] <= st["n_acc"] <= st["n_px"]
    assert st["n_hit"] <= st["n_gather"] and st["n_px"] == n * 640 * 480
    # the room is 6 x 3 x 7 m: all points inside it (raw units = mm), on its surfaces
    assert np.abs(xyz[:, 0]).max() < 3100 and np.abs(xyz[:, 1]).max() < 1600 and np.abs(xyz[:, 2]).max() < 3600
    # idempotence: a second run gives the identical cloud
    ctx.run()
    xyz2, bgr2, key2, tag2 = ctx.read_points()
    assert np.array_equal(key2, key) and np.array_equal(tag2, tag) and np.array_equal(xyz2, xyz)
    # halving the voxel size refines the cloud: parents of the fine keys == the coarse keys
    ctx.close()
    _, gp2 = both_params(voxel_size=20.0, T_dist=500.0)
    ctx = _ctx_from_device_sequence(seq, gp2, n)
    ctx.run()
    key20 = ctx.read_points()[2]
    parents = np.unique(np.trunc(np.trunc(xyz / 10.0) * 10.0 / 20.0).astype(np.int64), axis=0)
    # (every 1-cm winner lies in some 2-cm voxel that must be present)
    have = set(map(tuple, key20.view(np.int64).tolist()))
    fine_parents = set(map(tuple, np.trunc(xyz / 20.0).astype(np.int64).tolist()))
    assert fine_parents <= have
    ctx.close()


def test_c2_full_workload_against_oracle(fusion_path):
    """The whole bench workload (BASELINE.json configs[1]: 500 keyframes 640x480, the generator and
    seed bench.py uses), bit for bit against the oracle: 153.6 M pixels, 1.59 M points, ~5 s of oracle
    time on the GPU box's 16 host threads.  Once (on the default path): the oracle run dominates."""
    if fusion_path != "chain":
        pytest.skip("full-size oracle run once, on the default path")
    import torch
    n = 500
    seq = synth.make_sequence(n, 640, 480, K=(525.0, 525.0, 319.5, 239.5), device=torch.device("cuda", 0),
                              seed=20240315, components=1, radius_m=0.8)
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    ctx = _ctx_from_device_sequence(seq, gp, n)
    ctx.run()
    xyz, bgr, key, tag = ctx.read_points()
    st = ctx.stats()
    fb, fa = ctx.frame_counts()
    ctx.close()
    assert st["fused_chain"] == 1
    want = oracle.fuse(seq.frame_dicts(0, n), op, n_threads=16)
    assert want.key.shape[0] > 1500000
    assert_same_cloud(dict(xyz=xyz, bgr=bgr, key=key, tag=tag, stats=st, frame_before=fb, frame_after=fa), want)


def test_c2_full_workload_in_four_blocks_equals_the_single_run(fusion_path):
    """The same 500-keyframe workload cut into four blocks as a sequence split runs them -- rank 0 on the fused chain, the
    others in two phases, every rank with successors through sfm_fusion_run_chain / export_masks / sfm_fusion_finish --
    then the voxel exchange: the union is the single run's cloud bit for bit (which the test above pins to the oracle),
    and the ranks' counters add up to the single run's."""
    if fusion_path != "chain":
        pytest.skip("full-size run once, on the default path")
    import torch
    from test_gpu_sequence_split import _exchange, _run_split
    n = 500
    seq = synth.make_sequence(n, 640, 480, K=(525.0, 525.0, 319.5, 239.5), device=torch.device("cuda", 0),
                              seed=20240315, components=1, radius_m=0.8)
    _, gp = both_params(voxel_size=10.0, T_dist=500.0)
    ctx = _ctx_from_device_sequence(seq, gp, n)
    ctx.run()
    xyz, bgr, key, tag = ctx.read_points()
    st = ctx.stats()
    ctx.close()
    frames = seq.frame_dicts(0, n)
    plans, ctxs = _run_split(frames, gp, 4, two_phase_later=True)
    assert [c.stats()["precomputed"] for c in ctxs] == [0, 1, 1, 1]
    for k in ("n_px", "n_gather", "n_acc", "n_hit"):
        assert sum(c.stats()[k] for c in ctxs) == st[k], k
    xyz4, bgr4, key4, tag4 = _exchange(ctxs, 4)
    for c in ctxs:
        c.close()
    assert key.shape[0] > 1500000
    assert np.array_equal(key4, key) and np.array_equal(tag4, tag) and np.array_equal(bgr4, bgr)
    assert np.array_equal(xyz4.view(np.uint64), xyz.view(np.uint64))
