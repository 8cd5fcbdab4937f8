"""GPU parity on the edge cases the algorithm has: empty and ragged inputs, keyframes without
neighbours, zero-depth pathologies (NaN points), mixed intrinsics and sizes, strided host
images, many neighbours (more than one neighbour per wave), explicit capacities, repeated
runs, and the C-ABI's error behaviour."""
import numpy as np
import pytest

from oracle import oracle
from sfm_amd import fusion, synth
from helpers import assert_same_cloud, both_params, golden_names, load_golden, run_gpu
import scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_names())
def test_golden_fixtures(name):
    frames, params, want = load_golden(name)
    _, gp = both_params(**params)
    assert_same_cloud(run_gpu(frames, gp), want)


def test_plane_pair_with_reference_defaults():
    frames = scenes.plane_pair()
    op, gp = both_params()
    assert_same_cloud(run_gpu(frames, gp), oracle.fuse(frames, op))


def test_empty_context_and_empty_images():
    _, gp = both_params()
    g = run_gpu([], gp)
    assert g["key"].shape == (0, 3) and g["stats"]["n_vox"] == 0
    frames = [scenes.frame(np.zeros((0, 0)), nbrs=[1]), scenes.frame(np.full((6, 9), 30), K=(10, 10, 4, 3), nbrs=[0])]
    op, gp = both_params()
    assert_same_cloud(run_gpu(frames, gp), oracle.fuse(frames, op))


def test_keyframes_without_neighbours_yield_nothing():
    frames = [scenes.frame(np.full((20, 30), 35), K=(40, 40, 14.5, 9.5), nbrs=[])]
    op, gp = both_params()
    g = run_gpu(frames, gp)
    assert g["key"].shape[0] == 0
    assert_same_cloud(g, oracle.fuse(frames, op))
    # identical poses: the only neighbour is filtered out (KAT-6)
    f0 = scenes.frame(np.full((12, 16), 30), K=(20, 20, 7.5, 5.5), nbrs=[1])
    f1 = scenes.frame(np.full((12, 16), 30), K=(20, 20, 7.5, 5.5), nbrs=[0])
    g = run_gpu([f0, f1], gp)
    assert g["key"].shape[0] == 0 and g["frame_before"].tolist() == [0, 0]


def test_zero_depth_observations_make_nan_points():
    frames = scenes.plane_pair()
    frames[1]["depth"] = frames[1]["depth"].copy()
    frames[1]["depth"][3:30, 4:40:3] = 0
    op, gp = both_params(T_dist=100.0, voxel_size=0.5)
    c = oracle.fuse(frames, op)
    assert np.isnan(c.xyz).any() and (c.key == np.uint64(0x8000000000000000)).any()
    assert_same_cloud(run_gpu(frames, gp), c)


def test_mixed_sizes_intrinsics_and_many_neighbours():
    rng = np.random.default_rng(3)
    frames = []
    n = 11
    for i in range(n):
        H, W = (24 + 3 * (i % 3), 33 + 5 * (i % 2))
        d = (35 + rng.integers(0, 4, size=(H, W))).astype(np.uint16)
        K = (40.0 + i, 41.0 - 0.5 * i, W / 2 - 0.5 + 0.1 * i, H / 2 - 0.5)
        frames.append(scenes.frame(d, K=K, C=(0.4 * i, 0.05 * i, 0.02 * i), yaw=0.7 * i,
                                   nbrs=[j for j in range(n) if j != i]))       # 10 neighbours > 8 waves
    op, gp = both_params(T_dist=8.0, voxel_size=0.8)
    c = oracle.fuse(frames, op)
    assert c.key.shape[0] > 200
    assert_same_cloud(run_gpu(frames, gp), c)


def test_strided_host_rows_and_device_pointers():
    seq = synth.make_sequence(4, 50, 30, K=(131.25, 131.25, 24.5, 14.5), window=2)
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    ctx = fusion.FusionContext(gp)
    for f in frames:
        big_d = np.zeros((30, 64), dtype=np.uint16)
        big_c = np.zeros((30, 70, 3), dtype=np.uint8)
        big_d[:, :50] = f["depth"]
        big_c[:, :50] = f["bgr"]
        ctx.add_keyframe(big_d[:, :50], big_c[:, :50], f["K"], f["T_wk"], f["T_kw"])   # row strides 128 / 210 B
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run()
    xyz, bgr, key, tag = ctx.read_points()
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag) and np.array_equal(bgr, want.bgr)
    assert np.array_equal(xyz, want.xyz)
    ctx.close()
    import torch
    ctx = fusion.FusionContext(gp)
    keep = []
    for f in frames:
        d = torch.from_numpy(f["depth"].view(np.int16).copy()).cuda()
        c = torch.from_numpy(f["bgr"].copy()).cuda()
        keep += [d, c]
        ctx.add_keyframe_device(d.data_ptr(), c.data_ptr(), 50, 30, f["K"], f["T_wk"], f["T_kw"])
    torch.cuda.synchronize()
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run()
    _, _, key2, tag2 = ctx.read_points()
    assert np.array_equal(key2, want.key) and np.array_equal(tag2, want.tag)
    ctx.close()


def test_rerun_is_idempotent_and_capacities_grow():
    seq = synth.make_sequence(6, 96, 72, K=(196.8, 196.8, 47.5, 35.5), window=2)
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    for _ in range(3):
        ctx.run()
        xyz, bgr, key, tag = ctx.read_points()
        assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag) and np.array_equal(xyz, want.xyz)
    ctx.close()
    # capacities far too small: explicit ones must fail loudly, automatic ones must grow
    _, tiny = both_params(voxel_size=10.0, T_dist=500.0, max_points=64)
    with pytest.raises(fusion.FusionError) as e:
        run_gpu(frames, tiny)
    assert e.value.code == fusion.SFM_E_CAPACITY
    _, tiny = both_params(voxel_size=1.0, T_dist=500.0, table_capacity=1024)
    with pytest.raises(fusion.FusionError) as e:
        run_gpu(frames, tiny)
    assert e.value.code == fusion.SFM_E_TABLE_FULL


def _shifted(frames, offset):
    """The same scene somewhere else in the world: every pose translated by `offset` (raw units)."""
    out = []
    for f in frames:
        T_wk = np.array(f["T_wk"], dtype=np.float64)
        T_wk[:3, 3] += offset
        T_kw = np.array(f["T_kw"], dtype=np.float64)
        T_kw[:3, 3] = -T_kw[:3, :3] @ T_wk[:3, 3]
        out.append(dict(f, T_wk=T_wk, T_kw=T_kw))
    return out


def test_voxel_indices_beyond_the_default_key_range_stay_exact():
    """Voxel indices outside +-(2^20 - 1) (dense_reconstruction.cpp:270-274 has no limit): the run notices, scans the
    index ranges and packs the 64-bit key for exactly those -- the cloud still equals the oracle's three size_t."""
    frames = scenes.plane_pair()
    op, gp = both_params(voxel_size=1e-6)      # 40 / 1e-6 = 4e7 voxels from the origin > 2^20, on one axis
    want = oracle.fuse(frames, op)
    assert int(want.key.view(np.int64).max()) > 2 ** 21
    assert_same_cloud(run_gpu(frames, gp), want)
    # a room far from the origin in all three axes, negative side included: small extents, huge offsets
    seq = synth.make_sequence(6, 96, 72, K=(196.8, 196.8, 47.5, 35.5), window=2)
    frames = _shifted(seq.frame_dicts(), np.array([5.0e8, -3.0e8, 2.0e9]))
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    k = want.key.view(np.int64)
    assert k[:, 0].min() > 4e7 and k[:, 1].max() < -2e7 and k[:, 2].min() > 1e8 and want.key.shape[0] > 5000
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    assert ctx.key_range() == ((-1048575,) * 3, (1048575,) * 3)
    ctx.run()
    lo, hi = ctx.key_range()
    for ax in range(3):
        assert lo[ax] == k[:, ax].min() or lo[ax] < k[:, ax].min()    # (the two-phase scan sees rejected-by-mask pixels too)
        assert hi[ax] >= k[:, ax].max() and hi[ax] - lo[ax] < 4000
    xyz, bgr, key, tag = ctx.read_points()
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag)
    assert np.array_equal(xyz.view(np.uint64), want.xyz.view(np.uint64)) and np.array_equal(bgr, want.bgr)
    st = ctx.stats()
    assert st["n_vox"] == want.counters["n_vox"] and st["n_hit"] == want.counters["n_hit"]
    # the ctx keeps the ranges: the next run needs no scan and gives the same cloud
    ctx.run()
    assert ctx.key_range() == (lo, hi) and np.array_equal(ctx.read_points()[2], want.key)
    # ranges that do not hold the cloud at hand are refused and change nothing
    with pytest.raises(fusion.FusionError) as e:
        ctx.set_key_range((0, 0, 0), (100, 100, 100))
    assert e.value.code == fusion.SFM_E_KEY_RANGE and ctx.key_range() == (lo, hi)
    # wider ones re-encode it in place
    ctx.set_key_range([x - 1000 for x in lo], [x + 5 for x in hi])
    assert np.array_equal(ctx.read_points()[2], want.key) and np.array_equal(ctx.read_points()[3], want.tag)
    ctx.close()


def test_a_cloud_no_exact_64_bit_key_can_hold_is_an_error():
    seq = synth.make_sequence(4, 48, 36, K=(98.4, 98.4, 23.5, 17.5), window=2)
    frames = seq.frame_dicts()
    _, gp = both_params(voxel_size=1e-6, T_dist=500.0)   # metres of wall in nanometre voxels: about 30 bits per axis
    with pytest.raises(fusion.FusionError) as e:
        run_gpu(frames, gp)
    assert e.value.code == fusion.SFM_E_KEY_RANGE and "63" in str(e.value)


def test_argument_errors():
    ctx = fusion.FusionContext()
    f = scenes.plane_pair()[0]
    ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    with pytest.raises(fusion.FusionError):
        ctx.num_points()                       # no run yet
    ctx.set_neighbors(0, [0])
    with pytest.raises(fusion.FusionError):
        ctx.run()                              # own neighbour
    ctx.set_neighbors(0, [5])
    with pytest.raises(fusion.FusionError):
        ctx.run()                              # unknown neighbour
    ctx.close()


def test_more_neighbours_than_the_fused_kernels_hold():
    """The reference has no neighbour cap (map.cpp:152-160).  39 and 71 mutual neighbours per keyframe: more than
    SFM_MAX_NEIGHBORS, so the run takes the two-phase chain (a pixel's neighbours are a loop there) on every path."""
    far = []
    for i in range(40):
        far.append(scenes.frame(np.full((4, 4), 30), K=(5, 5, 1.5, 1.5), C=(0.1 * i, 0, 0),
                                nbrs=[j for j in range(40) if j != i]))
    op, gp = both_params()
    g = run_gpu(far, gp)
    assert g["stats"]["precomputed"] == 1
    assert_same_cloud(g, oracle.fuse(far, op))
    seq = synth.make_sequence(72, 64, 48, K=(131.25, 131.25, 31.5, 23.5), window=71, yaw_step_deg=0.5)
    frames = seq.frame_dicts()
    assert min(len(f["nbrs"]) for f in frames) == 71
    op, gp = both_params(voxel_size=20.0, T_dist=500.0)
    want = oracle.fuse(frames, op, n_threads=8)
    assert want.key.shape[0] > 1000 and want.counters["n_hit"] > 10 * want.counters["n_acc"]
    g = run_gpu(frames, gp)
    assert g["stats"]["precomputed"] == 1
    assert_same_cloud(g, want)
    # 33 neighbours on ONE keyframe only is enough to switch the run over; 32 is not
    seq = synth.make_sequence(40, 48, 36, K=(98.4, 98.4, 23.5, 17.5), window=2, yaw_step_deg=0.5)
    frames = seq.frame_dicts()
    frames[3]["nbrs"] = [j for j in range(34) if j != 3]
    assert_same_cloud(run_gpu(frames, gp), oracle.fuse(frames, op))
    frames[3]["nbrs"] = [j for j in range(33) if j != 3]
    g = run_gpu(frames, gp)
    assert_same_cloud(g, oracle.fuse(frames, op))


def _run_raw(frames, gp):
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run()
    xyz, bgr, key, tag = ctx.read_points()
    st = ctx.stats()
    fb, fa = ctx.frame_counts()
    ctx.close()
    return dict(xyz=xyz, bgr=bgr, key=key, tag=tag, stats=st, frame_before=fb, frame_after=fa)


def test_division_operands_outside_the_shared_range_fall_back_to_plain_division(fusion_path):
    """fuse_chain_kernel divides in pairs with a shared reciprocal, exact for operands within
    [2^-400, 2^400]; anything else must end on the kernel pair (plain divisions), never in a wrong bit.  (The
    two-phase chain picks verdict_kernel's plain-division instantiation instead and keeps the barrier chain.)"""
    two_phase = fusion_path == "verdict"
    frames = scenes.plane_pair()
    for f in frames:
        f["T_kw"] = f["T_kw"] * 2.0 ** -500          # camera coordinates of ~2^-495: same projections, exactly
    op, gp = both_params()
    want = oracle.fuse(frames, op)
    assert want.key.shape[0] > 100
    g = _run_raw(frames, gp)
    assert g["stats"]["fused_chain"] == int(two_phase)   # refused on the host (chain_div_ok): the pair runs
    assert g["stats"]["precomputed"] == int(two_phase)
    assert_same_cloud(g, want)
    frames = scenes.plane_pair()
    op, gp = both_params(base_line=2.0 ** -450)      # disparity^2 ~ 2^-890: refused on the host before the launch
    want = oracle.fuse(frames, op)
    g = _run_raw(frames, gp)
    assert g["stats"]["fused_chain"] == int(two_phase)
    assert_same_cloud(g, want)


def test_run_on_the_callers_stream():
    """`stream` of sfm_fusion_create: the chain runs on a stream the caller owns (here a torch stream that is
    not the default one); the library's second stream and its events hang off it."""
    import torch
    seq = synth.make_sequence(5, 96, 72, K=(196.8, 196.8, 47.5, 35.5), window=2)
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    s = torch.cuda.Stream()
    ctx = fusion.FusionContext(gp, 0, s.cuda_stream)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    for _ in range(2):
        ctx.run()
        xyz, bgr, key, tag = ctx.read_points()
        assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag) and np.array_equal(bgr, want.bgr)
        assert np.array_equal(xyz.view(np.uint64), want.xyz.view(np.uint64))
    ctx.close()
    s.synchronize()
