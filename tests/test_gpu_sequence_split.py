"""ONE keyframe sequence over several ranks, on one GPU: every rank's ctx fuses its block of keyframes starting from
the masks its predecessors left (sfm_fusion_set_active_range / import_masks / export_masks through the C ABI),
then the voxel exchange.  The union must be the single-process oracle cloud bit for bit, and the traffic
counters must add up to the oracle's."""
import numpy as np
import pytest

from oracle import oracle
from sfm_amd import dist_fusion, fusion, synth
from helpers import both_params

pytestmark = pytest.mark.gpu


def _run_split(frames, gp, world, W=None, H=None, cost=dist_fusion.PIPELINE_COST, two_phase_later=False):
    """W, H None: every keyframe has its own size (the mask buffers are packed back to back).  two_phase_later: ranks > 0
    compute their verdicts first (sfm_fusion_precompute), as SequenceSplit has them do while they wait."""
    import torch
    nbrs = [f["nbrs"] for f in frames]
    px = [f["depth"].shape[0] * f["depth"].shape[1] for f in frames]
    plans = dist_fusion.plan_sequence(nbrs, px, world, cost)
    ctxs, mail = [], {}
    for r, plan in enumerate(plans):
        ctx = fusion.FusionContext(gp)
        for g in plan.frames:
            f = frames[g]
            ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
        for g in range(plan.first, plan.end):
            ctx.set_neighbors(plan.local[g], [plan.local[j] for j in frames[g]["nbrs"]])
        ctx.set_global_order(plan.frames)
        ctx.set_active_range(*plan.active_local)
        if two_phase_later and r > 0:
            ctx.precompute()
        for q, fr in plan.recvs.items():
            buf = mail[(q, r)]
            off = 0
            for first, count in dist_fusion._runs([plan.local[g] for g in fr]):
                nb = sum(px[plan.frames[i]] for i in range(first, first + count))
                ctx.import_masks(first, count, buf.data_ptr() + off, nb)
                off += nb
        # a rank with successors hands its masks on between the two halves of its run, as SequenceSplit does
        if plan.sends:
            ctx.run_chain()
        else:
            ctx.run()
        for t, fr in plan.sends.items():
            buf = torch.empty(max(1, sum(px[g] for g in fr)), dtype=torch.uint8, device="cuda")
            off = 0
            for first, count in dist_fusion._runs([plan.local[g] for g in fr]):
                nb = sum(px[plan.frames[i]] for i in range(first, first + count))
                ctx.export_masks(first, count, buf.data_ptr() + off, nb)
                off += nb
            mail[(r, t)] = buf
        if plan.sends:
            ctx.finish()
        ctxs.append(ctx)
    return plans, ctxs


def _exchange(ctxs, world):
    import torch
    counts = [ctx.partition_counts(world) for ctx in ctxs]
    sends = []
    for r, ctx in enumerate(ctxs):
        n = sum(counts[r])
        buf = torch.empty(max(n, 1) * fusion.SFM_RECORD_BYTES, dtype=torch.uint8, device="cuda")
        ctx.export_records(buf.data_ptr(), n * fusion.SFM_RECORD_BYTES)
        sends.append(buf)
    torch.cuda.synchronize()
    got = []
    for owner, ctx in enumerate(ctxs):
        pieces = []
        for r in range(world):
            start = sum(counts[r][:owner]) * fusion.SFM_RECORD_BYTES
            pieces.append(sends[r][start:start + counts[r][owner] * fusion.SFM_RECORD_BYTES])
        recv = torch.cat(pieces)
        torch.cuda.synchronize()
        ctx.merge(recv.data_ptr(), recv.numel() // fusion.SFM_RECORD_BYTES)
        got.append(ctx.read_points())
    xyz = np.concatenate([g[0] for g in got]); bgr = np.concatenate([g[1] for g in got])
    key = np.concatenate([g[2] for g in got]); tag = np.concatenate([g[3] for g in got])
    return oracle.sort_by_key(xyz, bgr, key, tag)


@pytest.mark.parametrize("world,closure,cost", [(2, False, dist_fusion.PIPELINE_COST), (3, True, None),
                                                (5, False, dist_fusion.PIPELINE_COST), (4, True, (10.0, 40.0, 30.0))])
def test_block_split_of_one_sequence_equals_single_process(world, closure, cost):
    W, H = 96, 72
    seq = synth.make_sequence(41, W, H, K=(78.75, 78.75, 47.5, 35.5), window=3, yaw_step_deg=2.0)
    frames = seq.frame_dicts()
    if closure:
        frames[2]["nbrs"] = sorted(frames[2]["nbrs"] + [40])     # a long-range edge across two blocks
    assert len(dist_fusion.components([f["nbrs"] for f in frames])) == 1
    op, gp = both_params(voxel_size=20.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    plans, ctxs = _run_split(frames, gp, world, W, H, cost)
    stats = [c.stats() for c in ctxs]
    for k in ("n_px", "n_gather", "n_acc", "n_hit"):
        assert sum(s[k] for s in stats) == want.counters[k], k
    fb = np.zeros(len(frames), dtype=np.uint64)
    fa = np.zeros(len(frames), dtype=np.uint64)
    for plan, ctx in zip(plans, ctxs):
        b, a = ctx.frame_counts()
        a0, a1 = plan.active_local
        assert b[:a0].sum() == 0 and b[a1:].sum() == 0     # keyframes outside the block produce nothing
        fb[plan.first:plan.end] = b[a0:a1]
        fa[plan.first:plan.end] = a[a0:a1]
    assert np.array_equal(fb, want.frame_before) and np.array_equal(fa, want.frame_after)
    xyz, bgr, key, tag = _exchange(ctxs, world)
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag)
    assert np.array_equal(bgr, want.bgr) and np.array_equal(xyz.view(np.uint64), want.xyz.view(np.uint64))
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("seed,world", [(1, 2), (4, 3), (8, 2), (12, 3), (17, 4), (22, 2), (33, 3)])
def test_block_split_of_random_scenes(seed, world):
    """The random scenes of tests/test_gpu_fuzz.py -- per-keyframe image sizes and intrinsics, arbitrary and asymmetric
    neighbour lists (edges that jump over whole blocks, keyframes that name a neighbour which does not name them) --
    cut into blocks: the union is the oracle's cloud."""
    from test_gpu_fuzz import random_scene
    frames, params = random_scene(seed)
    op, gp = both_params(**params)
    want = oracle.fuse(frames, op)
    plans, ctxs = _run_split(frames, gp, world)
    stats = [c.stats() for c in ctxs]
    for k in ("n_px", "n_gather", "n_acc", "n_hit"):
        assert sum(s[k] for s in stats) == want.counters[k], k
    xyz, bgr, key, tag = _exchange(ctxs, world)
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag)
    assert np.array_equal(bgr, want.bgr) and np.array_equal(xyz.view(np.uint64), want.xyz.view(np.uint64))
    for c in ctxs:
        c.close()


def test_active_range_and_mask_calls_check_their_arguments():
    import torch
    seq = synth.make_sequence(4, 48, 36, K=(98.4, 98.4, 23.5, 17.5), window=2)
    frames = seq.frame_dicts()
    _, gp = both_params(voxel_size=30.0, T_dist=500.0)
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    buf = torch.zeros(4 * 48 * 36, dtype=torch.uint8, device="cuda")
    for bad in ((3, 2), (0, 5)):
        with pytest.raises(fusion.FusionError):
            ctx.set_active_range(*bad)
    with pytest.raises(fusion.FusionError):
        ctx.export_masks(0, 1, buf.data_ptr(), 48 * 36)          # no run yet
    with pytest.raises(fusion.FusionError):
        ctx.import_masks(0, 2, buf.data_ptr(), 48 * 36)          # size does not match two keyframes
    with pytest.raises(fusion.FusionError):
        ctx.import_masks(3, 2, buf.data_ptr(), 2 * 48 * 36)      # names a fifth keyframe
    # a fully masked keyframe yields nothing; clearing the import brings its points back
    ctx.run()
    n_all = ctx.frame_counts()[0].copy()
    buf[: 48 * 36] = 255
    ctx.import_masks(2, 1, buf.data_ptr(), 48 * 36)
    ctx.run()
    assert ctx.frame_counts()[0][2] == 0 and n_all[2] > 0
    ctx.clear_imported_masks()
    ctx.run()
    assert np.array_equal(ctx.frame_counts()[0], n_all)
    # an empty block is a valid rank
    ctx.set_active_range(2, 2)
    ctx.run()
    assert ctx.num_points() == 0 and ctx.stats()["n_px"] == 0
    ctx.close()


def _all_masks(ctx, n, px):
    import torch
    buf = torch.empty(n * px, dtype=torch.uint8, device="cuda")
    ctx.export_masks(0, n, buf.data_ptr(), n * px)
    return buf.cpu().numpy()


def test_a_run_in_two_halves():
    """sfm_fusion_run_chain + sfm_fusion_finish: the masks are final (and may be exported) after the first half, the
    second half completes the run as sfm_fusion_run would have -- here including a retry (the first guess of the
    output capacity, one slot per 16 pixels, is too small for a cloud with nearly one voxel per accepted pixel)."""
    W, H = 96, 72
    seq = synth.make_sequence(12, W, H, K=(78.75, 78.75, 47.5, 35.5), window=3, yaw_step_deg=2.0)
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=0.5, T_dist=500.0)
    want = oracle.fuse(frames, op)
    assert want.key.shape[0] > max(len(frames) * W * H // 16, 4096)     # the first attempt cannot hold the cloud
    ctx = fusion.FusionContext(gp)
    with pytest.raises(fusion.FusionError):
        ctx.finish()                                                     # nothing in flight
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run_chain()
    early = _all_masks(ctx, len(frames), W * H)
    for call in (lambda: ctx.set_neighbors(0, []), lambda: ctx.set_active_range(0, 3), ctx.clear_imported_masks,
                 ctx.precompute, ctx.run_chain, ctx.num_points, ctx.read_points, lambda: ctx.partition_counts(2),
                 lambda: ctx.add_keyframe(frames[0]["depth"], frames[0]["bgr"], frames[0]["K"], frames[0]["T_wk"], frames[0]["T_kw"])):
        with pytest.raises(fusion.FusionError):
            call()                                                       # refused while the run is in flight
    ctx.finish()
    late = _all_masks(ctx, len(frames), W * H)
    assert np.array_equal(early, late) and early.any()
    xyz, bgr, key, tag = ctx.read_points()
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag) and np.array_equal(bgr, want.bgr)
    assert np.array_equal(xyz.view(np.uint64), want.xyz.view(np.uint64))
    fb, fa = ctx.frame_counts()
    assert np.array_equal(fb, want.frame_before) and np.array_equal(fa, want.frame_after)
    st = ctx.stats()
    for k in ("n_px", "n_gather", "n_acc", "n_hit", "n_vox"):
        assert st[k] == want.counters[k], k
    # the second run fits at once; sfm_fusion_run after run_chain is the same as finish
    ctx.run_chain()
    ctx.run()
    assert np.array_equal(ctx.read_points()[2], want.key)
    with pytest.raises(fusion.FusionError):
        ctx.finish()
    ctx.close()


def test_a_chain_that_cannot_hold_its_barrier_falls_back_inside_the_first_half(fusion_path, monkeypatch):
    """With a barrier time limit of one tick the persistent chain gives up at once: sfm_fusion_run_chain itself must
    rerun on the path without a device-wide barrier before it reports the masks final."""
    if fusion_path == "pair":
        pytest.skip("the kernel pair has no device-wide barrier")
    monkeypatch.setenv("SFM_CHAIN_TIMEOUT_TICKS", "1")
    W, H = 64, 48
    seq = synth.make_sequence(8, W, H, K=(131.25, 131.25, 31.5, 23.5), window=2)
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=30.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run_chain()
    early = _all_masks(ctx, len(frames), W * H)
    ctx.finish()
    assert np.array_equal(early, _all_masks(ctx, len(frames), W * H))
    xyz, bgr, key, tag = ctx.read_points()
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag)
    assert np.array_equal(xyz.view(np.uint64), want.xyz.view(np.uint64))
    assert ctx.stats()["chain_fallbacks"] == 1 and ctx.stats()["fused_chain"] == 0
    ctx.close()


def test_imports_made_after_a_clear_go_straight_into_the_arena_and_still_last():
    """sfm_fusion_clear_imported_masks zeroes the mask arena ahead of the run and the imports that follow go straight
    into it (nothing but the imports between a predecessor's masks and the chain); the imports must still hold for a
    SECOND run without another clear, for a keyframe imported twice (ORed), and be gone after the next clear."""
    import torch
    W, H = 48, 36
    seq = synth.make_sequence(6, W, H, K=(98.4, 98.4, 23.5, 17.5), window=2)
    frames = seq.frame_dicts()
    _, gp = both_params(voxel_size=30.0, T_dist=500.0)
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run()                                   # builds the layout; the clear below can zero ahead
    plain = ctx.frame_counts()[0].copy()
    assert plain[2] > 0 and plain[3] > 0
    left = torch.zeros(H, W, dtype=torch.uint8, device="cuda")
    left[:, : W // 2] = 255
    right = torch.zeros(H, W, dtype=torch.uint8, device="cuda")
    right[:, W // 2:] = 255
    ctx.clear_imported_masks()
    ctx.import_masks(2, 1, left.data_ptr(), W * H)      # keyframe 2: left half ...
    ctx.import_masks(3, 1, left.data_ptr(), W * H)
    ctx.import_masks(2, 1, right.data_ptr(), W * H)     # ... and right half: fully masked
    for _ in range(2):                                   # the second run starts from the same imports
        ctx.run()
        got = ctx.frame_counts()[0]
        assert got[2] == 0 and 0 < got[3] < plain[3]
        m = torch.empty(2 * W * H, dtype=torch.uint8, device="cuda")
        ctx.export_masks(2, 2, m.data_ptr(), 2 * W * H)
        m = m.cpu().numpy().reshape(2, H, W)
        assert m[0].all() and m[1][:, : W // 2].all()
    half = ctx.frame_counts()[0].copy()
    ctx.clear_imported_masks()
    ctx.run()
    assert np.array_equal(ctx.frame_counts()[0], plain)
    # the same imports on a fresh ctx that has never run (no arena to zero ahead at the clear): same counts
    ctx2 = fusion.FusionContext(gp)
    for f in frames:
        ctx2.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx2.set_neighbors(i, f["nbrs"])
    ctx2.clear_imported_masks()
    ctx2.import_masks(2, 1, left.data_ptr(), W * H)
    ctx2.import_masks(3, 1, left.data_ptr(), W * H)
    ctx2.import_masks(2, 1, right.data_ptr(), W * H)
    ctx2.run()
    assert np.array_equal(ctx2.frame_counts()[0], half)
    ctx.close()
    ctx2.close()
