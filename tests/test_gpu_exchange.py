"""Multi-GPU data path on one device: two contexts play two ranks (one covisibility component
each); the all-to-all is done by hand with torch tensors.  Checks the partition, export and
merge kernels through the C ABI against the single-process oracle."""
import numpy as np
import pytest

from oracle import oracle
from sfm_amd import dist_fusion, fusion, synth
from helpers import both_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_component_sharded_fusion_equals_single_process(world):
    import torch
    seq = synth.make_sequence(4 * world, 64, 48, K=(131.25, 131.25, 31.5, 23.5), window=2, components=world)
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=30.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    shards = dist_fusion.shard_components(seq.nbrs, [64 * 48] * len(frames), world)
    ctxs, counts = [], []
    for r in range(world):
        mine = shards[r]
        pos = {g: i for i, g in enumerate(mine)}
        ctx = fusion.FusionContext(gp)
        for g in mine:
            f = frames[g]
            ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
        for g in mine:
            ctx.set_neighbors(pos[g], [pos[j] for j in frames[g]["nbrs"]])
        ctx.set_global_order(mine)
        ctx.run()
        ctxs.append(ctx)
        counts.append(ctx.partition_counts(world))
    sends = []
    for r, ctx in enumerate(ctxs):
        n = sum(counts[r])
        assert n == ctx.num_points()
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
        recv = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.merge(recv.data_ptr(), recv.numel() // fusion.SFM_RECORD_BYTES)
        got.append(ctx.read_points())
    xyz = np.concatenate([g[0] for g in got]); bgr = np.concatenate([g[1] for g in got])
    key = np.concatenate([g[2] for g in got]); tag = np.concatenate([g[3] for g in got])
    xyz, bgr, key, tag = oracle.sort_by_key(xyz, bgr, key, tag)
    assert np.array_equal(key, want.key), "union of the owners' key sets differs from the single-process cloud"
    assert np.array_equal(tag, want.tag) and np.array_equal(bgr, want.bgr) and np.array_equal(xyz, want.xyz)
    # owners are disjoint
    assert sum(g[2].shape[0] for g in got) == want.key.shape[0]
    for ctx in ctxs:
        ctx.close()


def test_read_points_reports_global_orders_without_merge():
    seq = synth.make_sequence(4, 48, 36, K=(98.4, 98.4, 23.5, 17.5), window=2)
    frames = seq.frame_dicts()
    _, gp = both_params(voxel_size=30.0, T_dist=500.0)
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run()
    t_local = ctx.read_points()[3]
    ctx.set_global_order([10, 11, 15, 40])
    t_glob = ctx.read_points()[3]
    lut = np.array([10, 11, 15, 40], dtype=np.uint64)
    assert np.array_equal(t_glob, (lut[(t_local >> np.uint64(32)).astype(int)] << np.uint64(32)) | (t_local & np.uint64(0xFFFFFFFF)))
    with pytest.raises(fusion.FusionError):
        ctx.set_global_order([3, 2, 1, 0])
    ctx.close()


def test_ranks_with_different_key_ranges_agree_on_their_hull():
    """Two components far apart: each rank's run ends with its own key packing; the exchange (VoxelExchange over a
    one-rank gloo group per ctx is not possible, so its steps by hand) sets the hull on both before the records
    travel, and the owners' union is the oracle's cloud."""
    import torch
    seq = synth.make_sequence(8, 64, 48, K=(131.25, 131.25, 31.5, 23.5), window=2, components=2)
    frames = seq.frame_dicts()
    for i, off in ((0, np.array([4.0e8, 0.0, 0.0])), (1, np.array([-2.0e8, 6.0e4, -3.0e4]))):
        for f in frames[4 * i:4 * i + 4]:
            f["T_wk"] = np.array(f["T_wk"]); f["T_wk"][:3, 3] += off
            f["T_kw"] = np.array(f["T_kw"]); f["T_kw"][:3, 3] = -f["T_kw"][:3, :3] @ f["T_wk"][:3, 3]
    op, gp = both_params(voxel_size=30.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    ctxs = []
    for r in range(2):
        ctx = fusion.FusionContext(gp)
        for f in frames[4 * r:4 * r + 4]:
            ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
        for i, f in enumerate(frames[4 * r:4 * r + 4]):
            ctx.set_neighbors(i, [j - 4 * r for j in f["nbrs"]])
        ctx.set_global_order(list(range(4 * r, 4 * r + 4)))
        ctx.run()
        ctxs.append(ctx)
    ranges = [c.key_range() for c in ctxs]
    assert ranges[0] != ranges[1]
    counts_before = [c.partition_counts(2) for c in ctxs]         # owners do not depend on the packing
    lo = tuple(min(r[0][a] for r in ranges) for a in range(3))
    hi = tuple(max(r[1][a] for r in ranges) for a in range(3))
    for c in ctxs:
        c.set_key_range(lo, hi)
    counts = [c.partition_counts(2) for c in ctxs]
    assert counts == counts_before
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
        for r in range(2):
            start = sum(counts[r][:owner]) * fusion.SFM_RECORD_BYTES
            pieces.append(sends[r][start:start + counts[r][owner] * fusion.SFM_RECORD_BYTES])
        recv = torch.cat(pieces)
        torch.cuda.synchronize()
        ctx.merge(recv.data_ptr(), recv.numel() // fusion.SFM_RECORD_BYTES)
        got.append(ctx.read_points())
    xyz = np.concatenate([g[0] for g in got]); bgr = np.concatenate([g[1] for g in got])
    key = np.concatenate([g[2] for g in got]); tag = np.concatenate([g[3] for g in got])
    xyz, bgr, key, tag = oracle.sort_by_key(xyz, bgr, key, tag)
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag) and np.array_equal(xyz, want.xyz)
    for c in ctxs:
        c.close()


def test_a_default_packing_and_a_scanned_one_meet_in_the_hull_of_the_clouds():
    """one rank's cloud is far from the origin (its run scanned and packs tightly), the other's is near it
    and still on the default packing, whose range is +-(2^20 - 1) on every axis whatever the cloud is.  The hull of the
    PACKINGS needs 25 + 21 + 21 bits and is refused; the hull of the index ranges the clouds really have fits.  The
    exchange (dist_fusion.VoxelExchange) takes the second; here its steps by hand on one GPU."""
    import torch
    seq = synth.make_sequence(8, 64, 48, K=(131.25, 131.25, 31.5, 23.5), window=2, components=2)
    frames = seq.frame_dicts()
    off = np.array([4.0e8, 0.0, 0.0])                      # component 0: x indices around 1.3e7 (24 bits)
    for f in frames[:4]:
        f["T_wk"] = np.array(f["T_wk"]); f["T_wk"][:3, 3] += off
        f["T_kw"] = np.array(f["T_kw"]); f["T_kw"][:3, 3] = -f["T_kw"][:3, :3] @ f["T_wk"][:3, 3]
    op, gp = both_params(voxel_size=30.0, T_dist=500.0)
    want = oracle.fuse(frames, op)
    ctxs = []
    for r in range(2):
        ctx = fusion.FusionContext(gp)
        for f in frames[4 * r:4 * r + 4]:
            ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
        for i, f in enumerate(frames[4 * r:4 * r + 4]):
            ctx.set_neighbors(i, [j - 4 * r for j in f["nbrs"]])
        ctx.set_global_order(list(range(4 * r, 4 * r + 4)))
        ctx.run()
        ctxs.append(ctx)
    packs = [c.key_range() for c in ctxs]
    default = ((-1048575,) * 3, (1048575,) * 3)
    assert packs[0] != default and packs[1] == default
    p_lo = tuple(min(p[0][a] for p in packs) for a in range(3))
    p_hi = tuple(max(p[1][a] for p in packs) for a in range(3))
    with pytest.raises(fusion.FusionError):                  # the packings' hull: more than 63 bits
        ctxs[1].set_key_range(p_lo, p_hi)
    real = [c.cloud_index_range() for c in ctxs]
    for (lo, hi), got in zip(real, (c.read_points()[2] for c in ctxs)):
        idx = got.astype(np.int64)                           # (no indefinite index in these clouds)
        assert lo == tuple(int(x) for x in idx.min(axis=0)) and hi == tuple(int(x) for x in idx.max(axis=0))
    lo = tuple(min(r[0][a] for r in real) for a in range(3))
    hi = tuple(max(r[1][a] for r in real) for a in range(3))
    counts_before = [c.partition_counts(2) for c in ctxs]
    for c in ctxs:
        c.set_key_range(lo, hi)
    counts = [c.partition_counts(2) for c in ctxs]
    assert counts == counts_before
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
        for r in range(2):
            start = sum(counts[r][:owner]) * fusion.SFM_RECORD_BYTES
            pieces.append(sends[r][start:start + counts[r][owner] * fusion.SFM_RECORD_BYTES])
        recv = torch.cat(pieces)
        torch.cuda.synchronize()
        ctx.merge(recv.data_ptr(), recv.numel() // fusion.SFM_RECORD_BYTES)
        got.append(ctx.read_points())
    xyz = np.concatenate([g[0] for g in got]); bgr = np.concatenate([g[1] for g in got])
    key = np.concatenate([g[2] for g in got]); tag = np.concatenate([g[3] for g in got])
    xyz, bgr, key, tag = oracle.sort_by_key(xyz, bgr, key, tag)
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag) and np.array_equal(xyz, want.xyz)
    for c in ctxs:
        c.close()
