"""ONE keyframe sequence (one covisibility component) over several ranks, on CPU: the planning, the block-wise
oracle, and a world_size-2 gloo run of sfm_amd.dist_fusion.SequenceSplit + VoxelExchange -- the code bench.py drives
over RCCL -- whose union cloud must equal the single-process oracle cloud bit for bit.  The per-rank fusion is
played by the oracle's block-wise variant (orc_fuse_ex); the product's GPU kernels are covered by
tests/test_gpu_sequence_split.py through the same driver."""
import ctypes
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle
from sfm_amd import dist_fusion, synth
from helpers import assert_clouds_equal
from test_dist_gloo import NumpyCtx, pack_key, part_of

PARAMS = dict(voxel_size=40.0, T_dist=500.0)


def _sequence(n=10, window=2, closure=False):
    seq = synth.make_sequence(n, 48, 36, K=(98.4, 98.4, 23.5, 17.5), window=window, yaw_step_deg=2.5)
    frames = seq.frame_dicts()
    if closure:  # a long-range edge, one-directional on purpose: keyframe 1 names the last keyframe, not vice versa
        frames[1]["nbrs"] = sorted(frames[1]["nbrs"] + [n - 1])
    return frames


class OracleBlockCtx(NumpyCtx):
    """FusionContext's sequence-split and exchange interface on top of the block-wise oracle."""

    def __init__(self, frames_local, active_local, global_ids):
        self.frames, self.active, self.gids = frames_local, active_local, global_ids
        self.seed = {}
        self.masks = None
        self.in_flight = False
        self.halves = 0
        self.cloud = None

    def clear_imported_masks(self):
        self.seed = {}

    def import_masks(self, first, count, ptr, nbytes):
        off = 0
        for i in range(first, first + count):
            H, W = self.frames[i]["depth"].shape
            m = np.frombuffer(ctypes.string_at(ptr + off, H * W), dtype=np.uint8).reshape(H, W)
            self.seed[i] = np.maximum(self.seed.get(i, 0), m)
            off += H * W
        assert off == nbytes

    def run(self):
        p = oracle.default_params(**PARAMS)
        self.cloud, self.masks = oracle.fuse(self.frames, p, active=self.active, mask_seed=self.seed, want_masks=True)
        NumpyCtx.__init__(self, self.cloud, self.gids)

    # the run in two halves, as SequenceSplit drives a rank with successors: masks after the first, cloud after the second
    def run_chain(self):
        assert not self.in_flight
        p = oracle.default_params(**PARAMS)
        self.cloud, self.masks = oracle.fuse(self.frames, p, active=self.active, mask_seed=self.seed, want_masks=True)
        self.in_flight = True
        self.halves += 1

    def finish(self):
        assert self.in_flight
        self.in_flight = False
        NumpyCtx.__init__(self, self.cloud, self.gids)

    def export_masks(self, first, count, ptr, nbytes):
        buf = b"".join(self.masks[i].tobytes() for i in range(first, first + count))
        assert len(buf) == nbytes
        ctypes.memmove(ptr, buf, nbytes)


def _local_frames(frames, plan):
    return [dict(frames[g], nbrs=[plan.local[j] for j in frames[g]["nbrs"]] if plan.first <= g < plan.end else [])
            for g in plan.frames]


def test_plan_covers_every_keyframe_once_and_names_the_mask_traffic():
    frames = _sequence(11, window=3, closure=True)
    nbrs = [f["nbrs"] for f in frames]
    plans = dist_fusion.plan_sequence(nbrs, [48 * 36] * 11, 3, cost=None)
    assert [p.first for p in plans] == [0, 4, 7] and plans[-1].end == 11
    assert sum(p.end - p.first for p in plans) == 11
    for r, p in enumerate(plans):
        assert p.frames == sorted(p.frames) and set(range(p.first, p.end)) <= set(p.frames)
        a, b = p.active_local
        assert p.frames[a:b] == list(range(p.first, p.end))
        for t, fr in p.sends.items():
            assert t > r and plans[t].recvs[r] == fr and all(plans[t].first <= g < plans[t].end for g in fr)
    assert 10 in plans[0].sends[2]                     # the long edge: rank 0 marks rank 2's keyframe directly
    assert dist_fusion.plan_sequence(nbrs, [1] * 11, 1)[0].sends == {}


def test_pipelined_block_shares():
    """Default block sizes: rank r's verdicts (phase A) are ready just when rank r-1 stops chaining, so the blocks
    grow along the ranks; the modelled makespan beats equal blocks, and any cost gives a valid cover."""
    f, a, b, tail = dist_fusion.PIPELINE_COST

    def makespan(blocks):
        return dist_fusion.modelled_makespan([(e - s) / 2000.0 for s, e in blocks], dist_fusion.PIPELINE_COST, 2000)

    for world in (2, 3, 4, 8):
        shares = dist_fusion.block_shares(world)
        assert abs(sum(shares) - 1.0) < 1e-12 and all(s > 0 for s in shares)
        assert all(shares[r + 1] > shares[r] for r in range(1, world - 2))
        # the last rank's voxel filter is nobody's to hide: its block is at most the just-in-time size, shorter where
        # the model gains from it (with these costs: at eight ranks)
        jit = shares[-2] * (1.0 + b / a) if world > 2 else shares[0] * f / a
        assert shares[-1] <= jit * (1 + 1e-12) and (world < 8 or shares[-1] < 0.9 * jit)
        blocks = dist_fusion.split_blocks([921600] * 2000, world, dist_fusion.PIPELINE_COST)
        assert blocks[0][0] == 0 and blocks[-1][1] == 2000
        assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        equal = dist_fusion.split_blocks([921600] * 2000, world)
        assert max(e - s for s, e in equal) - min(e - s for s, e in equal) <= 1
        assert makespan(blocks) < makespan(equal)
        # no rank waits for its own phase A: it is done when the chain arrives
        t = f * (blocks[0][1] - blocks[0][0])
        for first, end in blocks[1:]:
            assert a * (end - first) <= t * 1.02
            t += b * (end - first)
    assert dist_fusion.block_shares(1) == [1.0]
    assert dist_fusion.block_shares(4, None) == [0.25] * 4
    # three costs (no filter term): the plain geometric shares
    sh = dist_fusion.block_shares(4, (10.0, 40.0, 30.0))
    assert abs(sh[1] / sh[0] - 0.25) < 1e-12 and abs(sh[3] / sh[2] - 1.75) < 1e-12
    # more ranks than keyframes: empty blocks are allowed, the cover stays exact
    blocks = dist_fusion.split_blocks([1] * 3, 8, dist_fusion.PIPELINE_COST)
    assert blocks[0][0] == 0 and blocks[-1][1] == 3 and sum(e - s for s, e in blocks) == 3


def test_blockwise_oracle_equals_the_whole_oracle():
    """orc_fuse_ex run block after block, each seeded with its predecessors' masks, is orc_fuse."""
    for closure in (False, True):
        frames = _sequence(10, closure=closure)
        p = oracle.default_params(**PARAMS)
        want = oracle.fuse(frames, p)
        for world in (2, 3):
            masks = {}
            parts = []
            for first, end in dist_fusion.split_blocks([1] * 10, world):
                c, m = oracle.fuse(frames, p, active=(first, end), mask_seed=masks, want_masks=True)
                masks = dict(enumerate(m))
                parts.append(c)
            assert sum(int(c.frame_before.sum()) for c in parts) == int(want.frame_before.sum())
            for k in ("n_px", "n_gather", "n_acc", "n_hit"):
                assert sum(c.counters[k] for c in parts) == want.counters[k], k
            fb = sum(c.frame_before for c in parts)
            assert np.array_equal(fb, want.frame_before)
            # the global voxel filter over the blocks' clouds = smallest tag per key
            key = np.concatenate([c.key for c in parts]); tag = np.concatenate([c.tag for c in parts])
            xyz = np.concatenate([c.xyz for c in parts])
            o = np.lexsort((tag, key[:, 2], key[:, 1], key[:, 0]))
            key, tag, xyz = key[o], tag[o], xyz[o]
            first_of_key = np.ones(len(key), dtype=bool)
            first_of_key[1:] = (key[1:] != key[:-1]).any(axis=1)
            assert np.array_equal(key[first_of_key], want.key) and np.array_equal(tag[first_of_key], want.tag)
            assert np.array_equal(xyz[first_of_key], want.xyz)


def _worker(rank, world, port, out_dir, closure, n_kf=10):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _sequence(n_kf, closure=closure)
    plans = dist_fusion.plan_sequence([f["nbrs"] for f in frames], [48 * 36] * n_kf, world)
    plan = plans[rank]
    ctx = OracleBlockCtx(_local_frames(frames, plan), plan.active_local, plan.frames)
    split = dist_fusion.SequenceSplit(ctx, plan, lambda g: 48 * 36, rank, world, torch.device("cpu"))
    ex = dist_fusion.VoxelExchange(ctx, rank, world, torch.device("cpu"))
    for _ in range(2):  # a second pass starts from clean imported masks again
        split.run()
        n = ex.exchange_and_reduce()
    # a rank with successors ran in two halves (masks out in between), the others in one piece
    assert ctx.halves == (2 if plan.sends else 0) and not ctx.in_flight
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), ctx.result)
    np.save(os.path.join(out_dir, f"traffic{rank}.npy"), np.array([split.bytes_sent, split.bytes_received]))
    t = torch.tensor([n], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "total.npy"), np.array([int(t.item())]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(tmp_path, world, closure):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), closure), nprocs=world, join=True)
    frames = _sequence(10, closure=closure)
    assert len(dist_fusion.components([f["nbrs"] for f in frames])) == 1   # ONE component: the mask chain crosses ranks
    want = oracle.fuse(frames, oracle.default_params(**PARAMS))
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    assert int(np.load(tmp_path / "total.npy")[0]) == want.key.shape[0] == len(got)
    for r in range(world):
        part = np.load(tmp_path / f"rank{r}.npy")
        assert (part_of(part["key"], world) == r).all()
    got = got[np.argsort(got["key"], kind="stable")]
    wk = pack_key(want.key)
    wo = np.argsort(wk, kind="stable")
    assert np.array_equal(got["key"], wk[wo]) and np.array_equal(got["tag"], want.tag[wo])
    for ax, name in enumerate("xyz"):
        assert np.array_equal(got[name], want.xyz[wo, ax])
    traffic = [np.load(tmp_path / f"traffic{r}.npy") for r in range(world)]
    assert traffic[0][0] > 0 and traffic[0][1] == 0 and traffic[-1][1] > 0 and traffic[-1][0] == 0
    # winners come from both blocks, and the masks really crossed: without the hand-over the later block
    # would accept pixels that the earlier one had masked
    assert ((want.tag >> np.uint64(32)) >= np.uint64(5)).any() and ((want.tag >> np.uint64(32)) < np.uint64(5)).any()
    blocks = dist_fusion.split_blocks([1] * 10, world)
    alone = oracle.fuse(frames, oracle.default_params(**PARAMS), active=blocks[-1])
    seeded_total = int(want.frame_before[blocks[-1][0]:].sum())
    assert int(alone.frame_before.sum()) > seeded_total


def test_two_ranks_one_sequence_equals_single_process(tmp_path):
    _check(tmp_path, 2, closure=False)


def test_three_ranks_with_a_long_range_edge(tmp_path):
    _check(tmp_path, 3, closure=True)


def test_more_ranks_than_the_split_has_blocks_for(tmp_path):
    """SequenceSplit followed by VoxelExchange end to end with a rank whose block is EMPTY (3 keyframes
    over 4 ranks) and an edge that jumps a block (keyframe 0 names keyframe 2: the rank between them is skipped by that
    mask transfer).  The empty rank still takes part in every collective and owns its share of the voxels."""
    world, n_kf = 4, 3
    frames = _sequence(n_kf, closure=True)                 # closure: keyframe 1 names the last keyframe as well
    nbrs = [f["nbrs"] for f in frames]
    plans = dist_fusion.plan_sequence(nbrs, [48 * 36] * n_kf, world)
    sizes = [p.end - p.first for p in plans]
    assert sum(sizes) == n_kf and 0 in sizes, sizes
    owner = {g: r for r, p in enumerate(plans) for g in range(p.first, p.end)}
    assert any(owner[j] - owner[i] > 1 or (owner[j] > owner[i] and 0 in sizes[owner[i] + 1:owner[j]])
               for i in range(n_kf) for j in nbrs[i] if j > i), "no edge jumps a rank"
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, n_kf), nprocs=world, join=True)
    want = oracle.fuse(frames, oracle.default_params(**PARAMS))
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    assert int(np.load(tmp_path / "total.npy")[0]) == want.key.shape[0] == len(got) > 0
    for r in range(world):
        part = np.load(tmp_path / f"rank{r}.npy")
        assert (part_of(part["key"], world) == r).all()
    got = got[np.argsort(got["key"], kind="stable")]
    wk = pack_key(want.key)
    wo = np.argsort(wk, kind="stable")
    assert np.array_equal(got["key"], wk[wo]) and np.array_equal(got["tag"], want.tag[wo])
    for ax, name in enumerate("xyz"):
        assert np.array_equal(got[name], want.xyz[wo, ax])


def test_oracle_helper_is_consistent():
    frames = _sequence(6)
    p = oracle.default_params(**PARAMS)
    a = oracle.fuse(frames, p)
    b, masks = oracle.fuse(frames, p, active=(0, 6), want_masks=True)
    assert_clouds_equal(a, b)
    # (the reference marks earlier neighbours too, :152-154; nobody reads those masks again)
    assert len(masks) == 6 and all(m.any() for m in masks)


def test_the_cpp_planner_and_the_python_planner_cut_alike():
    """sfm_amd/cpp/dist_fusion.hpp plan_sequence against dist_fusion.plan_sequence: same blocks, same mask traffic, for
    several rank counts, with and without an edge that jumps blocks (every rank of a job computes the plan itself; the two
    hosts must never disagree about it).  `dist_reconstruct --plan` touches neither the GPU nor RCCL."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "sfm_amd", "cpp", "dist_reconstruct")
    assert os.path.exists(exe), "run ./build.sh"
    for n, window, world, closure in [(41, 4, 4, 30), (2000, 4, 8, 0), (2000, 4, 16, 1500), (10, 2, 3, 0), (3, 1, 8, 0), (500, 4, 2, 0)]:
        r = subprocess.run([exe, "--plan", str(n), str(window), str(world), str(closure)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        nbrs = [[j for j in range(i - window, i + window + 1) if 0 <= j < n and j != i] for i in range(n)]
        if closure > 0 and n > 2 and closure < n:
            nbrs[2] = sorted(set(nbrs[2] + [closure]))
        want = []
        for rank, p in enumerate(dist_fusion.plan_sequence(nbrs, [921600] * n, world)):
            line = "%d block %d %d frames %d" % (rank, p.first, p.end, len(p.frames))
            line += "".join(" send %d:%s," % (t, ",".join(map(str, f))) for t, f in p.sends.items())
            line += "".join(" recv %d:%s," % (q, ",".join(map(str, f))) for q, f in p.recvs.items())
            want.append(line)
        assert r.stdout.splitlines() == want, (n, window, world, closure)
