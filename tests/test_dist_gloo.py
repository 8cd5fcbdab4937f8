"""world_size-2 run of the multi-GPU host path on CPU (gloo).  The collective wiring
(partition counts -> all-to-all of counts -> all-to-all of 48-byte records -> owner reduce) is
sfm_amd.dist_fusion.VoxelExchange, the same code bench.py drives over RCCL; the per-rank
fusion and the owner reduce are played by a numpy stand-in built on the CPU oracle (tests may
use the oracle as a checker/stand-in; the product's GPU kernels are covered by the -m gpu
tests, including tests/test_gpu_exchange.py)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REC = np.dtype([("key", "<u8"), ("tag", "<u8"), ("x", "<f8"), ("y", "<f8"), ("z", "<f8"),
                ("bgr", "<u4"), ("pad", "<u4")])


def mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def pack_key(key3):
    """numpy mirror of the kernel's 3x21-bit packing (indefinite key -> code 0x100000)."""
    code = np.where(key3 == np.uint64(0x8000000000000000), np.uint64(0x100000), key3 & np.uint64(0x1FFFFF))
    return (code[:, 0] << np.uint64(42)) | (code[:, 1] << np.uint64(21)) | code[:, 2]


def part_of(packed, n_parts):
    with np.errstate(over="ignore"):
        return ((mix64(packed ^ np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(17)) % np.uint64(n_parts)).astype(np.int64)


class NumpyCtx:
    """Stand-in with FusionContext's exchange interface."""

    def __init__(self, cloud, global_order):
        tag = (np.asarray(global_order, dtype=np.uint64)[(cloud.tag >> np.uint64(32)).astype(np.int64)]
               << np.uint64(32)) | (cloud.tag & np.uint64(0xFFFFFFFF))
        self.rec = np.zeros(cloud.key.shape[0], dtype=REC)
        self.rec["key"] = pack_key(cloud.key)
        self.rec["tag"] = tag
        self.rec["x"], self.rec["y"], self.rec["z"] = cloud.xyz[:, 0], cloud.xyz[:, 1], cloud.xyz[:, 2]
        self.rec["bgr"] = (cloud.bgr[:, 0] | (cloud.bgr[:, 1] << 8) | (cloud.bgr[:, 2] << 16)).astype(np.uint32)
        self.order = None
        self.result = None

    def partition_counts(self, n):
        p = part_of(self.rec["key"], n)
        self.order = np.argsort(p, kind="stable")
        return np.bincount(p, minlength=n).tolist()

    def export_records(self, ptr, nbytes):
        buf = self.rec[self.order].tobytes()
        assert nbytes == len(buf)
        ctypes.memmove(ptr, buf, nbytes)

    def merge(self, ptr, n):
        raw = ctypes.string_at(ptr, n * REC.itemsize) if n else b""
        rec = np.frombuffer(raw, dtype=REC)
        o = np.lexsort((rec["tag"], rec["key"]))
        rec = rec[o]
        first = np.ones(len(rec), dtype=bool)
        first[1:] = rec["key"][1:] != rec["key"][:-1]
        self.result = rec[first]

    def num_points(self):
        return len(self.result)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle
    from sfm_amd import dist_fusion, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seq = synth.make_sequence(8, 48, 36, K=(98.4, 98.4, 23.5, 17.5), window=2, components=2)
    weights = [48 * 36] * 8
    shards = dist_fusion.shard_components(seq.nbrs, weights, world)
    mine = shards[rank]
    pos = {g: i for i, g in enumerate(mine)}
    frames = seq.frame_dicts()
    local = [dict(frames[g], nbrs=[pos[j] for j in frames[g]["nbrs"]]) for g in mine]
    p = oracle.default_params(voxel_size=40.0, T_dist=500.0)
    ctx = NumpyCtx(oracle.fuse(local, p), mine)
    ex = dist_fusion.VoxelExchange(ctx, rank, world, torch.device("cpu"))
    n = ex.exchange_and_reduce()
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), ctx.result)
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


def test_two_rank_exchange_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from sfm_amd import synth
    seq = synth.make_sequence(8, 48, 36, K=(98.4, 98.4, 23.5, 17.5), window=2, components=2)
    want = oracle.fuse(seq.frame_dicts(), oracle.default_params(voxel_size=40.0, T_dist=500.0))
    got = np.concatenate([np.load(tmp_path / f"rank{r}.npy") for r in range(world)])
    assert int(np.load(tmp_path / "total.npy")[0]) == want.key.shape[0] == len(got)
    # owners hold disjoint key sets; their union is the single-process cloud
    for r in range(world):
        part = np.load(tmp_path / f"rank{r}.npy")
        assert (part_of(part["key"], world) == r).all()
    o = np.argsort(got["key"], kind="stable")
    got = got[o]
    wk = pack_key(want.key)
    wo = np.argsort(wk, kind="stable")
    assert np.array_equal(got["key"], wk[wo])
    assert np.array_equal(got["tag"], want.tag[wo])
    assert np.array_equal(got["x"], want.xyz[wo, 0]) and np.array_equal(got["z"], want.xyz[wo, 2])
    # the voxel filter really crossed the components (both trajectories see the same room)
    ranks_of_winner = (want.tag >> np.uint64(32)) >= np.uint64(4)
    assert ranks_of_winner.any() and (~ranks_of_winner).any()
