"""Multi-GPU dense fusion: one process per GPU, keyframes sharded by covisibility component,
per-GPU voxel tables merged by a voxel-key hash partition + all-to-all + per-owner reduce.

Why components: an accepted pixel masks the pixels it re-observed in its neighbour keyframes
(reference src/dense_reconstruction.cpp:152-154), so keyframe k depends on every earlier
keyframe it shares an edge with -- transitively, on its whole connected component of the
filtered-neighbour graph.  Cutting a component across GPUs would change the result set
(SURVEY.md D3).  Whole components are independent, so they are the sharding unit.  The only
cross-GPU step is the global first-wins voxel filter (:57), an order-free min-reduce.

The reference has no counterpart (single process, no collectives).  torch.distributed is used
for plumbing only: RCCL all_to_all_single over xGMI (backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.distributed as dist

RECORD_BYTES = 48


def components(nbrs: Sequence[Sequence[int]]) -> list:
    """Connected components of the (undirected) neighbour graph; each ascending, components
    ordered by their smallest keyframe."""
    n = len(nbrs)
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for i, nb in enumerate(nbrs):
        for j in nb:
            ra, rb = find(i), find(j)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    groups = {}
    for i in range(n):
        groups.setdefault(find(i), []).append(i)
    return [groups[k] for k in sorted(groups)]


def shard_components(nbrs: Sequence[Sequence[int]], weights: Sequence[int], world: int) -> list:
    """Assigns whole components to `world` ranks, heaviest first onto the lightest rank.
    Returns per rank the ascending list of global keyframe indices."""
    comps = components(nbrs)
    load = [0] * world
    out = [[] for _ in range(world)]
    for comp in sorted(comps, key=lambda c: (-sum(weights[i] for i in c), c[0])):
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].extend(comp)
        load[r] += sum(weights[i] for i in comp)
    return [sorted(x) for x in out]


class VoxelExchange:
    """Exchange + per-owner reduce of the per-rank winners.

    `ctx` provides partition_counts(n), export_records(ptr, nbytes), merge(ptr, n) and
    num_points() -- sfm_amd.fusion.FusionContext on a GPU, a numpy stand-in in the CPU tests.
    """

    def __init__(self, ctx, rank: int, world: int, device, group=None):
        self.ctx, self.rank, self.world, self.device, self.group = ctx, rank, world, device, group
        self.n_points = 0
        self.sent_records = 0
        self.recv_records = 0
        self._keep = None

    def exchange_and_reduce(self):
        import os
        import time
        timing = os.environ.get("SFM_EXCHANGE_TIMING") == "1"   # diagnostic: host-side time of every step
        marks = []

        def mark(name):
            if timing:
                if isinstance(self.device, torch.device) and self.device.type == "cuda":
                    torch.cuda.synchronize(self.device)
                marks.append((name, time.perf_counter()))

        mark("start")
        counts = self.ctx.partition_counts(self.world)
        mark("partition_counts")
        n_local = sum(counts)
        send = torch.empty(max(n_local, 1) * RECORD_BYTES, dtype=torch.uint8, device=self.device)
        self.ctx.export_records(send.data_ptr(), n_local * RECORD_BYTES)
        mark("export_records")
        send_counts = torch.tensor(counts, dtype=torch.int64, device=self.device)
        recv_counts = torch.empty_like(send_counts)
        # gloo has no all-to-all on device tensors: stage through the host there (CPU tests and
        # the single-GPU rehearsal of the N>1 path); RCCL ("nccl") moves device memory directly
        staged = send.is_cuda and dist.get_backend(self.group) == "gloo"
        if staged:
            send_counts = send_counts.cpu()
            recv_counts = recv_counts.cpu()
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        rc = [int(x) for x in recv_counts.tolist()]
        mark("all_to_all counts")
        n_recv = sum(rc)
        recv = torch.empty(max(n_recv, 1) * RECORD_BYTES, dtype=torch.uint8, device=self.device)
        in_splits = [c * RECORD_BYTES for c in counts]
        out_splits = [r * RECORD_BYTES for r in rc]
        if staged:
            h_recv = torch.empty(n_recv * RECORD_BYTES, dtype=torch.uint8)
            dist.all_to_all_single(h_recv, send[: n_local * RECORD_BYTES].cpu(), out_splits, in_splits,
                                   group=self.group)
            recv[: n_recv * RECORD_BYTES].copy_(h_recv)
        else:
            dist.all_to_all_single(recv[: n_recv * RECORD_BYTES], send[: n_local * RECORD_BYTES],
                                   out_splits, in_splits, group=self.group)
        if recv.is_cuda:
            torch.cuda.current_stream(recv.device).synchronize()
        mark("all_to_all records")
        self.ctx.merge(recv.data_ptr(), n_recv)
        mark("merge")
        if timing and self.rank == 0:
            print("exchange timing (ms): " + ", ".join(
                "%s %.3f" % (marks[i][0], (marks[i][1] - marks[i - 1][1]) * 1e3) for i in range(1, len(marks))), flush=True)
        self._keep = (send, recv)
        self.n_points = self.ctx.num_points()
        self.sent_records, self.recv_records = n_local, n_recv
        return self.n_points
