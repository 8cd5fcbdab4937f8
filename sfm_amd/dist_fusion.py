"""Multi-GPU dense fusion: one process per GPU, per-GPU voxel tables merged by a voxel-key hash partition +
all-to-all + per-owner reduce.  Two ways to shard the keyframes, both exact:

* by covisibility component (shard_components): components never exchange anything but voxels;
* ONE component -- one sequence -- by contiguous keyframe blocks (plan_sequence, SequenceSplit): the visibility
  masks are the only state a keyframe hands to later ones, so a rank fuses its block starting from the masks its
  predecessors left and hands the masks of later keyframes on (point-to-point, rank to rank, in rank order).

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


# Cost of a keyframe in microseconds in the roles a block can play, measured on one MI355X on the C3 workload
# (DESIGN.md section 6): fused chain on rank 0, phase A (mask-independent verdicts) and phase B (accept chain) on the
# later ranks -- both chains as sfm_fusion_run_chain runs them, the voxel filter behind the chain and not under it --
# and that voxel filter (table + finalize kernels alone on the device).  Only the ratios matter, and only for speed:
# any split is exact.
PIPELINE_COST = (34.6, 44.5, 9.3, 9.3)


def modelled_makespan(shares, cost=PIPELINE_COST, units: float = 1.0) -> float:
    """The pipeline model behind block_shares: rank 0 chains its block from time 0 (f per unit), rank r > 0 computes
    its verdicts from time 0 (a per unit) and chains (b per unit) when they are done AND its predecessor has stopped
    chaining; a rank's voxel filter (t per unit, the optional fourth cost) runs behind its own chain.  Returns the time
    the last rank is done, for a sequence of `units` keyframes."""
    f, a, b = cost[:3]
    t = cost[3] if len(cost) > 3 else 0.0
    end = worst = 0.0
    for r, s in enumerate(shares):
        n = s * units
        start = end if r == 0 else max(end, a * n)
        end = start + (f if r == 0 else b) * n
        worst = max(worst, end + t * n)
    return worst


def block_shares(world: int, cost=PIPELINE_COST) -> list:
    """Share of the sequence each rank's block gets.  cost=None: equal shares.  Otherwise the shares that let every
    rank finish its phase A just when its predecessor finishes chaining: rank 0 chains its block with the fused
    kernel (f per unit) while rank 1 computes verdicts (a per unit) -- so block 1 is f/a of block 0 --, and each
    later rank has had (1 + b/a) times longer than the one before it (b per unit = the accept chain), so the blocks
    grow geometrically.  With equal blocks the later ranks idle after their phase A and rank 0's slow fused block
    is as long as everyone's; with these shares the chain ends at f*n0*(1 + b/a)^(N-1) instead of (f + (N-1)*b)*n/N.
    A fourth cost, the voxel filter behind a rank's chain, is hidden under the successors' chains on every rank but
    the last: the last block is shortened (the others keep their ratios) as far as modelled_makespan gains from it."""
    if cost is None or world == 1:
        return [1.0 / world] * world
    f, a, b = cost[:3]
    shares = [1.0] + [f / a * (1.0 + b / a) ** (r - 1) for r in range(1, world)]
    total = sum(shares)
    shares = [s / total for s in shares]
    if len(cost) > 3 and cost[3] > 0.0:
        best, best_t = shares, modelled_makespan(shares, cost)
        for k in range(1, 40):   # the last block at 97.5 % ... 2.5 % of its just-in-time size
            cand = shares[:-1] + [shares[-1] * (1.0 - k / 40.0)]
            tot = sum(cand)
            cand = [s / tot for s in cand]
            t = modelled_makespan(cand, cost)
            if t < best_t:
                best, best_t = cand, t
        shares = best
    return shares


def split_blocks(weights: Sequence[int], world: int, cost=None, shares=None) -> list:
    """Cuts keyframes 0..n-1 into `world` contiguous blocks whose weights (pixels) follow block_shares(world, cost)
    (cost=None: about equal weight; shares: given outright, one per block, summing to 1).  Returns [(first, end)] per
    rank; a rank may get an empty block when there are fewer keyframes than ranks."""
    n = len(weights)
    total = sum(weights)
    if shares is None:
        shares = block_shares(world, cost)
    cuts = [0]
    acc = 0
    i = 0
    upto = 0.0
    for r in range(1, world):
        upto += shares[r - 1]
        want = total * upto
        while i < n and acc + weights[i] / 2 <= want:
            acc += weights[i]
            i += 1
        cuts.append(i)
    cuts.append(n)
    for r in range(1, len(cuts)):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class RankPlan:
    """What one rank does in a sequence split: the block it fuses, the keyframes its ctx holds, and the mask
    traffic.  All keyframe numbers are global processing-order indices."""

    def __init__(self, first, end, frames, sends, recvs):
        self.first, self.end = first, end   # active block [first, end)
        self.frames = frames                # ascending: the block and every neighbour of a keyframe in it
        self.sends = sends                  # {later rank: [keyframes of ITS block that this block marks]}
        self.recvs = recvs                  # {earlier rank: [keyframes of this block that IT marks]}
        self.local = {g: i for i, g in enumerate(frames)}

    @property
    def active_local(self):
        """The block as a range of ctx order indices (frames is ascending, the block contiguous in it)."""
        if self.first == self.end:
            return (0, 0)
        return (self.local[self.first], self.local[self.end - 1] + 1)


def plan_sequence(nbrs: Sequence[Sequence[int]], weights: Sequence[int], world: int, cost=PIPELINE_COST, shares=None) -> list:
    """Exact split of one keyframe sequence over `world` ranks by contiguous blocks.  Keyframe i marks pixels in
    the masks of its LATER neighbours (earlier masks are never read again): rank q therefore sends rank r > q the
    masks of those keyframes of r's block that a keyframe of q's block names as a neighbour.  Every rank computes
    the same plan from the same graph.  Block sizes follow block_shares(world, cost): small first blocks, so that
    the chain reaches the later ranks when their verdicts are ready (cost=None: equal blocks)."""
    blocks = split_blocks(weights, world, cost, shares)
    owner = {}
    for r, (a, b) in enumerate(blocks):
        for g in range(a, b):
            owner[g] = r
    marks = [dict() for _ in range(world)]  # marks[q][r] = set of keyframes
    for q, (a, b) in enumerate(blocks):
        for i in range(a, b):
            for j in nbrs[i]:
                if j >= b:
                    marks[q].setdefault(owner[j], set()).add(j)
    plans = []
    for r, (a, b) in enumerate(blocks):
        frames = set(range(a, b))
        for i in range(a, b):
            frames.update(nbrs[i])
        sends = {t: sorted(v) for t, v in sorted(marks[r].items())}
        recvs = {q: sorted(marks[q][r]) for q in range(r) if r in marks[q]}
        plans.append(RankPlan(a, b, sorted(frames), sends, recvs))
    return plans


def _runs(local_ids):
    """[(first, count)] of consecutive runs in an ascending list of ctx order indices."""
    out = []
    for i in local_ids:
        if out and out[-1][0] + out[-1][1] == i:
            out[-1][1] += 1
        else:
            out.append([i, 1])
    return [(a, n) for a, n in out]


class SequenceSplit:
    """One pass of the keyframe chain of ONE sequence over all ranks: receive the masks the earlier blocks left,
    fuse this rank's block, send the masks of later keyframes on.  `ctx` provides clear_imported_masks(),
    import_masks(first, count, ptr, nbytes), run() [or run_chain() + finish()], export_masks(first, count, ptr, nbytes) -- sfm_amd.fusion.
    FusionContext on a GPU, an oracle-backed stand-in in the CPU tests.  `px` maps a global keyframe to W*H.
    The mask buffers (device, and host staging where the backend is gloo) are made once per plan, here.
    After run(): `timing` = host-side milliseconds of the pass (phase_a_issue, wait = blocked in recv for the
    predecessors' masks, import, run = until the masks are final, send, finish = the rest of the run)."""

    def __init__(self, ctx, plan: RankPlan, px, rank: int, world: int, device, group=None, precompute=None):
        self.ctx, self.plan, self.px = ctx, plan, px
        # two-phase chain on the ranks that wait: their mask-independent verdicts (sfm_fusion_precompute) are
        # computed while the ranks before them chain; the first rank has nobody to wait for and runs the fused chain
        self.precompute = (rank > 0 and hasattr(ctx, "precompute")) if precompute is None else precompute
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.bytes_sent = 0
        self.bytes_received = 0
        self.timing = {}
        self._recv_buf, self._send_buf, self._host_buf = {}, {}, {}
        for peer, frames, bufs in [(q, f, self._recv_buf) for q, f in plan.recvs.items()] + \
                                  [(r, f, self._send_buf) for r, f in plan.sends.items()]:
            n = sum(px(g) for g in frames)
            bufs[peer] = torch.empty(max(n, 1), dtype=torch.uint8, device=device)
            if self._staged(bufs[peer]):
                self._host_buf[(id(bufs), peer)] = torch.empty(max(n, 1), dtype=torch.uint8).pin_memory() \
                    if torch.cuda.is_available() else torch.empty(max(n, 1), dtype=torch.uint8)

    def _staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"  # gloo moves host memory only

    def run(self):
        import time
        ctx, plan = self.ctx, self.plan
        t = {"phase_a_issue": 0.0, "wait": 0.0, "import": 0.0, "run": 0.0, "send": 0.0, "finish": 0.0}
        t0 = time.perf_counter()
        ctx.clear_imported_masks()
        if self.precompute:
            ctx.precompute()  # asynchronous: overlaps the wait for the predecessors' masks
        t["phase_a_issue"] = time.perf_counter() - t0
        for q, frames in plan.recvs.items():  # ascending ranks: the order the senders finish in
            n = sum(self.px(g) for g in frames)
            buf = self._recv_buf[q]
            t0 = time.perf_counter()
            if self._staged(buf):
                h = self._host_buf[(id(self._recv_buf), q)]
                dist.recv(h, src=q, group=self.group)
                buf.copy_(h)
            else:
                dist.recv(buf, src=q, group=self.group)
            if buf.is_cuda:
                torch.cuda.current_stream(buf.device).synchronize()
            t1 = time.perf_counter()
            t["wait"] += t1 - t0
            off = 0
            for first, count in _runs([plan.local[g] for g in frames]):
                nb = sum(self.px(plan.frames[i]) for i in range(first, first + count))
                ctx.import_masks(first, count, buf.data_ptr() + off, nb)
                off += nb
            t["import"] += time.perf_counter() - t1
            self.bytes_received += n
        t0 = time.perf_counter()
        # a rank with successors hands its masks on as soon as the frame loop has finished; the voxel filter of its last
        # keyframe groups (stream B) runs on under the sends and is collected by finish()
        split = bool(plan.sends) and hasattr(ctx, "run_chain")
        if split:
            ctx.run_chain()
        else:
            ctx.run()
        t1 = time.perf_counter()
        t["run"] = t1 - t0
        sent = []
        for r, frames in plan.sends.items():
            n = sum(self.px(g) for g in frames)
            buf = self._send_buf[r]
            off = 0
            for first, count in _runs([plan.local[g] for g in frames]):
                nb = sum(self.px(plan.frames[i]) for i in range(first, first + count))
                ctx.export_masks(first, count, buf.data_ptr() + off, nb)
                off += nb
            # non-blocking: a successor that is not in its recv yet (it takes its predecessors in rank order) does not
            # hold up the sends to the others, nor this rank's finish()
            if self._staged(buf):
                h = self._host_buf[(id(self._send_buf), r)]
                h.copy_(buf)
                sent.append(dist.isend(h, dst=r, group=self.group))
            else:
                sent.append(dist.isend(buf, dst=r, group=self.group))
            self.bytes_sent += n
        t2 = time.perf_counter()
        t["send"] = t2 - t1
        if split:
            ctx.finish()
        t["finish"] = time.perf_counter() - t2
        for w in sent:   # the send buffers are free again when run() returns
            w.wait()
        self.timing = {k: v * 1e3 for k, v in t.items()}


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
        # Every peer learns this rank's record count for it AND this rank's key packing (index range per axis) in one
        # all-to-all: ranks that saw voxel indices of different ranges pack differently, and records are only
        # comparable in one common packing -- the hull of all ranges.  The owner of a voxel does not depend on it.
        lo, hi = self.ctx.key_range() if hasattr(self.ctx, "key_range") else ((0, 0, 0), (0, 0, 0))
        send_counts = torch.tensor([[n, *lo, *hi] for n in counts], dtype=torch.int64, device=self.device)
        recv_counts = torch.empty_like(send_counts)
        # gloo has no all-to-all on device tensors: stage through the host there (CPU tests and
        # the single-GPU rehearsal of the N>1 path); RCCL ("nccl") moves device memory directly
        staged = send_counts.is_cuda and dist.get_backend(self.group) == "gloo"
        if staged:
            send_counts = send_counts.cpu()
            recv_counts = recv_counts.cpu()
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        got = recv_counts.tolist()
        rc = [int(row[0]) for row in got]
        packings = {(tuple(lo), tuple(hi))} | {(tuple(int(x) for x in row[1:4]), tuple(int(x) for x in row[4:7])) for row in got}
        if len(packings) > 1:
            # The ranks pack differently (every rank sees the same rows, so every rank takes this branch).  The common
            # packing is the hull of the index ranges the CLOUDS really have, not of the packings: a rank still on the
            # default packing holds +-(2^20 - 1) on every axis whatever its cloud is, and the hull of that with a
            # scanned range far from the origin can need more than 63 bits where the clouds' own hull does not.
            if hasattr(self.ctx, "cloud_index_range"):
                r_lo, r_hi = self.ctx.cloud_index_range()
            else:
                r_lo, r_hi = lo, hi
            mine = torch.tensor([*r_lo, *r_hi], dtype=torch.int64, device=self.device)
            if staged:
                mine = mine.cpu()
            everyone = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(everyone, mine, group=self.group)
            rows = [[int(x) for x in e.tolist()] for e in everyone]
            c_lo, c_hi = [], []
            for a in range(3):
                have = [(row[a], row[3 + a]) for row in rows if row[a] <= row[3 + a]]   # lo > hi: no index on this axis
                c_lo.append(min(x for x, _ in have) if have else 0)
                c_hi.append(max(y for _, y in have) if have else 0)
            self.ctx.set_key_range(tuple(c_lo), tuple(c_hi))       # re-encodes this rank's cloud
            again = self.ctx.partition_counts(self.world)          # (arms export_records for the re-encoded cloud)
            if again != counts:
                raise RuntimeError("voxel owners changed with the key packing: %r != %r" % (again, counts))
        mark("all_to_all counts")
        send = torch.empty(max(n_local, 1) * RECORD_BYTES, dtype=torch.uint8, device=self.device)
        self.ctx.export_records(send.data_ptr(), n_local * RECORD_BYTES)
        mark("export_records")
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
