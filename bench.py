#!/usr/bin/env python3
"""bench.py -- fused depth px/s of the MI355X dense-fusion path (BASELINE.json metric).

One "step" = one full sfm_fusion_run over the workload: masks cleared, every keyframe fused in
order, voxel first-wins filter, compacted cloud left in HBM.  Inputs are synthetic
(sfm_amd/synth.py) and already resident in HBM when the timed region starts.

N=1 workload = BASELINE.json configs[1]: 500 keyframes 640x480, TUM-like intrinsics, depth in
millimetres, 1 cm voxels (voxel_size=10 raw units), T_dist=500 raw units, temporal
neighbours +-4.  N>1 = configs[2], STRONG scaling: ONE sequence of 2000 keyframes 1280x720 (one covisibility
component) cut into N contiguous keyframe blocks, one per rank.  The visibility masks are the only state a
keyframe hands to later ones: a rank fuses its block starting from the masks its predecessors left and sends the
masks of the next rank's first keyframes on (point-to-point over RCCL, sfm_amd.dist_fusion.SequenceSplit); then
the per-rank voxel tables are exchanged by voxel-key hash partition (RCCL all-to-all) and min-reduced by their
owners.  The union of the owners' clouds is bit-identical to the single-GPU cloud.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

CORRIDOR = {"c5_10000kf_640x480_5mm": 0.05}  # metres per keyframe of a straight walk (voxel count grows linearly)
WORKLOADS = {
    # name: (keyframes, W, H, intrinsics, voxel_size, T_dist)
    "c2_500kf_640x480_1cm": (500, 640, 480, (525.0, 525.0, 319.5, 239.5), 10.0, 500.0),
    "c3_2000kf_1280x720_1cm": (2000, 1280, 720, (913.848, 913.602, 642.941, 371.196), 10.0, 500.0),
    "c5_10000kf_640x480_5mm": (10000, 640, 480, (525.0, 525.0, 319.5, 239.5), 5.0, 500.0),
    "tiny_24kf_160x120": (24, 160, 120, (131.25, 131.25, 79.75, 59.75), 10.0, 500.0),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def observe_bytes(st):
    """Algorithmic bytes of observe_frame_kernel (SURVEY.md 8d terms that it moves; with
    scan_bytes() the terms of fuse_chain_kernel, which does both jobs): 2 B per neighbour depth
    gather, one 16-B slot read+write (its record hand-off to the voxel table) per accepted pixel,
    1 B per mask store.  SURVEY's 3 B of colour per accepted pixel are not in it: the colour is read
    by finalize_kernel, for the voxels' winners only (total_bytes counts 3 B per output point)."""
    return 2 * st["n_gather"] + st["n_hit"] + 32 * st["n_acc"]


def scan_bytes(st):
    """scan_frame_kernel: depth u16 + mask u8 per pixel."""
    return 3 * st["n_px"]


def total_bytes(st):
    return scan_bytes(st) + observe_bytes(st) + (27 + 3) * st["n_vox"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2_500kf_640x480_1cm on one GPU, c3_2000kf_1280x720_1cm on several")
    ap.add_argument("--frames", type=int, default=0, help="override the sequence's keyframe count")
    ap.add_argument("--two-phase", action="store_true",
                    help="one GPU: run every step as verdict_kernel (phase A) + accept_chain_kernel (phase B), the path "
                         "the waiting ranks of a sequence split take; measures the two terms of the scaling model")
    ap.add_argument("--split-run", action="store_true",
                    help="one GPU: run every step as sfm_fusion_run_chain + sfm_fusion_finish (what a rank with successors "
                         "does: the voxel filter runs behind the chain, not under it) and report when the masks were final")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=500)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # SFM_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo -- a functional rehearsal of the N>1
    # path on a one-GPU box (RCCL refuses two ranks on one device); never a measurement
    rehearsal = os.environ.get("SFM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SFM_BENCH_FORCE_EXCHANGE=1: run the N>1 exchange (RCCL all-to-all + owner merge) even with one
    # rank -- a functional check of that code path on a one-GPU box, never a measurement
    force_exchange = os.environ.get("SFM_BENCH_FORCE_EXCHANGE") == "1"
    if world > 1 or force_exchange:
        import torch.distributed as dist
        if force_exchange and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from sfm_amd import dist_fusion, fusion, synth

    if args.workload is None:
        args.workload = "c2_500kf_640x480_1cm" if world == 1 else "c3_2000kf_1280x720_1cm"
    n_kf, W, H, K, voxel, t_dist = WORKLOADS[args.workload]
    if args.frames:
        n_kf = args.frames
    # the sequence's poses and neighbour lists are the same on every rank; a rank renders (ray-casts, on its own
    # GPU) only the keyframes its ctx holds: its block and the neighbours of the block's keyframes
    walk = CORRIDOR.get(args.workload, 0.0)
    probe = synth.make_sequence(n_kf, W, H, K=K, device="cpu", seed=20240315, components=1, radius_m=0.8, only=(0, 0),
                                corridor_step_m=walk)
    plan = dist_fusion.plan_sequence(probe.nbrs, [W * H] * n_kf, world)[rank]
    lo, hi = (plan.frames[0], plan.frames[-1] + 1) if plan.frames else (0, 0)
    seq = synth.make_sequence(n_kf, W, H, K=K, device=dev, seed=20240315, components=1, radius_m=0.8, only=(lo, hi),
                              corridor_step_m=walk)
    torch.cuda.synchronize()
    over = {}
    if os.environ.get("SFM_BENCH_TABLE_CAP"):   # tuning experiments only; the default run sizes automatically
        over["table_capacity"] = int(os.environ["SFM_BENCH_TABLE_CAP"])
    params = fusion.default_params(voxel_size=voxel, T_dist=t_dist, **over)
    stream = torch.cuda.current_stream().cuda_stream
    ctx = fusion.FusionContext(params, local_rank, stream)
    for g in plan.frames:
        ctx.add_keyframe_device(seq.depth[g - seq.first].data_ptr(), seq.bgr[g - seq.first].data_ptr(), W, H,
                                seq.K[g], seq.T_wk[g], seq.T_kw[g])
    for g in range(plan.first, plan.end):
        ctx.set_neighbors(plan.local[g], [plan.local[j] for j in seq.nbrs[g]])
    if world > 1:
        ctx.set_active_range(*plan.active_local)

    try:
        return _bench(args, ctx, seq, plan, dev, rank, world, n_kf, W, H, voxel, t_dist, force_exchange, rehearsal)
    finally:
        # on every path, before the interpreter's exit handlers run: the ctx's streams and events are
        # destroyed while the HIP runtime is still alive
        ctx.close()
        if world > 1 or force_exchange:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()
                dist.destroy_process_group()


def _bench(args, ctx, seq, plan, dev, rank, world, n_kf, W, H, voxel, t_dist, force_exchange, rehearsal):
    if world > 1 or force_exchange:
        import torch.distributed as dist
    from sfm_amd import dist_fusion
    walk = CORRIDOR.get(args.workload, 0.0)
    merger = split = None
    if world > 1 or force_exchange:
        ctx.set_global_order(np.asarray(plan.frames, dtype=np.uint32))  # tags carry the sequence's keyframe numbers
        merger = dist_fusion.VoxelExchange(ctx, rank, world, dev)
    if world > 1:
        split = dist_fusion.SequenceSplit(ctx, plan, lambda g: W * H, rank, world, dev)

    half = [0.0, 0.0]   # --split-run: host time until the masks were final, and until the run was complete

    def step():
        if split is not None:
            split.run()       # (phase A while waiting,) masks in from the earlier blocks, this block's chain, masks out
        else:
            if args.two_phase:
                ctx.precompute()
            if args.split_run:
                ts = time.perf_counter()
                ctx.run_chain()
                half[0] += time.perf_counter() - ts
                ctx.finish()
                half[1] += time.perf_counter() - ts
            else:
                ctx.run()
        if merger is not None:
            merger.exchange_and_reduce()

    def barrier():
        if world > 1 or force_exchange:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    half[0] = half[1] = 0.0
    # per-rank breakdown of the timed steps (host clocks; N > 1): what the scaling model of DESIGN.md section 6 is made of
    acc = {"phase_a_ms": 0.0, "wait_ms": 0.0, "phase_b_ms": 0.0, "run_ms": 0.0, "chain_host_ms": 0.0, "mask_send_ms": 0.0,
           "finish_ms": 0.0, "exchange_ms": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if split is None:
            step()
            continue
        split.run()
        tx = time.perf_counter()
        merger.exchange_and_reduce()
        acc["exchange_ms"] += (time.perf_counter() - tx) * 1e3
        sst = ctx.stats()
        acc["phase_a_ms"] += sst["precompute_ms"]          # device time of this rank's verdicts (0 on rank 0: fused chain)
        acc["phase_b_ms"] += sst["frame_kernels_ms"]       # device time of this rank's chain (fused on rank 0, accept on the others)
        acc["run_ms"] += sst["last_run_ms"]                # ... and of the whole run: chain, then the voxel filter behind it
        acc["finish_ms"] += split.timing["finish"]         # host: from the last mask sent to the run complete
        acc["wait_ms"] += split.timing["wait"]             # blocked in recv for the predecessors' masks (phase A runs under it)
        acc["chain_host_ms"] += split.timing["run"]
        acc["mask_send_ms"] += split.timing["send"]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1 or force_exchange:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st = ctx.stats()
    n_points = merger.n_points if merger is not None else st["n_vox"]
    if world > 1 or force_exchange:
        t = torch.tensor([n_points], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        n_points = int(t.item())

    # roofline of the dominant kernel: one extra, untimed, run with start/stop events around
    # every dispatch of it on the library's stream
    # (rank 0 alone: the other ranks wait, so that nothing else touches a GPU that a rehearsal shares)
    pst = st
    if rank == 0 and not os.environ.get("SFM_DIAG_ONE_LAUNCH"):   # (that diagnostic has no per-group launches to time)
        ctx.set_profiling(True)
        if args.split_run and split is None:
            ctx.run_chain()
            ctx.finish()
        else:
            ctx.run()
        pst = ctx.stats()
        ctx.set_profiling(False)
    if world > 1:
        dist.barrier()
    launches = max(1, pst["fuse_kernel_launches"])
    avg_launch_s = pst["fuse_kernel_sum_ms"] * 1e-3 / launches
    # the persistent fuse_chain_kernel does the work of both per-keyframe kernels for a group of
    # 32 keyframes per launch; on the fallback path the dominant kernel is observe_frame_kernel
    chain = bool(pst["fused_chain"])
    kernel_name = "fuse_chain_kernel" if chain else "observe_frame_kernel"
    kernel_bytes = observe_bytes(pst) + (scan_bytes(pst) if chain else 0)
    if pst["precomputed"]:
        # --two-phase (and ranks > 0 of a split sequence): the timed chain is accept_chain_kernel, which reads every
        # pixel's verdict word and its first four stored hit pixels (4 + 16 B, dense), the mask byte of every pixel
        # (1 B), the 8-byte voxel key of an accepted pixel, and writes its 16-byte list entry (tag + key: the fused
        # point stays in the verdict record, finalize_kernel reads the winners' there) and one mask byte per kept
        # observation in a later keyframe (<= N_hit)
        kernel_name = "accept_chain_kernel"
        kernel_bytes = 21 * pst["n_px"] + 24 * pst["n_acc"] + pst["n_hit"]
    bytes_per_launch = kernel_bytes / launches
    scan_avg_s = pst["scan_kernel_sum_ms"] * 1e-3 / launches
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    traffic = None
    # (profiles/traffic_latest.json: the default run's dominant kernel; traffic_two_phase.json: accept_chain_kernel of
    # `--two-phase`; both from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/profile_summary.py)
    for tname in ("traffic_latest.json", "traffic_two_phase.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if traffic is None and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and not args.frames and tj.get("kernel") == kernel_name and world == 1:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None

    # The roofs that bind this kernel, beside the contracted HBM figure (profiles/bounds_latest.json, made by
    # tools/profile_pass.sh + tools/profile_summary.py from rocprofv3 SQ counters and the micro-benchmarks):
    #   fp64   VALU-issue floor: wave instructions counted by SQ_INSTS_VALU x 4 cycles on 1024 SIMDs at 2.4 GHz
    #   chain  keyframes x the smallest link: the device-wide barrier alone, and the chain with all arithmetic
    #          removed (accept_chain_kernel)
    fp64_roof = chain_roof = None
    bpath = os.path.join(ROOT, "profiles", "bounds_latest.json")
    if os.path.exists(bpath) and avg_launch_s > 0:
        try:
            bj = json.load(open(bpath))
            if bj.get("workload") == args.workload and not args.frames and bj.get("kernel") == kernel_name and world == 1:
                floor = bj["valu_issue_floor_us"]
                fp64_roof = {"valu_wave_insts_per_launch": bj["valu_insts_per_launch"], "issue_floor_us": floor,
                             "frac": floor * 1e-6 / avg_launch_s,
                             "issue_floor_us_at_measured_rate": bj.get("valu_issue_floor_us_measured_rate"),
                             "frac_at_measured_rate": (bj["valu_issue_floor_us_measured_rate"] * 1e-6 / avg_launch_s
                                                       if bj.get("valu_issue_floor_us_measured_rate") else None),
                             "source": "profiles/%s_summary.json (rocprofv3 --pmc SQ_INSTS_VALU), tools/valu_bench.hip" % bj["tag"]}
                kpl = (plan.end - plan.first) / launches
                chain_roof = {"keyframes_per_launch": kpl, "barrier_us": bj.get("barrier_us"),
                              "barrier_floor_frac": (kpl * bj["barrier_us"] * 1e-6 / avg_launch_s) if bj.get("barrier_us") else None,
                              "accept_link_us": bj.get("accept_link_us"),
                              "accept_link_floor_frac": (kpl * bj["accept_link_us"] * 1e-6 / avg_launch_s)
                              if bj.get("accept_link_us") else None,
                              "us_per_keyframe": avg_launch_s * 1e6 / kpl,
                              "source": "tools/barrier_bench.hip; bench.py --two-phase (accept_chain_kernel)"}
        except Exception:
            fp64_roof = chain_roof = None

    px_total = n_kf * W * H   # the whole sequence, whatever the number of ranks: strong scaling
    value = px_total * args.steps / dt
    out = {
        "metric": "fused depth px/sec (whole node)",
        "value": value,
        "unit": "px/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "single" if world == 1 else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (exchange forced at one rank: functional check, not a measurement)" if force_exchange else "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a measurement)",
        "config": {"workload": args.workload, "keyframes": n_kf, "keyframes_per_gpu": n_kf / world, "width": W, "height": H,
                   "voxel_size_raw_units": voxel, "T_dist_raw_units": t_dist,
                   "depth_unit": "mm", "neighbours": "temporal +-4",
                   "trajectory": "straight walk along a corridor, %.2f m per keyframe" % walk if walk else "circle in a room",
                   "sharding": "one sequence in contiguous keyframe blocks, masks handed rank to rank, voxel all-to-all"
                   if world > 1 else "single GPU"},
        "points_per_s": n_points * args.steps / dt,
        "output_points": n_points,
        "device_ms_per_step_rank0": st["last_run_ms"],
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "keyframes_per_launch": (plan.end - plan.first) / launches,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "bytes_per_launch": bytes_per_launch,
            "avg_launch_us": avg_launch_s * 1e6,
            "launches_per_step": launches,
            "scan_kernel": None if chain else {
                "avg_launch_us": scan_avg_s * 1e6, "bytes_per_launch": scan_bytes(pst) / launches,
                "achieved": scan_bytes(pst) / launches / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0},
            "whole_step_frac": total_bytes(st) / (st["last_run_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            if st["last_run_ms"] > 0 else None,
            "fp64": fp64_roof,
            "chain": chain_roof,
        },
        "table": {"capacity": st["table_capacity"], "used": st["table_used"],
                  "load_factor": st["table_used"] / max(1, st["table_capacity"]),
                  "max_probe": st["max_probe"]},
        "counters": {k: st[k] for k in ("n_px", "n_gather", "n_acc", "n_hit", "n_vox")},
        # 1/0: the persistent chain kernel ran; fallbacks > 0: a device-wide barrier timed out (another process held
        # CUs) and the timed runs went through the slower kernel pair -- such a line is NOT the path's figure
        "chain": {"fused_chain": st["fused_chain"], "fallbacks": max(st["chain_fallbacks"], pst["chain_fallbacks"])},
    }
    if st["chain_fallbacks"] or (not st["fused_chain"] and os.environ.get("SFM_FUSE_CHAIN", "1") != "0"):
        out["data"] += " (FALLBACK: the kernel pair ran instead of the chain kernel; not the path's figure)"
        print("bench.py: the persistent chain kernel fell back to the kernel pair", file=sys.stderr, flush=True)

    if st["precomputed"]:
        out["two_phase"] = {"phase_a_ms": st["precompute_ms"], "phase_b_ms": st["frame_kernels_ms"],
                            "note": "phase A = mask-independent verdicts of every pixel (divides by the number of GPUs); "
                                    "phase B = the ordered accept/mark/emit chain (does not)"}
    if args.split_run and split is None:
        out["split_run"] = {"masks_final_host_ms": half[0] / args.steps * 1e3, "run_complete_host_ms": half[1] / args.steps * 1e3,
                            "chain_device_ms": st["frame_kernels_ms"],
                            "note": "sfm_fusion_run_chain + sfm_fusion_finish: the voxel filter runs behind the chain"}
    if split is not None:
        out["mask_handover_bytes_rank0"] = split.bytes_sent // max(1, args.steps + args.warmup)
        mine = {k: v / args.steps for k, v in acc.items()}
        mine["keyframes"] = plan.end - plan.first
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        out["per_rank"] = rows
        # what DESIGN.md section 6 predicts for this plan (C3 costs per keyframe, scaled by the keyframe's pixels): the
        # time the last rank is done, without hand-overs and exchange -- to be held against ms_per_step and per_rank
        blocks = [r["keyframes"] for r in rows]
        scale = (W * H) / 921600.0
        out["model"] = {"cost_us_per_c3_keyframe": list(dist_fusion.PIPELINE_COST),
                        "last_rank_done_ms": dist_fusion.modelled_makespan([b / max(1, n_kf) for b in blocks],
                                                                          dist_fusion.PIPELINE_COST, n_kf) * scale / 1e3}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(seq, min(args.cpu_sample_frames, n_kf), W, H, voxel, t_dist)
    if rank == 0:
        print(json.dumps(out), flush=True)


def host_threads():
    """(threads the CPU baseline uses, CPUs the machine reports).  The threads are every CPU this process may use:
    its affinity mask, bounded by OMP_NUM_THREADS when the box sets it (a GPU box hands each GPU's job a share
    of the machine and says so there; os.cpu_count() reports the whole machine)."""
    host_cpus = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = host_cpus
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) >= 1:
        avail = min(avail, int(share))
    return max(1, avail), host_cpus


def cpu_baseline(seq, n_sample, W, H, voxel, t_dist):
    """Times the CPU oracle (a restatement of the reference, kind "port") on the first
    n_sample keyframes of the same workload, on this box's host cores."""
    from oracle import oracle
    frames = seq.frame_dicts(0, n_sample)
    p = oracle.default_params(voxel_size=voxel, T_dist=t_dist)
    threads, host_cpus = host_threads()
    t0 = time.perf_counter()
    c = oracle.fuse(frames, p, n_threads=threads)
    t_mt = time.perf_counter() - t0
    n1 = max(1, n_sample // 8)
    t0 = time.perf_counter()
    oracle.fuse(frames[:n1] if n1 == n_sample else seq.frame_dicts(0, n1), p, n_threads=1)
    t_1 = time.perf_counter() - t0
    return {
        "value": n_sample * W * H / t_mt,
        "unit": "px/s",
        "cores": threads,
        "host_cpus": host_cpus,
        "kind": "port",
        "sample": f"first {n_sample} keyframes of the workload, pixel loop over {threads} host threads "
                  f"(mirrors the reference's par_unseq loop); single-thread figure on the first {n1}",
        "value_1core": n1 * W * H / t_1,
        "seconds": t_mt + t_1,
        "sample_points": int(c.key.shape[0]),
        "note": "restatement without the reference's per-pixel shared_ptr/cv::Mat refcount traffic: "
                "expected to be faster than the real reference binary",
    }


if __name__ == "__main__":
    main()
