"""bench.py's contract on the GPU box: one JSON line with the fields the driver reads, at N=1 on a
small workload and -- as a functional rehearsal, every rank on GPU 0 over gloo -- through the N>1 path (ONE sequence
in keyframe blocks, masks handed rank to rank, exchange by voxel-key hash, owner reduce) as torch.distributed.run
launches it.  The rehearsal's cloud must have exactly as many points as the single-GPU run of the same sequence."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEEDED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
          "vs_baseline", "dtype", "data", "config", "roofline")


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_single_gpu_line(fusion_path):
    if fusion_path != "chain":
        pytest.skip("bench.py runs the default path")
    r = subprocess.run([sys.executable, "bench.py", "--workload", "tiny_24kf_160x120", "--steps", "2", "--warmup", "1",
                        "--cpu-sample-frames", "24"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    for k in NEEDED + ("cpu_baseline",):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "single" and j["vs_baseline"] is None
    assert j["value"] > 0 and j["dtype"] == "f64" and j["config"]["workload"] == "tiny_24kf_160x120"
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "fuse_chain_kernel" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["sample_points"] == j["output_points"]          # the CPU sample is the whole tiny workload: same cloud size


@pytest.mark.gpu
def test_two_rank_rehearsal_line(fusion_path):
    if fusion_path != "chain":
        pytest.skip("bench.py runs the default path")
    env = dict(os.environ, SFM_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", "bench.py", "--gpus", "2",
                        "--workload", "tiny_24kf_160x120", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    for k in NEEDED:
        assert k in j, k
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "REHEARSAL" in j["data"]
    assert j["output_points"] > 0 and j["value"] > 0 and j["mask_handover_bytes_rank0"] > 0
    assert j["config"]["keyframes"] == 24 and j["config"]["keyframes_per_gpu"] == 12
    # the per-rank breakdown the scaling model of DESIGN.md section 6 is checked against
    pr = j["per_rank"]
    assert len(pr) == 2 and sum(r["keyframes"] for r in pr) == 24
    for r in pr:
        for k in ("phase_a_ms", "wait_ms", "phase_b_ms", "run_ms", "chain_host_ms", "mask_send_ms", "finish_ms", "exchange_ms"):
            assert r[k] >= 0.0, (k, r)
    assert j["model"]["last_rank_done_ms"] > 0.0 and len(j["model"]["cost_us_per_c3_keyframe"]) == 4
    assert pr[0]["phase_a_ms"] == 0.0 and pr[1]["phase_a_ms"] > 0.0      # rank 0 runs the fused chain, rank 1 the two phases
    assert pr[0]["phase_b_ms"] > 0.0 and pr[1]["phase_b_ms"] > 0.0 and pr[1]["wait_ms"] > 0.0
    # the same sequence on one rank: the split must not change the cloud
    r1 = subprocess.run([sys.executable, "bench.py", "--workload", "tiny_24kf_160x120", "--steps", "1", "--warmup", "0",
                         "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert _last_json(r1.stdout)["output_points"] == j["output_points"]
