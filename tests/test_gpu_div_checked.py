"""The chain kernels divide in pairs with a shared reciprocal and WITHOUT an operand check (it costs more than the
sharing saves); the guard is a host-side bound on poses and intrinsics (chain_div_ok).  This test runs the library's
div-checked build -- the same kernels with the operand check compiled in (build.sh: libsfm_fusion_checked.so, never
loaded by the product) -- over the random scenes, the golden scenes and a room sequence, and requires that no kernel
ever meets an operand outside the range where the host said yes (kErrDivRange would show as a chain fallback), on the
fused chain and on the two-phase chain.  A second run turns the host-side bound off (a switch that only the checked
build has) on a scene it refuses: the kernels' checks must catch it and the run must end, exact, on the kernel pair."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKED = os.path.join(ROOT, "sfm_amd", "lib", "libsfm_fusion_checked.so")

SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from oracle import oracle
from sfm_amd import fusion, synth
import scenes
from helpers import assert_same_cloud, both_params, run_gpu
from test_gpu_fuzz import random_scene

assert fusion.version().endswith("div-checked"), fusion.version()
mode = sys.argv[1]
if mode == "sweep":
    cases = [random_scene(seed) for seed in range(40)]
    cases.append((scenes.plane_pair(), {}))
    seq = synth.make_sequence(12, 160, 120, K=(131.25, 131.25, 79.5, 59.5), window=4)
    cases.append((seq.frame_dicts(), dict(voxel_size=10.0, T_dist=500.0)))
    ran = 0
    for frames, params in cases:
        op, gp = both_params(**params)
        g = run_gpu(frames, gp)
        assert_same_cloud(g, oracle.fuse(frames, op))
        st = g["stats"]
        assert st["chain_fallbacks"] == 0, "a shared division met an operand outside its range"
        ran += st["fused_chain"]
    assert ran >= 35, ran          # the chain kernels really ran (scenes without keyframes aside)
    print("SWEEP-OK", ran)
elif mode == "trust_verdict":
    # the same on the two-phase chain (SFM_PRECOMPUTE=1): every division of that path is in phase A, whose error word
    # must survive the run's memset; the verdicts are recomputed with plain divisions and the run repeated
    frames = scenes.plane_pair()
    for f in frames:
        f["T_kw"] = f["T_kw"] * 2.0 ** -500
    op, gp = both_params()
    g = run_gpu(frames, gp)
    assert_same_cloud(g, oracle.fuse(frames, op))
    st = g["stats"]
    assert st["chain_fallbacks"] == 1 and st["precomputed"] == 1, st
    print("TRUST-VERDICT-OK")
else:
    frames = scenes.plane_pair()
    for f in frames:
        f["T_kw"] = f["T_kw"] * 2.0 ** -500          # camera coordinates of ~2^-495: outside the shared range
    op, gp = both_params()
    g = run_gpu(frames, gp)
    assert_same_cloud(g, oracle.fuse(frames, op))
    st = g["stats"]
    assert st["chain_fallbacks"] == 1 and st["fused_chain"] == 0, st   # caught in the kernel, rerun on the pair
    print("TRUST-OK")
'''


def _run(mode, **env):
    if not os.path.exists(CHECKED):
        pytest.fail("libsfm_fusion_checked.so is missing: run ./build.sh")
    e = dict(os.environ, SFM_FUSION_LIB=CHECKED, **env)
    e.pop("SFM_FUSE_CHAIN", None)
    for k in ("SFM_PRECOMPUTE", "SFM_CHAIN_DIV_TRUST"):
        if k not in env:
            e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + SCRIPT, mode], env=e, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_no_operand_leaves_the_shared_range_where_the_host_bound_says_yes(fusion_path):
    if fusion_path == "pair":
        pytest.skip("the kernel pair divides plainly")
    out = _run("sweep", **({"SFM_PRECOMPUTE": "1"} if fusion_path == "verdict" else {}))
    assert "SWEEP-OK" in out


def test_the_checked_build_catches_what_the_host_bound_refuses(fusion_path):
    if fusion_path == "pair":
        pytest.skip("the kernel pair divides plainly")
    if fusion_path == "verdict":   # phase A's error word used to be erased by the run's memset
        assert "TRUST-VERDICT-OK" in _run("trust_verdict", SFM_CHAIN_DIV_TRUST="1", SFM_PRECOMPUTE="1")
    else:
        assert "TRUST-OK" in _run("trust", SFM_CHAIN_DIV_TRUST="1")
