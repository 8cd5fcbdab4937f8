"""The C++ facade (sfm_amd/cpp/dense_reconstruction.hpp: the reference's two-call shape over
the C ABI) on the GPU: its text output must be the oracle's cloud, formatted like the
reference's writer."""
import io
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle
from sfm_amd import fusion
import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_example_matches_oracle():
    exe = os.path.join(ROOT, "sfm_amd", "cpp", "facade_example")
    assert os.path.exists(exe), "run ./build.sh"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    K = (60.0, 60.0, 23.5, 17.5)
    d = np.full((36, 48), 40)
    # edges (0,1)=250 and (1,2)=100: all three selected, only 0<->1 are neighbours
    frames = [scenes.frame(d, K=K, C=(0, 0, 0), nbrs=[1]), scenes.frame(d, K=K, C=(3, 0, 0), nbrs=[0]),
              scenes.frame(d, K=K, C=(6, 0, 0), nbrs=[])]
    want = oracle.fuse(frames, oracle.default_params(T_dist=5.0, voxel_size=1e-3))
    buf = io.StringIO()
    fusion.write_cloud(buf, want.xyz, want.bgr)
    assert r.stdout == buf.getvalue()
    assert f"points {want.key.shape[0]} before {want.counters['n_acc']}" in r.stderr
    # the reference's progress log (dense_reconstruction.cpp:49, :54-58, :118-122), line for line, counts from the oracle
    log = []
    for i in range(3):
        log += [f"Processing keyframe {i}/3", f"Points before voxel filter: {int(want.frame_before[i])}",
                f"Points after voxel filter: {int(want.frame_after[i])}", ""]
    log += ["processed all map frames", f"Points before voxel removal: {want.global_before}",
            f"Points after voxel removal: {want.global_after}"]
    assert r.stderr.split("\n")[:len(log)] == log, r.stderr
