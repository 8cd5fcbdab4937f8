"""Fusing from a keyframe replay dump: the C++ CLI
(sfm_amd/cpp/replay_reconstruct: mmap reader -> C++ facade -> C ABI -> writer) and the Python one
(python -m sfm_amd.replay) must both write exactly the oracle's cloud, as the reference's text
lines and as binary PLY.  The dump holds keyframes that the selection rule must drop."""
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import oracle
from sfm_amd import fusion, replay
from test_replay_format import small_map

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_cloud(m, **over):
    dense = fusion.DenseReconstruction(m)
    sel = dense._select()
    pos = {k: n for n, k in enumerate(sel)}
    frames = []
    for k in sel:
        kf = m.keyframes[k]
        frames.append(dict(depth=np.ascontiguousarray(kf.depth_img), bgr=np.ascontiguousarray(kf.img), K=tuple(kf.K),
                           T_wk=np.asarray(kf.T_wk), T_kw=np.asarray(kf.T_kw),
                           nbrs=[pos[j] for j in m.get_neighbors(k, 100)]))
    return oracle.fuse(frames, oracle.default_params(**over)), sel


def test_cli_from_replay_dump(tmp_path):
    m = small_map(n=6, W=64, H=48)
    want, sel = oracle_cloud(m, voxel_size=10.0, T_dist=500.0)
    # keyframe 6 has edges with counts 99 and 100: the == 100 edge selects it (KAT-11), but it is nobody's neighbour
    assert len(sel) == 7 and want.key.shape[0] > 500 and want.frame_before[6] == 0
    path = str(tmp_path / "scene.sfmr")
    replay.write_replay(path, m)
    buf = io.StringIO()
    fusion.write_cloud(buf, want.xyz, want.bgr)
    ply = io.BytesIO()
    fusion.write_ply(ply, want.xyz, want.bgr)

    exe = os.path.join(ROOT, "sfm_amd", "cpp", "replay_reconstruct")
    assert os.path.exists(exe), "run ./build.sh"
    out = str(tmp_path / "cpp.txt")
    r = subprocess.run([exe, path, out, "--voxel-size", "10", "--t-dist", "500"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == buf.getvalue()
    out = str(tmp_path / "cpp.ply")
    r = subprocess.run([exe, path, out, "--ply", "--voxel-size", "10", "--t-dist", "500"], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == ply.getvalue()

    env = dict(os.environ, PYTHONPATH=ROOT)
    out = str(tmp_path / "py.txt")
    r = subprocess.run([sys.executable, "-m", "sfm_amd.replay", path, out, "--voxel-size", "10", "--t-dist", "500"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert open(out).read() == buf.getvalue()
