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


def test_cpp_multi_gpu_host_at_one_rank(tmp_path, fusion_path):
    """sfm_amd/cpp/dist_reconstruct: the C++ host of the multi-GPU path (block plan, mask hand-over, RCCL exchange:
    allgather of counts and key ranges, grouped send/recv of records, owner merge) run as ONE rank -- every RCCL call
    is made, with itself as the only peer -- must write the one-GPU cloud."""
    if fusion_path != "chain":
        pytest.skip("one run of the executable")
    m = small_map(n=6, W=64, H=48)
    want, _ = oracle_cloud(m, voxel_size=10.0, T_dist=500.0)
    path = str(tmp_path / "scene.sfmr")
    replay.write_replay(path, m)
    buf = io.StringIO()
    fusion.write_cloud(buf, want.xyz, want.bgr)
    exe = os.path.join(ROOT, "sfm_amd", "cpp", "dist_reconstruct")
    assert os.path.exists(exe), "run ./build.sh"
    out = str(tmp_path / "dist.txt")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, path, out, "--voxel-size", "10", "--t-dist", "500"], capture_output=True, text=True, timeout=240,
                       env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "rank 0 of 1: %d owned points" % want.key.shape[0] in r.stdout
    assert open(out + ".rank0").read() == buf.getvalue()


def test_in_library_selection_and_neighbour_rules():
    """sfm_fusion_set_covisibility: all keyframes + raw edge counts in, the
    library applies count >= 100 / count > 100 itself.  Same cloud as the facade-side rules; tags
    carry map indices instead of positions in the selected order."""
    m = small_map(n=6, W=64, H=48)
    # two more keyframes that no edge >= 100 touches, in the middle of the order and at its end
    f = m.keyframes[2]
    m.keyframes.insert(3, fusion.KeyFrame(img=f.img, depth_img=f.depth_img, K=f.K, T_wk=f.T_wk, T_kw=f.T_kw))
    old = dict(m.covisibility_edge)
    m.covisibility_edge.clear()
    m._adj.clear()
    shift = lambda i: i + 1 if i >= 3 else i
    for (a, b), n in old.items():
        m.insert_edge(shift(a), shift(b), n)
    m.insert_edge(3, 0, 42)
    params = fusion.default_params(voxel_size=10.0, T_dist=500.0)
    d1 = fusion.DenseReconstruction(m, params)
    xyz1, bgr1, key1, tag1 = d1.fuse()
    st1 = d1.ctx.stats()
    d2 = fusion.DenseReconstruction(m, params)
    xyz2, bgr2, key2, tag2 = d2.fuse(in_library_rules=True)
    st2 = d2.ctx.stats()
    assert 3 not in d1.selected and len(d1.selected) == len(m.keyframes) - 1
    assert np.array_equal(key1, key2) and np.array_equal(bgr1, bgr2)
    assert np.array_equal(xyz1.view(np.uint64), xyz2.view(np.uint64))
    sel = np.asarray(d1.selected, dtype=np.uint64)
    assert np.array_equal((sel[(tag1 >> np.uint64(32)).astype(np.int64)] << np.uint64(32)) | (tag1 & np.uint64(0xFFFFFFFF)), tag2)
    for k in ("n_px", "n_gather", "n_acc", "n_hit", "n_vox", "global_before", "global_after"):
        assert st1[k] == st2[k], k
    want, _ = oracle_cloud(m, voxel_size=10.0, T_dist=500.0)
    assert np.array_equal(key1, want.key) and np.array_equal(xyz1.view(np.uint64), want.xyz.view(np.uint64))
    d1.ctx.close()
    d2.ctx.close()


def test_fusion_from_a_png_recording(tmp_path):
    """Disk to cloud: a recording written as color_N.png / depth_N.png, read
    back by RecordingReader, fused -- same cloud as fusing the arrays it was written from."""
    from PIL import Image
    from sfm_amd import recording, synth
    from helpers import assert_same_cloud, both_params, run_gpu
    seq = synth.make_sequence(5, 64, 48, K=(157.5, 157.5, 31.5, 23.5), window=2)
    frames = seq.frame_dicts()
    d = str(tmp_path)
    for i, f in enumerate(frames):
        Image.fromarray(np.ascontiguousarray(f["bgr"][:, :, ::-1])).save(f"{d}/color_{i}.png")
        Image.fromarray(f["depth"].astype(np.uint16)).save(f"{d}/depth_{i}.png")
    rd = recording.RecordingReader(d)
    loaded = []
    while (fr := rd.read_frames()) is not None:
        i = len(loaded)
        loaded.append(dict(frames[i], bgr=fr[0], depth=fr[1]))
    assert len(loaded) == len(frames)
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    assert_same_cloud(run_gpu(loaded, gp), oracle.fuse(frames, op))
