"""Keyframe replay dump: the Python and the C++ implementation of the format
agree byte for byte, dumps round-trip, malformed files are refused.  The GPU half (fusing from a
dump through the C++ CLI) is in tests/test_gpu_replay.py."""
import os
import struct
import subprocess

import numpy as np
import pytest

from sfm_amd import fusion, replay, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_map(n=5, W=40, H=30, extra_unselected=True):
    seq = synth.make_sequence(n, W, H, K=(105.0, 105.0, 19.5, 14.5), window=2)
    m = fusion.Map()
    for i in range(n):
        f = seq.frame_dicts()[i]
        m.add_keyframe(fusion.KeyFrame(img=f["bgr"], depth_img=f["depth"], K=f["K"], T_wk=f["T_wk"], T_kw=f["T_kw"]))
    for i in range(n):
        for j in seq.nbrs[i]:
            if i < j:
                m.insert_edge(i, j, 150 + i)
    if extra_unselected:   # a keyframe that only has weak edges: dumped, but never selected
        f = seq.frame_dicts()[0]
        k = m.add_keyframe(fusion.KeyFrame(img=f["bgr"], depth_img=f["depth"], K=f["K"], T_wk=f["T_wk"], T_kw=f["T_kw"]))
        m.insert_edge(0, k, 99)
        m.insert_edge(1, k, 100)   # == 100: selects both endpoints (KAT-11) but is not a neighbour
    return m


def test_round_trip(tmp_path):
    m = small_map()
    path = str(tmp_path / "a.sfmr")
    size = replay.write_replay(path, m)
    assert os.path.getsize(path) == size and size % 64 == 0
    rp = replay.Replay(path)
    assert len(rp.map.keyframes) == len(m.keyframes)
    assert rp.map.covisibility_edge == m.covisibility_edge
    assert list(rp.map.covisibility_edge) == sorted(m.covisibility_edge)        # canonical edge order
    for a, b in zip(m.keyframes, rp.map.keyframes):
        assert np.array_equal(a.img, b.img) and np.array_equal(a.depth_img, b.depth_img)
        assert tuple(a.K) == tuple(b.K)
        assert np.array_equal(np.asarray(a.T_wk), b.T_wk) and np.array_equal(np.asarray(a.T_kw), b.T_kw)
    for i in range(len(m.keyframes)):
        assert rp.map.get_neighbors(i, 100) == m.get_neighbors(i, 100)
    assert fusion.DenseReconstruction(rp.map)._select() == fusion.DenseReconstruction(m)._select()
    rp.close()


def test_empty_map_and_empty_images(tmp_path):
    m = fusion.Map()
    path = str(tmp_path / "e.sfmr")
    replay.write_replay(path, m)
    rp = replay.Replay(path)
    assert rp.map.keyframes == [] and rp.map.covisibility_edge == {}
    rp.close()
    m.add_keyframe(fusion.KeyFrame(img=np.zeros((0, 0, 3), np.uint8), depth_img=np.zeros((0, 0), np.uint16),
                                   K=(1, 1, 0, 0), T_wk=np.eye(4), T_kw=np.eye(4)))
    replay.write_replay(path, m)
    rp = replay.Replay(path)
    assert rp.map.keyframes[0].depth_img.shape == (0, 0)
    rp.close()


def test_malformed_files_are_refused(tmp_path):
    m = small_map(3, 16, 12, extra_unselected=False)
    path = str(tmp_path / "a.sfmr")
    replay.write_replay(path, m)
    raw = open(path, "rb").read()
    bad = str(tmp_path / "bad.sfmr")
    for mutate in (lambda b: b"NOTADUMP" + b[8:],                           # magic
                   lambda b: b[:8] + struct.pack("<I", 9) + b[12:],        # version
                   lambda b: b[:len(b) - 64],                              # truncated
                   lambda b: b[:32]):                                      # shorter than a header
        open(bad, "wb").write(mutate(raw))
        with pytest.raises(ValueError):
            replay.Replay(bad)
    # an edge that names a keyframe the dump does not hold
    hdr = struct.unpack_from("<IIIIQQQ", raw, 8)
    edge_off = hdr[5]
    open(bad, "wb").write(raw[:edge_off] + struct.pack("<III", 0, 77, 200) + raw[edge_off + 12:])
    with pytest.raises(ValueError):
        replay.Replay(bad)


def test_cpp_writer_produces_the_same_bytes(tmp_path):
    """replay_reconstruct --rewrite reads the dump with the C++ reader and writes it back with the
    C++ writer; no GPU is needed up to that point (the fusion that follows fails without one,
    after the copy has been written)."""
    exe = os.path.join(ROOT, "sfm_amd", "cpp", "replay_reconstruct")
    if not os.path.exists(exe):
        pytest.skip("sfm_amd/cpp/replay_reconstruct not built (./build.sh)")
    m = small_map()
    path, copy = str(tmp_path / "a.sfmr"), str(tmp_path / "b.sfmr")
    replay.write_replay(path, m)
    subprocess.run([exe, path, str(tmp_path / "out.txt"), "--rewrite", copy], capture_output=True)
    assert open(copy, "rb").read() == open(path, "rb").read()
