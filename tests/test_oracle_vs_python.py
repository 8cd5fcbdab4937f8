"""The C oracle against the independent pure-Python restatement: bit-exact on tiny scenes."""
import numpy as np
import pytest

from oracle import oracle
from sfm_amd import synth
import py_restatement as pr
import scenes


def _compare(frames, **prm):
    c = oracle.fuse(frames, oracle.default_params(**prm))
    r = pr.fuse(frames, **prm)
    assert r["frame_before"] == c.frame_before.tolist()
    assert r["frame_after"] == c.frame_after.tolist()
    assert r["global_before"] == c.global_before and r["global_after"] == c.global_after
    assert len(r["rows"]) == c.key.shape[0]
    for i, (key, tag, p, col) in enumerate(r["rows"]):
        assert list(key) == c.key[i].tolist()
        assert tag == int(c.tag[i])
        assert list(col) == c.bgr[i].tolist()
        got = c.xyz[i]
        for a, b in zip(p, got):
            assert (a != a and b != b) or a == b, (i, p, got)
    return c


def test_plane_pair_default_constants():
    c = _compare(scenes.plane_pair())
    assert c.key.shape[0] > 0


def test_plane_pair_wide_threshold_and_zero_depth():
    frames = scenes.plane_pair()
    frames[1]["depth"] = frames[1]["depth"].copy()
    frames[1]["depth"][5:9, 10:20] = 0
    frames[0]["depth"] = frames[0]["depth"].copy()
    frames[0]["depth"][20:22, 30:33] = 0
    c = _compare(frames, T_dist=100.0, voxel_size=1.0)
    assert np.isnan(c.xyz).any()


def test_synthetic_room_mm_units():
    seq = synth.make_sequence(4, 40, 30, K=(131.25, 131.25, 19.5, 14.5), window=2)
    c = _compare(seq.frame_dicts(), voxel_size=25.0, T_dist=500.0)
    assert c.key.shape[0] > 100


def test_rotated_views_and_custom_S():
    d = np.full((24, 32), 35)
    f0 = scenes.frame(d, K=(40, 41, 15.5, 11.5), C=(0, 0, 0), yaw=0.0, nbrs=[1, 2])
    f1 = scenes.frame(d, K=(40, 41, 15.5, 11.5), C=(2, 0.3, 0), yaw=4.0, nbrs=[0, 2])
    f2 = scenes.frame(d + 3, K=(38, 38, 16.0, 12.0), C=(-1.5, 0, 1.0), yaw=-3.0, nbrs=[1, 0])
    _compare([f0, f1, f2], T_dist=6.0, voxel_size=0.7, S=(0.07, 0.02, 8.0), base_line=0.4)
