"""Known-answer tests for the CPU oracle, derived by hand from the reference's source
(SURVEY.md section 8c).  The reference ships no tests or golden data: these pins are all
there is, hence "parity unpinned" in oracle/fusion_oracle.c."""
import math

import numpy as np
import pytest

from oracle import oracle
import scenes


def test_kat1_deproject():
    # pinhole.cpp:12-19: K=(500,500,320,240), (u,v,d)=(420,140,2000) -> (400,-400,2000)
    p = oracle.deproject((500, 500, 320, 240), 420, 140, 2000)
    assert p.tolist() == [400.0, -400.0, 2000.0]


def test_kat2_trace_closed_form():
    # f=525, u=v=0, d=1000: 0.1/525^2 + 40/525^2
    tr = oracle.cov_trace(0, 0, 1000, 525.0)
    assert tr == pytest.approx(0.1 / 525**2 + 40.0 / 525**2, rel=1e-13)
    assert tr == pytest.approx(1.45488e-4, rel=1e-5)
    # general closed form 0.05 (z/f)^2 * 2 + 10 z^4 (u^2+v^2+f^2)/(f^4 b^2), raw pixel coordinates
    for (u, v, d, f) in [(639, 479, 3000, 525.0), (10, 400, 777, 913.848), (0, 0, 65535, 100.0)]:
        z = d / 1000.0
        want = 0.1 * (z / f) ** 2 + 10.0 * z**4 * (u * u + v * v + f * f) / (f**4 * 0.25)
        assert oracle.cov_trace(u, v, d, f) == pytest.approx(want, rel=1e-12)


def test_kat3_cov_gate_location():
    # the gate trace > 0.5 opens at d ~ (f^2/80)^(1/4) m at pixel (0,0) and earlier in the corner
    f = 525.0
    d0 = 1000.0 * (f * f / 80.0) ** 0.25
    assert oracle.cov_trace(0, 0, d0 * 0.99, f) < 0.5 < oracle.cov_trace(0, 0, d0 * 1.01, f)
    assert oracle.cov_trace(639, 479, 5600, f) < 0.5 < oracle.cov_trace(639, 479, 5760, f)


def test_kat4_voxel_key_truncates_toward_zero():
    assert oracle.voxel_key(-0.12, 0.05) == 0xFFFFFFFFFFFFFFFE
    assert oracle.voxel_key(0.04, 0.05) == 0
    assert oracle.voxel_key(-0.04, 0.05) == 0
    assert oracle.voxel_key(0.05, 0.05) == 1
    assert oracle.voxel_key(12.349, 0.05) == int(12.349 / 0.05)


def test_kat5_nonfinite_and_out_of_range_keys():
    for x in (math.nan, math.inf, -math.inf, 1e300, -1e300):
        assert oracle.voxel_key(x, 0.05) == 0x8000000000000000
    # [2^63, 2^64) converts like an unsigned value (cvttsd2si pair)
    assert oracle.voxel_key(2.0**63 * 0.05 * 1.5, 0.05) == int(2.0**63 * 0.05 * 1.5 / 0.05)


def test_point_in_frame_bounds_and_floor():
    K = (100.0, 100.0, 32.0, 24.0)
    T = np.eye(4)
    assert oracle.point_in_frame(K, T, 64, 48, (0.0, 0.0, 10.0)) == (32, 24)
    assert oracle.point_in_frame(K, T, 64, 48, (0.049, -0.051, 10.0)) == (32, 23)     # floor, not round
    assert oracle.point_in_frame(K, T, 64, 48, (3.2, 0.0, 10.0)) is None              # px == cols
    assert oracle.point_in_frame(K, T, 64, 48, (3.1999, 0.0, 10.0)) == (63, 24)
    assert oracle.point_in_frame(K, T, 64, 48, (-3.2001, 0.0, 10.0)) is None          # px < 0
    assert oracle.point_in_frame(K, T, 64, 48, (0.0, 0.0, -10.0)) is None             # behind
    assert oracle.point_in_frame(K, T, 64, 48, (0.0, 0.0, 0.0)) is None               # 0/0: defined as invisible


def test_kat6_identical_poses_filtered_out():
    # same position => distance 0, not > min_dist_filter; view directions equal => dot 1 >= 0.30
    f0 = scenes.frame(np.full((12, 16), 30), K=(20, 20, 7.5, 5.5), nbrs=[1])
    f1 = scenes.frame(np.full((12, 16), 30), K=(20, 20, 7.5, 5.5), nbrs=[0])
    c = oracle.fuse([f0, f1])
    assert c.key.shape[0] == 0 and c.frame_before.tolist() == [0, 0]
    # same position but looking the other way: kept (dot = -1 < 0.30), still no visible pixel
    f1b = scenes.frame(np.full((12, 16), 30), K=(20, 20, 7.5, 5.5), yaw=180.0, nbrs=[0])
    assert oracle.neighbor_kept(f0["T_wk"], f1b["T_wk"])
    assert not oracle.neighbor_kept(f0["T_wk"], f1["T_wk"])


def test_kat7_two_frames_plane():
    W, H, d, shift, K = 48, 36, 40, 3.0, (60.0, 60.0, 23.5, 17.5)
    frames = scenes.plane_pair(W, H, d, shift, K)
    # tiny voxels: nothing merges; T_dist 5: the one-pixel offsets that floor() produces when a
    # re-projection rounds to just under an integer (0.67 units here) still count as close
    p = oracle.default_params(voxel_size=1e-3, T_dist=5.0)
    c = oracle.fuse(frames, p)
    # Re-observation of frame-0 pixel (u,v) in frame 1, in the reference's operation order
    # (ideal value px = u - shift*fx/d = u - 4.5, py = v; rounding decides the borders).
    def px01(u, s):
        return ((((u - K[2]) / K[0]) * d - s) / d) * K[0] + K[2]

    def py(v):
        return ((((v - K[3]) / K[1]) * d) / d) * K[1] + K[3]

    rows = [v for v in range(H) if 0 <= py(v) < H]
    vis0 = [u for u in range(W) if 0 <= px01(u, shift) < W]
    assert len(vis0) == 43 and len(rows) in (H - 1, H)
    assert c.frame_before[0] == len(vis0) * len(rows)
    # frame 1: the pixels frame 0 hit are masked; the others re-project into frame 0 at ~u+4.5
    hit = {(math.floor(px01(u, shift)), math.floor(py(v))) for u in vis0 for v in rows}
    def px10(u):
        return ((((u - K[2]) / K[0]) * d + shift) / d) * K[0] + K[2]

    acc1 = [(u, v) for v in rows for u in range(W) if (u, v) not in hit and 0 <= px10(u) < W]
    assert c.frame_before[1] == len(acc1)
    # the fused point of frame-0 pixel (u,v) is frame 1's observation at the floored pixel
    tag = (0 << 32) | (10 * W + 20)
    i = int(np.nonzero(c.tag == tag)[0][0])
    ix, iy = math.floor(px01(20, shift)), math.floor(py(10))
    assert ix == 15
    want_x = ((ix - K[2]) / K[0]) * d + shift
    want_y = ((iy - K[3]) / K[1]) * d
    assert c.xyz[i, 0] == pytest.approx(want_x, rel=1e-14)
    assert c.xyz[i, 1] == pytest.approx(want_y, rel=1e-14)
    assert c.xyz[i, 2] == pytest.approx(d, rel=1e-14)
    assert c.counters["n_hit"] == c.counters["n_acc"]          # one observation per accepted pixel


def test_kat8_first_wins():
    frames = scenes.plane_pair()
    fine = oracle.fuse(frames, oracle.default_params(voxel_size=1e-3))
    coarse = oracle.fuse(frames, oracle.default_params(voxel_size=5.0))
    assert coarse.key.shape[0] < fine.key.shape[0]
    # the survivor of every voxel is the smallest (frame, raster) tag among its members
    keys = [tuple(oracle.voxel_key(x, 5.0) for x in p) for p in fine.xyz]
    best = {}
    for k, t in zip(keys, fine.tag.tolist()):
        best[k] = min(best.get(k, t), t)
    got = {tuple(k): t for k, t in zip(coarse.key.tolist(), coarse.tag.tolist())}
    assert got == best
    assert coarse.global_before == int(coarse.frame_after.sum())


def test_kat9_zero_depth_observation_poisons_the_point():
    # a zero-depth neighbour pixel has disparity inf -> J = 0 -> trace 0 < T_cov -> kept with
    # w = 0; its "point" is the neighbour's camera centre.  T_dist is opened up so that the
    # observation also counts as close; then 1/w = inf reaches the weighted mean (:161-164).
    W, d, shift, K = 48, 40, 3.0, (60.0, 60.0, 23.5, 17.5)
    frames = scenes.plane_pair()
    ix = math.floor(((((20 - K[2]) / K[0]) * d - shift) / d) * K[0] + K[2])
    iy = math.floor(((((10 - K[3]) / K[1]) * d) / d) * K[1] + K[3])
    frames[1]["depth"] = frames[1]["depth"].copy()
    frames[1]["depth"][iy, ix] = 0
    c = oracle.fuse(frames, oracle.default_params(voxel_size=1e-3, T_dist=100.0))
    tag = (0 << 32) | (10 * W + 20)
    i = int(np.nonzero(c.tag == tag)[0][0])
    assert not np.isfinite(c.xyz[i]).any()    # (3,0,0)*inf / inf: NaN in every coordinate
    assert c.key[i].tolist() == [0x8000000000000000] * 3
    # with the reference's T_dist the camera centre is far: the pixel is rejected instead
    c = oracle.fuse(frames)
    assert tag not in c.tag.tolist()


def test_kat10_colour_order():
    from sfm_amd import fusion
    import io
    frames = scenes.plane_pair()
    frames[0]["bgr"][:] = (10, 20, 30)
    c = oracle.fuse(frames, oracle.default_params(voxel_size=1e-3))
    i = int(np.nonzero(c.tag >> np.uint64(32) == 0)[0][0])
    assert c.bgr[i].tolist() == [10, 20, 30]                    # stored B,G,R
    out = io.StringIO()
    fusion.write_cloud(out, c.xyz[i:i + 1], c.bgr[i:i + 1])
    assert out.getvalue().split()[3:] == ["30", "20", "10"]     # printed R G B (:63)


def test_kat11_edge_count_100_selects_but_is_not_a_neighbour():
    from sfm_amd import fusion
    m = fusion.Map()
    for _ in range(3):
        m.add_keyframe(fusion.KeyFrame(None, None, None, None, None))
    m.insert_edge(0, 1, 100)
    m.insert_edge(1, 2, 101)
    m.insert_edge(1, 2, 500)                                    # frozen at first insertion (map.cpp:113)
    dr = fusion.DenseReconstruction(m)
    assert dr._select() == [0, 1, 2]                            # count >= 100 (:34)
    assert m.get_neighbors(0, 100) == []                        # strictly > 100 (map.cpp:157)
    assert m.get_neighbors(1, 100) == [2]
    assert m.get_edge(1, 2) == 101


def test_empty_inputs():
    c = oracle.fuse([])
    assert c.key.shape == (0, 3)
    f = scenes.frame(np.zeros((0, 0)), nbrs=[])
    c = oracle.fuse([f])
    assert c.key.shape == (0, 3) and c.frame_before.tolist() == [0]
    # a keyframe without neighbours yields nothing (:180-182)
    c = oracle.fuse([scenes.frame(np.full((8, 8), 30), nbrs=[])])
    assert c.key.shape[0] == 0
