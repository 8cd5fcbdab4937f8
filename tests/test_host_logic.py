"""Host-side logic that needs no GPU: the reference-shaped containers, the text writer and
the component sharding used by the multi-GPU path."""
import io

import numpy as np

from sfm_amd import dist_fusion, fusion, synth


def test_writer_matches_ostream_default_formatting():
    # os << double: precision 6, %g style (src/dense_reconstruction.cpp:61-64)
    out = io.StringIO()
    xyz = np.array([[1599.915223, -8.8227014, 902.864993], [0.000012345678, 1e21, -0.5],
                    [float("nan"), float("inf"), 100000.0], [1234567.0, 0.1 + 0.2, 1e-5]])
    bgr = np.array([[1, 2, 3], [255, 0, 7], [9, 9, 9], [0, 0, 0]])
    fusion.write_cloud(out, xyz, bgr)
    lines = out.getvalue().splitlines()
    assert lines[0] == "1599.92 -8.8227 902.865 3 2 1"
    assert lines[1] == "1.23457e-05 1e+21 -0.5 7 0 255"
    assert lines[2] == "nan inf 100000 9 9 9"
    assert lines[3] == "1.23457e+06 0.3 1e-05 0 0 0"


def test_map_neighbours_and_selection():
    m = fusion.Map()
    for _ in range(5):
        m.add_keyframe(fusion.KeyFrame(None, None, None, None, None))
    m.insert_edge(0, 1, 150)
    m.insert_edge(1, 2, 100)
    m.insert_edge(2, 3, 99)
    m.insert_edge(3, 1, 101)
    dr = fusion.DenseReconstruction(m)
    assert dr._select() == [0, 1, 2, 3]          # 2 via the ==100 edge, 3 via the 101 edge, 4 never
    assert m.get_neighbors(1, 100) == [0, 3]     # ascending; the ==100 edge is not a neighbour
    assert m.get_neighbors(2, 100) == []
    assert m.get_neighbors(1, 15) == [0, 2, 3]


def test_components_and_sharding():
    nbrs = [[1], [0, 2], [1], [4], [3], [], [7], [6]]
    comps = dist_fusion.components(nbrs)
    assert comps == [[0, 1, 2], [3, 4], [5], [6, 7]]
    shards = dist_fusion.shard_components(nbrs, [10] * 8, 2)
    assert sorted(shards[0] + shards[1]) == list(range(8))
    for comp in comps:                           # never split a component
        assert any(set(comp) <= set(s) for s in shards)
    assert abs(len(shards[0]) - len(shards[1])) <= 2
    assert all(s == sorted(s) for s in shards)


def test_synthetic_sequence_shape_and_components():
    seq = synth.make_sequence(6, 32, 24, K=(65.6, 65.6, 15.5, 11.5), components=2, window=2)
    assert seq.depth.shape == (6, 24, 32) and seq.bgr.shape == (6, 24, 32, 3)
    assert dist_fusion.components(seq.nbrs) == [[0, 1, 2], [3, 4, 5]]
    for i in range(6):
        assert np.allclose(seq.T_wk[i] @ seq.T_kw[i], np.eye(4), atol=1e-9)
    assert seq.depth_u16(0).dtype == np.uint16 and seq.depth_u16(0).min() > 0
