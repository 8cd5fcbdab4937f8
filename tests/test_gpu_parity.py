"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact."""
import numpy as np
import pytest

from sfm_amd import synth
from oracle import oracle
from helpers import assert_same_cloud, both_params, run_gpu

pytestmark = pytest.mark.gpu

SMALL_K = (131.25, 131.25, 79.75, 59.75)


@pytest.mark.parametrize("n,W,H,K,window", [
    (6, 160, 120, SMALL_K, 2),
    (12, 160, 120, SMALL_K, 4),
    (5, 161, 77, SMALL_K, 4),       # ragged: W*H not a multiple of the 256-pixel tile
    (8, 320, 240, (262.5, 262.5, 159.5, 119.5), 3),
])
def test_mm_units_sequences(n, W, H, K, window):
    seq = synth.make_sequence(n, W, H, K=K, window=window)
    frames = seq.frame_dicts()
    op, gp = both_params(voxel_size=10.0, T_dist=500.0)
    c = oracle.fuse(frames, op)
    g = run_gpu(frames, gp)
    assert c.key.shape[0] > 1000
    assert_same_cloud(g, c)
