"""sfm_fusion_partition_counts_noalloc (the entry point that allocates nothing: static device scratch) against
sfm_fusion_partition_counts, through the C ABI, on two calls in a row -- the second on a smaller cloud with fewer
parts, so stale counters of the first call would show."""
import pytest

from sfm_amd import fusion, synth
from helpers import both_params

pytestmark = pytest.mark.gpu


def _ctx(n_frames, W, H, K, gp):
    seq = synth.make_sequence(n_frames, W, H, K=K, window=2)
    frames = seq.frame_dicts()
    ctx = fusion.FusionContext(gp)
    for f in frames:
        ctx.add_keyframe(f["depth"], f["bgr"], f["K"], f["T_wk"], f["T_kw"])
    for i, f in enumerate(frames):
        ctx.set_neighbors(i, f["nbrs"])
    ctx.run()
    return ctx


def test_no_alloc_partition_counts_match_the_allocating_entry_point():
    _, gp = both_params(voxel_size=20.0, T_dist=500.0)
    big = _ctx(8, 160, 120, (131.25, 131.25, 79.75, 59.75), gp)
    small = _ctx(3, 48, 36, (98.4, 98.4, 23.5, 17.5), gp)
    n_big, n_small = big.num_points(), small.num_points()
    assert n_big > 4 * n_small > 0
    # first call: many parts on the larger cloud
    want = big.partition_counts(1024)
    got = big.partition_counts_noalloc(1024)
    assert got == want and sum(got) == n_big
    # second call, right behind it: fewer parts, smaller cloud -- the array still holds the first call's counts
    want = small.partition_counts(5)
    got = small.partition_counts_noalloc(5)
    assert got == want and sum(got) == n_small
    # and once more on the first ctx with another part count
    assert big.partition_counts_noalloc(7) == big.partition_counts(7)
    # "allocates nothing": the device's free memory is the same before and after a burst of calls (hipMemGetInfo; a
    # hipMalloc of any size would take at least one allocation granule from it)
    import torch
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for parts in (1, 3, 64, 1024, 2):
        small.partition_counts_noalloc(parts)
        big.partition_counts_noalloc(parts)
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, (free_before, free_after)
    # beyond the static array's size: a clear error, nothing written
    with pytest.raises(fusion.FusionError) as e:
        big.partition_counts_noalloc(1025)
    assert e.value.code == fusion.SFM_E_CAPACITY and "1024" in str(e.value)
    # the allocating entry point is unchanged: it still arms export_records
    n = sum(big.partition_counts(4))
    buf = torch.empty(max(n, 1) * fusion.SFM_RECORD_BYTES, dtype=torch.uint8, device="cuda")
    big.export_records(buf.data_ptr(), n * fusion.SFM_RECORD_BYTES)
    big.close()
    small.close()
