"""The oracle reproduces the committed golden fixtures (guards the fixtures and the oracle
against drifting apart), single- and multi-threaded."""
import pytest

from oracle import oracle
from helpers import assert_clouds_equal, golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden(name):
    frames, params, want = load_golden(name)
    assert_clouds_equal(oracle.fuse(frames, oracle.default_params(**params)), want)


@pytest.mark.parametrize("name", ["room_mm_6x64x48", "room_mm_holes_5x61x37"])
def test_threaded_oracle_matches_golden(name):
    frames, params, want = load_golden(name)
    assert_clouds_equal(oracle.fuse(frames, oracle.default_params(**params), n_threads=4), want)


def test_there_are_fixtures():
    assert len(golden_names()) >= 5
