import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Every GPU test runs three times: on the persistent fuse_chain_kernel (one launch per group of
# keyframes, device-wide barrier between keyframes), on the scan_frame_kernel + observe_frame_kernel
# pair (two launches per keyframe), and on the two-phase chain (verdict_kernel for every pixel, then
# accept_chain_kernel; what a rank of a sequence split runs while it waits for its predecessors' masks).
# A ctx picks the path from SFM_FUSE_CHAIN / SFM_PRECOMPUTE when it is created; helpers.run_gpu checks
# through the stats which one really ran.
@pytest.fixture(autouse=True)
def fusion_path(request, monkeypatch):
    mode = getattr(request, "param", None)
    if mode is not None:
        monkeypatch.setenv("SFM_FUSE_CHAIN", "0" if mode == "pair" else "1")
        monkeypatch.setenv("SFM_PRECOMPUTE", "1" if mode == "verdict" else "0")
    yield mode


def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is not None:
        metafunc.parametrize("fusion_path", ["chain", "pair", "verdict"], indirect=True)
