"""div_pair (shared-reciprocal pair of divisions in fuse_chain_kernel) against plain IEEE division on
the GPU, bit for bit, over a billion operand triples including every combination of special values
(tests/hip/div_probe.hip, built by build.sh)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_div_pair_matches_ieee_division():
    exe = os.path.join(ROOT, "build", "div_probe")
    assert os.path.exists(exe), "run ./build.sh"
    r = subprocess.run([exe, "4096"], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    words = r.stdout.split()
    assert words[0] == "mismatches" and int(words[1]) == 0
    assert int(words[3]) > 0 and int(words[5]) == 1024 * 256 * 4096     # some triples flagged, all counted
