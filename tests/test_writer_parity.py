"""Writer parity: the reference writes the cloud with `std::ostream << double`
(src/dense_reconstruction.cpp:61-64).  tests/cpp/writer_probe.cpp exposes exactly that formatter
(host-only C++, built with g++), and these tests require the Python writer, the fast C++ writer
and the PLY writers to agree with it byte for byte."""
import io
import os
import struct
import subprocess

import numpy as np
import pytest

from sfm_amd import fusion

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "build", "writer_probe")


@pytest.fixture(scope="module")
def probe():
    src = os.path.join(ROOT, "tests", "cpp", "writer_probe.cpp")
    hdr = os.path.join(ROOT, "sfm_amd", "cpp", "cloud_writer.hpp")
    if not os.path.exists(PROBE) or os.path.getmtime(PROBE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(PROBE), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-pthread", src, "-o", PROBE], check=True)
    return PROBE


def tricky_doubles():
    rng = np.random.default_rng(7)
    v = [0.0, -0.0, 1.0, -1.0, 0.1 + 0.2, 1e-5, 9.9999995e-5, 1e-4, 0.00012345649, 0.00012345651,
         999999.0, 999999.4, 999999.5, 1e6, 1234565.0, 123456.5, 12345.65, 0.5, 2.5e-310, 5e-324,
         1.7976931348623157e308, float("inf"), float("-inf"), 1e15, 1e16, 1e17, 100000.0, 100000.5, 4096.125]
    out = [struct.unpack("<Q", struct.pack("<d", x))[0] for x in v]
    out += [0xFFF8000000000000, 0x7FF8000000000000]                     # x86 indefinite NaN, positive quiet NaN
    out += [int(x) for x in rng.integers(0, 2 ** 63, size=4000, dtype=np.uint64)]          # any bit pattern
    out += [struct.unpack("<Q", struct.pack("<d", float(x)))[0] for x in rng.normal(0, 3000, size=4000)]   # cloud-like
    out += [struct.unpack("<Q", struct.pack("<d", float(x)))[0]
            for x in (rng.integers(-10 ** 7, 10 ** 7, size=2000) / 10 ** rng.integers(0, 9, size=2000))]  # short decimals
    return out


def test_format_double_is_ostream_default_formatting(probe):
    bits = tricky_doubles()
    res = subprocess.run([probe, "fmt"], input="\n".join("%016x" % b for b in bits), capture_output=True, text=True,
                         check=True)
    want = res.stdout.split("\n")[:-1]
    assert len(want) == len(bits)
    got = [fusion.format_double(struct.unpack("<d", struct.pack("<Q", b))[0]) for b in bits]
    bad = [(hex(b), g, w) for b, g, w in zip(bits, got, want) if g != w]
    assert not bad, bad[:10]


def _cloud(n=3000):
    rng = np.random.default_rng(11)
    xyz = rng.normal(0, 2500, size=(n, 3))
    xyz[5] = [np.nan, 1.0, -0.0]
    xyz[5, 0] = np.frombuffer(struct.pack("<Q", 0xFFF8000000000000), dtype=np.float64)[0]
    xyz[6] = [1e6, 999999.5, 1e-5]
    bgr = rng.integers(0, 256, size=(n, 3)).astype(np.int32)
    return xyz, bgr


def _dump(tmp_path, xyz, bgr):
    p = tmp_path / "cloud.bin"
    with open(p, "wb") as f:
        f.write(struct.pack("<Q", len(xyz)))
        f.write(np.ascontiguousarray(xyz, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(bgr, dtype="<i4").tobytes())
    return str(p)


def test_text_writers_agree_byte_for_byte(probe, tmp_path):
    xyz, bgr = _cloud()
    path = _dump(tmp_path, xyz, bgr)
    ref = subprocess.run([probe, "text", path], capture_output=True, check=True).stdout     # os << double ...
    fast = subprocess.run([probe, "fast", path], capture_output=True, check=True).stdout
    assert fast == ref
    buf = io.StringIO()
    fusion.write_cloud(buf, xyz, bgr)
    assert buf.getvalue().encode("ascii") == ref
    assert ref.split(b"\n")[5].startswith(b"-nan 1 -0 ")                                   # glibc prints the NaN's sign
    csv = subprocess.run([probe, "csv", path], capture_output=True, check=True).stdout
    buf = io.StringIO()
    fusion.write_sparse_csv(buf, xyz, bgr)
    assert buf.getvalue().encode("ascii") == csv


def test_threaded_fast_writer_keeps_the_order(probe, tmp_path):
    xyz, bgr = _cloud(200_000)                       # several 64-Ki blocks: formatted in parallel, written in order
    path = _dump(tmp_path, xyz, bgr)
    ref = subprocess.run([probe, "text", path], capture_output=True, check=True).stdout
    fast = subprocess.run([probe, "fast", path], capture_output=True, check=True).stdout
    assert fast == ref and ref.count(b"\n") == 200_000


def test_ply_writers_agree_and_round_trip(probe, tmp_path):
    xyz, bgr = _cloud(1000)
    path = _dump(tmp_path, xyz, bgr)
    cpp = subprocess.run([probe, "ply", path], capture_output=True, check=True).stdout
    buf = io.BytesIO()
    fusion.write_ply(buf, xyz, bgr)
    assert buf.getvalue() == cpp
    x2, c2 = fusion.read_ply(io.BytesIO(cpp))
    assert np.array_equal(x2.view(np.uint64), xyz.view(np.uint64)) and np.array_equal(c2, bgr)


def test_empty_cloud_writes_nothing(probe, tmp_path):
    path = _dump(tmp_path, np.zeros((0, 3)), np.zeros((0, 3), np.int32))
    assert subprocess.run([probe, "text", path], capture_output=True, check=True).stdout == b""
    buf = io.StringIO()
    fusion.write_cloud(buf, np.zeros((0, 3)), np.zeros((0, 3), np.int32))
    assert buf.getvalue() == ""
