"""RecordingReader: the C++ reader's own PNG decoder (zlib + row filters) and the
Python reader (Pillow) against PNGs written here with every row-filter type forced, for the pixel
formats cv::imread(IMREAD_COLOR / IMREAD_ANYDEPTH) accepts; the reference's stop-at-first-missing-
frame behaviour (src/recording.cpp:19-35)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from sfm_amd import recording

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "build", "recording_probe")


@pytest.fixture(scope="module")
def probe():
    src = os.path.join(ROOT, "tests", "cpp", "recording_probe.cpp")
    hdr = os.path.join(ROOT, "sfm_amd", "cpp", "recording.hpp")
    if not os.path.exists(PROBE) or os.path.getmtime(PROBE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(PROBE), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", src, "-lz", "-o", PROBE], check=True)
    return PROBE


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, samples, color_type, bit_depth, palette=None, filters=(0, 1, 2, 3, 4), idat_split=3):
    """Minimal PNG encoder.  samples: (H, W, C) integer array of sample values; rows cycle through
    `filters`, so every PNG filter type is exercised; the stream is cut into several IDAT chunks."""
    H, W, C = samples.shape
    if bit_depth == 16:
        rows = samples.astype(">u2").reshape(H, -1).view(np.uint8)
    else:
        rows = samples.astype(np.uint8).reshape(H, -1)
    bpp = C * bit_depth // 8
    raw = bytearray()
    prev = np.zeros(rows.shape[1], dtype=np.int64)
    for y in range(H):
        cur = rows[y].astype(np.int64)
        ft = filters[y % len(filters)]
        left = np.concatenate([np.zeros(bpp, np.int64), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int64), prev[:-bpp]])
        if ft == 0:
            pred = np.zeros_like(cur)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) // 2
        else:
            pred = np.array([_paeth(int(a), int(b), int(c)) for a, b, c in zip(left, prev, ul)], dtype=np.int64)
        raw.append(ft)
        raw += ((cur - pred) % 256).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    z = zlib.compress(bytes(raw), 6)
    cut = max(1, len(z) // idat_split)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, bit_depth, color_type, 0, 0, 0)))
        if palette is not None:
            f.write(chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes()))
        f.write(chunk(b"tEXt", b"Comment\x00written by tests/test_recording_reader.py"))
        for i in range(0, len(z), cut):
            f.write(chunk(b"IDAT", z[i:i + cut]))
        f.write(chunk(b"IEND", b""))


def run_probe(probe, directory, start=0):
    r = subprocess.run([probe, directory, str(start)], capture_output=True, check=True)
    buf, frames, pos = r.stdout, [], 0
    while pos < len(buf):
        W, H = struct.unpack_from("<II", buf, pos)
        pos += 8
        c = np.frombuffer(buf, np.uint8, W * H * 3, pos).reshape(H, W, 3)
        pos += W * H * 3
        W2, H2 = struct.unpack_from("<II", buf, pos)
        pos += 8
        d = np.frombuffer(buf, "<u2", W2 * H2, pos).reshape(H2, W2)
        pos += 2 * W2 * H2
        frames.append((c, d))
    return frames


def make_recording(directory, n=7, W=37, H=23):
    """Frames 0..n-1 in the pixel formats a recorder (or a converter) may have used."""
    rng = np.random.default_rng(5)
    truth = []
    for i in range(n):
        smooth = (np.add.outer(np.arange(H) * 3, np.arange(W) * 5)[:, :, None] + np.array([0, 40, 90])) % 256
        rgb = ((smooth + rng.integers(0, 8, size=(H, W, 3))) % 256).astype(np.uint8)
        depth = (1000 + 20 * np.add.outer(np.arange(H), np.arange(W)) + rng.integers(0, 50, size=(H, W))).astype(np.uint16)
        depth[rng.random((H, W)) < 0.05] = 0
        depth[0, 0] = 65535
        kind = i % 4
        if kind == 0:
            write_png(f"{directory}/color_{i}.png", rgb, 2, 8)
            want = rgb
        elif kind == 1:
            rgba = np.concatenate([rgb, rng.integers(0, 256, size=(H, W, 1))], axis=2)
            write_png(f"{directory}/color_{i}.png", rgba, 6, 8)
            want = rgb
        elif kind == 2:
            g = rgb[:, :, :1]
            write_png(f"{directory}/color_{i}.png", g, 0, 8)
            want = np.repeat(g, 3, axis=2)
        else:
            pal = rng.integers(0, 256, size=(256, 3)).astype(np.uint8)
            idx = rgb[:, :, :1]
            write_png(f"{directory}/color_{i}.png", idx, 3, 8, palette=pal)
            want = pal[idx[:, :, 0]]
        write_png(f"{directory}/depth_{i}.png", depth[:, :, None], 0, 16, filters=(4, 3, 1, 2, 0) if i % 2 else (0, 1, 2, 3, 4))
        truth.append((np.ascontiguousarray(want[:, :, ::-1]), depth))
    return truth


def test_cpp_and_python_readers_decode_every_filter_and_format(probe, tmp_path):
    d = str(tmp_path)
    truth = make_recording(d)
    frames = run_probe(probe, d)
    assert len(frames) == len(truth)
    rd = recording.RecordingReader(d)
    for (c, z), (tc, tz) in zip(frames, truth):
        assert np.array_equal(c, tc) and np.array_equal(z, tz)          # own decoder == what was encoded
        pc, pz = rd.read_frames()
        assert np.array_equal(pc, tc) and np.array_equal(pz, tz)        # Pillow == what was encoded
        assert pz.dtype == np.uint16 and pc.dtype == np.uint8
    assert rd.read_frames() is None and rd.frame_counter == len(truth)


def test_reading_stops_at_the_first_missing_or_unreadable_frame(probe, tmp_path):
    d = str(tmp_path)
    make_recording(d, n=6)
    os.remove(f"{d}/depth_4.png")
    assert len(run_probe(probe, d)) == 4                                  # frames 0..3, like the reference's loop
    assert len(run_probe(probe, d, start=2)) == 2                         # start_frame
    assert len(run_probe(probe, d, start=5)) == 1
    rd = recording.RecordingReader(d, 2)
    assert rd.read_frames() is not None and rd.read_frames() is not None and rd.read_frames() is None
    assert rd.frame_counter == 4                                          # not advanced by the failure
    # an 8-bit depth file, a truncated file, a non-PNG: "imread failed"
    write_png(f"{d}/depth_1.png", np.zeros((4, 4, 1), np.uint8), 0, 8)
    assert len(run_probe(probe, d)) == 1
    assert recording.RecordingReader(d, 1).read_frames() is None
    raw = open(f"{d}/color_0.png", "rb").read()
    open(f"{d}/color_0.png", "wb").write(raw[:len(raw) // 2])
    assert len(run_probe(probe, d)) == 0
    assert recording.RecordingReader(d).read_frames() is None
    open(f"{d}/color_0.png", "wb").write(b"not a png at all")
    assert len(run_probe(probe, d)) == 0
