"""RecordingReader: the reference's reader of recorded logs,
src/recording.cpp:12-38 -- numbered `color_N.png` / `depth_N.png`, colour as CV_8UC3 in B,G,R order
(cv::imread IMREAD_COLOR), depth as CV_16UC1 (IMREAD_ANYDEPTH).  OpenCV is not in this image; PNG
decoding is done by Pillow here and by a small decoder of its own in the C++ twin
(sfm_amd/cpp/recording.hpp: zlib inflate + the five row filters).  tests/test_recording_reader.py
checks both against PNGs written filter type by filter type by a test-local encoder.
"""
from __future__ import annotations

import os

import numpy as np


def read_png_bgr(path):
    """cv::imread(path, IMREAD_COLOR): (H,W,3) uint8 in B,G,R order, or None."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            if im.format != "PNG" or im.mode in ("I;16", "I;16B", "I"):
                return None
            rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def read_png_depth16(path):
    """cv::imread(path, IMREAD_ANYDEPTH) of a 16-bit greyscale PNG: (H,W) uint16, or None."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            if im.format != "PNG" or im.mode not in ("I;16", "I;16B", "I"):
                return None
            a = np.asarray(im)
    except (OSError, ValueError):
        return None
    return np.ascontiguousarray(a.astype(np.uint16))


class RecordingReader:
    """sfm::utils::RecordingReader: sequential {colour, depth} frames of a recording directory."""

    def __init__(self, directory: str, start_frame: int = 0):
        self.directory = directory
        self.frame_counter = int(start_frame)

    def read_frames(self):
        """(colour BGR uint8, depth uint16) of the next frame, or None when a file is missing or
        unreadable (:19-33); the counter only advances on success (:35)."""
        c = os.path.join(self.directory, f"color_{self.frame_counter}.png")
        d = os.path.join(self.directory, f"depth_{self.frame_counter}.png")
        if not (os.path.exists(c) and os.path.exists(d)):
            return None
        color, depth = read_png_bgr(c), read_png_depth16(d)
        if color is None or depth is None:
            return None
        self.frame_counter += 1
        return color, depth
