#!/usr/bin/env python3
"""Generates the golden fixtures in this directory.

There is no reference build and the reference ships no test data, so the
expected outputs come from the CPU oracle (oracle/fusion_oracle.c); for every fixture small
enough, the independent Python restatement (tests/py_restatement.py) must agree bit for bit
before the file is written.  Inputs: synthetic (sfm_amd/synth.py, tests/scenes.py).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle  # noqa: E402
from sfm_amd import synth  # noqa: E402
import py_restatement as pr  # noqa: E402
import scenes  # noqa: E402


def save(name, frames, params, check_python=True):
    c = oracle.fuse(frames, oracle.default_params(**params))
    if check_python:
        r = pr.fuse(frames, **params)
        assert [list(x[0]) for x in r["rows"]] == c.key.tolist(), name
        assert [x[1] for x in r["rows"]] == c.tag.tolist(), name
        got = np.array([x[2] for x in r["rows"]], dtype=np.float64).reshape(-1, 3)
        assert np.array_equal(got.view(np.uint64), c.xyz.view(np.uint64)) or \
            np.array_equal(np.nan_to_num(got, nan=7.0), np.nan_to_num(c.xyz, nan=7.0)), name
    out = dict(n_frames=len(frames), param_names=np.array(sorted(params)),
               param_values=np.array([np.atleast_1d(np.asarray(params[k], dtype=np.float64)) for k in sorted(params)],
                                     dtype=object),
               xyz=c.xyz, bgr=c.bgr, key=c.key, tag=c.tag, frame_before=c.frame_before,
               frame_after=c.frame_after, global_before=c.global_before, global_after=c.global_after,
               counters=np.array([c.counters[k] for k in ("n_px", "n_gather", "n_acc", "n_hit", "n_vox")],
                                 dtype=np.uint64))
    for i, f in enumerate(frames):
        out[f"depth_{i}"] = f["depth"]
        out[f"bgr_{i}"] = f["bgr"]
        out[f"K_{i}"] = np.asarray(f["K"], dtype=np.float64)
        out[f"T_wk_{i}"] = np.asarray(f["T_wk"], dtype=np.float64)
        out[f"T_kw_{i}"] = np.asarray(f["T_kw"], dtype=np.float64)
        out[f"nbrs_{i}"] = np.asarray(f["nbrs"], dtype=np.uint32)
    out["param_values"] = np.array([float(np.atleast_1d(params[k])[0]) if np.ndim(params[k]) == 0 else np.nan
                                    for k in sorted(params)])
    if "S" in params:
        out["S"] = np.asarray(params["S"], dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(frames)} frames, {c.key.shape[0]} points, {os.path.getsize(path)} bytes")


def main():
    save("plane_pair_default", scenes.plane_pair(), {})
    fr = scenes.plane_pair()
    fr[1]["depth"] = fr[1]["depth"].copy()
    fr[1]["depth"][5:9, 10:20] = 0
    fr[0]["depth"] = fr[0]["depth"].copy()
    fr[0]["depth"][20:22, 30:33] = 0
    save("plane_pair_zero_depth_nan", fr, dict(T_dist=100.0, voxel_size=1.0))
    seq = synth.make_sequence(6, 64, 48, K=(131.25, 131.25, 31.5, 23.5), window=3)
    save("room_mm_6x64x48", seq.frame_dicts(), dict(voxel_size=10.0, T_dist=500.0))
    seq = synth.make_sequence(5, 61, 37, K=(131.25, 130.0, 30.1, 18.2), window=4, holes=0.05, seed=7)
    save("room_mm_holes_5x61x37", seq.frame_dicts(), dict(voxel_size=25.0, T_dist=4000.0, S=(0.07, 0.03, 9.0)))
    # reference-default constants: depth in units of ~5 cm so that T_dist=0.5 / voxel 0.05 bite
    seq = synth.make_sequence(6, 64, 48, K=(131.25, 131.25, 31.5, 23.5), window=2, unit=20.0, components=2)
    save("room_refdefault_2comp_6x64x48", seq.frame_dicts(), {})


if __name__ == "__main__":
    main()
