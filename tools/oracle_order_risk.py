#!/usr/bin/env python3
"""How much do the unverifiable Eigen evaluation orders matter?  Runs the CPU oracle with each alternative order
(oracle/fusion_oracle.c, orc_set_variant) on the golden scenes and on the first 24 keyframes of the C2 bench
workload, and reports per variant: voxel keys that appear or disappear, common voxels whose winner (tag) changes,
accepted-pixel count change, and the coordinate differences of the common winners in ulps.
Writes profiles/r02_oracle_order_risk.json and prints the table DESIGN.md section 2 quotes.  CPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle import oracle
from sfm_amd import synth
import helpers

VARIANTS = [(1, "trace (d0+d1)+d2"), (2, "row m0x+(m1y+m2z)"), (4, "3x3 coeff (a0b0+a1b1)+a2b2"),
            (7, "all three orders"), (8, "FMA-contracted"), (15, "all orders + FMA")]


def ulps(a, b):
    """|a - b| in units of the last place of a (finite values); NaN pairs count 0."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both = np.isfinite(a) & np.isfinite(b)
    ia = a.view(np.int64).astype(np.int64)
    ib = b.view(np.int64).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**63) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**63) - ib, ib)
    d = np.abs(ia - ib)
    return np.where(both, d, 0)


def compare(base, c):
    kb = {tuple(k): i for i, k in enumerate(base.key.tolist())}
    kc = {tuple(k): i for i, k in enumerate(c.key.tolist())}
    common = [k for k in kb if k in kc]
    ib = np.array([kb[k] for k in common], dtype=np.int64)
    ic = np.array([kc[k] for k in common], dtype=np.int64)
    same_tag = base.tag[ib] == c.tag[ic]
    u = ulps(base.xyz[ib][same_tag], c.xyz[ic][same_tag]) if len(common) else np.zeros((0, 3))
    return dict(points=int(base.key.shape[0]), keys_lost=len(kb) - len(common), keys_new=len(kc) - len(common),
                winners_changed=int((~same_tag).sum()), accepted_delta=int(c.counters["n_acc"] - base.counters["n_acc"]),
                xyz_max_ulp=int(u.max()) if u.size else 0, xyz_mean_ulp=float(u.mean()) if u.size else 0.0,
                xyz_changed_frac=float((u > 0).mean()) if u.size else 0.0)


def scenes(full=True):
    for name in helpers.golden_names():
        frames, params, _ = helpers.load_golden(name)
        yield name, frames, oracle.default_params(**{k: v for k, v in params.items()})
    seq = synth.make_sequence(8, 160, 120, K=(131.25, 131.25, 79.75, 59.75), window=3)
    yield "room_mm_8x160x120", seq.frame_dicts(), oracle.default_params(voxel_size=10.0, T_dist=500.0)
    if full:
        seq = synth.make_sequence(24, 640, 480)
        yield "c2_prefix_24x640x480", seq.frame_dicts(), oracle.default_params(voxel_size=10.0, T_dist=500.0)


def main():
    out = {}
    for name, frames, p in scenes():
        base = oracle.fuse(frames, p, n_threads=8)
        out[name] = {}
        for flags, label in VARIANTS:
            with oracle.variant(flags):
                c = oracle.fuse(frames, p, n_threads=8)
            out[name][label] = compare(base, c)
    path = os.path.join(ROOT, "profiles", "r02_oracle_order_risk.json")
    json.dump(out, open(path, "w"), indent=1)
    print("| scene | variant | points | keys lost / new | winners changed | accepted delta | xyz max ulp | xyz changed |")
    print("|---|---|---|---|---|---|---|---|")
    for name, rows in out.items():
        for label, r in rows.items():
            print("| %s | %s | %d | %d / %d | %d | %+d | %d | %.1f %% |" % (
                name, label, r["points"], r["keys_lost"], r["keys_new"], r["winners_changed"], r["accepted_delta"],
                r["xyz_max_ulp"], 100 * r["xyz_changed_frac"]))


if __name__ == "__main__":
    main()
