"""Synthetic RGB-D keyframe sequences for the dense-fusion path.

A static scene (box room + two pillars) is ray-cast from a smooth trajectory with torch, so
the same code fills host arrays for the parity tests and HBM-resident tensors for bench.py.
Depth is the z-depth along the optical axis in raw units (`unit` raw units per metre,
1000 => millimetres like a TUM/RealSense sensor), rounded to uint16.  Colour is a hashed
texture of the hit point.  Neighbour lists are the temporal window +-1..+-window, ascending,
standing in for Map::get_neighbors(kf, 100) (reference src/map.cpp:152-160).

Nothing here is on the product's data path: it only fabricates inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

TUM_K = (525.0, 525.0, 319.5, 239.5)
HD_K = (913.848, 913.602, 642.941, 371.196)  # reference src/reconstruct.cpp:45 (commented set)


@dataclass
class Sequence:
    depth: torch.Tensor   # (N,H,W) int16 holding uint16 bit patterns
    bgr: torch.Tensor     # (N,H,W,3) uint8, B,G,R
    K: np.ndarray         # (N,4) fx,fy,cx,cy
    T_wk: np.ndarray      # (N,4,4) camera->world
    T_kw: np.ndarray      # (N,4,4) world->camera
    nbrs: list            # per frame: ascending list of neighbour frame indices
    first: int = 0        # make_sequence(only=...): depth[i - first] / bgr[i - first] hold frame i; K, poses, nbrs cover all

    @property
    def n_frames(self):
        return len(self.nbrs)

    def depth_u16(self, i) -> np.ndarray:
        return self.depth[i - self.first].cpu().numpy().view(np.uint16)

    def bgr_u8(self, i) -> np.ndarray:
        return self.bgr[i - self.first].cpu().numpy()

    def frame_dicts(self, lo=0, hi=None):
        """Host-side frame dicts (the oracle's input format)."""
        hi = self.n_frames if hi is None else hi
        out = []
        for i in range(lo, hi):
            out.append(dict(depth=self.depth_u16(i), bgr=self.bgr_u8(i), K=self.K[i],
                            T_wk=self.T_wk[i], T_kw=self.T_kw[i],
                            nbrs=[j - lo for j in self.nbrs[i] if lo <= j < hi]))
        return out


def _poses(n, unit, rng, yaw_step_deg, radius_m, phase, component):
    T_wk = np.zeros((n, 4, 4))
    T_kw = np.zeros((n, 4, 4))
    jit = rng.normal(0.0, 1.0, size=(n, 4))
    for i in range(n):
        phi = phase + math.radians(yaw_step_deg) * i
        pitch = math.radians(6.0) * math.sin(0.05 * i + component) + math.radians(0.2) * jit[i, 0]
        r = (radius_m + 0.05 * math.sin(0.11 * i)) * unit
        C = np.array([r * math.cos(phi) + 0.002 * unit * jit[i, 1],
                      (0.10 * math.sin(0.07 * i + 2.0 * component)) * unit + 0.002 * unit * jit[i, 2],
                      r * math.sin(phi) + 0.002 * unit * jit[i, 3]])
        fwd = np.array([math.cos(phi) * math.cos(pitch), math.sin(pitch), math.sin(phi) * math.cos(pitch)])
        down0 = np.array([0.0, 1.0, 0.0])
        right = np.cross(down0, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=1)
        T_wk[i, :3, :3] = R
        T_wk[i, :3, 3] = C
        T_wk[i, 3, 3] = 1.0
        T_kw[i, :3, :3] = R.T
        T_kw[i, :3, 3] = -R.T @ C
        T_kw[i, 3, 3] = 1.0
    return T_wk, T_kw


def _corridor_poses(n, unit, rng, step_m, component):
    """A long straight walk: the camera moves along +x, `step_m` per keyframe, looking at the side wall (+z) with a
    slowly swinging yaw and pitch.  Every keyframe sees new wall, so the number of occupied voxels grows linearly
    with the number of keyframes (BASELINE configs[4]: capacity / collision stress)."""
    T_wk = np.zeros((n, 4, 4))
    T_kw = np.zeros((n, 4, 4))
    jit = rng.normal(0.0, 1.0, size=(n, 4))
    for i in range(n):
        yaw = math.radians(12.0) * math.sin(0.013 * i + component) + math.radians(0.2) * jit[i, 0]
        pitch = math.radians(5.0) * math.sin(0.031 * i) + math.radians(0.2) * jit[i, 1]
        C = np.array([(i - 0.5 * (n - 1)) * step_m * unit + 0.002 * unit * jit[i, 2],
                      0.10 * math.sin(0.07 * i) * unit,
                      0.05 * math.sin(0.011 * i) * unit + 0.002 * unit * jit[i, 3]])
        fwd = np.array([math.sin(yaw) * math.cos(pitch), math.sin(pitch), math.cos(yaw) * math.cos(pitch)])
        down0 = np.array([0.0, 1.0, 0.0])
        right = np.cross(down0, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd], axis=1)
        T_wk[i, :3, :3] = R
        T_wk[i, :3, 3] = C
        T_wk[i, 3, 3] = 1.0
        T_kw[i, :3, :3] = R.T
        T_kw[i, :3, 3] = -R.T @ C
        T_kw[i, 3, 3] = 1.0
    return T_wk, T_kw


def make_sequence(n_frames, W, H, K=TUM_K, *, unit=1000.0, seed=20240315, window=4,
                  yaw_step_deg=1.5, radius_m=0.8, holes=0.0, device="cpu", chunk=16,
                  components=1, room_m=(3.0, 1.5, 3.5), only=None, corridor_step_m=0.0) -> Sequence:
    """`components` > 1 lays out that many independent trajectories through the same room,
    one after the other in frame order, with no neighbour edges between them (the unit the
    multi-GPU path shards by).  corridor_step_m > 0: a straight walk along a corridor instead of the
    circle in the room (_corridor_poses).  only=(lo, hi): poses and neighbour lists of the whole sequence, images of frames
    [lo, hi) only -- a rank of a sequence split renders just the keyframes its ctx holds (same pixels as the full
    sequence would have; needs holes == 0, whose rectangles are drawn frame after frame)."""
    rng = np.random.default_rng(seed)
    dev = torch.device(device)
    fx, fy, cx, cy = K
    per = [n_frames // components + (1 if c < n_frames % components else 0) for c in range(components)]
    Ts_wk, Ts_kw, nbrs = [], [], []
    base = 0
    if corridor_step_m > 0:  # the room becomes a corridor along x that holds the whole walk (pillars: near its middle)
        room_m = (0.5 * n_frames * corridor_step_m + 3.0, room_m[1], 1.9)
    for c, n_c in enumerate(per):
        if corridor_step_m > 0:
            a, b = _corridor_poses(n_c, unit, rng, corridor_step_m, c)
        else:
            a, b = _poses(n_c, unit, rng, yaw_step_deg, radius_m * (1.0 + 0.07 * c),
                          phase=2.0 * math.pi * c / max(components, 1) + 0.013 * c, component=c)
        Ts_wk.append(a)
        Ts_kw.append(b)
        for i in range(n_c):
            nbrs.append([base + j for j in range(i - window, i + window + 1)
                         if j != i and 0 <= j < n_c])
        base += n_c
    T_wk = np.concatenate(Ts_wk) if Ts_wk else np.zeros((0, 4, 4))
    T_kw = np.concatenate(Ts_kw) if Ts_kw else np.zeros((0, 4, 4))

    B = torch.tensor([room_m[0] * unit, room_m[1] * unit, room_m[2] * unit], dtype=torch.float64, device=dev)
    pillars = torch.tensor([
        [[1.6, -1.5, 0.9], [2.0, 1.5, 1.3]],
        [[-2.2, -0.2, -1.9], [-1.7, 1.5, -1.4]],
    ], dtype=torch.float64, device=dev) * unit
    u = torch.arange(W, dtype=torch.float64, device=dev)
    v = torch.arange(H, dtype=torch.float64, device=dev)
    dirs_cam = torch.stack([((u - cx) / fx)[None, :].expand(H, W),
                            ((v - cy) / fy)[:, None].expand(H, W),
                            torch.ones(H, W, dtype=torch.float64, device=dev)], dim=-1)  # (H,W,3)

    r_lo, r_hi = (0, n_frames) if only is None else (max(0, only[0]), min(n_frames, only[1]))
    if only is not None and holes > 0:
        raise ValueError("only= needs holes == 0")
    depth = torch.empty((max(r_hi - r_lo, 0), H, W), dtype=torch.int16, device=dev)
    bgr = torch.empty((max(r_hi - r_lo, 0), H, W, 3), dtype=torch.uint8, device=dev)
    inf = torch.tensor(float("inf"), dtype=torch.float64, device=dev)
    hole_rng = np.random.default_rng(seed + 1)
    for lo in range(r_lo, r_hi, chunk):
        hi = min(r_hi, lo + chunk)
        R = torch.tensor(T_wk[lo:hi, :3, :3], device=dev)           # (n,3,3)
        Cc = torch.tensor(T_wk[lo:hi, :3, 3], device=dev)           # (n,3)
        d = torch.einsum("nij,hwj->nhwi", R, dirs_cam)              # (n,H,W,3)
        o = Cc[:, None, None, :]
        safe = torch.where(d == 0, torch.full_like(d, 1e-300), d)
        # room: exit distance of an interior ray
        t_axis = (torch.sign(d) * B - o) / safe
        t_axis = torch.where(d == 0, inf, t_axis)
        t = t_axis.min(dim=-1).values
        # pillars: slab test
        for pb in pillars:
            t1 = (pb[0] - o) / safe
            t2 = (pb[1] - o) / safe
            tmin = torch.minimum(t1, t2).max(dim=-1).values
            tmax = torch.maximum(t1, t2).min(dim=-1).values
            hit = (tmax >= tmin) & (tmin > 0)
            t = torch.where(hit & (tmin < t), tmin, t)
        zi = torch.round(t).clamp(0, 65535).to(torch.int32)
        P = o + d * t[..., None]
        cell = torch.floor(P / (0.064 * unit)).to(torch.int64)
        h = (cell[..., 0] * 73856093) ^ (cell[..., 1] * 19349663) ^ (cell[..., 2] * 83492791)
        h = h ^ (h >> 13)
        col = torch.stack([h & 255, (h >> 8) & 255, (h >> 16) & 255], dim=-1).to(torch.uint8)
        if holes > 0:
            for k in range(hi - lo):
                n_rect = max(1, int(round(holes * 20)))
                for _ in range(n_rect):
                    hw = max(1, int(W * math.sqrt(holes / n_rect)))
                    hh = max(1, int(H * math.sqrt(holes / n_rect)))
                    x0 = int(hole_rng.integers(0, max(1, W - hw + 1)))
                    y0 = int(hole_rng.integers(0, max(1, H - hh + 1)))
                    zi[k, y0:y0 + hh, x0:x0 + hw] = 0
        depth[lo - r_lo:hi - r_lo] = torch.where(zi >= 32768, zi - 65536, zi).to(torch.int16)
        bgr[lo - r_lo:hi - r_lo] = col
    Ks = np.tile(np.asarray(K, dtype=np.float64), (n_frames, 1))
    return Sequence(depth, bgr, Ks, T_wk, T_kw, nbrs, r_lo)
