"""Small hand-built scenes shared by the CPU and GPU tests."""
import numpy as np


def pose(C=(0.0, 0.0, 0.0), yaw_deg=0.0):
    """Camera looking along +z, rotated about +y by yaw; returns (T_wk, T_kw)."""
    a = np.deg2rad(yaw_deg)
    R = np.array([[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = C
    Ti = np.eye(4)
    Ti[:3, :3] = R.T
    Ti[:3, 3] = -R.T @ np.asarray(C, dtype=np.float64)
    return T, Ti


def frame(depth, bgr=None, K=(500.0, 500.0, 320.0, 240.0), C=(0, 0, 0), yaw=0.0, nbrs=()):
    depth = np.asarray(depth, dtype=np.uint16)
    H, W = depth.shape
    if bgr is None:
        yy, xx = np.mgrid[0:H, 0:W]
        bgr = np.stack([(xx * 7 + 1) % 256, (yy * 5 + 2) % 256, (xx + yy + 3) % 256], -1).astype(np.uint8)
    T_wk, T_kw = pose(C, yaw)
    return dict(depth=depth, bgr=np.asarray(bgr, dtype=np.uint8), K=np.asarray(K, dtype=np.float64),
                T_wk=T_wk, T_kw=T_kw, nbrs=list(nbrs))


def plane_pair(W=48, H=36, d=40, shift=3.0, K=(60.0, 60.0, 23.5, 17.5)):
    """KAT-7: two frames facing a fronto-parallel plane at depth d (raw units), the second
    translated by `shift` along +x; mutual neighbours."""
    f0 = frame(np.full((H, W), d), K=K, C=(0, 0, 0), nbrs=[1])
    f1 = frame(np.full((H, W), d), K=K, C=(shift, 0, 0), nbrs=[0])
    return [f0, f1]
