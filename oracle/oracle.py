"""ctypes wrapper over oracle/libfusion_oracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under sfm_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfusion_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("base_line", C.c_double),
        ("T_cov", C.c_double),
        ("voxel_size", C.c_double),
        ("T_dist", C.c_double),
        ("min_dist_filter", C.c_double),
        ("dir_dot", C.c_double),
        ("S", C.c_double * 3),
        ("depth_div", C.c_double),
        ("close_frac", C.c_double),
    ]


class OrcFrame(C.Structure):
    _fields_ = [
        ("W", C.c_int32),
        ("H", C.c_int32),
        ("depth", C.c_void_p),
        ("depth_stride", C.c_uint64),
        ("bgr", C.c_void_p),
        ("bgr_stride", C.c_uint64),
        ("K", C.c_double * 4),
        ("T_wk", C.c_double * 16),
        ("T_kw", C.c_double * 16),
        ("nbrs", C.c_void_p),
        ("n_nbrs", C.c_uint32),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("n_points", C.c_uint64),
        ("xyz", C.POINTER(C.c_double)),
        ("bgr", C.POINTER(C.c_int32)),
        ("key", C.POINTER(C.c_uint64)),
        ("tag", C.POINTER(C.c_uint64)),
        ("frame_before", C.POINTER(C.c_uint64)),
        ("frame_after", C.POINTER(C.c_uint64)),
        ("global_before", C.c_uint64),
        ("global_after", C.c_uint64),
        ("n_px", C.c_uint64),
        ("n_gather", C.c_uint64),
        ("n_acc", C.c_uint64),
        ("n_hit", C.c_uint64),
        ("n_vox", C.c_uint64),
        ("n_live", C.c_uint64),
        ("n_rej_cov", C.c_uint64),
        ("n_rej_noobs", C.c_uint64),
        ("n_rej_far", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc, strict IEEE)."""
    src = os.path.join(_HERE, "fusion_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libfusion_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_default_params.argtypes = [C.POINTER(OrcParams)]
        L.orc_cov_trace.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_cov_trace.restype = C.c_double
        L.orc_deproject.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.c_double,
                                    C.POINTER(C.c_double)]
        L.orc_project.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_point_in_frame.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int,
                                         C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_point_in_frame.restype = C.c_int
        L.orc_voxel_key.argtypes = [C.c_double, C.c_double]
        L.orc_voxel_key.restype = C.c_uint64
        L.orc_neighbor_kept.argtypes = [C.POINTER(OrcParams), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_neighbor_kept.restype = C.c_int
        L.orc_fuse.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcFrame), C.c_uint32, C.c_int,
                               C.POINTER(OrcResult)]
        L.orc_fuse.restype = C.c_int
        L.orc_fuse_ex.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcFrame), C.c_uint32, C.c_int, C.c_uint32,
                                  C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(OrcResult)]
        L.orc_fuse_ex.restype = C.c_int
        L.orc_free_result.argtypes = [C.POINTER(OrcResult)]
        L.orc_set_variant.argtypes = [C.c_int]
        L.orc_set_variant.restype = None
        L.orc_get_variant.restype = C.c_int
        _lib = L
    return _lib


def default_params(**over) -> OrcParams:
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    for k, v in over.items():
        if k == "S":
            p.S[0], p.S[1], p.S[2] = v
        else:
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
    return p


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def cov_trace(u, v, d, f, params=None) -> float:
    p = params or default_params()
    return lib().orc_cov_trace(C.byref(p), int(u), int(v), float(d), float(f))


def deproject(K, px, py, depth):
    K = np.ascontiguousarray(K, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_deproject(_dp(K), int(px), int(py), float(depth), _dp(out))
    return out


def point_in_frame(K, T_kw, W, H, pw):
    K = np.ascontiguousarray(K, dtype=np.float64)
    T = np.asfortranarray(np.asarray(T_kw, dtype=np.float64)).ravel(order="F").copy()
    pw = np.ascontiguousarray(pw, dtype=np.float64)
    x, y = C.c_int(0), C.c_int(0)
    ok = lib().orc_point_in_frame(_dp(K), _dp(T), int(W), int(H), _dp(pw), C.byref(x), C.byref(y))
    return (x.value, y.value) if ok else None


def voxel_key(coord, voxel_size) -> int:
    return int(lib().orc_voxel_key(float(coord), float(voxel_size)))


def neighbor_kept(T_wk_f, T_wk_lf, params=None) -> bool:
    p = params or default_params()
    a = np.asarray(T_wk_f, dtype=np.float64).ravel(order="F").copy()
    b = np.asarray(T_wk_lf, dtype=np.float64).ravel(order="F").copy()
    return bool(lib().orc_neighbor_kept(C.byref(p), _dp(a), _dp(b)))


@dataclass
class Cloud:
    """Fused cloud, sorted by voxel key (k0, k1, k2 unsigned lexicographic)."""
    xyz: np.ndarray            # (N,3) float64
    bgr: np.ndarray            # (N,3) int32, B,G,R
    key: np.ndarray            # (N,3) uint64
    tag: np.ndarray            # (N,)  uint64
    frame_before: np.ndarray   # (F,) uint64
    frame_after: np.ndarray    # (F,) uint64
    global_before: int = 0
    global_after: int = 0
    counters: dict = field(default_factory=dict)


def sort_by_key(xyz, bgr, key, tag):
    order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
    return xyz[order], bgr[order], key[order], tag[order]


def fuse(frames, params=None, n_threads: int = 1, active=None, mask_seed=None, want_masks=False):
    """frames: list of dicts with depth (H,W) u16, bgr (H,W,3) u8, K (4,), T_wk (4,4),
    T_kw (4,4), nbrs (list of frame indices, the count-filtered Map::get_neighbors list).

    active=(first, end), mask_seed={frame: (H,W) u8}, want_masks: the block-wise variant orc_fuse_ex (checks the
    multi-GPU split of one sequence); with want_masks the result is (Cloud, [mask per frame])."""
    p = params or default_params()
    n = len(frames)
    arr = (OrcFrame * max(n, 1))()
    keep = []
    for i, f in enumerate(frames):
        depth = np.ascontiguousarray(f["depth"], dtype=np.uint16)
        bgr = np.ascontiguousarray(f["bgr"], dtype=np.uint8)
        H, W = depth.shape
        assert bgr.shape == (H, W, 3)
        nb = np.ascontiguousarray(f.get("nbrs", []), dtype=np.uint32)
        keep += [depth, bgr, nb]
        a = arr[i]
        a.W, a.H = W, H
        a.depth = depth.ctypes.data
        a.depth_stride = depth.strides[0] if H else 2 * W
        a.bgr = bgr.ctypes.data
        a.bgr_stride = bgr.strides[0] if H else 3 * W
        K = np.asarray(f["K"], dtype=np.float64)
        for j in range(4):
            a.K[j] = K[j]
        Twk = np.asarray(f["T_wk"], dtype=np.float64).ravel(order="F")
        Tkw = np.asarray(f["T_kw"], dtype=np.float64).ravel(order="F")
        for j in range(16):
            a.T_wk[j] = Twk[j]
            a.T_kw[j] = Tkw[j]
        a.nbrs = nb.ctypes.data if nb.size else None
        a.n_nbrs = nb.size
    res = OrcResult()
    first, end = active if active is not None else (0, n)
    seeds = (C.c_void_p * max(n, 1))()
    outs = (C.c_void_p * max(n, 1))()
    masks_out = []
    for i, f in enumerate(frames):
        H, W = np.asarray(f["depth"]).shape
        if mask_seed is not None and i in mask_seed:
            m = np.ascontiguousarray(mask_seed[i], dtype=np.uint8).reshape(H, W)
            keep.append(m)
            seeds[i] = m.ctypes.data if m.size else None
        if want_masks:
            m = np.zeros((H, W), dtype=np.uint8)
            masks_out.append(m)
            outs[i] = m.ctypes.data if m.size else None
    rc = lib().orc_fuse_ex(C.byref(p), arr, n, int(n_threads), int(first), int(end), seeds, outs, C.byref(res))
    if rc != 0:
        raise ValueError(f"orc_fuse failed rc={rc} (bad neighbour index?)")
    try:
        N = res.n_points
        xyz = np.ctypeslib.as_array(res.xyz, shape=(max(N, 1) * 3,))[: 3 * N].reshape(N, 3).copy()
        bgr = np.ctypeslib.as_array(res.bgr, shape=(max(N, 1) * 3,))[: 3 * N].reshape(N, 3).copy()
        key = np.ctypeslib.as_array(res.key, shape=(max(N, 1) * 3,))[: 3 * N].reshape(N, 3).copy()
        tag = np.ctypeslib.as_array(res.tag, shape=(max(N, 1),))[:N].copy()
        fb = np.ctypeslib.as_array(res.frame_before, shape=(max(n, 1),))[:n].copy()
        fa = np.ctypeslib.as_array(res.frame_after, shape=(max(n, 1),))[:n].copy()
        xyz, bgr, key, tag = sort_by_key(xyz, bgr, key, tag)
        cloud = Cloud(xyz, bgr, key, tag, fb, fa, int(res.global_before), int(res.global_after),
                      dict(n_px=int(res.n_px), n_gather=int(res.n_gather), n_acc=int(res.n_acc),
                           n_hit=int(res.n_hit), n_vox=int(res.n_vox), n_live=int(res.n_live),
                           n_rej_cov=int(res.n_rej_cov), n_rej_noobs=int(res.n_rej_noobs),
                           n_rej_far=int(res.n_rej_far)))
        return (cloud, masks_out) if want_masks else cloud
    finally:
        lib().orc_free_result(C.byref(res))


class variant:
    """Context manager: run the oracle with alternative Eigen evaluation orders / FMA contraction (flags in
    fusion_oracle.c).  Only for measuring how much the unverifiable choices matter; flags 0 is the oracle."""

    def __init__(self, flags: int):
        self.flags = int(flags)

    def __enter__(self):
        self.prev = lib().orc_get_variant()
        lib().orc_set_variant(self.flags)
        return self

    def __exit__(self, *exc):
        lib().orc_set_variant(self.prev)
        return False


def algorithmic_bytes(c: dict) -> int:
    """SURVEY.md 8(d) / BASELINE.md: B_alg from the oracle's counters."""
    return (3 * c["n_px"] + 2 * c["n_gather"] + 3 * c["n_acc"] + 1 * c["n_hit"]
            + 32 * c["n_acc"] + 27 * c["n_vox"])
