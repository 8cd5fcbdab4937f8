"""ctypes binding of libsfm_fusion.so (include/sfm_fusion.h) and a host-side mirror of the
reference's dense-reconstruction interface.

Reference interface mirrored here (paths relative to /root/reference/reconstruction):
  * `DenseReconstruction(sfm::Map&)` / `reconstruct(std::ostream&)`
    (include/reconstruction/dense_reconstruction.hpp:26-31, src/dense_reconstruction.cpp:29-65)
  * the containers it reads: `Map::covisibility_edge`, `Map::get_neighbors`
    (include/reconstruction/map.hpp:143, src/map.cpp:152-160) and the KeyFrame accessors
    `img`, `get_depth_image()`, `camera_calibration()`, `transform()`, `world_to_camera()`
    (src/keyframe.cpp:31-54).

There is NO CPU fallback: loading fails loudly if the HIP library has not been built, and
every compute call needs a GPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsfm_fusion.so")

SFM_OK = 0
SFM_E_INVALID, SFM_E_HIP, SFM_E_TABLE_FULL, SFM_E_KEY_RANGE = -1, -2, -3, -4
SFM_E_CAPACITY, SFM_E_UNSUPPORTED, SFM_E_NOMEM = -5, -6, -7
SFM_MEM_HOST, SFM_MEM_DEVICE = 0, 1
SFM_MAX_NEIGHBORS = 32

EXPORTS = [
    "sfm_fusion_default_params", "sfm_fusion_create", "sfm_fusion_destroy",
    "sfm_fusion_add_keyframe", "sfm_fusion_set_neighbors", "sfm_fusion_run",
    "sfm_fusion_num_points", "sfm_fusion_read_points", "sfm_fusion_get_stats",
    "sfm_fusion_get_frame_counts", "sfm_fusion_last_error", "sfm_fusion_version",
    "sfm_fusion_set_profiling", "sfm_fusion_set_global_order", "sfm_fusion_partition_counts",
    "sfm_fusion_export_records", "sfm_fusion_merge", "sfm_fusion_set_covisibility",
    "sfm_fusion_partition_counts_noalloc", "sfm_fusion_set_active_range", "sfm_fusion_import_masks",
    "sfm_fusion_clear_imported_masks", "sfm_fusion_export_masks", "sfm_fusion_precompute",
    "sfm_fusion_add_keyframe_se3", "sfm_fusion_se3_exp", "sfm_fusion_get_key_range", "sfm_fusion_set_key_range",
    "sfm_fusion_get_cloud_index_range", "sfm_fusion_run_chain", "sfm_fusion_finish",
]
SFM_RECORD_BYTES = 48


class FusionParams(C.Structure):
    """sfm_fusion_params; defaults = dense_reconstruction.hpp:44-56."""
    _fields_ = [
        ("base_line", C.c_double), ("T_cov", C.c_double), ("voxel_size", C.c_double),
        ("T_dist", C.c_double), ("min_dist_filter", C.c_double), ("dir_dot", C.c_double),
        ("S", C.c_double * 3), ("depth_div", C.c_double), ("close_frac", C.c_double),
        ("table_capacity", C.c_uint64), ("max_points", C.c_uint64),
    ]


class FusionStats(C.Structure):
    _fields_ = [
        ("n_frames", C.c_uint64), ("global_before", C.c_uint64), ("global_after", C.c_uint64),
        ("n_px", C.c_uint64), ("n_gather", C.c_uint64), ("n_acc", C.c_uint64),
        ("n_hit", C.c_uint64), ("n_vox", C.c_uint64), ("table_capacity", C.c_uint64),
        ("table_used", C.c_uint64), ("max_probe", C.c_uint64),
        ("last_run_ms", C.c_double), ("frame_kernels_ms", C.c_double),
        ("fuse_kernel_launches", C.c_uint64), ("fuse_kernel_sum_ms", C.c_double),
        ("scan_kernel_sum_ms", C.c_double), ("fused_chain", C.c_uint64), ("precomputed", C.c_uint64),
        ("precompute_ms", C.c_double), ("chain_fallbacks", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class FusionError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sfm_fusion error {code}: {msg}")
        self.code = code


_lib = None


def load_library():
    """Loads the HIP library; raises if it was not built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SFM_FUSION_LIB", LIB_PATH)  # A/B builds of the same HIP library
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run ./build.sh (hipcc --offload-arch=gfx950). "
                          "sfm_amd has no CPU fallback.")
    L = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    dp = C.POINTER(C.c_double)
    L.sfm_fusion_default_params.argtypes = [C.POINTER(FusionParams)]
    L.sfm_fusion_default_params.restype = None
    L.sfm_fusion_create.argtypes = [C.POINTER(FusionParams), C.c_int, vp, C.POINTER(vp)]
    L.sfm_fusion_destroy.argtypes = [vp]
    L.sfm_fusion_destroy.restype = None
    L.sfm_fusion_add_keyframe.argtypes = [vp, u32, i32, i32, vp, C.c_size_t, vp, C.c_size_t, dp, dp, dp, C.c_int]
    L.sfm_fusion_add_keyframe_se3.argtypes = [vp, u32, i32, i32, vp, C.c_size_t, vp, C.c_size_t, dp, dp, C.c_int]
    L.sfm_fusion_se3_exp.argtypes = [dp, dp, dp]
    L.sfm_fusion_se3_exp.restype = None
    L.sfm_fusion_set_neighbors.argtypes = [vp, u32, C.POINTER(u32), u32]
    L.sfm_fusion_set_covisibility.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), u64, u32, u32]
    L.sfm_fusion_run.argtypes = [vp]
    L.sfm_fusion_run_chain.argtypes = [vp]
    L.sfm_fusion_finish.argtypes = [vp]
    L.sfm_fusion_num_points.argtypes = [vp, C.POINTER(u64)]
    L.sfm_fusion_read_points.argtypes = [vp, vp, vp, vp, vp]
    L.sfm_fusion_get_stats.argtypes = [vp, C.POINTER(FusionStats)]
    L.sfm_fusion_get_frame_counts.argtypes = [vp, vp, vp]
    L.sfm_fusion_set_profiling.argtypes = [vp, C.c_int]
    L.sfm_fusion_set_global_order.argtypes = [vp, C.POINTER(u32), u32]
    L.sfm_fusion_get_key_range.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sfm_fusion_set_key_range.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sfm_fusion_get_cloud_index_range.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sfm_fusion_partition_counts.argtypes = [vp, u32, C.POINTER(u64)]
    L.sfm_fusion_partition_counts_noalloc.argtypes = [vp, u32, C.POINTER(u64)]
    L.sfm_fusion_export_records.argtypes = [vp, vp, u64]
    L.sfm_fusion_merge.argtypes = [vp, vp, u64]
    L.sfm_fusion_set_active_range.argtypes = [vp, u32, u32]
    L.sfm_fusion_import_masks.argtypes = [vp, u32, u32, vp, u64]
    L.sfm_fusion_clear_imported_masks.argtypes = [vp]
    L.sfm_fusion_export_masks.argtypes = [vp, u32, u32, vp, u64]
    L.sfm_fusion_precompute.argtypes = [vp]
    L.sfm_fusion_last_error.argtypes = [vp]
    L.sfm_fusion_last_error.restype = C.c_char_p
    L.sfm_fusion_version.restype = C.c_char_p
    _lib = L
    return L


def default_params(**over) -> FusionParams:
    p = FusionParams()
    load_library().sfm_fusion_default_params(C.byref(p))
    for k, v in over.items():
        if k == "S":
            p.S[0], p.S[1], p.S[2] = v
        else:
            if not hasattr(p, k):
                raise KeyError(k)
            setattr(p, k, v)
    return p


def version() -> str:
    """The library's version string (the div-checked test build says so here)."""
    return load_library().sfm_fusion_version().decode()


def se3_exp(se3):
    """Sophus::SE3<double>::exp(se3).matrix() and its inverse as (4,4) arrays; se3 = (upsilon, omega)
    (the bundle-adjustment write-back, reference src/map.cpp:321-324)."""
    v = np.ascontiguousarray(se3, dtype=np.float64).reshape(6)
    T = np.zeros(16)
    Ti = np.zeros(16)
    dp = C.POINTER(C.c_double)
    load_library().sfm_fusion_se3_exp(v.ctypes.data_as(dp), T.ctypes.data_as(dp), Ti.ctypes.data_as(dp))
    return T.reshape(4, 4).T.copy(), Ti.reshape(4, 4).T.copy()


def _colmajor16(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(4, 4).T).ravel()


class FusionContext:
    """Thin owner of one sfm_fusion_ctx."""

    def __init__(self, params: FusionParams | None = None, device: int = 0, stream: int | None = None):
        self._L = load_library()
        self._h = C.c_void_p()
        p = params if params is not None else default_params()
        rc = self._L.sfm_fusion_create(C.byref(p), int(device), C.c_void_p(stream or 0), C.byref(self._h))
        if rc != SFM_OK:
            raise FusionError(rc, "sfm_fusion_create failed (is a HIP device present?)")
        self.n_frames = 0

    def _check(self, rc):
        if rc != SFM_OK:
            raise FusionError(rc, self._L.sfm_fusion_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.sfm_fusion_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_keyframe(self, depth, bgr, K, T_wk, T_kw):
        """depth (H,W) uint16, bgr (H,W,3) uint8 numpy arrays (rows may be strided)."""
        depth = np.asarray(depth)
        bgr = np.asarray(bgr)
        if depth.dtype != np.uint16 or bgr.dtype != np.uint8:
            raise TypeError("depth must be uint16 and bgr uint8")
        H, W = depth.shape
        if bgr.shape != (H, W, 3):
            raise ValueError("bgr shape must be (H, W, 3)")
        if W and depth.strides[1] != 2:
            depth = np.ascontiguousarray(depth)
        if W and (bgr.strides[2] != 1 or bgr.strides[1] != 3):
            bgr = np.ascontiguousarray(bgr)
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        a, b = _colmajor16(T_wk), _colmajor16(T_kw)
        dp = C.POINTER(C.c_double)
        self._check(self._L.sfm_fusion_add_keyframe(
            self._h, self.n_frames, W, H, depth.ctypes.data, depth.strides[0] if H else 2 * W,
            bgr.ctypes.data, bgr.strides[0] if H else 3 * W,
            Kc.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp), SFM_MEM_HOST))
        self.n_frames += 1
        return self.n_frames - 1

    def add_keyframe_se3(self, depth, bgr, K, se3_kw):
        """Host images, pose as the 6-vector of the bundle adjustment (T_kw = SE3::exp(se3_kw))."""
        depth = np.ascontiguousarray(depth, dtype=np.uint16)
        bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
        H, W = depth.shape
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        v = np.ascontiguousarray(se3_kw, dtype=np.float64).reshape(6)
        dp = C.POINTER(C.c_double)
        self._check(self._L.sfm_fusion_add_keyframe_se3(
            self._h, self.n_frames, W, H, depth.ctypes.data, 2 * W, bgr.ctypes.data, 3 * W,
            Kc.ctypes.data_as(dp), v.ctypes.data_as(dp), SFM_MEM_HOST))
        self.n_frames += 1
        return self.n_frames - 1

    def add_keyframe_device(self, depth_ptr, bgr_ptr, W, H, K, T_wk, T_kw, depth_stride=None, bgr_stride=None):
        """Device-resident images (raw device pointers, e.g. torch.Tensor.data_ptr())."""
        Kc = np.ascontiguousarray(K, dtype=np.float64)
        a, b = _colmajor16(T_wk), _colmajor16(T_kw)
        dp = C.POINTER(C.c_double)
        self._check(self._L.sfm_fusion_add_keyframe(
            self._h, self.n_frames, W, H, C.c_void_p(depth_ptr), depth_stride or 2 * W,
            C.c_void_p(bgr_ptr), bgr_stride or 3 * W,
            Kc.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp), SFM_MEM_DEVICE))
        self.n_frames += 1
        return self.n_frames - 1

    def set_neighbors(self, idx, nbrs):
        nb = np.ascontiguousarray(nbrs, dtype=np.uint32)
        self._check(self._L.sfm_fusion_set_neighbors(
            self._h, int(idx), nb.ctypes.data_as(C.POINTER(C.c_uint32)), nb.size))

    def set_covisibility(self, edges, select_min=100, neighbor_min=100):
        """edges: iterable of (a, b, count) over the order indices of every keyframe added; the
        library applies the selection (>= select_min) and neighbour (> neighbor_min) rules."""
        e = np.ascontiguousarray(list(edges), dtype=np.uint32).reshape(-1, 3)
        a, b, n = (np.ascontiguousarray(e[:, i]) for i in range(3))
        p = C.POINTER(C.c_uint32)
        self._check(self._L.sfm_fusion_set_covisibility(
            self._h, a.ctypes.data_as(p), b.ctypes.data_as(p), n.ctypes.data_as(p), len(e), int(select_min),
            int(neighbor_min)))

    def set_global_order(self, order):
        o = np.ascontiguousarray(order, dtype=np.uint32)
        self._check(self._L.sfm_fusion_set_global_order(
            self._h, o.ctypes.data_as(C.POINTER(C.c_uint32)), o.size))

    def key_range(self):
        """(lo, hi): per axis the voxel indices the exact 64-bit key packing in use holds."""
        lo = (C.c_int64 * 3)()
        hi = (C.c_int64 * 3)()
        self._check(self._L.sfm_fusion_get_key_range(self._h, lo, hi))
        return tuple(int(x) for x in lo), tuple(int(x) for x in hi)

    def set_key_range(self, lo, hi):
        a = (C.c_int64 * 3)(*[int(x) for x in lo])
        b = (C.c_int64 * 3)(*[int(x) for x in hi])
        self._check(self._L.sfm_fusion_set_key_range(self._h, a, b))

    def cloud_index_range(self):
        """(lo, hi): per axis the smallest and largest voxel index of the cloud at hand (the indefinite index aside);
        lo > hi on an axis without any.  What ranks exchange before they agree on a common key packing."""
        lo = (C.c_int64 * 3)()
        hi = (C.c_int64 * 3)()
        self._check(self._L.sfm_fusion_get_cloud_index_range(self._h, lo, hi))
        return tuple(int(x) for x in lo), tuple(int(x) for x in hi)

    def partition_counts(self, n_parts: int):
        counts = (C.c_uint64 * n_parts)()
        self._check(self._L.sfm_fusion_partition_counts(self._h, n_parts, counts))
        return [int(x) for x in counts]

    def partition_counts_noalloc(self, n_parts: int):
        """The same counts through the entry point that allocates nothing (static device scratch, <= 1024 parts)."""
        counts = (C.c_uint64 * max(n_parts, 1))()
        self._check(self._L.sfm_fusion_partition_counts_noalloc(self._h, n_parts, counts))
        return [int(x) for x in counts][:n_parts]

    def export_records(self, dst_ptr: int, dst_bytes: int):
        self._check(self._L.sfm_fusion_export_records(self._h, C.c_void_p(dst_ptr), dst_bytes))

    def merge(self, records_ptr: int, n: int):
        self._check(self._L.sfm_fusion_merge(self._h, C.c_void_p(records_ptr), n))

    def precompute(self):
        """Phase A of the two-phase chain: the mask-independent verdicts of the active keyframes (asynchronous)."""
        self._check(self._L.sfm_fusion_precompute(self._h))

    def set_active_range(self, first: int, end: int):
        """Only keyframes [first, end) are fused by this ctx (one sequence over several GPUs)."""
        self._check(self._L.sfm_fusion_set_active_range(self._h, int(first), int(end)))

    def import_masks(self, first: int, count: int, src_ptr: int, nbytes: int):
        self._check(self._L.sfm_fusion_import_masks(self._h, int(first), int(count), C.c_void_p(src_ptr), int(nbytes)))

    def clear_imported_masks(self):
        self._check(self._L.sfm_fusion_clear_imported_masks(self._h))

    def export_masks(self, first: int, count: int, dst_ptr: int, nbytes: int):
        self._check(self._L.sfm_fusion_export_masks(self._h, int(first), int(count), C.c_void_p(dst_ptr), int(nbytes)))

    def set_profiling(self, enable: bool):
        self._check(self._L.sfm_fusion_set_profiling(self._h, int(bool(enable))))

    def run(self):
        self._check(self._L.sfm_fusion_run(self._h))

    def run_chain(self):
        """First half of run(): returns when the frame loop has finished.  The masks may be exported from here on;
        the voxel filter of the last keyframe groups still runs.  finish() completes the run."""
        self._check(self._L.sfm_fusion_run_chain(self._h))

    def finish(self):
        self._check(self._L.sfm_fusion_finish(self._h))

    def num_points(self) -> int:
        n = C.c_uint64()
        self._check(self._L.sfm_fusion_num_points(self._h, C.byref(n)))
        return n.value

    def read_points(self):
        n = self.num_points()
        xyz = np.empty((n, 3), dtype=np.float64)
        bgr = np.empty((n, 3), dtype=np.int32)
        key = np.empty((n, 3), dtype=np.uint64)
        tag = np.empty((n,), dtype=np.uint64)
        self._check(self._L.sfm_fusion_read_points(self._h, xyz.ctypes.data, bgr.ctypes.data,
                                                   key.ctypes.data, tag.ctypes.data))
        return xyz, bgr, key, tag

    def stats(self) -> dict:
        st = FusionStats()
        self._check(self._L.sfm_fusion_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def frame_counts(self):
        before = np.zeros(self.n_frames, dtype=np.uint64)
        after = np.zeros(self.n_frames, dtype=np.uint64)
        self._check(self._L.sfm_fusion_get_frame_counts(self._h, before.ctypes.data, after.ctypes.data))
        return before, after


# ---- host-side mirror of the reference containers ------------------------------------------


@dataclass(eq=False)
class KeyFrame:
    """The part of sfm::KeyFrame the fusion reads (src/keyframe.cpp:31-54)."""
    img: np.ndarray            # (H,W,3) uint8 BGR  -- KeyFrame::img
    depth_img: np.ndarray      # (H,W) uint16       -- get_depth_image()
    K: tuple                   # fx,fy,cx,cy        -- camera_calibration()
    T_wk: np.ndarray           # 4x4 camera->world  -- transform()
    T_kw: np.ndarray           # 4x4 world->camera  -- world_to_camera()

    def get_depth_image(self):
        return self.depth_img

    def camera_calibration(self):
        return self.K

    def transform(self):
        return self.T_wk

    def world_to_camera(self):
        return self.T_kw


@dataclass
class Map:
    """The part of sfm::Map the fusion reads: keyframes in insertion order, the covisibility
    edge counts (map.hpp:143) and get_neighbors (map.cpp:152-160)."""
    keyframes: list = field(default_factory=list)
    covisibility_edge: dict = field(default_factory=dict)   # (i, j) with i<j -> shared count
    _adj: dict = field(default_factory=dict)

    def add_keyframe(self, kf: KeyFrame) -> int:
        self.keyframes.append(kf)
        return len(self.keyframes) - 1

    def insert_edge(self, i: int, j: int, count: int):
        """covisibility_insert (map.cpp:108-119): the count is frozen at first insertion."""
        if i == j:
            return
        e = (min(i, j), max(i, j))
        if e in self.covisibility_edge:
            return
        self.covisibility_edge[e] = int(count)
        self._adj.setdefault(i, []).append(j)
        self._adj.setdefault(j, []).append(i)

    def get_edge(self, i, j) -> int:
        return self.covisibility_edge[(min(i, j), max(i, j))]

    def get_neighbors(self, i: int, min_shared_features: int = 15):
        """Strictly more than min_shared_features (map.cpp:157).  Canonical order: ascending
        keyframe index."""
        return sorted(j for j in self._adj.get(i, []) if self.get_edge(i, j) > min_shared_features)


class DenseReconstruction:
    """Same two-call shape as sfm::DenseReconstruction.

    reconstruct(os) selects every keyframe that is an endpoint of an edge with count >= 100
    (dense_reconstruction.cpp:32-38), processes them in keyframe-insertion order, and writes
    `x y z R G B` lines (:61-64) sorted by voxel key.
    """
    SELECT_MIN = 100    # dense_reconstruction.cpp:34
    NEIGHBOR_MIN = 100  # dense_reconstruction.cpp:70

    def __init__(self, map_: Map, params: FusionParams | None = None, device: int = 0):
        self.map = map_
        self.params = params
        self.device = device
        self.ctx = None
        self.selected = []

    def _select(self):
        sel = set()
        for (i, j), count in self.map.covisibility_edge.items():
            if count >= self.SELECT_MIN:
                sel.add(i)
                sel.add(j)
        return sorted(sel)

    def fuse(self, in_library_rules: bool = False):
        """Runs the GPU fusion; returns (xyz, bgr, key, tag) sorted by key.  Tags carry the
        position in the selected-keyframe order -- or, with in_library_rules, the keyframe's index
        in the map: then every keyframe and the raw edge counts are handed to the library, which
        applies the selection and neighbour rules itself (sfm_fusion_set_covisibility)."""
        self.selected = self._select()
        self.ctx = FusionContext(self.params, self.device)
        if in_library_rules:
            for kf in self.map.keyframes:
                self.ctx.add_keyframe(kf.get_depth_image(), kf.img, kf.camera_calibration(),
                                      kf.transform(), kf.world_to_camera())
            self.ctx.set_covisibility([(a, b, n) for (a, b), n in self.map.covisibility_edge.items()],
                                      self.SELECT_MIN, self.NEIGHBOR_MIN)
            self.ctx.run()
            return self.ctx.read_points()
        pos = {k: n for n, k in enumerate(self.selected)}
        for k in self.selected:
            kf = self.map.keyframes[k]
            self.ctx.add_keyframe(kf.get_depth_image(), kf.img, kf.camera_calibration(),
                                  kf.transform(), kf.world_to_camera())
        for k in self.selected:
            nb = [pos[j] for j in self.map.get_neighbors(k, self.NEIGHBOR_MIN)]
            self.ctx.set_neighbors(pos[k], nb)
        self.ctx.run()
        return self.ctx.read_points()

    def progress_log(self):
        """The reference's progress log (dense_reconstruction.cpp:49, :54-58, :118-122) as a string: same lines, same
        order, from the counts of the last fuse()."""
        before, after = self.ctx.frame_counts()
        st = self.ctx.stats()
        n = len(before)
        out = []
        for i in range(n):
            out += [f"Processing keyframe {i}/{n}", f"Points before voxel filter: {int(before[i])}",
                    f"Points after voxel filter: {int(after[i])}", ""]
        out += ["processed all map frames", f"Points before voxel removal: {st['global_before']}",
                f"Points after voxel removal: {st['global_after']}"]
        return "\n".join(out) + "\n"

    def reconstruct(self, os, log=sys.stdout):
        """Writes the cloud to `os` and, like the reference, its progress log to stdout (log=None: silent)."""
        xyz, bgr, _, _ = self.fuse()
        if log is not None:
            log.write(self.progress_log())
        write_cloud(os, xyz, bgr)


def format_double(x: float) -> str:
    """`os << double` with default precision(6), floatfield unset == printf("%g").  glibc prints
    the sign of a NaN ("-nan" for x86's indefinite NaN, which is what every NaN on this path is);
    Python's %g drops it."""
    x = float(x)
    if x != x:
        return "-nan" if math.copysign(1.0, x) < 0 else "nan"
    return "%g" % x


def cloud_lines(xyz, bgr, sep=" "):
    """The reference's writer loop, dense_reconstruction.cpp:61-64: x y z, then colour z,y,x
    (= R G B), one line per point.  sep=", " gives the sparse-map variant (map.cpp:365-374)."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    bgr = np.asarray(bgr).reshape(-1, 3)
    fmt = sep.join(["%g", "%g", "%g", "%d", "%d", "%d"])
    has_nan = np.isnan(xyz).any(axis=1) if len(xyz) else np.zeros(0, bool)
    out = []
    for i in range(len(xyz)):
        if has_nan[i]:
            out.append(sep.join([format_double(xyz[i, 0]), format_double(xyz[i, 1]), format_double(xyz[i, 2]),
                                 str(int(bgr[i, 2])), str(int(bgr[i, 1])), str(int(bgr[i, 0]))]))
        else:
            out.append(fmt % (xyz[i, 0], xyz[i, 1], xyz[i, 2], bgr[i, 2], bgr[i, 1], bgr[i, 0]))
    return out


def write_cloud(os, xyz, bgr):
    """Text cloud, byte-identical to the reference's writer (tests/test_writer_parity.py pins the
    formatting against `std::ostream << double`)."""
    lines = cloud_lines(xyz, bgr)
    if lines:
        os.write("\n".join(lines) + "\n")


def write_sparse_csv(os, xyz, bgr):
    """The comma-separated variant of the reference's sparse-map writer (map.cpp:365-374)."""
    lines = cloud_lines(xyz, bgr, sep=", ")
    if lines:
        os.write("\n".join(lines) + "\n")


def write_ply(fp, xyz, bgr):
    """Binary little-endian PLY, lossless: double x/y/z, uchar red/green/blue (no reference
    counterpart; same bytes as sfm_amd::write_ply_binary).  fp: a binary file object."""
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype="<f8").reshape(-1, 3))
    bgr = np.asarray(bgr).reshape(-1, 3)
    n = len(xyz)
    fp.write(("ply\nformat binary_little_endian 1.0\ncomment fused by sfm_amd (MI355X dense fusion)\n"
              "element vertex %d\nproperty double x\nproperty double y\nproperty double z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii"))
    rec = np.zeros(n, dtype=np.dtype([("xyz", "<f8", 3), ("rgb", "u1", 3)]))
    rec["xyz"] = xyz
    rec["rgb"] = bgr[:, ::-1].astype(np.uint8)
    fp.write(rec.tobytes())


def read_ply(fp):
    """Reads what write_ply / write_ply_binary wrote: (xyz float64 [n,3], bgr int32 [n,3])."""
    header = b""
    while not header.endswith(b"end_header\n"):
        line = fp.readline()
        if not line:
            raise ValueError("not a PLY written by sfm_amd")
        header += line
    n = int([ln.split()[2] for ln in header.decode("ascii").split("\n") if ln.startswith("element vertex")][0])
    rec = np.frombuffer(fp.read(27 * n), dtype=np.dtype([("xyz", "<f8", 3), ("rgb", "u1", 3)]))
    return rec["xyz"].copy(), rec["rgb"][:, ::-1].astype(np.int32)
