"""A second, independent restatement of the reference's dense fusion in pure Python, written
directly from /root/reference/reconstruction/src/dense_reconstruction.cpp (+ pinhole.cpp,
keyframe.cpp:275-294, features.cpp:76-80) with Python floats (IEEE binary64, no FMA).

It exists to cross-check oracle/fusion_oracle.c on tiny scenes: two restatements written
separately from the same source lines must agree bit for bit.  Slow; tests only.
Evaluation orders: see the header of oracle/fusion_oracle.c.
"""
import math

T_DEFAULT = dict(base_line=0.50, T_cov=0.5, voxel_size=0.05, T_dist=0.50, min_dist_filter=0.0,
                 dir_dot=0.30, S=(0.05, 0.05, 10.0), depth_div=1000.0, close_frac=0.50)


def mat3mul(A, B):
    # Eigen lazy coefficient product, scalar path: a0*b0 + (a1*b1 + a2*b2)
    return [[A[i][0] * B[0][j] + (A[i][1] * B[1][j] + A[i][2] * B[2][j]) for j in range(3)] for i in range(3)]


def calculate_jacobian(u, v, d, focal_length, prm):            # dense_reconstruction.cpp:292-299
    b = prm["base_line"]
    return [[b / d, 0.0, float(-u) * b / (d * d)],
            [0.0, b / d, float(-v) * b / (d * d)],
            [0.0, 0.0, -focal_length * b / (d * d)]]


def calculate_covariance_trace(u, v, d, focal_length, prm):    # :301-313, then .trace()
    disparity = (focal_length * prm["base_line"]) / (d / prm["depth_div"]) if d != 0 else math.inf
    J = calculate_jacobian(u, v, disparity, focal_length, prm)
    S = [[prm["S"][0], 0.0, 0.0], [0.0, prm["S"][1], 0.0], [0.0, 0.0, prm["S"][2]]]
    Jt = [[J[j][i] for j in range(3)] for i in range(3)]
    P = mat3mul(mat3mul(J, S), Jt)
    return P[0][0] + (P[1][1] + P[2][2])


def deproject(K, px, py, depth):                                # pinhole.cpp:8-30
    x = (float(px) - K[2]) / K[0]
    y = (float(py) - K[3]) / K[1]
    return (x * depth, y * depth, depth)


def apply(T, p):                                                # T * p.homogeneous(), rows 0..2
    return tuple(((T[r][0] * p[0] + T[r][1] * p[1]) + T[r][2] * p[2]) + T[r][3] for r in range(3))


def safe_div(a, b):
    if b == 0.0:
        if a == 0.0 or a != a:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)
    return a / b


def point_in_frame(kf, pw):                                     # keyframe.cpp:275-294
    c = apply(kf["T_kw"], pw)
    K = kf["K"]
    px = safe_div(c[0], c[2]) * K[0] + K[2]                     # pinhole.cpp:42-43
    py = safe_div(c[1], c[2]) * K[1] + K[3]
    H, W = kf["depth"].shape
    if px != px or py != py:
        return None                                             # defined here, UB in the reference
    if px < 0 or px >= W or py < 0 or py >= H:
        return None
    if c[2] < 0:
        return None
    return (int(math.floor(px)), int(math.floor(py)))


def voxel_key(coord, vs):                                       # :270-274, x86-64 cvttsd2si semantics
    q = safe_div(coord, vs)
    if q != q:
        return 0x8000000000000000
    if math.isinf(q):
        return 0x8000000000000000
    t = math.trunc(q)
    if -(1 << 63) <= t < (1 << 63):
        return t & 0xFFFFFFFFFFFFFFFF
    if (1 << 63) <= t < (1 << 64):
        return t
    return 0x8000000000000000


def neighbor_kept(f, lf, prm):                                  # :72-89
    a, b = f["T_wk"], lf["T_wk"]
    dx, dy, dz = b[0][3] - a[0][3], b[1][3] - a[1][3], b[2][3] - a[2][3]
    if math.sqrt((dx * dx + dy * dy) + dz * dz) > prm["min_dist_filter"]:
        return True
    dot = (a[0][2] * b[0][2] + a[1][2] * b[1][2]) + a[2][2] * b[2][2]
    return dot < prm["dir_dot"]


def fuse(frames, **over):
    """frames: list of dicts depth (H,W) u16 ndarray, bgr (H,W,3) u8 ndarray, K, T_wk, T_kw
    (nested lists or arrays), nbrs.  Returns dict with sorted cloud and counts."""
    prm = dict(T_DEFAULT)
    prm.update(over)
    fr = []
    for f in frames:
        fr.append(dict(depth=f["depth"], bgr=f["bgr"], K=[float(x) for x in f["K"]],
                       T_wk=[[float(x) for x in r] for r in f["T_wk"]],
                       T_kw=[[float(x) for x in r] for r in f["T_kw"]], nbrs=list(f.get("nbrs", []))))
    masks = [[[0] * f["depth"].shape[1] for _ in range(f["depth"].shape[0])] for f in fr]
    all_points, frame_before, frame_after = [], [], []
    for fi, f in enumerate(fr):
        local = [fr_i for fr_i in f["nbrs"] if neighbor_kept(f, fr[fr_i], prm)]
        H, W = f["depth"].shape
        pts = []
        for i in range(W * H):                                  # :95-110
            u, v = i % W, i // W
            if masks[fi][v][u] > 0:
                continue
            depth = int(f["depth"][v, u])
            if depth <= 0:
                continue
            tr = calculate_covariance_trace(u, v, float(depth), f["K"][0], prm)
            if tr > prm["T_cov"]:
                continue
            pw = apply(f["T_wk"], deproject(f["K"], u, v, float(depth)))
            if not local:
                continue
            obs = []
            for li in local:                                    # geometric_check :187-216
                lf = fr[li]
                pf = point_in_frame(lf, pw)
                if pf is None:
                    continue
                d = int(lf["depth"][pf[1], pf[0]])
                q = apply(lf["T_wk"], deproject(lf["K"], pf[0], pf[1], float(d)))
                w = calculate_covariance_trace(pf[0], pf[1], float(d), lf["K"][0], prm)
                if w < prm["T_cov"]:
                    obs.append((q, w, li, pf))
            if len(obs) < 1:
                continue
            close = 0
            for q, _, _, _ in obs:
                dx, dy, dz = q[0] - pw[0], q[1] - pw[1], q[2] - pw[2]
                if math.sqrt((dx * dx + dy * dy) + dz * dz) < prm["T_dist"]:
                    close += 1
            if close < len(obs) * prm["close_frac"]:
                continue
            for _, _, li, pf in obs:                            # :152-154
                masks[li][pf[1]][pf[0]] = 255
            wp, total = [0.0, 0.0, 0.0], 0.0                    # :158-165
            for q, w, _, _ in obs:
                inv = safe_div(1.0, w)
                for c in range(3):
                    wp[c] = wp[c] + q[c] * inv                  # 0*inf -> nan, as in C
                total = total + inv
            p = tuple(safe_div(wp[c], total) for c in range(3)) # inf/inf -> nan, as in C
            col = tuple(int(x) for x in f["bgr"][v, u])
            pts.append((p, col, (fi << 32) | i))
        frame_before.append(len(pts))
        kept = voxel_filter(pts, prm)                           # :119
        frame_after.append(len(kept))
        all_points.extend(kept)
    final = voxel_filter(all_points, prm)                       # :57
    rows = []
    for p, col, tag in final:
        key = tuple(voxel_key(p[c], prm["voxel_size"]) for c in range(3))
        rows.append((key, tag, p, col))
    rows.sort(key=lambda r: r[0])
    return dict(rows=rows, frame_before=frame_before, frame_after=frame_after,
                global_before=len(all_points), global_after=len(final))


def voxel_filter(points, prm):                                  # :263-290, first wins
    seen, out = set(), []
    for p, col, tag in points:
        key = tuple(voxel_key(p[c], prm["voxel_size"]) for c in range(3))
        if key not in seen:
            seen.add(key)
            out.append((p, col, tag))
    return out
