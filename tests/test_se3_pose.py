"""sfm_fusion_se3_exp, the pose conversion of the bundle-adjustment write-back (reference src/map.cpp:321-324:
tf = Sophus::SE3<double>::exp(se3_vec).matrix()), against the matrix exponential of the twist and against
closed-form known answers.  Host-only code of the library: runs without a GPU."""
import numpy as np
import pytest
from scipy.linalg import expm

from sfm_amd import fusion


def twist(se3):
    u, w = np.asarray(se3[:3], float), np.asarray(se3[3:], float)
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


def sophus_exp_py(se3):
    """A second restatement of Sophus::SE3d::exp(v).matrix() and of the rigid inverse, in plain Python floats, written
    from Sophus' published formulas (so3.hpp expAndTheta, se3.hpp exp) and Eigen's Quaternion::toRotationMatrix,
    separately from the library's C code: (T, T_inv) as 4x4 arrays.  Sophus itself is not in this image."""
    import math
    ux, uy, uz, wx, wy, wz = (float(x) for x in se3)
    theta_sq = wx * wx + wy * wy + wz * wz
    eps = 1e-10
    if theta_sq < eps * eps:
        theta = 0.0
        po4 = theta_sq * theta_sq
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * po4
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * po4
    else:
        theta = math.sqrt(theta_sq)
        half = 0.5 * theta
        imag = math.sin(half) / theta
        real = math.cos(half)
    qw, qx, qy, qz = real, imag * wx, imag * wy, imag * wz
    tx, ty, tz = 2.0 * qx, 2.0 * qy, 2.0 * qz
    twx, twy, twz = tx * qw, ty * qw, tz * qw
    txx, txy, txz = tx * qx, ty * qx, tz * qx
    tyy, tyz, tzz = ty * qy, tz * qy, tz * qz
    R = [[1.0 - (tyy + tzz), txy - twz, txz + twy],
         [txy + twz, 1.0 - (txx + tzz), tyz - twx],
         [txz - twy, tyz + twx, 1.0 - (txx + tyy)]]
    if abs(theta) < eps:
        V = [row[:] for row in R]
    else:
        Om = [[0.0, -wz, wy], [wz, 0.0, -wx], [-wy, wx, 0.0]]
        Om2 = [[Om[i][0] * Om[0][j] + Om[i][1] * Om[1][j] + Om[i][2] * Om[2][j] for j in range(3)] for i in range(3)]
        a = (1.0 - math.cos(theta)) / theta_sq
        b = (theta - math.sin(theta)) / (theta_sq * theta)
        V = [[(1.0 if i == j else 0.0) + a * Om[i][j] + b * Om2[i][j] for j in range(3)] for i in range(3)]
    t = [V[i][0] * ux + V[i][1] * uy + V[i][2] * uz for i in range(3)]
    T = np.eye(4)
    Ti = np.eye(4)
    for i in range(3):
        for j in range(3):
            T[i, j] = R[i][j]
            Ti[i, j] = R[j][i]
        T[i, 3] = t[i]
        Ti[i, 3] = -(R[0][i] * t[0] + R[1][i] * t[1] + R[2][i] * t[2])
    return T, Ti


def test_the_library_and_the_python_restatement_agree_bit_for_bit():
    rng = np.random.default_rng(11)
    cases = [np.concatenate([rng.normal(0, 3.0, 3), rng.normal(0, 1.2, 3)]) for _ in range(300)]
    cases += [np.array([0.3, -0.1, 0.7, w, -w / 2, w / 3]) for w in (0.0, 1e-12, 9.9e-11, 1.1e-10, 1e-8, 1e-5)]
    for se3 in cases:
        T, Ti = fusion.se3_exp(se3)
        Tp, Tip = sophus_exp_py(se3)
        assert np.array_equal(T.view(np.uint64), Tp.view(np.uint64)), se3
        assert np.array_equal(Ti.view(np.uint64), Tip.view(np.uint64)), se3


def test_known_answers():
    T, Ti = fusion.se3_exp([1.5, -2.0, 0.25, 0, 0, 0])              # pure translation
    assert np.array_equal(T, np.array([[1, 0, 0, 1.5], [0, 1, 0, -2.0], [0, 0, 1, 0.25], [0, 0, 0, 1.0]]))
    assert np.array_equal(Ti[:3, 3], [-1.5, 2.0, -0.25])
    T, _ = fusion.se3_exp([0, 0, 0, 0, 0, np.pi / 2])              # quarter turn about z
    assert np.allclose(T[:3, :3], [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15) and np.array_equal(T[:3, 3], [0, 0, 0])
    # a screw along z: translation along the axis is untouched, the in-plane part follows V
    th = 0.7
    T, _ = fusion.se3_exp([1.0, 0.0, 2.0, 0, 0, th])
    assert np.allclose(T[:3, 3], [np.sin(th) / th, (1 - np.cos(th)) / th, 2.0], rtol=0, atol=1e-15)


def test_against_the_matrix_exponential():
    rng = np.random.default_rng(5)
    for _ in range(200):
        se3 = np.concatenate([rng.normal(0, 3.0, 3), rng.normal(0, 1.2, 3)])
        T, Ti = fusion.se3_exp(se3)
        assert np.allclose(T, expm(twist(se3)), rtol=0, atol=5e-13)
        assert np.allclose(T @ Ti, np.eye(4), rtol=0, atol=5e-13)
        R = T[:3, :3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
        assert np.array_equal(T[3], [0, 0, 0, 1]) and np.array_equal(Ti[3], [0, 0, 0, 1])


def test_small_angle_branch():
    for w in (0.0, 1e-12, 9.9e-11, 1.1e-10, 1e-8, 1e-5):
        se3 = np.array([0.3, -0.1, 0.7, w, -w / 2, w / 3])
        T, _ = fusion.se3_exp(se3)
        # Sophus' own formulas, restated as they are: below |omega| = 1e-10 it takes V = R, and up to about 1e-7 the
        # (1 - cos th)/th^2 of V cancels to zero in double precision -- either way an error of the angle's size
        th = np.linalg.norm(se3[3:])
        assert np.allclose(T, expm(twist(se3)), rtol=0, atol=th + 1e-15 if th < 1e-6 else 1e-12)


@pytest.mark.gpu
def test_add_keyframe_se3_equals_add_keyframe_with_the_converted_matrices():
    from scipy.spatial.transform import Rotation
    from sfm_amd import synth
    seq = synth.make_sequence(5, 64, 48, K=(131.25, 131.25, 31.5, 23.5), window=2)
    frames = seq.frame_dicts()
    gp = fusion.default_params(voxel_size=30.0, T_dist=500.0)
    clouds = []
    se3s = []
    for use_se3 in (True, False):
        ctx = fusion.FusionContext(gp)
        for f in frames:
            # the keyframe's world->camera pose as a tangent vector: log of the rotation, V^-1 t by solving V u = t
            rv = Rotation.from_matrix(f["T_kw"][:3, :3]).as_rotvec()
            V = fusion.se3_exp(np.concatenate([[0, 0, 0], rv]))[0][:3, :3]
            th = np.linalg.norm(rv)
            Om = twist(np.concatenate([[0, 0, 0], rv]))[:3, :3]
            Vm = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * Om @ Om
            se3 = np.concatenate([np.linalg.solve(Vm, f["T_kw"][:3, 3]), rv])
            if use_se3:
                se3s.append(se3)
                ctx.add_keyframe_se3(f["depth"], f["bgr"], f["K"], se3)
            else:
                T_kw, T_wk = fusion.se3_exp(se3)
                ctx.add_keyframe(f["depth"], f["bgr"], f["K"], T_wk, T_kw)
        for i, f in enumerate(frames):
            ctx.set_neighbors(i, f["nbrs"])
        ctx.run()
        clouds.append(ctx.read_points())
        ctx.close()
    assert clouds[0][2].shape[0] > 500
    for a, b in zip(*clouds):
        assert np.array_equal(a, b)
    # ... and against the ORACLE fed with the matrices of the independent Python restatement of Sophus' formulas
    # (the comparison above is the HIP path against itself)
    from oracle import oracle
    ref = []
    for f, se3 in zip(frames, se3s):
        T_kw, T_wk = sophus_exp_py(se3)
        ref.append(dict(f, T_kw=T_kw, T_wk=T_wk))
    want = oracle.fuse(ref, oracle.default_params(voxel_size=30.0, T_dist=500.0))
    xyz, bgr, key, tag = clouds[0]
    assert np.array_equal(key, want.key) and np.array_equal(tag, want.tag) and np.array_equal(bgr, want.bgr)
    assert np.array_equal(xyz.view(np.uint64), want.xyz.view(np.uint64))
