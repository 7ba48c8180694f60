"""The oracle's primitives against NumPy / SciPy - implementations nobody in this repository wrote.

The reference ships no golden vectors and cannot be built in this image, so the oracle is "parity
unpinned" against the reference itself.  What CAN be pinned independently is every piece of third-party arithmetic the
reference leans on (Eigen's LLT, MatrixXd::inverse(), Quaterniond product / inverse() / toRotationMatrix(), kalman.cpp:104-107,
motion_model.cpp:102, full_feature_model.cpp:76-80) and the closed-form model equations the reference's text states
(motion_model.cpp:84-146, full_feature_model.cpp:67-101, camera.cpp:90-114, improc.cpp:55-134, monoslam.cpp:401-477): here
each is re-evaluated with LAPACK (numpy.linalg / scipy.linalg) and scipy.spatial.transform.Rotation and compared with what
oracle/dense.hpp and oracle/slam_oracle.hpp compute.  CPU only."""
import numpy as np
import pytest
import scipy.linalg
from scipy.spatial.transform import Rotation

import oracle_api as oa
from conftest import SHIPPED_CAM


def _spd(rng, n, cond=1e4):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return (q * np.geomspace(1.0, cond, n)) @ q.T


@pytest.mark.parametrize("n", [2, 8, 64, 200])
def test_llt_lower_equals_lapack_cholesky(n):
    rng = np.random.default_rng(100 + n)
    A = _spd(rng, n)
    A = 0.5 * (A + A.T)
    ok, L = oa.dense_llt(A)
    assert ok
    want = np.linalg.cholesky(A)
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    assert np.abs(L - want).max() <= 1e-12 * np.abs(want).max()
    assert np.abs(L @ L.T - A).max() <= 1e-13 * np.abs(A).max() * n


def test_llt_reads_only_the_lower_triangle_and_flags_a_bad_pivot():
    rng = np.random.default_rng(5)
    A = _spd(rng, 6)
    A = 0.5 * (A + A.T)
    junk = A.copy()
    junk[np.triu_indices(6, 1)] = 1e6                    # Eigen::LLT<.., Lower> never looks above the diagonal
    assert np.array_equal(oa.dense_llt(junk)[1], oa.dense_llt(A)[1])
    bad = A.copy()
    bad[3, 3] = -1.0
    assert oa.dense_llt(bad)[0] is False


@pytest.mark.parametrize("n", [2, 8, 64, 200])
def test_inverse_of_the_cholesky_factor_equals_lapack(n):
    """kalman.cpp:104-107: S_L = S.llt().matrixL(); S_Linv = S_L.inverse(); Sinv = S_Linv^T S_Linv."""
    rng = np.random.default_rng(200 + n)
    A = _spd(rng, n)
    A = 0.5 * (A + A.T)
    L = np.linalg.cholesky(A)
    Li = oa.dense_inverse(L)
    want = scipy.linalg.solve_triangular(L, np.eye(n), lower=True)
    scale = np.abs(want).max()
    assert np.abs(Li - want).max() <= 1e-11 * scale
    assert np.abs(np.triu(Li, 1)).max() <= 1e-13 * scale     # the general LU inverse of a lower triangle stays lower
    Sinv = Li.T @ Li
    assert np.abs(Sinv - np.linalg.inv(A)).max() <= 1e-9 * np.abs(Sinv).max()
    assert np.abs(Sinv @ A - np.eye(n)).max() <= 1e-9


@pytest.mark.parametrize("n", [3, 13, 50])
def test_general_inverse_equals_lapack_with_pivoting(n):
    rng = np.random.default_rng(300 + n)
    A = rng.standard_normal((n, n))
    A[0, 0] = 0.0                                        # forces a row exchange in the first column
    X = oa.dense_inverse(A)
    want = np.linalg.inv(A)
    assert np.abs(X - want).max() <= 1e-10 * np.abs(want).max()
    assert np.abs(A @ X - np.eye(n)).max() <= 1e-10 * np.linalg.cond(A)


def test_product_equals_numpy():
    rng = np.random.default_rng(7)
    A, B = rng.standard_normal((37, 53)), rng.standard_normal((53, 29))
    C = oa.dense_mul(A, B)
    assert np.abs(C - A @ B).max() <= 1e-13 * 53


def _scipy(q):        # (w, x, y, z) -> scipy's scalar-last
    return Rotation.from_quat([q[1], q[2], q[3], q[0]])


def _wxyz(r):
    x, y, z, w = r.as_quat()
    return np.array([w, x, y, z])


def test_unit_quaternion_ops_equal_scipy_rotation():
    rng = np.random.default_rng(11)
    for _ in range(50):
        a = rng.standard_normal(4); a /= np.linalg.norm(a)
        b = rng.standard_normal(4); b /= np.linalg.norm(b)
        prod, inv, R = oa.quat_ops(a, b)
        want = _wxyz(_scipy(a) * _scipy(b))
        assert min(np.abs(prod - want).max(), np.abs(prod + want).max()) <= 1e-14      # q and -q are the same rotation
        assert np.abs(R - _scipy(a).as_matrix()).max() <= 1e-14
        want_inv = _wxyz(_scipy(a).inv())
        assert min(np.abs(inv - want_inv).max(), np.abs(inv + want_inv).max()) <= 1e-14


def test_non_unit_quaternion_semantics_are_eigens():
    """Q11: inverse() = conjugate / squared norm; toRotationMatrix() does NOT normalise.  For q = s u (u unit) the nine
    entries of Eigen's formula are (1 - s^2) I + s^2 R(u) - checked against SciPy's R(u); and q * q.inverse() = identity,
    with the product evaluated independently as a 4x4 matrix-vector product in numpy."""
    rng = np.random.default_rng(13)

    def left_matrix(q):
        w, x, y, z = q
        return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])

    for _ in range(50):
        q = rng.standard_normal(4) * rng.uniform(0.3, 3.0)
        p = rng.standard_normal(4)
        s2 = float(q @ q)
        prod, inv, R = oa.quat_ops(q, p)
        assert np.abs(prod - left_matrix(q) @ p).max() <= 1e-14 * max(1.0, np.abs(prod).max())
        assert np.abs(inv - np.array([q[0], -q[1], -q[2], -q[3]]) / s2).max() <= 1e-15 * max(1.0, np.abs(inv).max())
        assert np.abs(left_matrix(q) @ inv - np.array([1.0, 0, 0, 0])).max() <= 1e-14
        want = (1.0 - s2) * np.eye(3) + s2 * _scipy(q / np.sqrt(s2)).as_matrix()
        assert np.abs(R - want).max() <= 1e-13 * max(1.0, s2)


def test_motion_model_state_transition_equals_scipy():
    """motion_model.cpp:84-146: r' = r + v dt, q' = q x q(omega dt), v' = v, omega' = omega - with SciPy's rotation-vector
    exponential standing in for math_util.cpp:61-80."""
    rng = np.random.default_rng(17)
    dt = 1.0 / 30.0
    for _ in range(20):
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        xv = np.concatenate([rng.standard_normal(3), q, rng.standard_normal(3) * 0.3, rng.standard_normal(3) * 0.4])
        f, F, Q = oa.motion_model(xv, dt)
        assert np.abs(f[:3] - (xv[:3] + xv[7:10] * dt)).max() <= 1e-15
        want_q = _wxyz(_scipy(q) * Rotation.from_rotvec(xv[10:13] * dt))
        assert min(np.abs(f[3:7] - want_q).max(), np.abs(f[3:7] + want_q).max()) <= 1e-14
        assert np.array_equal(f[7:], xv[7:])
        assert np.abs(Q - Q.T).max() <= 1e-16 * np.abs(Q).max() and np.linalg.eigvalsh(Q).min() >= -1e-15 * np.abs(Q).max()     # a covariance


def test_measurement_model_equals_numpy_projection():
    """full_feature_model.cpp:67-101 + camera.cpp:90-114: y_R = R(q)^T (y - r); u_c = (-f_u x / z, -f_v y / z);
    h = u_c / sqrt(1 + 2 k1 |u_c|^2) + centre; R = (sd (1 + |h - c| / |c|))^2 - rotation from SciPy."""
    cam = dict(SHIPPED_CAM)
    rng = np.random.default_rng(19)
    for _ in range(30):
        q = np.array([1.0, 0, 0, 0]) + 0.1 * rng.standard_normal(4); q /= np.linalg.norm(q)
        r = np.array([0.0, 0.0, -0.6]) + 0.05 * rng.standard_normal(3)
        y = np.array([rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.1), 0.0])
        m = oa.measurement_model(cam, np.concatenate([r, q]), y)
        yR = _scipy(q).as_matrix().T @ (y - r)
        uc = np.array([-cam["fku"] * yR[0] / yR[2], -cam["fkv"] * yR[1] / yR[2]])
        h = uc / np.sqrt(1.0 + 2.0 * cam["kd1"] * (uc @ uc)) + np.array([cam["u0"], cam["v0"]])
        assert np.abs(m["h"] - h).max() <= 1e-11
        c = np.array([cam["u0"], cam["v0"]])
        Rn = (cam["sd"] * (1.0 + np.linalg.norm(h - c) / np.linalg.norm(c))) ** 2
        assert abs(m["R"] - Rn) <= 1e-12 * Rn


def test_elliptical_search_equals_a_brute_force_numpy_scan():
    """monoslam.cpp:401-477 re-done with numpy primitives: every position inside the ellipse and the image, Pearson
    correlation from np.corrcoef (score = 2 (1 - rho)), np.std for the two sigma >= 10 gates, argmin, <= 0.40 to accept."""
    rng = np.random.default_rng(23)
    H, W = 120, 160
    base = rng.integers(0, 256, (H // 4 + 2, W // 4 + 2)).astype(np.float64)
    img = np.kron(base, np.ones((4, 4)))[:H, :W]
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    checked = 0
    for trial in range(12):
        cu, cv = int(rng.integers(30, W - 30)), int(rng.integers(30, H - 30))
        patch = img[cv - 5:cv + 6, cu - 5:cu + 6].copy()
        if trial % 3 == 2:
            patch = np.clip(patch.astype(int) * 0.7 + 20, 0, 255).astype(np.uint8)     # gain / offset: still the best match
        S = _spd(rng, 2, cond=4.0) * rng.uniform(4.0, 20.0)
        Si = np.linalg.inv(S)
        a, b, c = Si[0, 0], Si[0, 1], Si[1, 1]
        centre = np.array([cu + rng.uniform(-2, 2), cv + rng.uniform(-2, 2)])
        got = oa.elliptical_search(img, patch, centre, a, b, c)
        hw, hh = int(3.0 / np.sqrt(a - b * b / c)), int(3.0 / np.sqrt(c - b * b / a))
        assert (got["hw"], got["hh"]) == (hw, hh)
        uc, vc = int(centre[0] + 0.5), int(centre[1] + 0.5)
        best, where, ncand = np.inf, None, 0
        for du in range(-hw, hw + 1):                                  # u outer, v inner; "<=" keeps the LAST of equals (Q2)
            for dv in range(-hh, hh + 1):
                u, v = uc + du, vc + dv
                if u - 5 < 0 or v - 5 < 0 or u + 5 >= W or v + 5 >= H:
                    continue
                if not (a * du * du + 2 * b * du * dv + c * dv * dv < 9.0):
                    continue
                ncand += 1
                win = img[v - 5:v + 6, u - 5:u + 6].astype(np.float64)
                if win.std() < 10.0 or patch.std() < 10.0:
                    continue
                score = 2.0 * (1.0 - np.corrcoef(win.ravel(), patch.astype(np.float64).ravel())[0, 1])
                if score <= best + 1e-12:
                    best, where = score, (u, v)
        assert got["ncand"] == ncand
        assert got["ok"] == (best <= 0.40)
        if got["ok"]:
            assert (got["u"], got["v"]) == where and abs(got["corr"] - best) <= 1e-10
            checked += 1
    assert checked >= 8
