"""CPU checks of the restated Open3D routines (oracle/open3d_reg.py).  Open3D itself is not available
offline, so these are known-answer and property tests of the restatement (parity unpinned)."""
import numpy as np

from conftest import rot_angle_deg
from oracle import open3d_reg as o3


def _rot(rng, max_deg):
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    a = np.radians(rng.uniform(0, max_deg))
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def test_umeyama_known_answers():
    rng = np.random.default_rng(0)
    P = rng.standard_normal((50, 3))
    R, t = _rot(rng, 180), rng.standard_normal(3)
    T = o3.umeyama(P, P @ R.T + t)
    np.testing.assert_allclose(T[:3, :3], R, atol=1e-12)
    np.testing.assert_allclose(T[:3, 3], t, atol=1e-12)
    # a mirrored target must still give a proper rotation (Eigen::umeyama's sign fix)
    Q = P * np.array([1, 1, -1.0])
    T = o3.umeyama(P, Q)
    assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
    np.testing.assert_array_equal(o3.umeyama(np.zeros((0, 3)), np.zeros((0, 3))), np.eye(4))


def test_icp_recovers_small_motion_and_counts_iterations():
    rng = np.random.default_rng(1)
    dst = rng.uniform(-1, 1, (4000, 3))
    R, t = _rot(rng, 3.0), rng.uniform(-0.02, 0.02, 3)
    src = (dst[:3000] - t) @ R            # dst = R src + t
    T, fit, rmse, it = o3.icp_point_to_point(src, dst, 0.1)
    assert fit == 1.0 and rmse < 1e-9 and 1 <= it <= 30
    assert rot_angle_deg(T[:3, :3], R) < 1e-4 and np.linalg.norm(T[:3, 3] - t) < 1e-8
    # the init is honoured; no overlap -> no correspondences, identity update, stops after one iteration
    T2, fit2, _, it2 = o3.icp_point_to_point(src + 100.0, dst, 0.1, init=np.eye(4))
    assert fit2 == 0.0 and it2 == 1
    np.testing.assert_array_equal(T2, np.eye(4))


def test_ransac_samples_are_deterministic_and_uniform():
    a = o3.ransac_samples(3, 0, 1000, 777)
    b = o3.ransac_samples(3, 500, 500, 777)
    np.testing.assert_array_equal(a[500:], b)                    # counter based: any sub-range reproduces
    assert a.min() >= 0 and a.max() < 777
    assert not np.array_equal(a, o3.ransac_samples(4, 0, 1000, 777))
    big = o3.ransac_samples(0, 0, 50000, 100).reshape(-1)
    cnt = np.bincount(big, minlength=100)
    assert abs(cnt - 2000).max() < 250                           # ~5 sigma of a uniform draw


def test_ransac_finds_the_planted_model():
    rng = np.random.default_rng(2)
    n = 600
    X = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    R, t = _rot(rng, 180), rng.uniform(-1, 1, 3)
    Y = (X @ R.T + t).astype(np.float32)
    out = rng.random(n) < 0.6
    Y[out] = rng.uniform(-3, 3, (int(out.sum()), 3)).astype(np.float32)
    T, h, c, rmse = o3.ransac_correspondence(X, Y, 0.05, 3000, seed=5)
    assert c >= int((~out).sum()) and rmse < 0.05
    assert rot_angle_deg(T[:3, :3], R) < 0.5 and np.linalg.norm(T[:3, 3] - t) < 0.02
    # ties: same count -> lower error -> lower index; re-running is deterministic
    T2, h2, c2, _ = o3.ransac_correspondence(X, Y, 0.05, 3000, seed=5)
    assert (h, c) == (h2, c2) and np.array_equal(T, T2)
