"""CPU restatement of the two Open3D registration routines behind the reference's safeguard and ICP.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Arithmetic unpinned against Open3D**: Open3D (requirements.txt pins
open3d==0.17.0) is not installed in this image and its source is not under /root/reference, so the
functions below restate the published algorithm of Open3D 0.17.0
(cpp/open3d/pipelines/registration/Registration.cpp and TransformationEstimation.cpp).  What IS pinned, by the
reference's own code: the call sites -- tests/golden/make_golden_register.py runs
core/deep_global_registration.py:50-64, 302-322 unchanged over a stand-in `open3d` module with Open3D 0.17's
signatures that records every argument (source / target order, correspondences, 2 * voxel, rigid point-to-point,
ransac_n = 4, criteria (4000000, 80000 -> confidence 1.0), ICP from T with Open3D's defaults) and computes with the
functions below (tests/golden/register_e2e_o3d.npz, tests/test_oracle_register_golden.py).  The functions and the
reference call they stand for:

* `icp_point_to_point`        <- `o3d.pipelines.registration.registration_icp(source, target,
                                 max_correspondence_distance=2*voxel, init=T)` at
                                 core/deep_global_registration.py:317-322 (defaults: point-to-point
                                 estimation without scaling, ICPConvergenceCriteria(1e-6, 1e-6, 30))
* `ransac_correspondence`     <- `registration_ransac_based_on_correspondence(pcd0, pcd1, corres,
                                 2*voxel, TransformationEstimationPointToPoint(False), ransac_n=4,
                                 RANSACConvergenceCriteria(4000000, 80000))` at :50-64, 302-315.  The
                                 second criteria argument is the *confidence* in Open3D >= 0.12 and is
                                 clamped to 1.0, so the early exit never triggers and all max_iteration
                                 hypotheses are evaluated; no checkers are passed; the best 4-point
                                 hypothesis is returned as is (no refit on its inliers).

Open3D draws its RANSAC samples from per-thread std::mt19937 streams, which nothing can reproduce
bit-for-bit.  The samples here come from the counter-based generator `ransac_samples` (shared with
the HIP kernel), and the per-correspondence inlier test is evaluated in f32 with a fixed operation
order (no fma) so that the consensus counts are exactly reproducible; what is preserved from Open3D
is the hypothesis distribution (uniform draws with replacement), the estimator (Umeyama/Kabsch without
scaling), the consensus criterion (most inliers with distance < threshold, ties -> lower RMSE) and
the returned transformation.
"""
import numpy as np


def umeyama(src, dst):
    """Eigen::umeyama(src, dst, with_scaling=false) -> 4x4 float64 (TransformationEstimationPointToPoint)."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    T = np.eye(4)
    n = len(src)
    if n == 0:
        return T
    ms, md = src.mean(0), dst.mean(0)
    sigma = (dst - md).T @ (src - ms) / n
    U, _, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2] = -1
    R = U @ np.diag(S) @ Vt
    T[:3, :3] = R
    T[:3, 3] = md - R @ ms
    return T


def _evaluate_icp(P, tree, dst, max_dist):
    d, j = tree.query(P, k=1, distance_upper_bound=max_dist)
    ok = np.isfinite(d)
    n = int(ok.sum())
    fitness = n / max(len(P), 1)
    rmse = float(np.sqrt((d[ok] ** 2).sum() / n)) if n else 0.0
    return np.nonzero(ok)[0], j[ok], fitness, rmse


def icp_point_to_point(src, dst, max_dist, init=None, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    """RegistrationICP (Registration.cpp), point-to-point.  Returns (T [4,4] f64, fitness, inlier_rmse,
    iterations run)."""
    from scipy.spatial import cKDTree
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    T = np.eye(4) if init is None else np.asarray(init, np.float64).copy()
    P = src @ T[:3, :3].T + T[:3, 3]
    tree = cKDTree(dst)
    ci, cj, fit, rmse = _evaluate_icp(P, tree, dst, max_dist)
    it = 0
    for it in range(1, max_iter + 1):
        upd = umeyama(P[ci], dst[cj])
        T = upd @ T
        P = P @ upd[:3, :3].T + upd[:3, 3]
        pf, pr = fit, rmse
        ci, cj, fit, rmse = _evaluate_icp(P, tree, dst, max_dist)
        if abs(pf - fit) < rel_fitness and abs(pr - rmse) < rel_rmse:
            break
    return T, fit, rmse, it


# ---- RANSAC ----------------------------------------------------------------------------------------
def _mix32(x):
    """32-bit finaliser (murmur3 fmix32) on uint32 arrays."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x85ebca6b)).astype(np.uint32)
    x ^= x >> np.uint32(13)
    x = (x * np.uint32(0xc2b2ae35)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def ransac_samples(seed, first, count, n_corr, ransac_n=4):
    """Correspondence indices of hypotheses first .. first+count-1: [count, ransac_n] int64.  Draw j of
    hypothesis h is mix32(mix32(h * 4 + j + 0x9e3779b9 * seed) ^ 0x68bc21eb) mapped to [0, n_corr) by the
    high half of the 32 x 32 -> 64 bit product (uniform up to 2^-32, with replacement)."""
    with np.errstate(over='ignore'):
        h = (np.arange(first, first + count, dtype=np.uint64)[:, None] * np.uint64(ransac_n)
             + np.arange(ransac_n, dtype=np.uint64)[None, :])
        key = (h + np.uint64(0x9e3779b9) * np.uint64(seed)).astype(np.uint32)
        r = _mix32(_mix32(key) ^ np.uint32(0x68bc21eb))
    return ((r.astype(np.uint64) * np.uint64(n_corr)) >> np.uint64(32)).astype(np.int64)


def ransac_correspondence(X, Y, max_dist, num_hyp, seed=0, ransac_n=4, chunk=2048):
    """X, Y [N,3]: corresponding points (xyz0[idx0], xyz1[idx1]).  Returns (T [4,4] f64, best hypothesis
    index, inlier count, inlier rmse).  Consensus test in f32, fixed op order:
        p = (R00*x + R01*y) + R02*z + t0 ...;  e = p - y;  d2 = (e0*e0 + e1*e1) + e2*e2;  inlier: d2 < thr2
    with R, t the f32-rounded hypothesis.  Best = most inliers; ties -> lower sum of d2 (accumulated
    sequentially in float32 over the inliers in correspondence order); remaining ties -> lower
    hypothesis index."""
    X32 = np.asarray(X, np.float32)
    Y32 = np.asarray(Y, np.float32)
    X64, Y64 = X32.astype(np.float64), Y32.astype(np.float64)
    n = len(X32)
    thr2 = np.float32(np.float32(max_dist) * np.float32(max_dist))
    best = (-1, np.inf, -1, np.eye(4))
    for first in range(0, num_hyp, chunk):
        cnt = min(chunk, num_hyp - first)
        S = ransac_samples(seed, first, cnt, n, ransac_n)
        for h in range(cnt):
            T = umeyama(X64[S[h]], Y64[S[h]])
            R = T[:3, :3].astype(np.float32)
            t = T[:3, 3].astype(np.float32)
            p0 = (R[0, 0] * X32[:, 0] + R[0, 1] * X32[:, 1]) + R[0, 2] * X32[:, 2] + t[0]
            p1 = (R[1, 0] * X32[:, 0] + R[1, 1] * X32[:, 1]) + R[1, 2] * X32[:, 2] + t[1]
            p2 = (R[2, 0] * X32[:, 0] + R[2, 1] * X32[:, 1]) + R[2, 2] * X32[:, 2] + t[2]
            e0, e1, e2 = p0 - Y32[:, 0], p1 - Y32[:, 1], p2 - Y32[:, 2]
            d2 = (e0 * e0 + e1 * e1) + e2 * e2
            inl = d2 < thr2
            c = int(inl.sum())
            if c < best[0]:
                continue
            err = float(np.add.accumulate(d2[inl], dtype=np.float32)[-1]) if c else 0.0
            if c > best[0] or err < best[1]:
                best = (c, err, first + h, T)
    c, err, h, T = best
    return T, h, c, (float(np.sqrt(err / c)) if c > 0 else 0.0)
