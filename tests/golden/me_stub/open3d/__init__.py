"""A stand-in for `import open3d` so that `core/deep_global_registration.py` and `util/pointcloud.py` of the reference
IMPORT and RUN in this container (tests/golden/make_golden_register.py).

It has the classes those modules name at import time and in `preprocess`'s isinstance test, and -- since round 5 -- the
two Open3D ENTRY POINTS of the path with Open3D 0.17's signatures (keyword names, defaults, result objects):

* `pipelines.registration.registration_ransac_based_on_correspondence(source, target, corres,
  max_correspondence_distance, estimation_method, ransac_n, checkers, criteria)`
* `pipelines.registration.registration_icp(source, target, max_correspondence_distance, init, estimation_method,
  criteria)`

What the golden run pins through them is the reference's GLUE (core/deep_global_registration.py:50-64, 302-322):
which cloud is source and which target, that the correspondences are `(idx0, idx1)` rows over the VOXELISED clouds,
the distance `2 * voxel_size`, `TransformationEstimationPointToPoint(False)`, `ransac_n = 4`, the criteria arguments,
that ICP starts from T and runs after either branch, and that `.transformation` is what is returned.  Every call is
recorded in `CALLS` (argument values as the reference passed them) and the generator stores them in the fixture.

The ARITHMETIC behind the two entry points is `oracle/open3d_reg.py` (a restatement of Open3D 0.17's published
algorithms with a counter-based sample generator -- Open3D's per-thread mt19937 streams cannot be reproduced by
anything); it stays unpinned against Open3D itself, and its header says so.  `RANSAC_CAP` (set by the generator)
bounds the number of hypotheses the stand-in evaluates -- the reference hard-codes 4 000 000 (:61), which the CPU
restatement would need ten minutes for; the count the reference ASKED for is recorded, the count evaluated is stored
next to it and the tests give the same count to both sides.
"""
import types

import numpy as np

CALLS = []            # one dict per entry-point call, in order
RANSAC_CAP = None     # int: evaluate at most this many hypotheses (see above)
RANSAC_SEED = 0


class _PointCloud:
    def __init__(self):
        self.points = None
        self.colors = None


class _TransformationEstimationPointToPoint:
    def __init__(self, with_scaling=False):
        self.with_scaling = bool(with_scaling)


class _RANSACConvergenceCriteria:
    """Open3D 0.17: RANSACConvergenceCriteria(max_iteration=100000, confidence=0.999); the confidence is clamped to
    [0, 1] by the constructor (RANSACConvergenceCriteria in Registration.h)."""
    def __init__(self, max_iteration=100000, confidence=0.999):
        self.max_iteration = int(max_iteration)
        self.confidence_given = float(confidence)
        self.confidence = max(0.0, min(1.0, float(confidence)))


class _ICPConvergenceCriteria:
    def __init__(self, relative_fitness=1e-6, relative_rmse=1e-6, max_iteration=30):
        self.relative_fitness, self.relative_rmse, self.max_iteration = relative_fitness, relative_rmse, max_iteration


class _RegistrationResult:
    def __init__(self, T, fitness, inlier_rmse):
        self.transformation = T
        self.fitness = fitness
        self.inlier_rmse = inlier_rmse


def _points(pcd):
    assert isinstance(pcd, _PointCloud), 'the entry points take open3d.geometry.PointCloud objects'
    return np.asarray(pcd.points)


def _ransac_based_on_correspondence(source, target, corres, max_correspondence_distance,
                                    estimation_method=None, ransac_n=3, checkers=(), criteria=None):
    from oracle import open3d_reg
    estimation_method = estimation_method or _TransformationEstimationPointToPoint(False)
    criteria = criteria or _RANSACConvergenceCriteria()
    src, dst, corres = _points(source), _points(target), np.asarray(corres)
    assert corres.ndim == 2 and corres.shape[1] == 2
    assert isinstance(estimation_method, _TransformationEstimationPointToPoint) and not estimation_method.with_scaling, \
        'the restated estimator is point-to-point without scaling'
    assert ransac_n == 4 and len(checkers) == 0, 'oracle/open3d_reg.py restates the 4-point, checker-free call only'
    n_hyp = criteria.max_iteration if RANSAC_CAP is None else min(criteria.max_iteration, int(RANSAC_CAP))
    CALLS.append(dict(fn='registration_ransac_based_on_correspondence', n_source=len(src), n_target=len(dst),
                      source_dtype=str(src.dtype), corres=corres.copy(),
                      max_correspondence_distance=float(max_correspondence_distance),
                      with_scaling=estimation_method.with_scaling, ransac_n=int(ransac_n), n_checkers=len(checkers),
                      max_iteration=criteria.max_iteration, confidence_given=criteria.confidence_given,
                      confidence=criteria.confidence, hypotheses_evaluated=n_hyp))
    # confidence 1.0 => the early exit of Open3D's loop never triggers: every hypothesis is evaluated
    assert criteria.confidence >= 1.0, 'the restatement has no early exit; the reference passes 80000 -> 1.0'
    T, h, count, rmse = open3d_reg.ransac_correspondence(src[corres[:, 0]], dst[corres[:, 1]],
                                                         max_correspondence_distance, n_hyp, seed=RANSAC_SEED,
                                                         ransac_n=ransac_n)
    CALLS[-1].update(hypothesis=int(h), inliers=int(count), inlier_rmse=float(rmse))
    return _RegistrationResult(T, count / max(len(corres), 1), rmse)


def _icp(source, target, max_correspondence_distance, init=None, estimation_method=None, criteria=None):
    from oracle import open3d_reg
    estimation_method = estimation_method or _TransformationEstimationPointToPoint(False)
    criteria = criteria or _ICPConvergenceCriteria()
    init = np.identity(4) if init is None else np.asarray(init, np.float64)
    src, dst = _points(source), _points(target)
    assert isinstance(estimation_method, _TransformationEstimationPointToPoint) and not estimation_method.with_scaling
    CALLS.append(dict(fn='registration_icp', n_source=len(src), n_target=len(dst), source_dtype=str(src.dtype),
                      source_head=src[:4].copy(), target_head=dst[:4].copy(),
                      max_correspondence_distance=float(max_correspondence_distance), init=init.copy(),
                      max_iteration=criteria.max_iteration, relative_fitness=criteria.relative_fitness,
                      relative_rmse=criteria.relative_rmse))
    T, fit, rmse, it = open3d_reg.icp_point_to_point(src, dst, max_correspondence_distance, init=init,
                                                     max_iter=criteria.max_iteration,
                                                     rel_fitness=criteria.relative_fitness,
                                                     rel_rmse=criteria.relative_rmse)
    CALLS[-1].update(iterations=int(it), fitness=float(fit), inlier_rmse=float(rmse))
    return _RegistrationResult(T, fit, rmse)


geometry = types.SimpleNamespace(PointCloud=_PointCloud)
utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a, np.float64),     # Eigen::Vector3d storage
                                Vector2iVector=lambda a: np.asarray(a, np.int32))
pipelines = types.SimpleNamespace(registration=types.SimpleNamespace(
    registration_icp=_icp, registration_ransac_based_on_correspondence=_ransac_based_on_correspondence,
    TransformationEstimationPointToPoint=_TransformationEstimationPointToPoint,
    RANSACConvergenceCriteria=_RANSACConvergenceCriteria, ICPConvergenceCriteria=_ICPConvergenceCriteria,
    RegistrationResult=_RegistrationResult))
