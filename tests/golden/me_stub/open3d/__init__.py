"""A stand-in for `import open3d` so that `core/deep_global_registration.py` and `util/pointcloud.py` of the reference
IMPORT in this container (tests/golden/make_golden_register.py).  It has the classes those modules name at import time
and in `preprocess`'s isinstance test; the two Open3D ALGORITHMS of the path (RANSAC on correspondences, ICP) are not
restated here -- the golden run takes the learned branch with `use_icp = False`, and calling either raises."""
import types


class _PointCloud:
    def __init__(self):
        self.points = None


def _absent(*a, **k):
    raise NotImplementedError('Open3D is not installed in this container; the stand-in has no algorithms')


geometry = types.SimpleNamespace(PointCloud=_PointCloud)
utility = types.SimpleNamespace(Vector3dVector=lambda a: a, Vector2iVector=lambda a: a)
pipelines = types.SimpleNamespace(registration=types.SimpleNamespace(
    registration_icp=_absent, registration_ransac_based_on_correspondence=_absent,
    TransformationEstimationPointToPoint=_absent, RANSACConvergenceCriteria=_absent))
