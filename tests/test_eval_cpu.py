"""Evaluation harness and file formats -- host-side, CPU only."""
import os

import numpy as np
import pytest

from deepglobalregistration_amd import eval as ev


def _pose(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    T = np.eye(4); T[:3, :3] = q; T[:3, 3] = rng.standard_normal(3)
    return T


def test_rte_rre_matches_reference_definition():
    import math
    import sys
    rng = np.random.default_rng(0)
    A, B = _pose(rng), _pose(rng)
    s, rte, rre = ev.rte_rre(A, B, 0.3, 15)
    assert rte == pytest.approx(np.linalg.norm(A[:3, 3] - B[:3, 3]))
    assert rre == pytest.approx(math.degrees(math.acos((np.trace(A[:3, :3].T @ B[:3, :3]) - 1) / 2)))
    assert s == 0
    np.testing.assert_array_equal(ev.rte_rre(None, B, 0.3, 15), [0, np.inf, np.inf])
    assert ev.rte_rre(B, B, 0.3, 15)[0] == 1 and ev.rte_rre(B, B, 0.3, 15)[2] < 1e-5
    try:   # the reference's own function, when the reference tree is present (not on the GPU box)
        sys.path.insert(0, '/root/reference/scripts')
        src = open('/root/reference/scripts/test_3dmatch.py').read()
    except OSError:
        return
    ns = {'np': np, 'math': math}
    start = src.index('def rte_rre'); end = src.index('def analyze_stats')
    exec(src[start:end], ns)
    np.testing.assert_allclose(ev.rte_rre(A, B, 0.3, 15), ns['rte_rre'](A, B, 0.3, 15), rtol=1e-12)


def test_trajectory_kitti_and_ply_round_trips(tmp_path):
    rng = np.random.default_rng(1)
    recs = [([0, 1, 37], _pose(rng)), ([5, 12, 37], _pose(rng))]
    ev.write_trajectory(tmp_path / 'gt.log', recs)
    back = ev.read_trajectory(tmp_path / 'gt.log')
    assert [m for m, _ in back] == [m for m, _ in recs]
    for (_, a), (_, b) in zip(back, recs):
        np.testing.assert_array_equal(a, b)
    # the layout the reference parses: tab / space separated, 4 rows per record
    (tmp_path / 'hand.log').write_text('0\t 1 \t 2\n1 0 0 0.5\n0 1 0 0\n 0 0 1 0\n0 0 0 1\n')
    (meta, pose), = ev.read_trajectory(tmp_path / 'hand.log')
    assert meta == [0, 1, 2] and pose[0, 3] == 0.5
    with pytest.raises(ValueError):
        (tmp_path / 'bad.log').write_text('0 1 2\n1 0 0 0\n')
        ev.read_trajectory(tmp_path / 'bad.log')
    xyz = rng.standard_normal((100, 3)).astype(np.float32)
    ev.write_kitti_bin(tmp_path / 'a.bin', xyz, rng.random(100))
    np.testing.assert_array_equal(ev.read_kitti_bin(tmp_path / 'a.bin'), xyz)
    pts = rng.standard_normal((57, 3))
    for binary in (True, False):
        ev.write_ply(tmp_path / 'c.ply', pts, binary=binary)
        np.testing.assert_array_equal(ev.read_ply(tmp_path / 'c.ply'), pts)
        np.testing.assert_array_equal(ev.load_cloud(str(tmp_path / 'c.ply')), pts)
    # float vertices with extra properties and a face element, as 3DMatch fragments have
    v = np.zeros(3, dtype=[('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('red', 'u1'), ('nx', '<f4')])
    v['x'], v['y'], v['z'] = [1, 2, 3], [4, 5, 6], [7, 8, 9]
    hdr = (b'ply\nformat binary_little_endian 1.0\ncomment made by hand\nelement vertex 3\nproperty float x\n'
           b'property float y\nproperty float z\nproperty uchar red\nproperty float nx\nelement face 0\n'
           b'property list uchar int vertex_indices\nend_header\n')
    (tmp_path / 'f.ply').write_bytes(hdr + v.tobytes())
    np.testing.assert_array_equal(ev.read_ply(tmp_path / 'f.ply'), [[1, 4, 7], [2, 5, 8], [3, 6, 9]])
    with pytest.raises(ValueError):
        ev.load_cloud('cloud.pcd')


class _Oracle:
    """Stand-in method: returns the ground truth for even pairs and the identity for odd ones."""

    def __init__(self, answers):
        self.answers, self.k = answers, 0

    def register(self, xyz0, xyz1):
        T = self.answers[self.k]
        self.k += 1
        return T


def test_harness_on_a_synthetic_trajectory_dataset(tmp_path):
    rng = np.random.default_rng(2)
    root = tmp_path / 'threedmatch'
    poses = {}
    for s in ('kitchen', 'lab'):
        (root / s).mkdir(parents=True)
        (root / f'{s}-evaluation').mkdir()
        for i in range(3):
            ev.write_ply(root / s / f'cloud_bin_{i}.ply', rng.standard_normal((40, 3)))
        recs = [([0, 1, 3], _pose(rng)), ([1, 2, 3], _pose(rng))]
        poses[s] = recs
        ev.write_trajectory(root / f'{s}-evaluation' / 'gt.log', recs)
    ds = ev.ThreeDMatchTrajectory(str(root))
    assert len(ds) == 4 and ds.scenes == ['kitchen', 'lab']
    s, a, b, T = ds[3]
    assert s == 'lab' and a.shape == (40, 3) and np.array_equal(T, poses['lab'][1][1])
    answers = [np.linalg.inv(ds.files[k][3]) if k % 2 == 0 else np.eye(4) for k in range(4)]
    lines = []
    stats, scene_means, summary = ev.evaluate([_Oracle(answers)], ['stub'], ds, out=lines.append)
    assert stats.shape == (1, 4, 5)
    np.testing.assert_array_equal(stats[0, :, 0], [1, 0, 1, 0])
    np.testing.assert_array_equal(stats[0, :, 4], [0, 0, 1, 1])
    assert summary['stub']['recall'] == 0.5 and summary['stub']['mean_successful'][1] < 1e-9
    assert scene_means.shape == (1, 2, 3) and scene_means[0, 0, 0] == 0.5 and lines


def test_kitti_pairs_ground_truth_and_loop(tmp_path):
    """A synthetic mini odometry sequence: a static world seen from a vehicle driving 2.5 m per frame."""
    rng = np.random.default_rng(3)
    root = tmp_path / 'dataset'
    (root / 'sequences' / '03' / 'velodyne').mkdir(parents=True)
    (root / 'poses').mkdir()
    world = rng.uniform(-60, 60, (4000, 3)); world[:, 2] = rng.uniform(-2, 3, 4000)
    V = ev.kitti.VELO2CAM
    poses = []
    for f in range(12):
        # camera-0 pose: drive along the camera z axis with a slow yaw about the camera y axis
        a = 0.02 * f
        P = np.eye(4)
        P[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        P[:3, 3] = [0.1 * f, 0.0, 2.5 * f]
        poses.append(P)
        velo_T_world = np.linalg.inv(P @ V)            # world -> velodyne frame f (world = camera-0 frame of frame 0)
        pts = world @ velo_T_world[:3, :3].T + velo_T_world[:3, 3]
        ev.write_kitti_bin(root / 'sequences' / '03' / 'velodyne' / f'{f:06d}.bin', pts)
    np.savetxt(root / 'poses' / '03.txt', np.array([P[:3].reshape(-1) for P in poses]))
    ds = ev.KITTIOdometryPairs(str(root), [3])
    # 10 m is first exceeded 4 frames later (10.008 m); the 3DFeatNet rule takes the frame before that one
    assert ds.files == [(3, 0, 3), (3, 4, 7)]        # frames 8..11 never get 10 m apart
    drive, x0, x1, T = ds[0]
    assert drive == 3 and x0.dtype == np.float32 and x0.shape == (4000, 3)
    np.testing.assert_allclose(x0.astype(np.float64) @ T[:3, :3].T + T[:3, 3], x1, atol=2e-4)   # x1 = T x0
    np.testing.assert_allclose(T, ev.relative_velodyne_pose(poses[0], poses[3]), atol=1e-12)
    # an ICP hook is applied the way the reference caches its refined poses (M @ reg)
    nudge = np.eye(4); nudge[0, 3] = 0.01
    ds2 = ev.KITTIOdometryPairs(str(root), [3], icp_refine=lambda s, d, init: nudge)
    np.testing.assert_allclose(ds2[0][3], T @ nudge, atol=1e-12)

    class Perfect:
        def register(self, a, b):
            return T_of[(len(a), float(a[0, 0]))]
    T_of = {(len(ds[k][1]), float(ds[k][1][0, 0])): ds[k][3] for k in range(len(ds))}
    lines = []
    stats, summary = ev.evaluate_kitti(Perfect(), ds, out=lines.append)
    assert stats.shape == (2, 5) and summary['recall'] == 1.0 and (stats[:, 4] == 3).all() and lines
    with pytest.raises(FileNotFoundError):
        ev.KITTIOdometryPairs(str(root), [7])


def test_trajectory_reader_and_rte_rre_match_the_reference(golden):
    """`read_trajectory` (util/file.py:69-90) and `rte_rre` (scripts/test_3dmatch.py:38-46): the reference's own
    functions were run on a generated gt.log / pose set by tests/golden/make_golden.py; the product's reader and
    metric must reproduce their outputs on the same inputs."""
    from deepglobalregistration_amd.eval import formats, metrics
    g = golden('eval_formats')
    path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'dgr_gt_golden.log')
    with open(path, 'w') as f:
        f.write(str(g['gt_log_text']))
    traj = formats.read_trajectory(path)
    os.unlink(path)
    assert len(traj) == len(g['traj_meta'])
    for rec, meta, pose in zip(traj, g['traj_meta'], g['traj_pose']):
        m, T = (rec.metadata, rec.pose) if hasattr(rec, 'pose') else rec
        np.testing.assert_array_equal(np.asarray(m), meta)
        np.testing.assert_array_equal(np.asarray(T), pose)        # both parse the same text: bit-identical
    np.testing.assert_allclose(g['traj_pose'], g['poses_written'], rtol=1e-8)
    for Tp, Tg, ref in zip(g['T_pred'], g['T_gt'], g['rte_rre']):
        np.testing.assert_allclose(metrics.rte_rre(Tp, Tg, 0.3, 15), ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(metrics.rte_rre(None, g['T_gt'][0], 0.3, 15), g['rte_rre_none'])
