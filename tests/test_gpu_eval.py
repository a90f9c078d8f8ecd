"""The evaluation harness driven through the HIP `register()` on the GPU.

`python -m deepglobalregistration_amd.eval` is run as the user would run it -- on a synthetic 3DMatch-layout directory
(`<scene>/cloud_bin_<i>.ply` + `<scene>-evaluation/gt.log`, scripts/test_3dmatch.py:87-156 /
dataloader/threedmatch_loader.py:144-196) and on a KITTI-odometry-layout one (`sequences/<d>/velodyne/*.bin` +
`poses/<d>.txt`, scripts/test_kitti.py:59-119 / dataloader/kitti_loader.py:132-158, ground truth refined by GPU ICP)
-- with a checkpoint FILE; the per-pair stats rows it saves must equal what direct `register()` calls on the same
files give (success, RTE, RRE bit for bit: both go through the same library and the same readers), and the scene
bookkeeping must be right.  The synthetic weights make the poses themselves meaningless; the pairs take the safeguard
(RANSAC) or the refinement branch as the untrained confidence decides, ICP included (`use_icp` default)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from deepglobalregistration_amd import eval as ev, synth
from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_cli(cwd, *flags):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, '-m', 'deepglobalregistration_amd.eval', *flags], env=env, cwd=cwd,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_threedmatch_cli_rows_equal_direct_register_calls(tmp_path):
    root = tmp_path / 'threedmatch'
    rng = np.random.default_rng(0)
    for si, s in enumerate(('7-scenes-redkitchen', 'sun3d-home_at')):
        (root / s).mkdir(parents=True)
        (root / f'{s}-evaluation').mkdir()
        a, b, T = synth.synth_pair(10 + si, n_raw=6000)
        c = a[rng.permutation(len(a))[:5000]] + rng.normal(scale=0.002, size=(5000, 3))
        for i, pts in enumerate((a, b, c)):
            ev.write_ply(root / s / f'cloud_bin_{i}.ply', pts)
        # gt.log poses map fragment j into fragment i (util/file.py:69-90); b = T a, c ~ a
        ev.write_trajectory(root / f'{s}-evaluation' / 'gt.log', [([0, 1, 3], np.linalg.inv(T)), ([0, 2, 3], np.eye(4))])
    ckpt = tmp_path / 'dgr.pth'
    torch.save(synth.synth_checkpoint(seed=3, voxel_size=0.05, feat_conv1_kernel_size=5), ckpt)
    out = tmp_path / 'stats.npz'
    text = _run_cli(str(tmp_path), '--threed_match_dir', str(root), '--weights', str(ckpt), '--out', str(out))
    res = np.load(out, allow_pickle=True)
    stats = res['stats']
    assert stats.shape == (1, 4, 5) and list(res['scenes']) == ['7-scenes-redkitchen', 'sun3d-home_at']
    np.testing.assert_array_equal(stats[0, :, 4], [0, 0, 1, 1])
    assert 'scene average' in text and (stats[0, :, 3] > 0).all()
    # the same pairs through direct register() calls in this process
    dgr = DeepGlobalRegistration({'weights': str(ckpt)}, torch.device('cuda'))
    ds = ev.ThreeDMatchTrajectory(str(root))
    for k in range(len(ds)):
        _, x0, x1, pose = ds[k]
        T = dgr.register(x0, x1)
        row = ev.rte_rre(T, np.linalg.inv(pose), 0.3, 15.0)
        np.testing.assert_array_equal(stats[0, k, :3], row)
    np.testing.assert_allclose(res['scene_means'][0, 0], stats[0, :2, :3].mean(0))


def test_kitti_cli_rows_equal_direct_register_calls(tmp_path):
    root = tmp_path / 'dataset'
    (root / 'sequences' / '08' / 'velodyne').mkdir(parents=True)
    (root / 'poses').mkdir()
    a, _, _ = synth.synth_pair(5, n_raw=20000, kind='outdoor')      # one LiDAR-shaped scan = the static world
    V = ev.kitti.VELO2CAM
    poses = []
    for f in range(8):
        P = np.eye(4)
        ang = 0.01 * f
        P[:3, :3] = [[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]
        P[:3, 3] = [0.05 * f, 0.0, 2.6 * f]
        poses.append(P)
        w2v = np.linalg.inv(P @ V) @ (poses[0] @ V)                  # velodyne frame 0 -> velodyne frame f
        ev.write_kitti_bin(root / 'sequences' / '08' / 'velodyne' / f'{f:06d}.bin', a @ w2v[:3, :3].T + w2v[:3, 3])
    np.savetxt(root / 'poses' / '08.txt', np.array([P[:3].reshape(-1) for P in poses]))
    ckpt = tmp_path / 'dgr_kitti.pth'
    torch.save(synth.synth_checkpoint(seed=4, voxel_size=0.3, feat_conv1_kernel_size=5), ckpt)
    out = tmp_path / 'kitti.npz'
    text = _run_cli(str(tmp_path), '--kitti_dir', str(root), '--drives', '8', '--weights', str(ckpt), '--out', str(out))
    stats = np.load(out)['stats']
    ds = ev.KITTIOdometryPairs(str(root), [8])
    assert len(ds) >= 1 and stats.shape == (len(ds), 5) and (stats[:, 4] == 8).all() and 'KITTI: recall' in text
    dgr = DeepGlobalRegistration({'weights': str(ckpt)}, torch.device('cuda'))
    from deepglobalregistration_amd import ops

    def refine(xyz0, xyz1, M):   # the hook of eval/__main__.py (kitti_loader.py:139-158)
        s5, _, _ = ops.voxelize(xyz0, 0.05)
        d5, _, _ = ops.voxelize(xyz1, 0.05)
        Mt = torch.as_tensor(M, dtype=torch.float32, device=s5.device)
        return ops.icp_point_to_point(s5 @ Mt[:3, :3].T + Mt[:3, 3], d5, 0.2, init=np.eye(4), max_iter=200)[0]
    ds = ev.KITTIOdometryPairs(str(root), [8], icp_refine=refine)
    for k in range(len(ds)):
        _, x0, x1, T_gt = ds[k]
        np.testing.assert_array_equal(stats[k, :3], ev.rte_rre(dgr.register(x0, x1), T_gt, 0.6, 5.0))
    # the ICP-refined ground truth stays at the odometry pose of this noise-free static world
    np.testing.assert_allclose(ds[0][3], ev.relative_velodyne_pose(poses[ds.files[0][1]], poses[ds.files[0][2]]), atol=5e-3)
