"""KITTI odometry pairs and the evaluation loop of scripts/test_kitti.py:59-119.

Layout (dataloader/kitti_loader.py:34, 120, 74): `<root>/sequences/<DD>/velodyne/<NNNNNN>.bin` (float32 x, y, z,
reflectance) and `<root>/poses/<DD>.txt` (one 3x4 camera-0 pose per frame, row-major).  Pairs are frames at
least `min_dist` metres apart (KITTINMPairDataset, :233-282, following 3DFeatNet); the ground truth is the
relative pose expressed in the velodyne frame with the fixed velodyne-to-camera calibration the reference hard
codes (:66-77), optionally refined by ICP on 5 cm voxels like the reference's cached `icp/` poses (:139-160)."""
import glob
import os
import time

import numpy as np

from .formats import read_kitti_bin
from .metrics import rte_rre

# dataloader/kitti_loader.py:69-75 (the constants of the reference, not a per-sequence calib.txt)
VELO2CAM = np.eye(4)
VELO2CAM[:3, :3] = np.array([7.533745e-03, -9.999714e-01, -6.166020e-04, 1.480249e-02, 7.280733e-04, -9.998902e-01,
                             9.998621e-01, 7.523790e-03, 1.480755e-02]).reshape(3, 3)
VELO2CAM[:3, 3] = [-4.069766e-03, -7.631618e-02, -2.717806e-01]


def relative_velodyne_pose(pose0, pose1):
    """4x4 transform taking velodyne-frame points of frame 0 into the velodyne frame of frame 1 from the two
    camera-0 poses: inv(V) inv(P1) P0 V -- the matrix `M` of kitti_loader.py:146-147."""
    return np.linalg.inv(VELO2CAM) @ np.linalg.inv(pose1) @ pose0 @ VELO2CAM


class KITTIOdometryPairs:
    """Items are (drive, xyz0 [N,3] f32, xyz1 [M,3] f32, T_gt [4,4] with x1 = T_gt x0)."""
    MIN_DIST = 10.0

    def __init__(self, root, drives, min_dist=None, icp_refine=None, exclude=((8, 15, 58),)):
        """`icp_refine(xyz0, xyz1, M) -> reg` gets the UNTRANSFORMED scans and the odometry pose M; it is expected to
        do what kitti_loader.py:139-154 does -- subsample both at 5 cm, move the selected points of cloud 0 by M,
        run ICP from the identity (max distance 0.2, 200 iterations) and return the ICP transformation -- or None
        for the raw odometry ground truth."""
        self.root = root
        self.min_dist = self.MIN_DIST if min_dist is None else float(min_dist)
        self.icp_refine = icp_refine
        self.files, self.poses = [], {}
        for drive in drives:
            drive = int(drive)
            names = glob.glob(os.path.join(root, 'sequences', f'{drive:02d}', 'velodyne', '*.bin'))
            if not names:
                raise FileNotFoundError(f'no scans for drive {drive} under {root}')
            frames = sorted(int(os.path.basename(n)[:-4]) for n in names)
            P = np.loadtxt(os.path.join(root, 'poses', f'{drive:02d}.txt')).reshape(-1, 3, 4)
            P = np.concatenate([P, np.tile([[[0, 0, 0, 1.0]]], (len(P), 1, 1))], axis=1)
            self.poses[drive] = P
            pos = P[:, :3, 3]
            have = set(frames)
            cur = frames[0]
            while cur in have:
                # first later frame (within the next 100) farther than min_dist, minus one: the 3DFeatNet rule
                d = np.linalg.norm(pos[cur:cur + 100] - pos[cur], axis=1)
                far = np.nonzero(d > self.min_dist)[0]
                if len(far) == 0:
                    cur += 1
                    continue
                nxt = int(far[0]) + cur - 1
                if nxt in have and (drive, cur, nxt) not in exclude:
                    self.files.append((drive, cur, nxt))
                cur = nxt + 1

    def __len__(self):
        return len(self.files)

    def scan(self, drive, t):
        return read_kitti_bin(os.path.join(self.root, 'sequences', f'{drive:02d}', 'velodyne', f'{t:06d}.bin'))

    def __getitem__(self, k):
        drive, t0, t1 = self.files[k]
        xyz0, xyz1 = self.scan(drive, t0), self.scan(drive, t1)
        M = relative_velodyne_pose(self.poses[drive][t0], self.poses[drive][t1])
        if self.icp_refine is not None:
            # the reference subsamples cloud 0, moves it by M, runs ICP from the identity and stores M @ reg (:142-158)
            reg = self.icp_refine(xyz0, xyz1, M)
            M = M @ reg
        return drive, xyz0, xyz1, M


def evaluate_kitti(method, dataset, te_thresh=0.6, re_thresh=5.0, out=print, log_every=10):
    """scripts/test_kitti.py:59-103: stats [N,5] = (success, RTE, RRE, time, drive); success = RTE < 0.6 m and
    RRE < 5 deg (:33-34).  The time is the method's own feature + registration timers when it has them."""
    n = len(dataset)
    stats = np.zeros((n, 5))
    for i in range(n):
        drive, xyz0, xyz1, T_gt = dataset[i]
        t0 = time.time()
        T = method.register(xyz0, xyz1)
        elapsed = time.time() - t0
        stats[i, :3] = rte_rre(T, T_gt, te_thresh, re_thresh)
        ft, rt = getattr(method, 'feat_timer', None), getattr(method, 'reg_timer', None)
        stats[i, 3] = (ft.diff + rt.diff) if (ft is not None and rt is not None) else elapsed
        stats[i, 4] = drive
        if log_every and i % log_every == 0:
            s = stats[:i + 1].mean(0)
            out(f'{i} / {n}: RTE {s[1]:.4f}, RRE {s[2]:.3f}, success {100 * s[0]:.1f} %, {s[3]:.3f} s/pair')
    ok = stats[stats[:, 0] > 0]
    summary = {'pairs': n, 'recall': float(stats[:, 0].mean()) if n else 0.0,
               'mean': stats.mean(0) if n else np.zeros(5), 'mean_successful': ok.mean(0) if len(ok) else np.zeros(5)}
    out(f'KITTI: recall {summary["recall"]:.4f} over {n} pairs; successful: RTE {summary["mean_successful"][1]:.4f} m, '
        f'RRE {summary["mean_successful"][2]:.3f} deg')
    return stats, summary
