"""Evaluation harness and on-disk formats around `DeepGlobalRegistration.register()`:
the metric of scripts/test_3dmatch.py:38-46, the `gt.log` trajectory files (util/file.py:69-90), KITTI
velodyne `.bin` scans (dataloader/kitti_loader.py:132-133) and `.ply` fragments (read by Open3D in the
reference, dataloader/threedmatch_loader.py:192-195).  Host-side Python like the reference's; no GPU code."""
from .formats import (load_cloud, read_kitti_bin, read_ply, read_trajectory, write_kitti_bin, write_ply,
                      write_trajectory)
from .harness import ThreeDMatchTrajectory, analyze_stats, evaluate
from .kitti import KITTIOdometryPairs, evaluate_kitti, relative_velodyne_pose
from .metrics import rte_rre

__all__ = ['rte_rre', 'read_trajectory', 'write_trajectory', 'read_kitti_bin', 'write_kitti_bin', 'read_ply',
           'write_ply', 'load_cloud', 'ThreeDMatchTrajectory', 'evaluate', 'analyze_stats', 'KITTIOdometryPairs',
           'evaluate_kitti', 'relative_velodyne_pose']
