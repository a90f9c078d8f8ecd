"""Registration error metrics of the reference's evaluation scripts."""
import numpy as np


def rte_rre(T_pred, T_gt, rte_thresh, rre_thresh, eps=1e-16):
    """scripts/test_3dmatch.py:38-46 (same in scripts/test_kitti.py): returns [success, RTE (m), RRE (deg)];
    a missing estimate counts as a failure with infinite errors."""
    if T_pred is None:
        return np.array([0, np.inf, np.inf])
    T_pred, T_gt = np.asarray(T_pred, np.float64), np.asarray(T_gt, np.float64)
    rte = float(np.linalg.norm(T_pred[:3, 3] - T_gt[:3, 3]))
    cos = (np.trace(T_pred[:3, :3].T @ T_gt[:3, :3]) - 1.0) / 2.0
    rre = float(np.degrees(np.arccos(np.clip(cos, -1 + eps, 1 - eps))))
    return np.array([float(rte < rte_thresh and rre < rre_thresh), rte, rre])
