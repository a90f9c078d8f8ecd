"""Evaluation loop in the spirit of scripts/test_3dmatch.py:87-156: every pair of a trajectory dataset is
registered by each method, and success / RTE / RRE / time are accumulated per pair and per scene."""
import os
import time

import numpy as np

from .formats import load_cloud, read_trajectory
from .metrics import rte_rre


class ThreeDMatchTrajectory:
    """The test split layout read by `ThreeDMatchTrajectoryDataset` (dataloader/threedmatch_loader.py:144-196):
    `<root>/<scene>/cloud_bin_<i>.ply` fragments and `<root>/<scene>-evaluation/gt.log` with one record per
    overlapping pair (i, j) whose pose maps fragment j into fragment i.  Items are (scene, xyz_i, xyz_j, T_gt)."""

    def __init__(self, root, scenes=None):
        self.root = root
        if scenes is None:
            scenes = sorted(d[:-len('-evaluation')] for d in os.listdir(root) if d.endswith('-evaluation'))
        self.scenes = list(scenes)
        self.files = []
        for s in self.scenes:
            traj = os.path.join(root, s + '-evaluation', 'gt.log')
            if not os.path.exists(traj):
                raise FileNotFoundError(traj)
            for meta, pose in read_trajectory(traj):
                self.files.append((s, meta[0], meta[1], pose))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, k):
        s, i, j, T = self.files[k]
        return (s, load_cloud(os.path.join(self.root, s, f'cloud_bin_{i}.ply')),
                load_cloud(os.path.join(self.root, s, f'cloud_bin_{j}.ply')), T)


def analyze_stats(stats, mask, method_names, out=print):
    """Mean [success, RTE, RRE, time, scene id] over the evaluated pairs, and over the successful ones."""
    mask = np.asarray(mask).reshape(-1) > 0
    summary = {}
    for m, name in enumerate(method_names):
        s = stats[m][mask]
        ok = s[s[:, 0] > 0]
        summary[name] = {'pairs': int(len(s)), 'recall': float(s[:, 0].mean()) if len(s) else 0.0,
                         'mean': s.mean(0) if len(s) else np.zeros(stats.shape[2]),
                         'mean_successful': ok.mean(0) if len(ok) else np.zeros(stats.shape[2])}
        out(f'{name}: recall {summary[name]["recall"]:.4f} over {len(s)} pairs; successful pairs: '
            f'RTE {summary[name]["mean_successful"][1]:.4f} m, RRE {summary[name]["mean_successful"][2]:.3f} deg, '
            f'{summary[name]["mean_successful"][3]:.4f} s')
    return summary


def evaluate(methods, method_names, dataset, success_rte_thresh=0.3, success_rre_thresh=15.0, out=print,
             summary_every=0):
    """`methods` expose `.register(xyz0, xyz1) -> T [4,4]`.  The ground truth of a gt.log record maps the
    second fragment into the first, and `register` estimates first -> second, hence `T_gt = inv(pose)`
    (scripts/test_3dmatch.py:106).  Returns (stats [methods, pairs, 5], per-scene means, summary)."""
    n = len(dataset)
    scenes = list(getattr(dataset, 'scenes', []))
    stats = np.zeros((len(methods), n, 5))
    mask = np.zeros((n, 1), int)
    for k in range(n):
        sname, xyz0, xyz1, pose = dataset[k]
        if sname not in scenes:
            scenes.append(sname)
        T_gt = np.linalg.inv(pose)
        for m, method in enumerate(methods):
            t0 = time.time()
            T = method.register(xyz0, xyz1)
            stats[m, k, :3] = rte_rre(T, T_gt, success_rte_thresh, success_rre_thresh)
            stats[m, k, 3] = time.time() - t0
            stats[m, k, 4] = scenes.index(sname)
        mask[k] = 1
        if summary_every and k % summary_every == summary_every - 1:
            analyze_stats(stats, mask, method_names, out)
    summary = analyze_stats(stats, mask, method_names, out)
    scene_means = np.zeros((len(methods), len(scenes), 3))
    for m in range(len(methods)):
        for sid in range(len(scenes)):
            sel = stats[m, :, 4] == sid
            if sel.any():
                scene_means[m, sid] = stats[m, sel, :3].mean(0)
    return stats, scene_means, summary
