"""`oracle.pipeline.register` -- the CPU restatement of `register()` (core/deep_global_registration.py:238-324) that
tests/test_gpu_register_e2e.py holds the HIP path to -- executed on the CPU on a small pair through both of its
branches: it has to be deterministic, take the branch the gate dictates, and find the pose when the harness gives it
ground-truth matches / logits (untrained synthetic weights give neither)."""
import numpy as np
import pytest

from conftest import rot_angle_deg
from oracle import pipeline as opipe

VOXEL = 0.05


@pytest.fixture(scope='module')
def case():
    from deepglobalregistration_amd import synth
    ck = synth.synth_checkpoint(seed=0, voxel_size=VOXEL, feat_conv1_kernel_size=3)
    x0, x1, T_gt = synth.synth_pair(6, n_raw=2500)
    return ck, x0, x1, T_gt


def _gt_matches(T_gt, seed):
    from deepglobalregistration_amd import synth
    return lambda p0, p1, F0, F1, idx1: np.where((g := synth.gt_correspondences(p0, p1, T_gt, VOXEL, frac=1.0, seed=seed)) >= 0, g, idx1)


def test_safeguard_branch_is_deterministic_and_finds_the_pose(case):
    ck, x0, x1, T_gt = case
    kw = dict(clip_weight_thresh=0.97, ransac_hypotheses=3000, ransac_seed=5, idx1_fn=_gt_matches(T_gt, 6))
    a = opipe.register(ck, x0, x1, **kw)
    b = opipe.register(ck, x0, x1, **kw)
    assert a['status'] == 'safeguard' and not a['confident'] and a['wsum'] < a['wsum_threshold']
    assert a['ransac'] == b['ransac'] and np.array_equal(a['T'], b['T'])          # counter-based sampler: reproducible
    assert a['T'].dtype == np.float64 and np.array_equal(a['T'][3], [0, 0, 0, 1])
    assert a['ransac']['inliers'] > 0.2 * len(a['idx1']) and 0 <= a['ransac']['hypothesis'] < 3000
    assert a['icp']['iterations'] >= 1 and a['icp']['fitness'] > 0.1
    assert rot_angle_deg(a['T'][:3, :3], T_gt[:3, :3]) < 3.0 and np.linalg.norm(a['T'][:3, 3] - T_gt[:3, 3]) < 0.15
    # another seed draws other hypotheses; without ICP the RANSAC estimate itself is returned
    c = opipe.register(ck, x0, x1, **dict(kw, ransac_seed=6, use_icp=False))
    assert c['ransac']['hypothesis'] != a['ransac']['hypothesis'] or not np.array_equal(c['T'], a['T_before_icp'])
    assert np.array_equal(c['T'], c['T_before_icp']) and 'icp' not in c


def test_learned_branch_and_the_low_confidence_exit(case):
    from deepglobalregistration_amd import synth
    ck, x0, x1, T_gt = case
    forced = lambda xa, xb, logit: synth.gt_forced_logits(xa, xb, T_gt, VOXEL)
    o = opipe.register(ck, x0, x1, idx1_fn=_gt_matches(T_gt, 6), forced_logit_fn=forced, use_icp=False)
    assert o['status'] == 'ok' and o['confident'] and o['wsum'] >= o['wsum_threshold']
    assert o['stats']['iterations'] >= 1 and np.array_equal(o['T'], o['T_before_icp'])
    assert np.array_equal(o['T'][:3, :3], np.asarray(o['R'], np.float64)) and o['logit_net'].shape == o['logit'].shape
    assert rot_angle_deg(o['T'][:3, :3], T_gt[:3, :3]) < 2.0 and np.linalg.norm(o['T'][:3, 3] - T_gt[:3, 3]) < 0.1
    # gate fails and no safeguard asked for: the identity, like the reference before it calls its safeguard
    n = opipe.register(ck, x0, x1, clip_weight_thresh=0.97, safeguard=False, use_icp=False)
    assert n['status'] == 'low_confidence' and np.array_equal(n['T'], np.identity(4))
