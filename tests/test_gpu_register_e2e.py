"""End-to-end parity of `DeepGlobalRegistration.register(xyz0, xyz1)` (core/deep_global_registration.py:238-324): ONE 4x4 of
the HIP path against ONE 4x4 of `oracle.pipeline.register` from the same raw points -- own voxels -> own features ->
own matches -> own logits -> gate -> refinement OR safeguard RANSAC -> ICP -> float64 transform, nothing teacher-forced
between the stages except where the docstrings say so (and then by the same function on both sides).

The only tolerance for a discrete disagreement: a 1-NN result may differ from the oracle's where the two candidates
are a rounding tie on the oracle's own features (the HIP features differ from the oracle's by ~1e-6); the oracle then
continues from the HIP matches.  On the pairs below no row needs it, and the tests say so when that changes."""
import numpy as np
import pytest
import torch

from conftest import rot_angle_deg
from helpers import assert_refine_parity
from oracle import knn as oknn
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu
VOXEL = 0.05
HYP = 20000          # the reference hard-codes 4 000 000 hypotheses (:61); both sides get a count the CPU oracle finishes


@pytest.fixture(scope='module')
def ck():
    from deepglobalregistration_amd import synth
    return synth.synth_checkpoint(seed=0, voxel_size=VOXEL, feat_conv1_kernel_size=7)


def _tie_checked(hip_idx1, note):
    """idx1_fn of the oracle: accept the HIP matches where they differ from the oracle's own search only by rounding ties."""
    def fn(p0, p1, F0, F1, idx1):
        mism = np.nonzero(idx1 != hip_idx1)[0]
        note['mismatches'] = len(mism)
        if len(mism):
            assert len(mism) <= 0.01 * len(idx1), len(mism)
            d_a = oknn.knn_sqdist_f64(F0[mism], F1, hip_idx1[mism])
            d_b = oknn.knn_sqdist_f64(F0[mism], F1, idx1[mism])
            assert np.all(np.abs(d_a - d_b) <= 1e-4), np.abs(d_a - d_b).max()
        return hip_idx1
    return fn


def _gt_matches(T_gt, seed):
    """Harness: a share of the matches replaced by ground-truth ones AFTER the search (untrained descriptors are not
    repeatable across views) -- one numpy function for both sides."""
    from deepglobalregistration_amd import synth

    def np_fn(p0, p1, idx1):
        g = synth.gt_correspondences(p0, p1, T_gt, VOXEL, seed=seed)
        return np.where(g >= 0, g, idx1)

    def dev_fn(x0, x1, idx1):
        return torch.from_numpy(np_fn(x0.cpu().numpy(), x1.cpu().numpy(), idx1.cpu().numpy())).to(x0.device)
    return np_fn, dev_fn


def _dgr(ck, **cfg):
    from helpers import harness_dgr
    return harness_dgr(dict({'weights': ck, 'ransac_max_iteration': HYP, 'ransac_seed': 3}, **cfg), torch.device('cuda'))


@pytest.mark.parametrize('gt_share', [False, True])
def test_register_safeguard_branch_matches_oracle(ck, gt_share):
    """Gate fails -> safeguard RANSAC over the pair's own putative correspondences -> ICP (:302-322).  Nothing is forced
    in the first case (untrained weights: ~0 % correct matches, RANSAC returns its best consensus of garbage -- still one
    deterministic hypothesis on both sides); in the second a share of the matches is ground truth, so the branch also
    has to find the pose.  `clip_weight_thresh` = 0.97 (a runtime config key, config.py:63) zeroes every weight of the
    untrained inlier net: the gate fails without touching the logits."""
    from deepglobalregistration_amd import synth
    x0, x1, T_gt = synth.synth_pair(2, n_raw=6000)
    dgr = _dgr(ck, clip_weight_thresh=0.97)
    assert dgr.use_icp
    np_fn, dev_fn = _gt_matches(T_gt, seed=2)
    if gt_share:
        dgr.harness_matches = dev_fn
    T = dgr.register(x0, x1)
    assert T.shape == (4, 4) and T.dtype == np.float64 and dgr.last_status == 'safeguard'
    hip_idx1 = dgr.last_corres_idx1.cpu().numpy()
    note = {}
    tie = _tie_checked(hip_idx1, note)
    o = opipe.register(ck, x0, x1, clip_weight_thresh=0.97, ransac_hypotheses=HYP, ransac_seed=3,
                       idx1_fn=(lambda p0, p1, F0, F1, idx1: tie(p0, p1, F0, F1, np_fn(p0, p1, idx1) if gt_share else idx1)))
    assert o['status'] == 'safeguard' and not o['confident']
    # own logits against the oracle's own logits (the 6-D net on the pair's own correspondences)
    dl = np.abs(dgr.last_logit.cpu().numpy().reshape(-1) - o['logit_net'].reshape(-1)).max() / max(1.0, np.abs(o['logit_net']).max())
    assert dl < 1e-4, dl
    # the same hypothesis wins with the same consensus, the same ICP follows
    assert dgr.last_stats['ransac_hypothesis'] == o['ransac']['hypothesis']
    assert dgr.last_stats['ransac_inliers'] == o['ransac']['inliers']
    assert dgr.last_icp['iterations'] == o['icp']['iterations']
    assert abs(dgr.last_icp['fitness'] - o['icp']['fitness']) < 1e-12
    np.testing.assert_allclose(T, o['T'], atol=1e-6, rtol=0)
    print(f'e2e safeguard (gt_share={gt_share}): N0={len(hip_idx1)} 1-NN tie mismatches {note["mismatches"]}, dlogit {dl:.1e}, '
          f'hypothesis {o["ransac"]["hypothesis"]} inliers {o["ransac"]["inliers"]}, ICP {o["icp"]["iterations"]} it, '
          f'max|T - T_oracle| {np.abs(T - o["T"]).max():.1e}')
    if gt_share:
        assert rot_angle_deg(T[:3, :3], T_gt[:3, :3]) < 2.0 and np.linalg.norm(T[:3, 3] - T_gt[:3, 3]) < 0.1


def test_register_learned_branch_matches_oracle(ck):
    """Gate passes -> weighted Procrustes + SE(3) refinement -> ICP (:283-300, 317-322).  Harness (same functions on both
    sides): a share of ground-truth matches after the search, logits +-4 from the ground truth after the inlier net --
    the networks still run and their own outputs are compared before they are replaced.  The refinement is free-running on
    both sides (each stops by its own counter), so T before ICP is held to the band of tests/helpers.assert_refine_parity;
    ICP then re-estimates T from the nearest-neighbour correspondences alone, which pulls the two sides together again."""
    from deepglobalregistration_amd import ops, synth
    x0, x1, T_gt = synth.synth_pair(4, n_raw=6000)
    dgr = _dgr(ck, clip_weight_thresh=0.05)
    np_fn, dev_fn = _gt_matches(T_gt, seed=4)
    net_logit = {}

    def dev_logits(xa, xb, logit):
        net_logit['hip'] = logit.detach().cpu().numpy().reshape(-1)
        return torch.from_numpy(synth.gt_forced_logits(xa.cpu().numpy(), xb.cpu().numpy(), T_gt, VOXEL)).to(xa.device)
    dgr.harness_matches, dgr.harness_logits = dev_fn, dev_logits
    dgr.use_icp = False
    T_reg = dgr.register(x0, x1)                      # the learned estimate alone ...
    assert dgr.last_status == 'ok'
    st = dict(dgr.last_stats)
    dgr.use_icp = True
    T = dgr.register(x0, x1)                          # ... and register() as shipped
    hip_idx1 = dgr.last_corres_idx1.cpu().numpy()
    note = {}
    tie = _tie_checked(hip_idx1, note)
    o = opipe.register(ck, x0, x1, clip_weight_thresh=0.05,
                       idx1_fn=lambda p0, p1, F0, F1, idx1: tie(p0, p1, F0, F1, np_fn(p0, p1, idx1)),
                       forced_logit_fn=lambda xa, xb, logit: synth.gt_forced_logits(xa, xb, T_gt, VOXEL))
    assert o['status'] == 'ok' and o['confident']
    dl = np.abs(net_logit['hip'] - o['logit_net'].reshape(-1)).max() / max(1.0, np.abs(o['logit_net']).max())
    assert dl < 1e-4, dl
    # before ICP: the free-running refinement against the oracle's (1e-4, or the reference's own terminal band)
    X, Y = o['xyz0'], o['xyz1'][o['idx1']]
    assert_refine_parity(X, Y, o['weights'], T_reg[:3, :3], T_reg[:3, 3], st, break_threshold_ratio=1e-4,
                         quantization_size=2 * VOXEL)
    # after ICP: one 4x4 against one 4x4
    assert abs(dgr.last_icp['fitness'] - o['icp']['fitness']) < 2e-3
    dT = np.abs(T - o['T']).max()
    print(f'e2e learned branch: N0={len(hip_idx1)} tie mismatches {note["mismatches"]}, dlogit {dl:.1e}, refinement iterations hip '
          f'{st["iterations"]} oracle {o["stats"]["iterations"]}, max|T - T_oracle| before ICP '
          f'{np.abs(T_reg - o["T_before_icp"]).max():.1e}, after ICP {dT:.1e} (ICP iterations {dgr.last_icp["iterations"]} / {o["icp"]["iterations"]})')
    assert dT <= 1e-4, dT
    assert rot_angle_deg(T[:3, :3], T_gt[:3, :3]) < 2.0 and np.linalg.norm(T[:3, 3] - T_gt[:3, 3]) < 0.1


def test_register_equals_the_reference_run():
    """No oracle and no harness here: `register()` of the HIP path against the committed run of the REFERENCE's own
    `register()` (tests/golden/make_golden_register.py: core/deep_global_registration.py imported from /root/reference
    over stand-ins for its MinkowskiEngine / Open3D imports; learned branch, `use_icp = False`) on the same raw points
    and the same seeded checkpoint -- matches exact, logits 1e-4, T 1e-4 (or, should the free-running refinement stop
    elsewhere, the band of tests/helpers.assert_refine_parity)."""
    from test_oracle_register_golden import golden_case
    g, ck = golden_case()
    dgr = _dgr(ck, clip_weight_thresh=float(g['clip_weight_thresh']))
    dgr.use_icp = False
    T = dgr.register(g['xyz0'], g['xyz1'])
    assert dgr.last_status == 'ok' and T.dtype == np.float64
    idx1 = dgr.last_corres_idx1.cpu().numpy().reshape(-1)
    assert np.array_equal(idx1, g['idx1'].reshape(-1)), int((idx1 != g['idx1'].reshape(-1)).sum())
    dl = np.abs(dgr.last_logit.cpu().numpy().reshape(-1) - g['logit'].reshape(-1)).max() / np.abs(g['logit']).max()
    dT = float(np.abs(T - g['T']).max())
    print(f'HIP register() vs the reference run: N0={len(idx1)}, matches equal, dlogit {dl:.1e}, max|T - T_reference| {dT:.1e}, '
          f'refinement iterations {dgr.last_stats["iterations"]}')
    assert dl < 1e-4, dl
    if dT > 1e-4:
        w = 1 / (1 + np.exp(-g['logit'].reshape(-1).astype(np.float64)))
        w[w < float(g['clip_weight_thresh'])] = 0
        assert_refine_parity(g['p0'], g['p1'][idx1], w.astype(np.float32), T[:3, :3], T[:3, 3], dict(dgr.last_stats),
                             break_threshold_ratio=1e-4, quantization_size=2 * float(g['voxel']))


def test_register_with_icp_reproduces_the_recorded_run():
    """`register()` as shipped (`use_icp = True`, the reference's default) against the reference's own run through its
    ICP call site (:317-322; tests/golden/register_e2e_o3d.npz case `icp`)."""
    from test_oracle_register_golden import golden_case, golden_o3d
    g, ck = golden_case()
    o3 = golden_o3d()
    dgr = _dgr(ck, clip_weight_thresh=float(g['clip_weight_thresh']))
    assert dgr.use_icp
    T = dgr.register(g['xyz0'], g['xyz1'])
    assert dgr.last_status == 'ok' and T.dtype == np.float64
    assert np.array_equal(dgr.last_corres_idx1.cpu().numpy().reshape(-1), g['idx1'].reshape(-1))
    dT = float(np.abs(T - o3['icp_T']).max())
    print(f'HIP register() + ICP vs the reference run: max|T - T_reference| {dT:.1e}, ICP iterations '
          f'{dgr.last_icp["iterations"]} (reference {int(o3["icp_call_iterations"])})')
    assert dT < 1e-4, dT


def test_register_safeguard_reproduces_the_recorded_run():
    """Gate fails -> safeguard RANSAC -> ICP against the reference's own run through both Open3D call sites (:50-64,
    302-322; case `safeguard` of register_e2e_o3d.npz): same matches, same winning hypothesis and consensus, T to 1e-6."""
    from test_oracle_register_golden import golden_case, golden_o3d
    g, ck = golden_case()
    o3 = golden_o3d()
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    dgr = DeepGlobalRegistration({'weights': ck, 'clip_weight_thresh': float(g['clip_weight_thresh']), 'keep_intermediates': True,
                                  'ransac_max_iteration': int(o3['ransac_cap']), 'ransac_seed': int(o3['ransac_seed'])},
                                 torch.device('cuda'))
    T = dgr.register(o3['sg_xyz0'], o3['sg_xyz1'])
    assert dgr.last_status == 'safeguard' and T.dtype == np.float64
    assert np.array_equal(dgr.last_corres_idx1.cpu().numpy().reshape(-1), o3['sg_idx1'].reshape(-1))
    dl = np.abs(dgr.last_logit.cpu().numpy().reshape(-1) - o3['sg_logit'].reshape(-1)).max() / np.abs(o3['sg_logit']).max()
    assert dl < 1e-4, dl
    assert dgr.last_stats['ransac_hypothesis'] == int(o3['sg_ransac_call_hypothesis'])
    assert dgr.last_stats['ransac_inliers'] == int(o3['sg_ransac_call_inliers'])
    assert dgr.last_icp['iterations'] == int(o3['sg_icp_call_iterations'])
    dT = float(np.abs(T - o3['sg_T']).max())
    print(f'HIP register() safeguard vs the reference run: hypothesis {dgr.last_stats["ransac_hypothesis"]}, '
          f'{dgr.last_stats["ransac_inliers"]} inliers, max|T - T_reference| {dT:.1e}')
    assert dT < 1e-6, dT
