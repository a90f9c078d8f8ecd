"""GPU parity at the other BASELINE.json configurations:
  configs[2]  KITTI-shaped LiDAR pair, 120k points, 30 cm voxels, conv1 k = 5 (scripts/train_kitti.sh:17,19)
  configs[4]  dense pair, 200k points, 2.5 cm voxels, refinement on / off (core/registration.py:135-194)
Each through the fused batched call (`dgr_register_batch`) against the CPU oracle on identical inputs."""
import numpy as np
import pytest
import torch

from conftest import rot_angle_deg
from helpers import assert_iteration_matched, assert_refine_parity, oracle_pair_counts, rel_err
from oracle import knn as oknn
from oracle import pipeline as opipe
from oracle import registration as oreg
from oracle import resunet as oresunet

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _prepare(voxel, ks, n_raw, kind):
    from deepglobalregistration_amd import ops, synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    ck = synth.synth_checkpoint(seed=0, voxel_size=voxel, feat_conv1_kernel_size=ks)
    dgr = DeepGlobalRegistration({'weights': ck, 'clip_weight_thresh': 0.05}, torch.device('cuda'))
    a, b, T_gt = synth.synth_pair(0, n_raw=n_raw, kind=kind)
    xa, ca, _ = dgr.preprocess(a)
    xb, cb, _ = dgr.preprocess(b)
    gt = synth.gt_correspondences(xa.cpu().numpy(), xb.cpu().numpy(), T_gt, voxel, seed=0)
    ovr = torch.from_numpy(gt).cuda()
    off0, off1 = [0, len(xa)], [0, len(xb)]
    dgr.register_voxelized(ca, xa, off0, cb, xb, off1, override_idx1=ovr)
    idx1 = ops.batch_output('cuda', 'idx1').cpu().numpy()
    forced = synth.gt_forced_logits(xa.cpu().numpy(), xb.cpu().numpy()[idx1], T_gt, voxel)
    return dict(ck=ck, dgr=dgr, raw=(a, b), T_gt=T_gt, xa=xa, ca=ca, xb=xb, cb=cb, off0=off0, off1=off1, ovr=ovr,
                idx1=idx1, forced=forced)


def test_config2_kitti_shaped_pair_matches_oracle():
    from deepglobalregistration_amd import ops
    voxel, ks = 0.3, 5
    w = _prepare(voxel, ks, 120000, 'outdoor')
    ck, dgr = w['ck'], w['dgr']
    # voxelisation of the LiDAR-shaped cloud (rings, 80 m range, negative coordinates) = the oracle's
    op0, oc0, _ = opipe.preprocess(w['raw'][0], voxel)
    op1, oc1, _ = opipe.preprocess(w['raw'][1], voxel)
    np.testing.assert_array_equal(w['ca'].cpu().numpy(), oc0)
    np.testing.assert_array_equal(w['cb'].cpu().numpy(), oc1)
    assert len(oc0) > 8000 and len(oc1) > 8000
    T, status, stats = dgr.register_voxelized(w['ca'], w['xa'], w['off0'], w['cb'], w['xb'], w['off1'], override_idx1=w['ovr'],
                                              forced_logits=torch.from_numpy(w['forced']).cuda())
    assert status.tolist() == [0]
    idx1 = ops.batch_output('cuda', 'idx1').cpu().numpy()
    assert np.array_equal(idx1, w['idx1'])
    F0 = ops.batch_output('cuda', 'F0').reshape(-1, 32).cpu().numpy()
    F1 = ops.batch_output('cuda', 'F1').reshape(-1, 32).cpu().numpy()
    logit = ops.batch_output('cuda', 'logit').cpu().numpy()
    # FCGF with the 5^3 conv1 (125 offsets) and the 6-D net, with kernel-map pair counts
    m0, m1 = oresunet.SparseMaps(oc0, 3, ks), oresunet.SparseMaps(oc1, 3, ks)
    oF0 = oresunet.resunet_forward(ck['state_dict'], oc0, np.ones((len(oc0), 1), np.float32), 3, ks, True, maps=m0)
    oF1 = oresunet.resunet_forward(ck['state_dict'], oc1, np.ones((len(oc1), 1), np.float32), 3, ks, True, maps=m1)
    assert np.abs(F0 - oF0).max() < TOL and np.abs(F1 - oF1).max() < TOL
    c6, f6 = opipe.inlier_inputs(op0, op1, oc0, oc1, np.arange(len(oc0)), idx1)
    m6 = oresunet.SparseMaps(c6, 6, 3)
    ologit = oresunet.resunet_forward(ck['state_dict_inlier'], c6, f6, 6, 3, False, maps=m6).reshape(-1)
    assert rel_err(logit, ologit) < TOL
    fc, inl = dgr.fcgf_model._handle(), dgr.inlier_model._handle()
    fc.forward(w['ca'], torch.ones(len(oc0), 1, device='cuda'))
    assert [s['pairs'] for s in fc.layer_stats()] == oracle_pair_counts(m0, ks)
    d6, g6 = ops.inlier_inputs(w['ca'], w['xa'], w['cb'], w['xb'], torch.from_numpy(idx1).cuda(), 'coords')
    inl.forward(d6, g6)
    assert [s['pairs'] for s in inl.layer_stats()] == oracle_pair_counts(m6, 3)
    # 1-NN on the oracle's features: exact up to f64 ties
    i1 = dgr.fcgf_feature_matching(torch.from_numpy(oF0).cuda(), torch.from_numpy(oF1).cuda())[1].cpu().numpy()
    oi1 = oknn.find_knn(oF0, oF1, nn_max_n=250).reshape(-1)
    bad = np.nonzero(i1 != oi1)[0]
    if len(bad):
        assert np.abs(oknn.knn_sqdist_f64(oF0[bad], oF1, i1[bad]) - oknn.knn_sqdist_f64(oF0[bad], oF1, oi1[bad])).max() <= 2e-6
    # registration (quantisation 2 x 0.3 m) vs the oracle; KITTI success thresholds 0.6 m / 5 deg (scripts/test_kitti.py:33-34)
    ow, owsum, thr = opipe.confidence_gate(w['forced'], 0.05)
    assert owsum >= thr
    st = {'iterations': int(stats[0, 0]), 'loss': float(stats[0, 1]), 'break_count': int(stats[0, 2])}
    assert_refine_parity(op0, op1[idx1], ow, T[0, :3, :3], T[0, :3, 3], st, break_threshold_ratio=1e-4, quantization_size=2 * voxel)
    assert_iteration_matched(op0, op1[idx1], ow, break_threshold_ratio=1e-4, quantization_size=2 * voxel)
    assert rot_angle_deg(T[0, :3, :3], w['T_gt'][:3, :3]) < 5.0 and np.linalg.norm(T[0, :3, 3] - w['T_gt'][:3, 3]) < 0.6


def test_config4_dense_pair_refinement_on_off():
    from deepglobalregistration_amd import ops
    voxel, ks = 0.025, 7
    w = _prepare(voxel, ks, 200000, 'indoor')
    ck, dgr = w['ck'], w['dgr']
    n0, n1 = w['off0'][1], w['off1'][1]
    assert n0 > 80000 and n1 > 80000, (n0, n1)            # the stress size: ~100k voxels per fragment
    X0, X1 = w['xa'].cpu().numpy(), w['xb'].cpu().numpy()
    forced = torch.from_numpy(w['forced']).cuda()
    ow, owsum, thr = opipe.confidence_gate(w['forced'], 0.05)
    assert owsum >= thr
    Y = X1[w['idx1']]
    # refinement OFF (skip_refinement): the result is the weighted Procrustes estimate (core/registration.py:91-113)
    T_off, status, stats_off = dgr.register_voxelized(w['ca'], w['xa'], w['off0'], w['cb'], w['xb'], w['off1'], override_idx1=w['ovr'],
                                                      forced_logits=forced, skip_refinement=True)
    assert status.tolist() == [0] and int(stats_off[0, 0]) == 0
    assert np.array_equal(ops.batch_output('cuda', 'idx1').cpu().numpy(), w['idx1'])
    Rp, tp = (np.asarray(v) for v in oreg.weighted_procrustes(X0, Y, ow))
    assert np.abs(T_off[0, :3, :3] - Rp).max() < TOL and np.abs(T_off[0, :3, 3] - tp.reshape(-1)).max() < TOL
    # refinement ON (core/registration.py:161-194)
    T_on, status, stats = dgr.register_voxelized(w['ca'], w['xa'], w['off0'], w['cb'], w['xb'], w['off1'], override_idx1=w['ovr'],
                                                 forced_logits=forced)
    assert status.tolist() == [0] and int(stats[0, 0]) > 0
    F0 = ops.batch_output('cuda', 'F0').reshape(-1, 32).cpu().numpy()     # before any other library call on the context
    F1 = ops.batch_output('cuda', 'F1').reshape(-1, 32).cpu().numpy()
    logit = ops.batch_output('cuda', 'logit').cpu().numpy().reshape(-1)
    st = {'iterations': int(stats[0, 0]), 'loss': float(stats[0, 1]), 'break_count': int(stats[0, 2])}
    assert_refine_parity(X0, Y, ow, T_on[0, :3, :3], T_on[0, :3, 3], st, break_threshold_ratio=1e-4, quantization_size=2 * voxel)
    assert_iteration_matched(X0, Y, ow, break_threshold_ratio=1e-4, quantization_size=2 * voxel)
    for T in (T_off, T_on):
        assert rot_angle_deg(T[0, :3, :3], w['T_gt'][:3, :3]) < 2.0 and np.linalg.norm(T[0, :3, 3] - w['T_gt'][:3, 3]) < 0.1
    # a later call reused the workspace: the library refuses to hand out the stale outputs (ValueError), it does not read them
    with pytest.raises(ValueError, match='gone'):
        ops.batch_output('cuda', 'F0')
    # the brute-force stress: 1-NN over ~100k x ~100k features, checked on a row sample against the oracle's chunked search
    i1 = dgr.fcgf_feature_matching(torch.from_numpy(F0).cuda(), torch.from_numpy(F1).cuda())[1].cpu().numpy()
    rows = np.random.default_rng(0).choice(n0, 1500, replace=False)
    oi = oknn.find_knn(F0[rows], F1, nn_max_n=250).reshape(-1)
    bad = np.nonzero(i1[rows] != oi)[0]
    if len(bad):
        assert np.abs(oknn.knn_sqdist_f64(F0[rows][bad], F1, i1[rows][bad]) - oknn.knn_sqdist_f64(F0[rows][bad], F1, oi[bad])).max() <= 2e-6
    # FCGF features of fragment 0 at this size vs the oracle (one ~100k-voxel forward on the CPU)
    oc0 = w['ca'].cpu().numpy()
    oF0 = oresunet.resunet_forward(ck['state_dict'], oc0, np.ones((n0, 1), np.float32), 3, ks, True)
    assert np.abs(F0 - oF0).max() < TOL
    # ... and the 6-D logits of all ~100k correspondences (the inlier net at its largest BASELINE size: one forward of the
    # oracle over ~100k 6-D rows, K = 729)
    oc1 = w['cb'].cpu().numpy()
    c6, f6 = opipe.inlier_inputs(X0, X1, oc0, oc1, np.arange(n0), w['idx1'])
    ologit = oresunet.resunet_forward(ck['state_dict_inlier'], c6, f6, 6, 3, False).reshape(-1)
    assert logit.shape == ologit.shape
    err = rel_err(logit, ologit)
    print(f'configs[4]: {n0} 6-D rows, max |dlogit| / max |logit| = {err:.1e}')
    assert err < TOL
