"""GPU parity of the whole path: DeepGlobalRegistration.register() (stage-wise, reference-like
call sequence) and the fused batched pipeline against the CPU oracle on identical inputs."""
import numpy as np
import pytest
import torch

from conftest import rot_angle_deg
from helpers import assert_iteration_matched, assert_refine_parity, rel_err
from oracle import pipeline as opipe
from oracle import registration as oreg
from oracle import resunet as oresunet

pytestmark = pytest.mark.gpu
VOXEL = 0.05


@pytest.fixture(scope='module')
def setup():
    from deepglobalregistration_amd import synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    ck = synth.synth_checkpoint(seed=0, voxel_size=VOXEL, feat_conv1_kernel_size=7)
    dgr = DeepGlobalRegistration({'weights': ck, 'clip_weight_thresh': 0.05}, torch.device('cuda'))
    pairs = [synth.synth_pair(s, n_raw=6000) for s in (0, 1)]
    return ck, dgr, pairs


def test_stagewise_matches_oracle(setup):
    from deepglobalregistration_amd import ops, synth
    ck, dgr, pairs = setup
    xyz0, xyz1, T_gt = pairs[0]
    p0, c0, f0 = dgr.preprocess(xyz0)
    p1, c1, f1 = dgr.preprocess(xyz1)
    op0, oc0, of0 = opipe.preprocess(xyz0, VOXEL)
    op1, oc1, of1 = opipe.preprocess(xyz1, VOXEL)
    np.testing.assert_array_equal(c0.cpu().numpy(), oc0)
    np.testing.assert_array_equal(p1.cpu().numpy(), op1)
    F0 = dgr.fcgf_feature_extraction(f0, c0)
    F1 = dgr.fcgf_feature_extraction(f1, c1)
    oF0 = oresunet.resunet_forward(ck['state_dict'], oc0, of0, 3, 7, True)
    oF1 = oresunet.resunet_forward(ck['state_dict'], oc1, of1, 3, 7, True)
    assert np.abs(F0.cpu().numpy() - oF0).max() < 1e-4
    assert np.abs(F1.cpu().numpy() - oF1).max() < 1e-4
    # matching on the ORACLE's features (teacher forcing) so that later stages see identical inputs
    i0, i1 = dgr.fcgf_feature_matching(torch.from_numpy(oF0).cuda(), torch.from_numpy(oF1).cuda())
    from oracle import knn as oknn
    oi1 = oknn.find_knn(oF0, oF1, nn_max_n=250).reshape(-1)
    mism = np.nonzero(i1.cpu().numpy() != oi1)[0]
    if len(mism):   # only genuine rounding ties may differ: both candidates equally near in float64
        d_a = oknn.knn_sqdist_f64(oF0[mism], oF1, i1.cpu().numpy()[mism])
        d_b = oknn.knn_sqdist_f64(oF0[mism], oF1, oi1[mism])
        assert np.all(np.abs(d_a - d_b) <= 2e-6), (len(mism), np.abs(d_a - d_b).max())
    # untrained weights give ~0 % correct matches: override a share with ground-truth matches
    gt = synth.gt_correspondences(op0, op1, T_gt, VOXEL)
    oi1 = np.where(gt >= 0, gt, oi1)
    i1 = torch.from_numpy(oi1).cuda()
    feats6 = dgr.inlier_feature_generation(p0, p1, c0, c1, F0, F1, i0, i1)
    coords6, feats6b = ops.inlier_inputs(c0, p0, c1, p1, i1, 'coords')
    ocoords6, ofeats6 = opipe.inlier_inputs(op0, op1, oc0, oc1, np.arange(len(oi1)), oi1)
    np.testing.assert_array_equal(coords6.cpu().numpy(), ocoords6)
    np.testing.assert_allclose(feats6.cpu().numpy(), ofeats6, atol=2e-6)
    logit = dgr.inlier_prediction(feats6, coords6)
    ologit = oresunet.resunet_forward(ck['state_dict_inlier'], ocoords6, ofeats6, 6, 3, False)
    assert rel_err(logit.cpu().numpy(), ologit) < 1e-4
    # gate + registration on identical (teacher-forced) logits
    forced = synth.gt_forced_logits(op0, op1[oi1], T_gt, VOXEL)
    w, wsum = ops.sigmoid_clip_sum(torch.from_numpy(forced).cuda(), 0.05)
    ow, owsum, thr = opipe.confidence_gate(forced, 0.05)
    np.testing.assert_allclose(w.cpu().numpy(), ow, atol=1e-6)
    assert abs(wsum - owsum) < 1e-2 * max(1.0, owsum) * 1e-2
    assert owsum >= thr, 'synthetic pair should pass the confidence gate'
    from deepglobalregistration_amd.core.registration import GlobalRegistration
    R, t, st = GlobalRegistration(p0, ops.gather_rows3(p1, i1), weights=w, break_threshold_ratio=1e-4,
                                  quantization_size=2 * VOXEL)
    Ro, to, sto = assert_refine_parity(op0, op1[oi1], ow, R.cpu().numpy(), t.cpu().numpy(), st,
                                       break_threshold_ratio=1e-4, quantization_size=2 * VOXEL)
    # the same pipeline-shaped input with the iteration count pinned on both sides: 1e-4
    assert_iteration_matched(op0, op1[oi1], ow, break_threshold_ratio=1e-4, quantization_size=2 * VOXEL)
    # and the estimate is close to the ground truth pose
    assert rot_angle_deg(Ro, T_gt[:3, :3]) < 2.0 and np.linalg.norm(to - T_gt[:3, 3]) < 0.1


def test_register_api(setup):
    ck, dgr, pairs = setup
    xyz0, xyz1, _ = pairs[0]
    T = dgr.register(xyz0, xyz1)
    assert T.shape == (4, 4) and T.dtype == np.float64
    np.testing.assert_array_equal(T[3], [0, 0, 0, 1])
    assert dgr.last_status in ('ok', 'safeguard')
    assert dgr.feat_timer.diff > 0 and dgr.reg_timer.avg > 0
    with pytest.raises(Exception, match='Unrecognized pcd type'):
        dgr.register([1, 2, 3], xyz1)


def test_fused_batch_matches_stagewise_and_oracle(setup):
    from deepglobalregistration_amd import ops, synth
    ck, dgr, pairs = setup
    # voxelise both pairs into one batch
    x0, c0, x1, c1, off0, off1 = [], [], [], [], [0], [0]
    for p, (a, b, _) in enumerate(pairs):
        xa, ca, _ = dgr.preprocess(a, batch_index=p)
        xb, cb, _ = dgr.preprocess(b, batch_index=p)
        x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
        off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
    C0, X0, C1, X1 = torch.cat(c0), torch.cat(x0), torch.cat(c1), torch.cat(x1)
    # untrained weights: override a share of the matches with ground-truth ones (batch-global rows),
    # run once to obtain the final correspondences, then force GT-derived logits
    ovr = np.concatenate([
        (lambda g, o: np.where(g >= 0, g + o, -1))(
            synth.gt_correspondences(X0[off0[p]:off0[p + 1]].cpu().numpy(), X1[off1[p]:off1[p + 1]].cpu().numpy(),
                                     pairs[p][2], VOXEL, seed=p), off1[p])
        for p in range(2)])
    ovr_t = torch.from_numpy(ovr).cuda()
    T, status, stats = dgr.register_voxelized(C0, X0, off0, C1, X1, off1, override_idx1=ovr_t)
    idx1 = ops.batch_output('cuda', 'idx1').cpu().numpy()
    forced = np.concatenate([
        synth.gt_forced_logits(X0[off0[p]:off0[p + 1]].cpu().numpy(),
                               X1.cpu().numpy()[idx1[off0[p]:off0[p + 1]]], pairs[p][2], VOXEL)
        for p in range(2)])
    T, status, stats = dgr.register_voxelized(C0, X0, off0, C1, X1, off1, override_idx1=ovr_t,
                                              forced_logits=torch.from_numpy(forced).cuda())
    idx1_b = ops.batch_output('cuda', 'idx1').cpu().numpy()
    assert np.array_equal(idx1_b, idx1)            # the search is deterministic (packed-key atomicMin: value, then index)
    logit = ops.batch_output('cuda', 'logit').cpu().numpy()
    F0 = ops.batch_output('cuda', 'F0').reshape(-1, 32).cpu().numpy()
    assert status.tolist() == [0, 0]
    for p in range(2):
        s0, e0, s1, e1 = off0[p], off0[p + 1], off1[p], off1[p + 1]
        # batched features == per-pair oracle features (pairs do not interact through the batch)
        oc0 = C0[s0:e0].cpu().numpy().copy(); oc0[:, 0] = 0
        oF0 = oresunet.resunet_forward(ck['state_dict'], oc0, np.ones((e0 - s0, 1), np.float32), 3, 7, True)
        assert np.abs(F0[s0:e0] - oF0).max() < 1e-4
        # 6-D logits vs the oracle on the HIP path's own correspondences
        li = idx1[s0:e0] - s1
        assert li.min() >= 0 and li.max() < e1 - s1
        oc1 = C1[s1:e1].cpu().numpy().copy(); oc1[:, 0] = 0
        oc6, of6 = opipe.inlier_inputs(X0[s0:e0].cpu().numpy(), X1[s1:e1].cpu().numpy(), oc0, oc1,
                                       np.arange(e0 - s0), li)
        ologit = oresunet.resunet_forward(ck['state_dict_inlier'], oc6, of6, 6, 3, False)
        assert rel_err(logit[s0:e0], ologit.reshape(-1)) < 1e-4
        # registration vs the oracle on identical correspondences + forced logits
        ow, owsum, thr = opipe.confidence_gate(forced[s0:e0], 0.05)
        st = {'iterations': int(stats[p, 0]), 'loss': float(stats[p, 1]), 'break_count': int(stats[p, 2])}
        assert_refine_parity(X0[s0:e0].cpu().numpy(), X1[s1:e1].cpu().numpy()[li], ow, T[p, :3, :3], T[p, :3, 3],
                             st, break_threshold_ratio=1e-4, quantization_size=2 * VOXEL)
        assert rot_angle_deg(T[p, :3, :3], pairs[p][2][:3, :3]) < 2.0
        assert abs(stats[p, 3] - owsum) < 1e-3 * owsum


def test_low_confidence_status(setup):
    from deepglobalregistration_amd import ops
    ck, dgr, pairs = setup
    xa, ca, _ = dgr.preprocess(pairs[0][0])
    xb, cb, _ = dgr.preprocess(pairs[0][1])
    forced = torch.full((len(xa),), -6.0).cuda()          # every correspondence rejected
    T, status, stats = dgr.register_voxelized(ca, xa, [0, len(xa)], cb, xb, [0, len(xb)], forced_logits=forced)
    assert status.tolist() == [1]
    np.testing.assert_array_equal(T[0], np.eye(4))


def test_collated_batch_and_batched_knn(setup):
    """The data-loader layout (dataloader/base_loader.py:40-98) and find_knn_gpu_batch (core/knn.py:106-140)."""
    from deepglobalregistration_amd.core.knn import find_knn_batch, find_knn_gpu, find_knn_gpu_batch
    ck, dgr, pairs = setup
    x0, c0, x1, c1, len_batch = [], [], [], [], []
    for p, (a, b, _) in enumerate(pairs):
        xa, ca, _ = dgr.preprocess(a, batch_index=p)
        xb, cb, _ = dgr.preprocess(b, batch_index=p)
        x0.append(xa.cpu().numpy()); c0.append(ca); x1.append(xb.cpu().numpy()); c1.append(cb)
        len_batch.append([len(xa), len(xb)])
    batch = {'pcd0': x0, 'pcd1': x1, 'sinput0_C': torch.cat(c0), 'sinput1_C': torch.cat(c1), 'len_batch': len_batch}
    forced = torch.full((sum(n for n, _ in len_batch),), -6.0).cuda()
    T, status, stats = dgr.register_collated(batch, forced_logits=forced)
    assert T.shape == (len(pairs), 4, 4) and status.tolist() == [1] * len(pairs)
    with pytest.raises(ValueError):
        dgr.register_collated(dict(batch, len_batch=[[1, 1]]))
    # non-degenerate: a third of the rows carry ground-truth matches, logits +-4 from the ground truth -> every pair registers, the
    # collated call returns what the per-pair calls return (rows of different pairs never interact) and the pose
    from deepglobalregistration_amd import synth
    ov, fl, off1 = [], [], 0
    for p, (a, b, T_gt) in enumerate(pairs):
        gt = synth.gt_correspondences(x0[p], x1[p], T_gt, VOXEL)
        keep = np.random.default_rng(p).random(len(gt)) < 0.7
        ov.append(np.where((gt >= 0) & keep, gt + off1, -1))
        fl.append(np.where((gt >= 0) & keep, 4.0, -4.0).astype(np.float32))
        off1 += len(x1[p])
    ov_d, fl_d = torch.from_numpy(np.concatenate(ov)).cuda(), torch.from_numpy(np.concatenate(fl)).cuda()
    T, status, stats = dgr.register_collated(batch, forced_logits=fl_d, override_idx1=ov_d)
    assert status.tolist() == [0] * len(pairs)
    o1 = 0
    for p, (a, b, T_gt) in enumerate(pairs):
        assert rot_angle_deg(T[p, :3, :3], T_gt[:3, :3]) < 1.0 and np.linalg.norm(T[p, :3, 3] - T_gt[:3, 3]) < 0.05
        ca = c0[p].clone(); ca[:, 0] = 0
        cb = c1[p].clone(); cb[:, 0] = 0
        ovp = torch.from_numpy(np.where(ov[p] >= 0, ov[p] - o1, -1)).cuda()
        T1, s1, st1 = dgr.register_voxelized(ca, torch.from_numpy(x0[p]).cuda(), [0, len(x0[p])], cb,
                                             torch.from_numpy(x1[p]).cuda(), [0, len(x1[p])],
                                             forced_logits=torch.from_numpy(fl[p]).cuda(), override_idx1=ovp)
        assert s1.tolist() == [0] and np.array_equal(T1[0], T[p]) and int(st1[0, 0]) == int(stats[p, 0])
        o1 += len(x1[p])
    # batched 1-NN = per-pair 1-NN, shifted by the pair's first row when concatenated
    rng = np.random.default_rng(0)
    lens = [[700, 900], [1300, 1100]]
    F0 = torch.from_numpy(rng.standard_normal((2000, 32)).astype(np.float32)).cuda()
    F1 = torch.from_numpy(rng.standard_normal((2000, 32)).astype(np.float32)).cuda()
    per = find_knn_gpu_batch(F0, F1, lens, nn_max_n=250)
    assert [tuple(t.shape) for t in per] == [(700, 1), (1300, 1)]
    cat, dist = find_knn_batch(F0, F1, lens, nn_max_n=250, return_distance=True, concat_results=True)
    ref1 = find_knn_gpu(F0[700:], F1[900:], nn_max_n=250)
    assert torch.equal(cat[700:], ref1 + 900) and torch.equal(per[1], ref1) and dist.shape == (2000, 1)
    with pytest.raises(ValueError):
        find_knn_batch(F0, F1, lens, search_method='tree')


def test_ragged_and_tiny_batches(setup):
    """Edge cases through the fused batched call: a tiny pair (tens of voxels), a one-voxel fragment, very
    different fragment sizes inside one batch.  Nothing may crash; hopeless pairs come back as low confidence
    with T = I, and the normal pair next to them is unaffected."""
    from deepglobalregistration_amd import synth
    ck, dgr, pairs = setup
    rng = np.random.default_rng(5)
    tiny0 = rng.uniform(0, 0.4, (60, 3)); tiny1 = rng.uniform(0, 0.4, (45, 3))
    single = np.array([[0.01, 0.02, 0.03]])
    big0, big1, T_gt = pairs[0]
    batch = [(tiny0, tiny1), (big0, single), (big0, big1)]
    x0, c0, x1, c1, off0, off1 = [], [], [], [], [0], [0]
    for p, (a, b) in enumerate(batch):
        xa, ca, _ = dgr.preprocess(a, batch_index=p)
        xb, cb, _ = dgr.preprocess(b, batch_index=p)
        x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
        off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
    assert len(x1[1]) == 1
    C0, X0, C1, X1 = torch.cat(c0), torch.cat(x0), torch.cat(c1), torch.cat(x1)
    T, status, stats = dgr.register_voxelized(C0, X0, off0, C1, X1, off1)
    assert T.shape == (3, 4, 4) and np.isfinite(T).all()
    assert status[0] == 1                                    # 60 correspondences: wsum < 200, never enough support
    np.testing.assert_array_equal(T[0], np.eye(4))
    # every voxel of the big fragment matched to ONE point: a degenerate (rank-0) covariance; with untrained
    # weights the gate may let it through -- whatever the status, the result must be a finite rigid transform
    assert status[1] in (0, 1, 2) and abs(np.linalg.det(T[1, :3, :3]) - 1) < 1e-4
    # the same big pair alone gives the same answer as inside the ragged batch (pairs are independent units)
    xa, ca, _ = dgr.preprocess(big0); xb, cb, _ = dgr.preprocess(big1)
    T1, s1, st1 = dgr.register_voxelized(ca, xa, [0, len(xa)], cb, xb, [0, len(xb)])
    assert s1[0] == status[2]
    np.testing.assert_allclose(T1[0], T[2], atol=1e-4)


def test_checkpoint_file_legacy_keys_and_kernel_shapes(tmp_path):
    """`__init__` (:88-131): a checkpoint FILE (torch.save), the legacy un-prefixed config keys (:104-112) and
    kernel-volume-1 kernels stored as [1, Cin, Cout] (MinkowskiEngine 0.4) give the same networks as the in-memory
    dict with [Cin, Cout] kernels: identical registration of one pair, bit for bit."""
    from deepglobalregistration_amd import synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    ck = synth.synth_checkpoint(seed=3, voxel_size=VOXEL, feat_conv1_kernel_size=5)
    legacy = {'config': dict(ck['config']), 'state_dict': {}, 'state_dict_inlier': {}}
    cfg = legacy['config']
    cfg['model'], cfg['model_n_out'], cfg['conv1_kernel_size'] = (cfg.pop('feat_model'), cfg.pop('feat_model_n_out'),
                                                                  cfg.pop('feat_conv1_kernel_size'))
    reshaped = 0
    for part in ('state_dict', 'state_dict_inlier'):
        for k, v in ck[part].items():
            t = torch.as_tensor(v)
            if k.endswith('.kernel') and t.dim() == 2:
                t = t[None]                       # [Cin, Cout] -> [1, Cin, Cout]
                reshaped += 1
            legacy[part][k] = t
    assert reshaped >= 2                          # conv1_tr / final of both nets
    path = tmp_path / 'ckpt.pth'
    torch.save(legacy, str(path))
    a = DeepGlobalRegistration({'weights': ck, 'use_icp': False}, torch.device('cuda'))
    b = DeepGlobalRegistration({'weights': str(path), 'use_icp': False}, torch.device('cuda'))
    x0, x1, _ = synth.synth_pair(5, n_raw=6000)
    Ta, Tb = a.register(x0, x1), b.register(x0, x1)
    assert a.last_status == b.last_status and np.array_equal(Ta, Tb)
    with pytest.raises(ValueError):
        DeepGlobalRegistration({'weights': 3}, torch.device('cuda'))
