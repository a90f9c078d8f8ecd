"""GPU parity of the ICP and the safeguard RANSAC (o3d.hip) against the CPU restatement of the Open3D
algorithms (oracle/open3d_reg.py), through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import rot_angle_deg
from oracle import open3d_reg as o3

pytestmark = pytest.mark.gpu


def _rot(rng, max_deg):
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    a = np.radians(rng.uniform(0, max_deg))
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def _voxel_cloud(seed, n_raw=20000):
    from deepglobalregistration_amd import synth
    x0, x1, T = synth.synth_pair(seed, n_raw)
    q0 = np.unique(np.floor(x0 / 0.05).astype(np.int64), axis=0, return_index=True)[1]
    q1 = np.unique(np.floor(x1 / 0.05).astype(np.int64), axis=0, return_index=True)[1]
    return x0[np.sort(q0)].astype(np.float32), x1[np.sort(q1)].astype(np.float32), T


@pytest.mark.parametrize('case', ['gt_init', 'perturbed_init', 'identity_init_far', 'max_iter_2'])
def test_icp_matches_oracle(case):
    from deepglobalregistration_amd import ops
    rng = np.random.default_rng(3)
    src, dst, T_gt = _voxel_cloud(1)
    init = T_gt.copy()
    kw = {}
    if case == 'perturbed_init':
        init[:3, :3] = _rot(rng, 2.0) @ init[:3, :3]
        init[:3, 3] += rng.uniform(-0.03, 0.03, 3)
    elif case == 'identity_init_far':
        init = None                                     # no overlap after the identity: fitness ~ 0
        src = src + np.float32(50.0)
    elif case == 'max_iter_2':
        init[:3, 3] += 0.04
        kw = {'max_iter': 2}
    To, fo, ro, ito = o3.icp_point_to_point(src, dst, 0.1, init=init, **kw)
    T, f, r, it = ops.icp_point_to_point(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(), 0.1, init=init, **kw)
    assert it == ito
    assert abs(f - fo) < 1e-12 and abs(r - ro) < 1e-9
    np.testing.assert_allclose(T, To, atol=1e-9)
    np.testing.assert_array_equal(T[3], [0, 0, 0, 1])
    if case == 'gt_init':   # voxel-quantised planar scene: ICP stays near the truth, it does not sharpen it
        assert rot_angle_deg(T[:3, :3], T_gt[:3, :3]) < 2.0 and np.linalg.norm(T[:3, 3] - T_gt[:3, 3]) < 0.1


def test_icp_argument_errors():
    from deepglobalregistration_amd import ops
    a = torch.zeros(10, 3).cuda()
    with pytest.raises(ValueError):
        ops.icp_point_to_point(torch.zeros(0, 3).cuda(), a, 0.1)
    with pytest.raises(ValueError):
        ops.icp_point_to_point(a, a, -1.0)
    with pytest.raises(ValueError):
        ops.icp_point_to_point(torch.zeros(10, 2).cuda(), a, 0.1)


def _corr_problem(seed, n, outlier_frac):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    R, t = _rot(rng, 180), rng.uniform(-1, 1, 3)
    Y = (X @ R.T + t + rng.normal(0, 0.005, (n, 3))).astype(np.float32)
    out = rng.random(n) < outlier_frac
    Y[out] = rng.uniform(-3, 3, (int(out.sum()), 3)).astype(np.float32)
    return X, Y, R, t, out


def test_ransac_matches_oracle_exactly():
    from deepglobalregistration_amd import ops
    X, Y, R, t, out = _corr_problem(4, 1500, 0.7)
    for seed, hyp in ((0, 2000), (9, 2500)):
        To, ho, co, ro = o3.ransac_correspondence(X, Y, 0.1, hyp, seed=seed)
        T, h, c, r = ops.ransac_correspondence(torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda(), 0.1, hyp, seed=seed)
        assert (h, c) == (ho, co)                      # same hypothesis wins with the same consensus
        assert abs(r - ro) < 1e-6
        np.testing.assert_allclose(T, To, atol=1e-9)
        assert rot_angle_deg(T[:3, :3], R) < 3.0


def test_ransac_full_size_properties():
    """BASELINE size: 4 000 000 hypotheses over ~27 k correspondences with 80 % outliers."""
    from deepglobalregistration_amd import ops
    X, Y, R, t, out = _corr_problem(6, 27000, 0.8)
    Xd, Yd = torch.from_numpy(X).cuda(), torch.from_numpy(Y).cuda()
    T, h, c, r = ops.ransac_correspondence(Xd, Yd, 0.1, 4000000, seed=1)
    assert 0 <= h < 4000000 and c >= 0.9 * int((~out).sum()) and r < 0.1
    assert rot_angle_deg(T[:3, :3], R) < 1.5 and np.linalg.norm(T[:3, 3] - t) < 0.05
    # the winner's consensus is what the oracle computes for that very hypothesis
    To, _, co, ro = o3.ransac_correspondence(X, Y, 0.1, 1, seed=1) if h == 0 else (None, None, None, None)
    S = o3.ransac_samples(1, h, 1, len(X))[0]
    Th = o3.umeyama(X[S].astype(np.float64), Y[S].astype(np.float64))
    np.testing.assert_allclose(T, Th, atol=1e-9)
    # deterministic
    T2, h2, c2, _ = ops.ransac_correspondence(Xd, Yd, 0.1, 4000000, seed=1)
    assert (h, c) == (h2, c2) and np.array_equal(T, T2)
    # a sub-range of the hypotheses can only do as well or worse
    _, _, c3, _ = ops.ransac_correspondence(Xd, Yd, 0.1, 100000, seed=1)
    assert c3 <= c


def test_register_runs_safeguard_and_icp():
    """Untrained weights fail the confidence gate -> register() takes the safeguard branch (:302-315) and,
    with use_icp (the reference default), refines by ICP (:317-322)."""
    from deepglobalregistration_amd import synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    ck = synth.synth_checkpoint(0)
    dgr = DeepGlobalRegistration({'weights': ck}, torch.device('cuda'))
    assert dgr.use_icp
    x0, x1, T_gt = synth.synth_pair(0, 20000)
    T = dgr.register(x0, x1)
    assert T.shape == (4, 4) and T.dtype == np.float64
    assert dgr.last_status in ('ok', 'safeguard')
    assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-6      # f32 rotation when the learned path ran
    assert 'iterations' in dgr.last_icp


def test_batched_call_with_safeguard_and_icp():
    """Pairs rejected by the gate inside the fused batched call are rescued by the safeguard (status 3) and
    refined by ICP; with 20 % ground-truth matches among the putative correspondences RANSAC finds the pose."""
    from deepglobalregistration_amd import ops, synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    dgr = DeepGlobalRegistration({'weights': synth.synth_checkpoint(0)}, torch.device('cuda'))
    pairs = [synth.synth_pair(s, 20000) for s in (0, 1)]
    x0, c0, x1, c1, off0, off1, ov = [], [], [], [], [0], [0], []
    for p, (a, b, T_gt) in enumerate(pairs):
        xa, ca, _ = dgr.preprocess(a, batch_index=p)
        xb, cb, _ = dgr.preprocess(b, batch_index=p)
        gt = synth.gt_correspondences(xa.cpu().numpy(), xb.cpu().numpy(), T_gt, 0.05)
        keep = np.random.default_rng(p).random(len(gt)) < 0.5
        ov.append(np.where((gt >= 0) & keep, gt + off1[-1], -1))
        x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
        off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
    forced = torch.full((off0[-1],), -6.0).cuda()          # the gate rejects everything
    T, status, _ = dgr.register_voxelized(torch.cat(c0), torch.cat(x0), off0, torch.cat(c1), torch.cat(x1), off1,
                                          forced_logits=forced, override_idx1=torch.from_numpy(np.concatenate(ov)).cuda(),
                                          safeguard=True, icp=True)
    assert status.tolist() == [3, 3]
    for p, (_, _, T_gt) in enumerate(pairs):
        assert rot_angle_deg(T[p, :3, :3], T_gt[:3, :3]) < 2.0 and np.linalg.norm(T[p, :3, 3] - T_gt[:3, 3]) < 0.1
    # the two steps run inside dgr_register_batch are the stand-alone entry points: same hypotheses, same ICP
    idx1 = ops.batch_output('cuda', 'idx1')      # the override keeps the 1-NN answer where it holds -1
    X1 = torch.cat(x1)
    for p in range(2):
        sl = slice(off0[p], off0[p + 1])
        Y = ops.gather_rows3(X1, idx1[sl])
        Tr, _, _, _ = ops.ransac_correspondence(x0[p], Y, 2 * dgr.voxel_size, 4000000, seed=dgr.ransac_seed)
        Ti, _, _, _ = ops.icp_point_to_point(x0[p], x1[p], 2 * dgr.voxel_size, init=Tr)
        assert np.abs(Ti - T[p]).max() < 1e-6          # the batched call returns float32 at the ABI
    # without the flags the rejected pairs stay at identity with status 1 (register_batch proper, :288-300)
    T0, status0, _ = dgr.register_voxelized(torch.cat(c0), torch.cat(x0), off0, torch.cat(c1), torch.cat(x1), off1,
                                            forced_logits=forced,
                                            override_idx1=torch.from_numpy(np.concatenate(ov)).cuda())
    assert status0.tolist() == [1, 1] and np.array_equal(T0[0], np.eye(4))


def test_register_equals_the_fused_call():
    """`register()` (stage by stage, the reference's own control flow, :238-324) and ONE `dgr_register_batch` with
    the safeguard / ICP flags return the same transformation for the same pair -- gate, safeguard RANSAC (same
    counter-based sampler and seed) and ICP included."""
    from deepglobalregistration_amd import synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    dgr = DeepGlobalRegistration({'weights': synth.synth_checkpoint(0)}, torch.device('cuda'))
    x0, x1, _ = synth.synth_pair(2, 20000)
    dgr.last_status = None
    T = dgr.register(x0, x1)
    assert dgr.last_status in ('ok', 'safeguard')  # untrained weights: whichever branch the gate takes, both calls take it
    xa, ca, _ = dgr.preprocess(x0)
    xb, cb, _ = dgr.preprocess(x1)
    Tf, status, _ = dgr.register_voxelized(ca, xa, [0, len(xa)], cb, xb, [0, len(xb)], safeguard=True, icp=True)
    assert status.tolist() == [3 if dgr.last_status == 'safeguard' else 0]     # (code, no flag: the ICP ran)
    # the fused call hands the float64 results of its RANSAC / ICP stages on (dgr_register_batch_f64): the same
    # arithmetic as the step-by-step register() -- equal to rounding of the f32 -> f64 hand-over of the initial T
    assert Tf.dtype == np.float64 and np.abs(Tf[0] - T).max() < 1e-9
