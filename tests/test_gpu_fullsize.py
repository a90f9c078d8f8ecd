"""GPU parity at the BASELINE size (BASELINE.json configs[1]: 50k raw points per fragment, 5 cm voxels,
conv1 k = 7 -- 27 k / 21 k voxels per pair): both ResUNetBN2C nets of the fused batched call
(`dgr_register_batch`, B = 1 and B = 4) against the CPU oracle's `resunet_forward`
(model/resunet.py:598-649), every kernel-map pair count against the oracle's maps, and the batch
invariance of a pair's outputs.  Tolerance 1e-4 of the activation scale (f32 conv stack)."""
import numpy as np
import pytest
import torch

from helpers import oracle_pair_counts, rel_err
from oracle import pipeline as opipe
from oracle import resunet as oresunet

pytestmark = pytest.mark.gpu
VOXEL, KS, NRAW = 0.05, 7, 50000
TOL = 1e-4


@pytest.fixture(scope='module')
def world():
    from deepglobalregistration_amd import synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    ck = synth.synth_checkpoint(seed=0, voxel_size=VOXEL, feat_conv1_kernel_size=KS)
    dgr = DeepGlobalRegistration({'weights': ck, 'clip_weight_thresh': 0.05}, torch.device('cuda'))
    pairs = [synth.synth_pair(s, n_raw=NRAW) for s in range(6)]
    vox = []
    for p, (a, b, _) in enumerate(pairs):
        xa, ca, _ = dgr.preprocess(a, batch_index=p)
        xb, cb, _ = dgr.preprocess(b, batch_index=p)
        vox.append((xa, ca, xb, cb))
    return {'ck': ck, 'dgr': dgr, 'pairs': pairs, 'vox': vox, 'cache': {}}


def _batch(world, ids):
    """Voxelised pairs `ids` as one batch (batch column renumbered 0..B-1) + the harness's GT match override."""
    from deepglobalregistration_amd import synth
    x0, c0, x1, c1, off0, off1, ovr = [], [], [], [], [0], [0], []
    for q, p in enumerate(ids):
        xa, ca, xb, cb = world['vox'][p]
        ca, cb = ca.clone(), cb.clone()
        ca[:, 0] = q; cb[:, 0] = q
        g = synth.gt_correspondences(xa.cpu().numpy(), xb.cpu().numpy(), world['pairs'][p][2], VOXEL, seed=p)
        ovr.append(np.where(g >= 0, g + off1[-1], -1))
        x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
        off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
    return (torch.cat(c0), torch.cat(x0), off0, torch.cat(c1), torch.cat(x1), off1,
            torch.from_numpy(np.concatenate(ovr)).cuda())


def _run(world, ids):
    from deepglobalregistration_amd import ops
    C0, X0, off0, C1, X1, off1, ovr = _batch(world, ids)
    T, status, stats = world['dgr'].register_voxelized(C0, X0, off0, C1, X1, off1, override_idx1=ovr)
    out = {k: ops.batch_output('cuda', k).cpu().numpy() for k in ('idx1', 'logit', 'F0', 'F1')}
    out['F0'] = out['F0'].reshape(-1, 32); out['F1'] = out['F1'].reshape(-1, 32)
    out.update(off0=off0, off1=off1, T=T, status=status)
    return out


def _oracle_pair(world, p, idx1_local):
    """Oracle features of both fragments and oracle logits of pair p on the given correspondences."""
    ck = world['ck']
    xa, ca, xb, cb = [t.cpu().numpy() for t in world['vox'][p]]
    ca, cb = ca.copy(), cb.copy()
    ca[:, 0] = 0; cb[:, 0] = 0
    m0 = oresunet.SparseMaps(ca, 3, KS)
    m1 = oresunet.SparseMaps(cb, 3, KS)
    oF0 = oresunet.resunet_forward(ck['state_dict'], ca, np.ones((len(ca), 1), np.float32), 3, KS, True, maps=m0)
    oF1 = oresunet.resunet_forward(ck['state_dict'], cb, np.ones((len(cb), 1), np.float32), 3, KS, True, maps=m1)
    c6, f6 = opipe.inlier_inputs(xa, xb, ca, cb, np.arange(len(ca)), idx1_local)
    m6 = oresunet.SparseMaps(c6, 6, 3)
    ologit = oresunet.resunet_forward(ck['state_dict_inlier'], c6, f6, 6, 3, False, maps=m6)
    return oF0, oF1, ologit.reshape(-1), (m0, m1, m6), (c6, f6)


def test_fullsize_single_pair_nets_match_oracle(world):
    """B = 1: F0 / F1 / logit of the fused call vs the oracle; kernel-map pair counts of all 3 x 23 layers."""
    from deepglobalregistration_amd import ops
    r = _run(world, [0])
    world['cache'][0] = r
    n0, n1 = r['off0'][1], r['off1'][1]
    assert n0 > 20000 and n1 > 20000, (n0, n1)               # the BASELINE size, not a toy cloud
    oF0, oF1, ologit, (m0, m1, m6), (c6, f6) = _oracle_pair(world, 0, r['idx1'])
    dF = max(np.abs(r['F0'] - oF0).max(), np.abs(r['F1'] - oF1).max())
    assert dF < TOL, dF                                      # rows are unit vectors: absolute = relative
    dl = rel_err(r['logit'], ologit)
    assert dl < TOL, dl
    # kernel maps: pair counts per layer (stage-wise forwards of the same inputs expose the statistics)
    dgr = world['dgr']
    xa, ca, xb, cb = world['vox'][0]
    fc, inl = dgr.fcgf_model._handle(), dgr.inlier_model._handle()
    for c, m in ((ca, m0), (cb, m1)):
        F = fc.forward(c, torch.ones(len(c), 1, device='cuda'))
        assert [s['pairs'] for s in fc.layer_stats()] == oracle_pair_counts(m, KS)
    # stage-wise forward == fused forward, bitwise (same kernels, same order)
    assert np.array_equal(F.cpu().numpy(), r['F1'])
    d6, g6 = ops.inlier_inputs(ca, xa, cb, xb, torch.from_numpy(r['idx1']).cuda(), 'coords')   # device cosf, as in the fused call
    assert np.array_equal(d6.cpu().numpy(), c6) and np.abs(g6.cpu().numpy() - f6).max() < 2e-6
    lg = inl.forward(d6, g6)
    assert [s['pairs'] for s in inl.layer_stats()] == oracle_pair_counts(m6, 3)
    assert np.array_equal(lg.cpu().numpy().reshape(-1), r['logit'])
    print(f'full size B=1: N0={n0} N1={n1} |dF|={dF:.2e} rel|dlogit|={dl:.2e}')


def test_fullsize_batch_of_four_matches_oracle_and_single(world):
    """B = 4 (the benchmark's batch): pair 2 of the batch vs the oracle; pair 0 of the batch vs the same
    pair registered alone (pairs are independent units: the batch column only separates them)."""
    r = _run(world, [0, 1, 2, 3])
    off0, off1 = r['off0'], r['off1']
    s0, e0, s1, e1 = off0[2], off0[3], off1[2], off1[3]
    li = r['idx1'][s0:e0] - s1
    assert li.min() >= 0 and li.max() < e1 - s1
    oF0, oF1, ologit, _, _ = _oracle_pair(world, 2, li)
    dF = max(np.abs(r['F0'][s0:e0] - oF0).max(), np.abs(r['F1'][s1:e1] - oF1).max())
    dl = rel_err(r['logit'][s0:e0], ologit)
    assert dF < TOL and dl < TOL, (dF, dl)
    single = world['cache'].get(0) or _run(world, [0])
    n0, n1 = single['off0'][1], single['off1'][1]
    assert np.abs(r['F0'][:n0] - single['F0']).max() < 1e-6 and np.abs(r['F1'][:n1] - single['F1']).max() < 1e-6
    assert np.array_equal(r['idx1'][:n0], single['idx1'])
    assert rel_err(r['logit'][:n0], single['logit']) < 1e-6
    print(f'full size B=4: pair 2 |dF|={dF:.2e} rel|dlogit|={dl:.2e}')


def test_fullsize_batch_of_six_matches_oracle_and_single(world):
    """B = 6 (bench.py's default batch since round 5): the LAST pair of the batch vs the oracle, pair 0 of the batch vs the
    same pair registered alone."""
    r = _run(world, [0, 1, 2, 3, 4, 5])
    off0, off1 = r['off0'], r['off1']
    s0, e0, s1, e1 = off0[5], off0[6], off1[5], off1[6]
    li = r['idx1'][s0:e0] - s1
    assert li.min() >= 0 and li.max() < e1 - s1
    oF0, oF1, ologit, _, _ = _oracle_pair(world, 5, li)
    dF = max(np.abs(r['F0'][s0:e0] - oF0).max(), np.abs(r['F1'][s1:e1] - oF1).max())
    dl = rel_err(r['logit'][s0:e0], ologit)
    assert dF < TOL and dl < TOL, (dF, dl)
    single = world['cache'].get(0) or _run(world, [0])
    n0, n1 = single['off0'][1], single['off1'][1]
    assert np.abs(r['F0'][:n0] - single['F0']).max() < 1e-6 and np.abs(r['F1'][:n1] - single['F1']).max() < 1e-6
    assert np.array_equal(r['idx1'][:n0], single['idx1'])
    assert rel_err(r['logit'][:n0], single['logit']) < 1e-6
    print(f'full size B=6: pair 5 |dF|={dF:.2e} rel|dlogit|={dl:.2e}')
