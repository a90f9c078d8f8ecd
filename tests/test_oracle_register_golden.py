"""`oracle.pipeline.register` against a run of the reference's own `DeepGlobalRegistration.register()`
(core/deep_global_registration.py:238-324 imported from /root/reference by tests/golden/make_golden_register.py, with
stand-ins for the MinkowskiEngine / Open3D imports): every intermediate the reference's methods hand on, and T."""
import os

import numpy as np
import pytest

from oracle import pipeline as opipe

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'register_e2e.npz')


def golden_case():
    from deepglobalregistration_amd import synth
    g = np.load(GOLDEN)
    seed, ks = (int(v) for v in g['spec'])
    ck = synth.synth_checkpoint(seed=seed, voxel_size=float(g['voxel']), feat_conv1_kernel_size=ks)
    return g, ck


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())


def test_oracle_register_equals_the_reference_run():
    g, ck = golden_case()
    o = opipe.register(ck, g['xyz0'], g['xyz1'], clip_weight_thresh=float(g['clip_weight_thresh']), use_icp=False)
    # integer work and selections: exact
    assert np.array_equal(o['xyz0'], g['p0']) and np.array_equal(o['xyz1'], g['p1'])
    assert np.array_equal(o['coords0'], g['coords0']) and np.array_equal(o['coords1'], g['coords1'])
    assert np.array_equal(o['idx1'], g['idx1'].reshape(-1))
    assert np.array_equal(o['feats6'], g['feats6'])
    devs = {'F0': rel(o['F0'], g['F0']), 'F1': rel(o['F1'], g['F1']), 'logit': rel(o['logit'].reshape(-1), g['logit'].reshape(-1))}
    dT = float(np.abs(o['T'] - g['T']).max())          # (element-wise: arccos of a trace cannot resolve below 0.01 deg in f32)
    print({k: f'{v:.1e}' for k, v in devs.items()}, f'max |T - T_reference| = {dT:.2e}, iterations {o["stats"]["iterations"]}')
    assert max(devs.values()) < 5e-6
    assert o['status'] == 'ok' and o['confident']
    assert dT < 1e-5                                   # f32 refinement from logits that differ in the last bits
    # the golden pose is not the planted translation: outliers with arbitrary weights pull on it
    assert np.abs(g['T'] - g['T_gt']).max() > 1e-3


def test_same_logits_same_bits():
    """With the reference run's own logits in place of the oracle's (1e-6 apart), T is the reference's to the bit: what
    is left of the deviation above is the inlier net's summation order, not the glue."""
    g, ck = golden_case()
    o = opipe.register(ck, g['xyz0'], g['xyz1'], clip_weight_thresh=float(g['clip_weight_thresh']), use_icp=False,
                       forced_logit_fn=lambda x0, x1, logit: g['logit'])
    assert np.array_equal(o['T'], g['T'])
