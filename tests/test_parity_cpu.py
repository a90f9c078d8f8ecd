"""oracle/parity.py (the iteration-matched refinement comparison used by the GPU tests and by the bench line) on the
CPU: with the oracle itself as the implementation under test the deviation is exactly zero; with an implementation
that sums in another order (a row permutation of the same input) the deviation is what `reference_band` is meant to
measure -- it must come out below the band the function reports for that family of perturbations."""
import os

import numpy as np

from oracle import parity, registration as oreg


def _problem(n=600, seed=0, outliers=0.5):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    ang = 0.4
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    Y = (X @ R.T + np.float32([0.1, -0.2, 0.05]) + rng.normal(scale=0.01, size=(n, 3))).astype(np.float32)
    bad = rng.random(n) < outliers
    Y[bad] = rng.uniform(-1, 1, (bad.sum(), 3)).astype(np.float32)
    w = np.where(bad, 0.05, 0.95).astype(np.float32).reshape(-1, 1)
    return X, Y, w


def test_oracle_against_itself_is_exact():
    X, Y, w = _problem()

    def refine(Xn, Yn, wn, max_iter, max_break):
        R, t, st = oreg.global_registration(Xn, Yn, wn, max_iter=max_iter, max_break_count=max_break,
                                            quantization_size=0.1, break_threshold_ratio=1e-4)
        return R, t, st
    r = parity.iteration_matched(X, Y, w, refine, quantization_size=0.1, break_threshold_ratio=1e-4)
    assert r['dR'] == 0.0 and r['dt'] == 0.0 and r['band'] is None
    assert r['iterations_impl'] == r['iterations_oracle'] and r['iterations'] >= 1


def test_a_reordered_implementation_stays_inside_the_reported_band():
    X, Y, w = _problem(seed=3)
    perm = np.random.default_rng(0).permutation(len(X))

    def refine(Xn, Yn, wn, max_iter, max_break):   # the same algorithm, rows in another order (other f32 sums)
        R, t, st = oreg.global_registration(Xn[perm], Yn[perm], wn[perm], max_iter=max_iter, max_break_count=max_break,
                                            quantization_size=0.1, break_threshold_ratio=1e-4)
        return R, t, st
    r = parity.iteration_matched(X, Y, w, refine, always_band=True, quantization_size=0.1, break_threshold_ratio=1e-4)
    assert r['band'] is not None and max(r['dR'], r['dt']) <= max(1e-4, r['band'])
    assert abs(r['loss'] - r['loss_oracle']) <= 2e-3 * abs(r['loss_oracle']) + 1e-9


def test_bench_checker_legs_run_on_cpu(monkeypatch):
    """`bench.py`'s parity + CPU-baseline leg (the part of the driver's no-flag command that runs AFTER the timed
    region) end to end on a tiny pair, with the oracle itself standing in for the HIP outputs and the HIP refinement:
    every key the bench line promises is there, the deviations are rounding noise, the flags say so.  (Round 4 shipped a name
    collision in this function to the GPU box once: nothing on the CPU side executed it.)"""
    import argparse
    import bench   # (repo root is on sys.path: tests/conftest.py)
    from deepglobalregistration_amd import ops, synth
    from oracle import pipeline as opipe, registration as oreg, resunet as oresunet
    voxel = 0.05
    ck = synth.synth_checkpoint(seed=0, voxel_size=voxel, feat_conv1_kernel_size=3)
    a, b, T_gt = synth.synth_pair(1, n_raw=1500)
    p0, c0, _ = opipe.preprocess(a, voxel)
    p1, c1, _ = opipe.preprocess(b, voxel)
    F0 = oresunet.resunet_forward(ck['state_dict'], c0, np.ones((len(c0), 1), np.float32), 3, 3, True)
    F1 = oresunet.resunet_forward(ck['state_dict'], c1, np.ones((len(c1), 1), np.float32), 3, 3, True)
    g = synth.gt_correspondences(p0, p1, T_gt, voxel, frac=1.0)
    idx1 = np.where(g >= 0, g, 0)
    c6, f6 = opipe.inlier_inputs(p0, p1, c0, c1, np.arange(len(p0)), idx1)
    logit = oresunet.resunet_forward(ck['state_dict_inlier'], c6, f6, 6, 3, False).reshape(-1)
    forced = synth.gt_forced_logits(p0, p1[idx1], T_gt, voxel).reshape(-1)

    def se3_refine(X, Y, w, q, max_iter, max_break, ratio):       # the "implementation under test" = the oracle
        R, t, st = oreg.global_registration(X.numpy(), Y.numpy(), w.numpy(), max_iter=max_iter, max_break_count=max_break,
                                            break_threshold_ratio=ratio, quantization_size=q)
        return R, t, st
    monkeypatch.setattr(ops, 'se3_refine', se3_refine)
    monkeypatch.setattr(os, 'cpu_count', lambda: 4)   # (the leg sets torch's thread count from it: see conftest.py)
    args = argparse.Namespace(voxel=voxel, no_refine=False)
    pair0 = {'xyz0': p0, 'coords0': c0, 'xyz1': p1, 'coords1': c1, 'idx1': idx1, 'F0': F0, 'F1': F1, 'logit': logit,
             'forced': forced, 'device': 'cpu'}
    parity, base = bench.oracle_parity_and_baseline(ck, args, pair0, True)
    # (two runs of the oracle agree to rounding only: its BLAS sums depend on the thread count the process is at)
    assert parity['dF'] < 1e-6 and parity['dlogit_rel'] < 1e-6 and parity['features_logits_within_1e-4']
    if 'dR' in parity:                        # the tiny pair passes the gate with all-ground-truth matches
        assert parity['dR'] < 1e-6 and parity['dt'] < 1e-6 and parity['rt_within_1e-4'] and parity['within_1e-4'] and parity['ok']
    assert base['kind'] == 'port' and base['value'] > 0 and base['cores'] >= 1
    assert set(base['stage_s']) == {'fcgf', 'inlier_net', 'knn', 'registration'}
    assert len(base['stage_runs_s']['fcgf']) == 3 and len(base['stage_runs_s']['inlier_net']) == 3
    assert len(base['stage_runs_s']['registration']) == 3
