"""oracle/parity.py (the iteration-matched refinement comparison used by the GPU tests and by the bench line) on the
CPU: with the oracle itself as the implementation under test the deviation is exactly zero; with an implementation
that sums in another order (a row permutation of the same input) the deviation is what `reference_band` is meant to
measure -- it must come out below the band the function reports for that family of perturbations."""
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
