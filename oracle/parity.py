"""Iteration-matched parity of an SE(3) refinement implementation against the reference algorithm.

TEST INFRASTRUCTURE (see oracle/__init__.py): used by `tests/helpers.py` and by the `parity` leg of `bench.py`
(outside the timed region).  The measurement, without any assertion:

* the oracle (`oracle.registration.global_registration` = `core/registration.py:135-194`) runs freely and stops
  after k iterations; then BOTH sides run exactly k iterations (`max_iter = k`, the break counter out of reach), so
  the discrete stopping logic is out of the picture;
* `dR` = max |R - R_oracle|, `dt` = max |t - t_oracle| (relative to max(1, |t|): metre-scale translations count
  absolutely, KITTI-scale ones relatively);
* `band` = how far the REFERENCE moves from ITSELF at the same k when only its floating-point summation order /
  the last bit of its inputs change (a row permutation, and the source and target points scaled by 1 +- j ulp).
  Adam starts at the weighted-Procrustes estimate, a stationary point of the loss whenever every inlier residual is
  below the SmoothL1 knee: the first step is +-lr per parameter with signs decided by rounding noise, and the
  iterates oscillate around the optimum afterwards.  `band` is only measured when the
  deviation exceeds `tol` (it costs 9 x 4 oracle runs) unless `always_band` is set.
"""
import numpy as np

from . import registration as oreg


def reference_band(X, Y, w, k, Ro, to, ts, counts=None, n_ulps=8, **kw):
    """Largest |R|, |t| / ts excursion of the oracle from itself over the perturbation family at iteration counts
    `counts` (default: k, k - 7, k - 15, k - 30)."""
    perm = np.random.default_rng(0).permutation(len(X))
    variants = [(X[perm], Y[perm], w[perm])]
    for j in range(1, n_ulps + 1):
        variants.append((X * np.float32(1 + ((-1) ** j) * j * 2.0 ** -23), Y * np.float32(1 + (j % 3 - 1) * 2.0 ** -23), w))
    band = 0.0
    kw2 = dict(kw, max_break_count=10 ** 9)
    if counts == 'short':
        counts = sorted({k, max(1, k - 15)})
    for kk in (counts or sorted({k, max(1, k - 7), max(1, k - 15), max(1, k - 30)})):
        kw3 = dict(kw2, max_iter=kk)
        Rb, tb = (Ro, to) if kk == k else oreg.global_registration(X, Y, w, **kw3)[:2]
        for Xv, Yv, wv in variants:
            Rp, tp, _ = oreg.global_registration(Xv, Yv, wv, **kw3)
            band = max(band, float(np.abs(Rp - Rb).max()), float(np.abs(tp.reshape(-1) - tb.reshape(-1)).max()) / ts)
    return band


def iteration_matched(X, Y, w, refine, tol=1e-4, always_band=False, band_counts=None, band_ulps=8, **kw):
    """`refine(X, Y, w, max_iter, max_break_count) -> (R [3,3], t [3], stats dict)` is the implementation under test.
    `kw` = the oracle's keyword arguments (quantization_size, break_threshold_ratio).  Returns a dict."""
    X, Y = np.asarray(X, np.float32), np.asarray(Y, np.float32)
    w = np.asarray(w, np.float32).reshape(-1, 1)
    free = oreg.global_registration(X, Y, w, **kw)[2]
    k = max(1, free['iterations'])
    kw2 = dict(kw, max_iter=k, max_break_count=10 ** 9)
    Ro, to, so = oreg.global_registration(X, Y, w, **kw2)
    R, t, st = refine(X, Y, w, k, 10 ** 9)
    ts = max(1.0, float(np.abs(to).max()))
    dR = float(np.abs(np.asarray(R, np.float64).reshape(3, 3) - Ro).max())
    dt = float(np.abs(np.asarray(t, np.float64).reshape(-1) - to.reshape(-1)).max()) / ts
    band = None
    if always_band or max(dR, dt) > tol:
        band = reference_band(X, Y, w, k, Ro, to, ts, counts=band_counts, n_ulps=band_ulps, **kw)
    return {'iterations': k, 'free_running_iterations_oracle': free['iterations'], 'dR': dR, 'dt': dt, 'band': band,
            'loss': float(st['loss']), 'loss_oracle': float(so['loss']), 'iterations_impl': int(st['iterations']),
            'iterations_oracle': int(so['iterations']), 't_scale': ts, 'tolerance': tol, 'R_oracle': Ro, 't_oracle': to}
