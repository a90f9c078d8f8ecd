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
import torch

from . import registration as oreg


def perturbation_family(X, Y, w, n_ulps=8):
    """The inputs the reference is re-run on to see how far it moves from itself: a row permutation and the source /
    target points scaled by 1 +- j ulp."""
    perm = np.random.default_rng(0).permutation(len(X))
    variants = [(X[perm], Y[perm], w[perm])]
    for j in range(1, n_ulps + 1):
        variants.append((X * np.float32(1 + ((-1) ** j) * j * 2.0 ** -23), Y * np.float32(1 + (j % 3 - 1) * 2.0 ** -23), w))
    return variants


def f64_arbiter(X, Y, w, k, R_impl, t_impl, R_f32, t_f32, ts, family=False, n_ulps=8, **kw):
    """The reference ALGORITHM evaluated in float64 on the same f32 inputs for exactly k iterations
    (`oracle.registration.global_registration(dtype=torch.float64)`), and how far the implementation under test and the
    f32 reference each end from it.  `family`: also the largest distance from the f64 result of the f32 reference run on
    the perturbation family (what a re-ordering of the f32 sums / a last-bit change of the inputs does to the reference
    itself, measured against the same yardstick)."""
    kw2 = dict(kw, max_iter=k, max_break_count=10 ** 9)
    R8, t8, _ = oreg.global_registration(X, Y, w, dtype=torch.float64, **kw2)

    def dev(R, t):
        return max(float(np.abs(np.asarray(R, np.float64).reshape(3, 3) - R8).max()),
                   float(np.abs(np.asarray(t, np.float64).reshape(-1) - t8.reshape(-1)).max()) / ts)
    out = {'err_impl_f64': dev(R_impl, t_impl), 'err_f32_f64': dev(R_f32, t_f32), 'err_family_f64': None}
    if family:
        out['err_family_f64'] = max(dev(*oreg.global_registration(Xv, Yv, wv, **kw2)[:2])
                                    for Xv, Yv, wv in perturbation_family(X, Y, w, n_ulps))
    return out


def window_accuracy(X, Y, w, refine_from, W=4, starts=None, **kw):
    """LOCAL accuracy, free of the trajectory's chaotic amplification: the f32 reference runs freely and records its
    optimiser state before every step; from the state before step i, W steps are taken (a) by the f32 reference (its own
    trajectory), (b) by the reference algorithm in float64, (c) by the implementation under test
    (`refine_from(X, Y, w, state, max_iter) -> end state`).  Returns rows (i, e_f32, e_impl) with e = max |prm - prm_f64|
    over the 9 parameters after the W steps.  Start 0 is the weighted-Procrustes estimate, where the gradient is rounding
    noise and the first Adam step is +-lr per parameter whatever its size: listed, to be judged separately."""
    X, Y = np.asarray(X, np.float32), np.asarray(Y, np.float32)
    w = np.asarray(w, np.float32).reshape(-1, 1)
    k = max(1, oreg.global_registration(X, Y, w, **kw)[2]['iterations'])
    states = []
    oreg.global_registration(X, Y, w, states=states, **dict(kw, max_iter=k, max_break_count=10 ** 9))
    last = len(states) - 1
    if starts is None:
        starts = [i for i in (0, 1, 2, 3, 5, 8, 12, 20, 30, 45, 60, 80, 100, 130, 160, 200, 250, 300, 400, 600, 800) if i + W <= last]
    rows = []
    for i in starts:
        end32 = states[i + W]['prm']
        end64 = oreg.global_registration(X, Y, w, dtype=torch.float64, start=states[i],
                                         **dict(kw, max_iter=i + W, max_break_count=10 ** 9))[2]['prm']
        endim = np.asarray(refine_from(X, Y, w, states[i], i + W)['prm'], np.float64)
        rows.append((i, float(np.abs(end32 - end64).max()), float(np.abs(endim - end64).max())))
    return rows


def reference_band(X, Y, w, k, Ro, to, ts, counts=None, n_ulps=8, **kw):
    """Largest |R|, |t| / ts excursion of the oracle from itself over the perturbation family at iteration counts
    `counts` (default: k, k - 7, k - 15, k - 30)."""
    variants = perturbation_family(X, Y, w, n_ulps)
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
            'R_impl': np.asarray(R, np.float64).reshape(3, 3), 't_impl': np.asarray(t, np.float64).reshape(-1),
            'loss': float(st['loss']), 'loss_oracle': float(so['loss']), 'iterations_impl': int(st['iterations']),
            'iterations_oracle': int(so['iterations']), 't_scale': ts, 'tolerance': tol, 'R_oracle': Ro, 't_oracle': to}
