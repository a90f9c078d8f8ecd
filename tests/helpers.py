"""Shared helpers for the GPU parity tests."""
import os

import numpy as np


def random_cloud_coords(rng, n_pts, extent, D=3, batch=0, surface=True):
    """Unique integer voxel coordinates [N,1+D] (batch column first), roughly surface-like so that
    neighbour counts resemble real scans; includes negative coordinates."""
    if surface:
        # points on a few random planes + noise, then quantised
        pts = []
        for _ in range(4):
            o = rng.uniform(-extent, extent, D)
            a = rng.normal(size=D); a /= np.linalg.norm(a)
            b = rng.normal(size=D); b -= a * (a @ b); b /= np.linalg.norm(b)
            uv = rng.uniform(-extent, extent, (n_pts // 4, 2))
            pts.append(o + uv[:, :1] * a + uv[:, 1:] * b + rng.normal(scale=0.3, size=(n_pts // 4, D)))
        pts = np.floor(np.concatenate(pts)).astype(np.int32)
    else:
        pts = rng.integers(-extent, extent, (n_pts, D)).astype(np.int32)
    _, first = np.unique(pts, axis=0, return_index=True)
    pts = pts[np.sort(first)]
    return np.concatenate([np.full((len(pts), 1), batch, np.int32), pts], axis=1)


def kmap_set(k, i, o):
    return set(zip(np.asarray(k).tolist(), np.asarray(i).tolist(), np.asarray(o).tolist()))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def _report(line):
    """One line of the refinement-parity table: printed, and appended to $DGR_PARITY_REPORT/refine_parity.txt when
    that directory is named (the round's table is committed as profiles/r03_refine_parity.txt)."""
    print(line)
    rep = os.environ.get('DGR_PARITY_REPORT')
    if rep:
        os.makedirs(rep, exist_ok=True)
        with open(os.path.join(rep, 'refine_parity.txt'), 'a') as f:
            f.write(line + '\n')


def _case():
    return os.environ.get('PYTEST_CURRENT_TEST', '?').split('::', 1)[-1].replace(' (call)', '')


def assert_refine_parity(X, Y, w, R, t, stats, tol=1e-4, window=40, **kw):
    """FREE-RUNNING comparison: R, t of the HIP refinement vs the oracle (= the reference algorithm), each side
    stopping by its own break counter.

    Bound: `tol` (1e-4, the north_star tolerance) -- or, where the reference itself does not settle to
    that level, the reference's own terminal oscillation band.  Adam with lr = 0.1 * 0.999^i keeps
    oscillating around the optimum and amplifies the f32 rounding of the loss/gradient sums: on
    plateau-like inputs the oracle run on a mere row permutation of the same input differs from itself
    by 1e-4 .. 6e-3 in R/t at equal iteration count and stops ~10 iterations earlier or later
    (measured).  The HIP kernel accumulates the sums in f64 in a fixed order, i.e.
    it is one more such re-ordering and stops somewhere in the same band.  Criterion: equal final
    losses (2e-3 relative) and |dR|, |dt| <= max(tol, band), band = the largest excursion of the
    oracle's own (R, t) iterates from its final value over the last `window` iterations before the
    later of the two stopping points."""
    from oracle import registration as oreg
    X, Y, w = np.asarray(X), np.asarray(Y), np.asarray(w).reshape(-1, 1)
    R = np.asarray(R, np.float64).reshape(3, 3)
    t = np.asarray(t, np.float64).reshape(3)
    Ro, to, so = oreg.global_registration(X, Y, w, **kw)
    to = to.reshape(3)
    assert abs(so['loss'] - stats['loss']) <= 2e-3 * abs(so['loss']) + 1e-9, (so, stats)
    dR, dt = np.abs(R - Ro).max(), np.abs(t - to).max()
    band = None
    if max(dR, dt) >= tol:
        last = max(so['iterations'], stats['iterations']) + 1
        kw2 = dict(kw)
        kw2.update(max_iter=last, max_break_count=10 ** 9)
        trace = []
        oreg.global_registration(X, Y, w, trace=trace, **kw2)
        tail = trace[max(0, min(so['iterations'], stats['iterations']) - window):]
        band = max(max(np.abs(Ri - Ro).max(), np.abs(ti - to).max()) for Ri, ti in tail)
    arb = ''
    if band is not None:
        # the f64 arbiter, free-running as well (informational: the three sides stop at three iteration counts, and what
        # separates them is the terminal oscillation the band measures)
        import torch
        R8, t8, s8 = oreg.global_registration(X, Y, w, dtype=torch.float64, **kw)
        e = [max(np.abs(Ra - R8).max(), np.abs(np.asarray(ta).reshape(3) - t8.reshape(3)).max()) for Ra, ta in ((R, t), (Ro, to))]
        arb = f'  f64 arbiter stops at {s8["iterations"]}: |hip - f64| {e[0]:.1e}  |f32 ref - f64| {e[1]:.1e}'
    _report(f'free-running       {_case():70s} n={len(X):6d} iterations hip {stats["iterations"]:4d} oracle {so["iterations"]:4d}  '
            f'dR {dR:.1e} dt {dt:.1e}  ' + (f'oracle tail band {band:.1e}' if band is not None else 'inside 1e-4') + arb)
    if band is not None:
        assert dR <= max(tol, band) and dt <= max(tol, band), (dR, dt, band, so, stats)
    return Ro, to, so


def oracle_pair_counts(maps, conv1_ks):
    """Kernel-map pair count of every conv of ResUNetBN2C in forward order (23 entries), from the
    oracle's maps (`oracle.resunet.SparseMaps`); k = 1 convs count one pair per row."""
    n1 = len(maps.coords[1])
    same = {ts: len(maps.same(ts)[0]) for ts in (1, 2, 4, 8)}
    down = {ts: len(maps.down(ts)[0]) for ts in (1, 2, 4)}
    c1 = len(maps.same(1, conv1_ks)[0])
    return [c1, same[1], same[1], down[1], same[2], same[2], down[2], same[4], same[4], down[4], same[8], same[8],
            down[4], same[4], same[4], down[2], same[2], same[2], down[1], same[1], same[1], n1, n1]



def _hip_refine_from(Xn, Yn, wn, state, max_iter, kw):
    import torch
    from deepglobalregistration_amd import ops
    return ops.se3_refine_from(torch.from_numpy(Xn).cuda(), torch.from_numpy(Yn).cuda(), torch.from_numpy(wn).cuda(), state,
                               max_iter, kw.get('quantization_size', 1.0), 10 ** 9, kw.get('break_threshold_ratio', 1e-5))


def assert_window_accuracy(X, Y, w, factor=2.0, **kw):
    """LOCAL accuracy of the HIP refinement against the f64 arbiter (`oracle.parity.window_accuracy`): from the f32
    reference's own optimiser state before step i, four steps by the f32 reference, by the reference algorithm in float64
    and by the HIP kernel (`dgr_debug_se3_refine_from`).  Required: over the windows (the stationary start i = 0 listed but
    left out, see `window_accuracy`) the HIP kernel is not farther from the f64 steps than the f32 reference is:
    rms(e_hip) <= factor * rms(e_f32) + 1e-7.  Returns (rms_f32, rms_hip, rows)."""
    from oracle import parity
    rows = parity.window_accuracy(X, Y, w, lambda a, b, c, st, mi: _hip_refine_from(a, b, c, st, mi, kw), **kw)
    a = np.array([r for r in rows if r[0] > 0], np.float64).reshape(-1, 3)
    rms32, rmsh = float(np.sqrt((a[:, 1] ** 2).mean())), float(np.sqrt((a[:, 2] ** 2).mean()))
    worst = max(rows[1:], key=lambda r: r[2] / max(r[1], 1e-9)) if len(rows) > 1 else rows[0]
    _report(f'4-step windows     {_case():70s} n={len(np.asarray(X)):6d} {len(a):2d} windows from the reference\'s own states: '
            f'rms |f32 ref - f64| {rms32:.1e}  rms |hip - f64| {rmsh:.1e}  (start 0: {rows[0][1]:.1e} / {rows[0][2]:.1e}; '
            f'worst window i={worst[0]}: {worst[1]:.1e} / {worst[2]:.1e})')
    assert rmsh <= factor * rms32 + 1e-7, (rms32, rmsh, rows)
    return rms32, rmsh, rows


def assert_iteration_matched(X, Y, w, tol=1e-4, **kw):
    """Iteration-matched refinement parity on the given inputs (the measurement is `oracle.parity.iteration_matched`):
    the oracle runs freely (k iterations), then BOTH sides run exactly k iterations (max_iter = k, max_break_count =
    10^9; the stopping logic is out of the picture).

    Required: |dR| <= tol, |dt| <= tol max(1, |t|) against the f32 reference, equal final losses (2e-3).  Where that does
    not hold, an F64 ARBITER decides which side is off (round-5 verdict, item 4): the reference ALGORITHM evaluated in
    float64 on the same inputs for the same k iterations (`oracle.registration.global_registration(dtype=float64)`).
      (a) err_hip = |HIP - f64| <= max(tol, 1.5 err_f32), err_f32 = |f32 reference - f64|: HIP is not farther from exact
          arithmetic than the reference is -- accepted; or
      (b) the trajectory is chaotic on this input -- Adam starts at the weighted-Procrustes estimate, where the gradient
          is rounding noise and the first step is +-lr per parameter whatever its size, so a re-ordering of the f32 sums
          picks another sign pattern and some patterns are still 1e-3 away after 150 iterations (tools/diag_refine.py) --
          then BOTH of the following are required: err_hip <= max(tol, 1.5 err_family), err_family = the largest
          |f32 reference on a perturbed input - f64| over a row permutation and +-1..8 ulp input scalings (the reference
          itself lands that far from exact arithmetic when only its rounding changes), AND the LOCAL accuracy test
          passes on this input (`assert_window_accuracy`: four HIP steps from any state of the reference's trajectory
          are not farther from four f64 steps than four f32-reference steps are) -- i.e. the kernel's arithmetic is as
          good as the reference's and the distance is the reference's own sensitivity, not a kernel error.
    Returns (deviation from the f32 reference, err_hip_f64, err_f32_f64, err_family_f64 | None)."""
    import torch
    from deepglobalregistration_amd import ops
    from oracle import parity

    def refine(Xn, Yn, wn, max_iter, max_break):
        R, t, st = ops.se3_refine(torch.from_numpy(Xn).cuda(), torch.from_numpy(Yn).cuda(), torch.from_numpy(wn).cuda(),
                                  kw.get('quantization_size', 1.0), max_iter, max_break, kw.get('break_threshold_ratio', 1e-5))
        return R, t, st
    Xn, Yn = np.asarray(X, np.float32), np.asarray(Y, np.float32)
    wn = np.asarray(w, np.float32).reshape(-1, 1)
    r = parity.iteration_matched(Xn, Yn, wn, refine, tol=10.0, **kw)     # (tol = 10: no band runs; the arbiter below)
    d = max(r['dR'], r['dt'])
    arb = parity.f64_arbiter(Xn, Yn, wn, r['iterations'], r['R_impl'], r['t_impl'], r['R_oracle'], r['t_oracle'], r['t_scale'],
                             family=False, **kw)
    eh, e32, efam = arb['err_impl_f64'], arb['err_f32_f64'], None
    verdict = 'inside 1e-4 of the f32 reference' if d <= tol else 'hip not farther from f64 than the f32 reference (x1.5)'
    if d > tol and eh > max(tol, 1.5 * e32):
        efam = parity.f64_arbiter(Xn, Yn, wn, r['iterations'], r['R_impl'], r['t_impl'], r['R_oracle'], r['t_oracle'],
                                  r['t_scale'], family=True, **kw)['err_family_f64']
        verdict = f'chaotic input: f32 reference on perturbed inputs vs f64 {efam:.1e}'
    _report(f'iteration-matched  {_case():70s} n={len(Xn):6d} iterations {r["iterations"]:4d} (both sides)          '
            f'dR {r["dR"]:.1e} dt {r["dt"]:.1e}  |hip - f64| {eh:.1e}  |f32 ref - f64| {e32:.1e}  {verdict}'
            f'  loss hip {r["loss"]:.6e} oracle {r["loss_oracle"]:.6e}')
    assert r['iterations_impl'] == r['iterations_oracle'], r
    if d <= tol:
        assert abs(r['loss'] - r['loss_oracle']) <= 2e-3 * abs(r['loss_oracle']) + 1e-9, r
    elif efam is not None:
        assert eh <= max(tol, 1.5 * efam), (eh, e32, efam, r['iterations'])
        assert_window_accuracy(Xn, Yn, wn, **kw)
    return d, eh, e32, efam


def harness_dgr(config, device):
    """`DeepGlobalRegistration` with the parity tests' hooks: `harness_matches(xyz0, xyz1, idx1) -> idx1` replaces matches
    after the search ran, `harness_logits(xyz0, xyz1_matched, logit) -> logit` replaces logits after the inlier net ran
    (device tensors in and out; None = the product's behaviour), intermediates kept.  The product class only has the two
    identity extension points these override."""
    from deepglobalregistration_amd import ops
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration

    class HarnessDGR(DeepGlobalRegistration):
        harness_matches = None
        harness_logits = None

        def _post_matching(self, xyz0, xyz1, corres_idx1):
            if self.harness_matches is None:
                return corres_idx1
            return self.harness_matches(xyz0, xyz1, corres_idx1).long().reshape(-1)

        def _post_inlier_prediction(self, xyz0, xyz1, corres_idx1, logit):
            if self.harness_logits is None:
                return logit
            return self.harness_logits(xyz0, ops.gather_rows3(xyz1, corres_idx1), logit).float().reshape(-1, 1)

    return HarnessDGR(dict(config, keep_intermediates=True), device)
