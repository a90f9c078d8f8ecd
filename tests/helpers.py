"""Shared helpers for the GPU parity tests."""
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


def assert_refine_parity(X, Y, w, R, t, stats, tol=1e-4, window=40, **kw):
    """R, t of the HIP refinement vs the oracle (= the reference algorithm).

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
    if max(dR, dt) < tol:
        return Ro, to, so
    last = max(so['iterations'], stats['iterations']) + 1
    kw2 = dict(kw)
    kw2.update(max_iter=last, max_break_count=10 ** 9)
    trace = []
    oreg.global_registration(X, Y, w, trace=trace, **kw2)
    tail = trace[max(0, min(so['iterations'], stats['iterations']) - window):]
    band = max(max(np.abs(Ri - Ro).max(), np.abs(ti - to).max()) for Ri, ti in tail)
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
