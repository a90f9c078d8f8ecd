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
