"""Generate golden vectors from the REFERENCE's own importable modules.

Run in the build container only (needs /root/reference; the GPU box does not
have it):

    python tests/golden/make_golden.py

Imports `core/knn.py`, `core/metrics.py`, `core/loss.py`, `core/registration.py`
from /root/reference unchanged (they run on CPU torch) and stores seeded
input/output vectors as small .npz files next to this script.  The reference
has no tests or golden vectors of its own, so these are
the pins for the kNN / Procrustes / refinement stages.  MinkowskiEngine is not
importable, so the sparse-conv arithmetic cannot be pinned this way (see
oracle/__init__.py); the model classes built on it are, by
make_golden_model.py next to this script.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get('DGR_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    from core.knn import find_knn_gpu
    from core.loss import HighDimSmoothL1Loss
    from core.registration import (GlobalRegistration, ortho2rotation,
                                   weighted_procrustes)
    torch.manual_seed(0)
    torch.set_num_threads(1)          # summation order independent of the host core count
    rng = np.random.default_rng(1234)

    # (i) kNN ---------------------------------------------------------------
    out = {}
    for tag, (n0, n1, c) in {'a': (700, 900, 32), 'b': (257, 64, 32), 'c': (33, 1000, 16)}.items():
        F0 = rng.standard_normal((n0, c)).astype(np.float32)
        F1 = rng.standard_normal((n1, c)).astype(np.float32)
        F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
        F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
        ic, dc = find_knn_gpu(torch.from_numpy(F0), torch.from_numpy(F1), nn_max_n=250, knn=1,
                              return_distance=True)
        iu, du = find_knn_gpu(torch.from_numpy(F0), torch.from_numpy(F1), nn_max_n=-1, knn=1,
                              return_distance=True)
        out.update({f'{tag}_F0': F0, f'{tag}_F1': F1,
                    f'{tag}_idx_chunked': ic.numpy(), f'{tag}_dist_chunked': dc.numpy(),
                    f'{tag}_idx_unchunked': iu.numpy(), f'{tag}_dist_unchunked': du.numpy()})
    # exact duplicates in F1: torch CPU min returns the first minimal index
    F0 = rng.standard_normal((50, 32)).astype(np.float32)
    F1 = rng.standard_normal((40, 32)).astype(np.float32)
    F1[20:] = F1[:20]
    it = find_knn_gpu(torch.from_numpy(F0), torch.from_numpy(F1), nn_max_n=250)
    out.update(tie_F0=F0, tie_F1=F1, tie_idx_chunked=it.numpy())
    np.savez_compressed(os.path.join(HERE, 'knn.npz'), **out)

    # (ii) weighted Procrustes -----------------------------------------------
    out = {}
    eps = float(np.finfo(np.float32).eps)

    def rand_pose(r):
        q = r.standard_normal(4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        return R, r.uniform(-1, 1, 3)

    def problem(n, outlier_frac, noise, r):
        R, t = rand_pose(r)
        X = r.uniform(-2, 2, (n, 3))
        Y = X @ R.T + t + r.normal(scale=noise, size=(n, 3))
        nout = int(outlier_frac * n)
        Y[:nout] = r.uniform(-3, 3, (nout, 3))
        w = np.concatenate([r.uniform(0.0, 0.3, nout), r.uniform(0.6, 1.0, n - nout)])
        w[w < 0.05] = 0
        perm = r.permutation(n)
        return (X[perm].astype(np.float32), Y[perm].astype(np.float32),
                w[perm].astype(np.float32).reshape(-1, 1), R, t)

    cases = {'clean': problem(500, 0.0, 0.0, rng), 'noisy': problem(2000, 0.3, 0.01, rng),
             'zeros': problem(300, 0.5, 0.005, rng)}
    cases['zeros'][2][:100] = 0
    # reflection case: planar, mirrored target => det(U)det(V) < 0 branch
    Xp = rng.uniform(-1, 1, (200, 3)).astype(np.float32)
    Yp = Xp.copy()
    Yp[:, 2] *= -1
    cases['reflect'] = (Xp, Yp, np.ones((200, 1), np.float32), None, None)
    for tag, (X, Y, w, _, _) in cases.items():
        R, t = weighted_procrustes(torch.from_numpy(X), torch.from_numpy(Y), torch.from_numpy(w), eps)
        out.update({f'{tag}_X': X, f'{tag}_Y': Y, f'{tag}_w': w,
                    f'{tag}_R': R.numpy(), f'{tag}_t': t.numpy()})
    np.savez_compressed(os.path.join(HERE, 'procrustes.npz'), **out)

    # (iii) GlobalRegistration -------------------------------------------------
    out = {}
    gcases = {
        'clean': (problem(1500, 0.0, 0.002, rng), dict(break_threshold_ratio=1e-4, quantization_size=0.1)),
        'outliers70': (problem(4000, 0.7, 0.01, rng), dict(break_threshold_ratio=1e-4, quantization_size=0.1)),
        'exact': (problem(400, 0.0, 0.0, rng), dict(break_threshold_ratio=1e-4, quantization_size=0.1)),
        'maxiter': (problem(800, 0.5, 0.02, rng), dict(break_threshold_ratio=1e-12, quantization_size=0.1,
                                                        max_iter=60)),
        'default_q': (problem(600, 0.2, 0.01, rng), dict()),
    }
    for tag, ((X, Y, w, _, _), kw) in gcases.items():
        R, t, st = GlobalRegistration(torch.from_numpy(X), torch.from_numpy(Y),
                                      weights=torch.from_numpy(w.copy()), **kw)
        out.update({f'{tag}_X': X, f'{tag}_Y': Y, f'{tag}_w': w,
                    f'{tag}_R': R.detach().numpy(), f'{tag}_t': t.detach().numpy(),
                    f'{tag}_iterations': np.int64(st['iterations']),
                    f'{tag}_loss': np.float64(st['loss']),
                    f'{tag}_break_count': np.int64(st['break_count']),
                    f'{tag}_kw': np.array(repr(kw))})
        print(tag, st)
    np.savez_compressed(os.path.join(HERE, 'refine.npz'), **out)

    # (iv) HighDimSmoothL1Loss straddling s == 1 -------------------------------
    out = {}
    X = rng.standard_normal((64, 3)).astype(np.float32)
    dirs = rng.standard_normal((64, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    q = 0.1
    radii = np.concatenate([np.linspace(0.0, 0.099, 24), [0.0999999, 0.1, 0.1000001],
                            np.linspace(0.101, 0.5, 37)])
    Y = (X + dirs * radii[:, None]).astype(np.float32)
    w = rng.uniform(0, 1, (64, 1)).astype(np.float32)
    lw = HighDimSmoothL1Loss(torch.from_numpy(w), q)(torch.from_numpy(X), torch.from_numpy(Y))
    lu = HighDimSmoothL1Loss(None, q)(torch.from_numpy(X), torch.from_numpy(Y))
    per = []
    for i in range(64):
        per.append(HighDimSmoothL1Loss(None, q)(torch.from_numpy(X[i:i + 1]),
                                                torch.from_numpy(Y[i:i + 1])).item())
    out.update(X=X, Y=Y, w=w, q=np.float64(q), loss_weighted=np.float64(lw.item()),
               loss_unweighted=np.float64(lu.item()), per_point=np.array(per))
    np.savez_compressed(os.path.join(HERE, 'loss.npz'), **out)

    # (v) ortho2rotation -------------------------------------------------------
    P = rng.standard_normal((32, 6)).astype(np.float32)
    P[0] = 0                                   # both clamps hit
    P[1, :3] = 0                               # zero first vector
    P[2, 3:] = P[2, :3] * 2.5                  # parallel vectors -> ||u|| clamp
    P[3] = [1e-9, 0, 0, 0, 1e-9, 0]
    P[4] = [1, 0, 0, 0, 1, 0]
    Rm = ortho2rotation(torch.from_numpy(P)).numpy()
    np.savez_compressed(os.path.join(HERE, 'ortho6d.npz'), P=P, R=Rm)

    # ---- sections added in round 2: their own generators, so that the vectors above do not change ----
    # (vi) ortho2rotation backward (autograd through core/registration.py:16-64, incl. the clamped cases)
    r2 = np.random.default_rng(4321)
    P2 = r2.standard_normal((32, 6)).astype(np.float32)
    P2[0] = 0
    P2[1, :3] = 0
    P2[2, 3:] = P2[2, :3] * 2.5
    P2[3] = [1e-9, 0, 0, 0, 1e-9, 0]
    P2[4] = [1, 0, 0, 0, 1, 0]
    P2[5] = [3, 0, 0, 0, 0, 1e-9]
    G2 = r2.standard_normal((32, 3, 3)).astype(np.float32)
    Pt = torch.from_numpy(P2).clone().requires_grad_(True)
    Rt = ortho2rotation(Pt)
    (Rt * torch.from_numpy(G2)).sum().backward()
    np.savez_compressed(os.path.join(HERE, 'ortho6d_grad.npz'), P=P2, G=G2, R=Rt.detach().numpy(), dP=Pt.grad.numpy())

    # (vii) find_knn_gpu_batch (core/knn.py:106-140): per-pair 1-NN over a concatenated feature batch
    from core.knn import find_knn_gpu_batch
    r3 = np.random.default_rng(777)
    lens = [[700, 900], [1300, 1100], [33, 5]]
    n0, n1 = sum(a for a, _ in lens), sum(b for _, b in lens)
    F0 = r3.standard_normal((n0, 32)).astype(np.float32)
    F1 = r3.standard_normal((n1, 32)).astype(np.float32)
    F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
    F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    per = find_knn_gpu_batch(torch.from_numpy(F0), torch.from_numpy(F1), lens, nn_max_n=250)
    cat, dist = find_knn_gpu_batch(torch.from_numpy(F0), torch.from_numpy(F1), lens, nn_max_n=250,
                                   return_distance=True, concat_results=True)
    np.savez_compressed(os.path.join(HERE, 'knn_batch.npz'), F0=F0, F1=F1, len_batch=np.array(lens),
                        per_pair=np.concatenate([t.numpy().reshape(-1) for t in per]),
                        cat_idx=cat.numpy(), cat_dist=dist.numpy())

    # (viii) evaluation formats / metrics.  util/file.py imports cleanly; scripts/test_3dmatch.py pulls in open3d and
    # MinkowskiEngine at module level, so its `rte_rre` (:38-46) is executed from its own source text, alone.
    import ast
    import math
    import tempfile
    from util.file import read_trajectory
    r4 = np.random.default_rng(99)
    n_rec = 7
    poses = []
    lines = []
    for i in range(n_rec):
        Rr, tr = rand_pose(r4)
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = Rr, tr
        poses.append(T)
        lines.append(f'{i}\t{i + 1 + (i % 3)}\t{n_rec + 30}')
        for row in T:                       # the 3DMatch gt.log layout: tab-separated scientific notation
            lines.append('\t'.join(f'{v: .8e}' for v in row))
    text = '\n'.join(lines) + '\n'
    with tempfile.NamedTemporaryFile('w', suffix='.log', delete=False) as f:
        f.write(text)
    traj = read_trajectory(f.name)
    os.unlink(f.name)
    src = open(os.path.join(REF, 'scripts', 'test_3dmatch.py')).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == 'rte_rre'][0]
    ns = {'np': np, 'math': math}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'scripts/test_3dmatch.py', 'exec'), ns)
    ref_rte_rre = ns['rte_rre']
    Tp, Tg, res = [], [], []
    for i in range(24):
        Rg, tg = rand_pose(r4)
        a = Tg_ = np.eye(4)
        Tg_[:3, :3], Tg_[:3, 3] = Rg, tg
        ang = [0.0, 1e-9, 0.05, 0.2, 0.3, 1.0][i % 6]         # radians: identical, ~equal, below / above 15 degrees
        ax = r4.standard_normal(3); ax /= np.linalg.norm(ax)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        dR = np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * Kx @ Kx
        Tp_ = np.eye(4)
        Tp_[:3, :3] = dR @ Rg
        Tp_[:3, 3] = tg + r4.standard_normal(3) * [0.0, 0.05, 0.2, 0.4][i % 4]
        Tp.append(Tp_); Tg.append(Tg_.copy()); res.append(ref_rte_rre(Tp_, Tg_, 0.3, 15))
    np.savez_compressed(os.path.join(HERE, 'eval_formats.npz'), gt_log_text=np.array(text),
                        traj_meta=np.array([t.metadata for t in traj]), traj_pose=np.stack([t.pose for t in traj]),
                        poses_written=np.stack(poses), T_pred=np.stack(Tp), T_gt=np.stack(Tg), rte_rre=np.stack(res),
                        rte_rre_none=ref_rte_rre(None, Tg[0], 0.3, 15))
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
