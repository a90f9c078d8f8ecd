"""CPU restatement of the feature-space 1-NN search.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Pinned**: checked against the
reference's own `core/knn.py::find_knn_gpu` + `core/metrics.py::pdist` (both
import and run on CPU torch) by `tests/golden/make_golden.py`.

Follows `core/knn.py:23-74` (knn=1 only, the only value the hot path uses,
`core/deep_global_registration.py:175-179`) and `core/metrics.py:62-69`.
"""
import numpy as np
import torch


def pdist(A, B, dist_type='L2'):
    """core/metrics.py:62-69."""
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    if dist_type == 'L2':
        return torch.sqrt(D2 + 1e-7)
    if dist_type == 'SquareL2':
        return D2
    raise NotImplementedError('Not implemented')


def find_knn(F0, F1, nn_max_n=-1, return_distance=False):
    """core/knn.py:23-74 with knn=1.  Chunked branch (`nn_max_n > 1`): L2
    distance, indices shaped [N0,1]; unchunked branch: squared L2, [N0]."""
    F0 = torch.as_tensor(np.asarray(F0), dtype=torch.float32)
    F1 = torch.as_tensor(np.asarray(F1), dtype=torch.float32)
    if nn_max_n > 1:
        N = len(F0)
        C = int(np.ceil(N / nn_max_n))
        dists, inds = [], []
        for i in range(C):
            d = pdist(F0[i * nn_max_n:(i + 1) * nn_max_n], F1, 'L2')
            m, ind = d.min(dim=1, keepdim=True)
            dists.append(m)
            inds.append(ind)
        dists, inds = torch.cat(dists), torch.cat(inds)
    else:
        d = pdist(F0, F1, 'SquareL2')
        m, inds = d.min(dim=1)
        dists = m.unsqueeze(1)
    if return_distance:
        return inds.numpy(), dists.numpy()
    return inds.numpy()


def knn_sqdist_f64(F0, F1, idx):
    """Exact (float64) squared distances of the chosen pairs; used by tests to
    decide whether an index mismatch is a genuine rounding tie."""
    F0 = np.asarray(F0, np.float64)
    F1 = np.asarray(F1, np.float64)
    return ((F0 - F1[np.asarray(idx).reshape(-1)]) ** 2).sum(1)
