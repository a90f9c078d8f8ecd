"""`weighted_procrustes` / `GlobalRegistration` with the reference signatures
(core/registration.py:91-113, 135-194), executed by the persistent registration kernel."""
import numpy as np
import torch

from .. import ops

F32_EPS = float(np.finfo(np.float32).eps)   # HighDimSmoothL1Loss.eps, core/loss.py:44


def _to_device(a, like=None):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(a).float()
    if not a.is_cuda:
        a = a.to(like.device if (like is not None and like.is_cuda) else 'cuda')
    return a


def weighted_procrustes(X, Y, w, eps=F32_EPS):
    """X, Y [N,3], w [N] or [N,1] -> R [3,3] float32, t [3] float32 (CPU tensors, like the
    reference which finishes on the host at :105-112)."""
    X = _to_device(X)
    R, t = ops.weighted_procrustes(X, _to_device(Y, X), _to_device(w, X), eps)
    return torch.from_numpy(R), torch.from_numpy(t)


def GlobalRegistration(points, trans_points, weights=None, max_iter=1000, verbose=False,
                       stat_freq=20, max_break_count=20, break_threshold_ratio=1e-5, loss_fn=None,
                       quantization_size=1):
    """Returns (R [3,3], t [1,3], {'iterations','loss','break_count'}) like the reference."""
    if loss_fn is not None:
        raise NotImplementedError('custom loss functions are not supported; the kernel implements '
                                  'HighDimSmoothL1Loss (core/loss.py:42-61)')
    points = _to_device(points)
    trans_points = _to_device(trans_points, points)
    if weights is None:
        # argmin_se3_squared_dist (:67-88) == weighted Procrustes with unit weights (up to the eps in
        # the normalisation); HighDimSmoothL1Loss(None) is a plain mean == unit weights
        weights = torch.ones(points.shape[0], 1, device=points.device)
    weights = _to_device(weights, points)
    R, t, stats = ops.se3_refine(points, trans_points, weights, quantization_size, max_iter,
                                 max_break_count, break_threshold_ratio)
    return (torch.from_numpy(R).to(points.device), torch.from_numpy(t).reshape(1, 3).to(points.device), stats)
