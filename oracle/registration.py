"""CPU restatement of weighted Procrustes + robust SE(3) refinement.

TEST INFRASTRUCTURE (see oracle/__init__.py).  **Pinned**: checked against the
reference's own `core/registration.py` / `core/loss.py` (both import and run
on CPU torch) by `tests/golden/make_golden.py`.

* `weighted_procrustes`   follows `core/registration.py:91-113`
* `rot6d_to_matrix`       follows `core/registration.py:16-64` (`ortho2rotation`)
* `smooth_l1_highdim`     follows `core/loss.py:42-61` (`HighDimSmoothL1Loss`,
                          including its discontinuity at s == 1)
* `global_registration`   follows `core/registration.py:135-194`
                          (`GlobalRegistration` with weights given)

The optimiser is torch's own Adam / ExponentialLR driven by autograd, exactly
like the reference, so the trajectory (and the discrete stopping logic) is the
reference's up to floating-point summation order.
"""
import warnings

import numpy as np
import torch

F32_EPS = float(np.finfo(np.float32).eps)   # core/loss.py:44


def weighted_procrustes(X, Y, w, eps=F32_EPS, dtype=torch.float32):
    """X,Y [N,3] f32, w [N,1] f32 -> R [3,3] f32, t [3] f32.  (`dtype=torch.float64`: the same formulas evaluated in
    double on the same f32 inputs -- the ARBITER of tests/helpers.py, not the reference's arithmetic.)"""
    X, Y, w = (torch.as_tensor(np.asarray(a), dtype=torch.float32).to(dtype) for a in (X, Y, w))
    w = w.reshape(-1, 1)
    wn = w / (w.abs().sum() + eps)
    mx = (wn * X).sum(0, keepdim=True)
    my = (wn * Y).sum(0, keepdim=True)
    S = ((Y - my).t() @ (wn * (X - mx))).double()          # 3x3, SVD in f64 on the host
    U, _, V = torch.svd(S)
    sgn = torch.eye(3, dtype=torch.float64)
    if torch.det(U) * torch.det(V) < 0:
        sgn[2, 2] = -1
    R = (U @ sgn @ V.t()).to(dtype)
    t = (my.squeeze() - (R @ mx.t()).squeeze()).to(dtype)
    return R, t


def rot6d_to_matrix(p6):
    """[B,6] -> [B,3,3]: Gram-Schmidt on the two 3-vectors, clamps at 1e-8."""
    a, b = p6[:, 0:3], p6[:, 3:6]
    x = a / torch.clamp(torch.sqrt((a ** 2).sum(1, keepdim=True)), min=1e-8)
    coef = (x * b).sum(1, keepdim=True) / torch.clamp((x ** 2).sum(1, keepdim=True), min=1e-8)
    u = b - coef * x
    y = u / torch.clamp(torch.sqrt((u ** 2).sum(1, keepdim=True)), min=1e-8)
    z = torch.stack((x[:, 1] * y[:, 2] - x[:, 2] * y[:, 1],
                     x[:, 2] * y[:, 0] - x[:, 0] * y[:, 2],
                     x[:, 0] * y[:, 1] - x[:, 1] * y[:, 0]), dim=1)
    return torch.stack((x, y, z), dim=2)


def smooth_l1_highdim(P, Q, w, wsum, q, eps=F32_EPS):
    s = (((P - Q) / q) ** 2).sum(1, keepdim=True)
    half = 0.5 * (s < 1).to(s.dtype)
    per = (0.5 - half) * (torch.sqrt(s + eps) - 0.5) + half * s
    if w is None:
        return per.mean()
    return (per * w).sum() / wsum


def global_registration(X, Y, w, max_iter=1000, max_break_count=20,
                        break_threshold_ratio=1e-5, quantization_size=1.0, trace=None, dtype=torch.float32,
                        states=None, start=None):
    """Returns R [3,3], t [1,3] (float32 numpy; float64 when `dtype` is) and the stats dict
    {'iterations','loss','break_count'} of the reference.  `trace` (a list) receives (R, t) after
    every optimiser step (test instrumentation; not part of the reference).

    Test instrumentation beyond the reference (tests/helpers.py, the f64 arbiter):
    * `dtype=torch.float64`: the same algorithm -- same formulas, same Adam, same stopping logic -- evaluated in double
      on the same f32 inputs.  NOT the reference's arithmetic: it is the exact-arithmetic yardstick that says which of
      two f32 implementations is nearer the truth.
    * `states` (a list) receives, BEFORE every optimiser step i, the dict {'i', 'prm' [9], 'm' [9], 'v' [9],
      'loss_prev', 'breaks'} -- everything the next step depends on.
    * `start` (such a dict): the loop resumes at iteration start['i'] from that state instead of the weighted-Procrustes
      estimate (the values are cast to `dtype`), and `max_iter` is the iteration index to stop before."""
    X, Y, w = (torch.as_tensor(np.asarray(a), dtype=torch.float32).to(dtype) for a in (X, Y, w))
    w = w.reshape(-1, 1)
    wsum = w.sum()
    if start is None:
        R0, t0 = weighted_procrustes(X, Y, w, F32_EPS, dtype=dtype)
        rot6d = torch.nn.Parameter(torch.cat((R0[:, 0], R0[:, 1])).reshape(1, 6).clone())
        trans = torch.nn.Parameter(t0.reshape(1, 3).clone())
    else:
        prm = torch.as_tensor(np.asarray(start['prm'], np.float64)).to(dtype)
        rot6d = torch.nn.Parameter(prm[:6].reshape(1, 6).clone())
        trans = torch.nn.Parameter(prm[6:].reshape(1, 3).clone())

    def apply(P):
        return P @ rot6d_to_matrix(rot6d)[0].t() + trans

    opt = torch.optim.Adam([rot6d, trans], lr=1e-1)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.999)
    i0 = 0
    if start is not None:
        i0 = int(start['i'])
        m = torch.as_tensor(np.asarray(start['m'], np.float64)).to(dtype)
        v = torch.as_tensor(np.asarray(start['v'], np.float64)).to(dtype)
        if i0 > 0:
            for p_, sl in ((rot6d, slice(0, 6)), (trans, slice(6, 9))):
                opt.state[p_] = {'step': torch.tensor(float(i0)), 'exp_avg': m[sl].reshape(p_.shape).clone(),
                                 'exp_avg_sq': v[sl].reshape(p_.shape).clone()}
            with warnings.catch_warnings():   # (scheduler stepped before the first optimiser step, on purpose)
                warnings.simplefilter('ignore')
                for _ in range(i0):    # ExponentialLR after i0 steps: the scheduler's own recurrence, lr *= gamma
                    sched.step()
        loss_prev = float(start['loss_prev'])
        breaks = int(start['breaks'])
    else:
        loss_prev = smooth_l1_highdim(apply(X), Y, w, wsum, quantization_size).item()
        breaks = 0
    i = i0
    loss = None
    for i in range(i0, max_iter):
        loss = smooth_l1_highdim(apply(X), Y, w, wsum, quantization_size)
        if loss.item() < 1e-7:
            break
        if states is not None:
            st = [opt.state[p_] if p_ in opt.state and len(opt.state[p_]) else None for p_ in (rot6d, trans)]
            states.append({'i': i, 'prm': torch.cat((rot6d.detach().reshape(-1), trans.detach().reshape(-1))).double().numpy().copy(),
                           'm': np.concatenate([(s_['exp_avg'].reshape(-1).double().numpy() if s_ else np.zeros(n_))
                                                for s_, n_ in zip(st, (6, 3))]),
                           'v': np.concatenate([(s_['exp_avg_sq'].reshape(-1).double().numpy() if s_ else np.zeros(n_))
                                                for s_, n_ in zip(st, (6, 3))]),
                           'loss_prev': loss_prev, 'breaks': breaks})
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if trace is not None:
            with torch.no_grad():
                trace.append((rot6d_to_matrix(rot6d)[0].numpy().copy(), trans.detach().numpy().reshape(3).copy()))
        if abs(loss_prev - loss.item()) < loss_prev * break_threshold_ratio:
            breaks += 1
            if breaks >= max_break_count:
                break
        loss_prev = loss.item()
    R = rot6d_to_matrix(rot6d.detach())[0]
    return (R.numpy(), trans.detach().numpy(),
            {'iterations': i, 'loss': float(loss.item()), 'break_count': breaks,
             'prm': torch.cat((rot6d.detach().reshape(-1), trans.detach().reshape(-1))).double().numpy().copy()})
