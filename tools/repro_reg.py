#!/usr/bin/env python
"""Reproducibility stress of the registration kernel alone (dgr_se3_refine on fixed inputs), N runs."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepglobalregistration_amd import ops, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
a, b, Tg = synth.synth_pair(0, n_raw=20000)
xa, _, _ = ops.voxelize(a, 0.05); xb, _, _ = ops.voxelize(b, 0.05)
g = synth.gt_correspondences(xa.cpu().numpy(), xb.cpu().numpy(), Tg, 0.05, seed=0)
idx = np.where(g >= 0, g, (np.arange(len(g)) * 7919) % len(xb))
X, Y = xa, xb[torch.from_numpy(idx).cuda()]
w = torch.from_numpy(np.where(g >= 0, 0.98, 0.02).astype(np.float32)).cuda()
w[w < 0.05] = 0
ref, bad = None, 0
for it in range(N):
    R, t, st = ops.se3_refine(X, Y, w, 0.1, 1000, 20, 1e-4)
    cur = (R.tobytes(), t.tobytes(), st['iterations'], st['loss'])
    if ref is None: ref = cur
    elif cur != ref: bad += 1
print('registration kernel alone:', N, 'runs,', bad, 'differ from the first; iterations', ref[2])
