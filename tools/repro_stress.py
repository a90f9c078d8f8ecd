#!/usr/bin/env python
"""Run-to-run reproducibility stress: the same batch through dgr_register_batch N times (optionally next to a second
copy of this script on the same GPU for timing jitter); every stage output is compared bit for bit with the first run.
    python tools/repro_stress.py [N] [n_raw]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepglobalregistration_amd import ops, synth
from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n_raw = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
dev = torch.device('cuda')
ck = synth.synth_checkpoint(seed=0, voxel_size=0.05, feat_conv1_kernel_size=5)
dgr = DeepGlobalRegistration({'weights': ck}, dev)
x0, c0, x1, c1, off0, off1, ovr = [], [], [], [], [0], [0], []
for q, seed in enumerate((0, 1, 2)):
    a, b, Tg = synth.synth_pair(seed, n_raw=n_raw)
    xa, ca, _ = ops.voxelize(a, 0.05, 0, dev); xb, cb, _ = ops.voxelize(b, 0.05, 0, dev)
    ca = ca.clone(); cb = cb.clone(); ca[:, 0] = q; cb[:, 0] = q
    g = synth.gt_correspondences(xa.cpu().numpy(), xb.cpu().numpy(), Tg, 0.05, seed=seed)
    ovr.append(np.where(g >= 0, g + off1[-1], -1))
    x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
    off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
C0, X0, C1, X1 = torch.cat(c0), torch.cat(x0), torch.cat(c1), torch.cat(x1)
ovr = torch.from_numpy(np.concatenate(ovr)).to(dev)
ref, bad = None, {}
for it in range(N):
    T, status, stats = dgr.register_voxelized(C0, X0, off0, C1, X1, off1, override_idx1=ovr)
    cur = {k: ops.batch_output(dev, k).cpu().numpy() for k in ('idx1', 'logit', 'F0', 'F1', 'weights')}
    cur['T'] = T; cur['stats'] = stats
    if ref is None:
        ref = cur
        continue
    for k in cur:
        if not np.array_equal(cur[k], ref[k]):
            bad.setdefault(k, []).append(it)
            if k == 'stats' and len(bad[k]) <= 3:
                print('run', it, 'stats', cur[k].tolist(), 'ref', ref[k].tolist(), 'status', status.tolist(), flush=True)
print('runs', N, 'voxels', off0[-1], off1[-1], 'mismatching runs per output:', {k: (len(v), v[:5]) for k, v in bad.items()} or 'none')
