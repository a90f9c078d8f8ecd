"""Helper of tests/test_gpu_numeric_modes.py: forward of a 3-D and a 6-D ResUNet on fixed seeded inputs, outputs to
an .npz.  The arithmetic mode of the conv kernels is fixed when the library builds a net (environment variables),
so every mode needs its own process."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import random_cloud_coords   # noqa: E402
from deepglobalregistration_amd import ops, synth   # noqa: E402


def main(out):
    rng = np.random.default_rng(7)
    # 6-D inlier net (wide layers: rule-major split-operand kernel)
    c0 = random_cloud_coords(rng, 1600, 12, 3)
    c1 = c0[:, 1:] + rng.integers(-2, 3, (len(c0), 3)).astype(np.int32)
    coords6 = np.concatenate([c0, c1], axis=1).astype(np.int32)
    feats6 = np.cos(rng.uniform(-3, 3, (len(coords6), 6))).astype(np.float32)
    net6 = ops.NetHandle(synth.synth_state_dict(6, 6, 1, 3, 11), 6, 6, 1, 3, False)
    logit = net6.forward(torch.from_numpy(coords6).cuda(), torch.from_numpy(feats6).cuda()).cpu().numpy()
    inter6 = {'i6_' + n: net6.intermediate(n) for n in ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr')}
    kinds6 = list(dict.fromkeys(_kinds(net6, coords6, feats6)))
    # 3-D FCGF net (output-stationary kernel), features of very different magnitude per row
    c3 = random_cloud_coords(rng, 3000, 24, 3)
    feats3 = np.ones((len(c3), 1), np.float32)
    net3 = ops.NetHandle(synth.synth_state_dict(3, 1, 32, 7, 0), 3, 1, 32, 7, True)
    F = net3.forward(torch.from_numpy(c3).cuda(), torch.from_numpy(feats3).cuda()).cpu().numpy()
    kinds3 = list(dict.fromkeys(_kinds(net3, c3, feats3)))
    np.savez(out, logit=logit, F=F, coords6=coords6, feats6=feats6, c3=c3, kinds=np.array(kinds6 + kinds3), **inter6)


def _kinds(net, coords, feats):
    ops.set_profiling('cuda', True)
    net.forward(torch.from_numpy(coords).cuda(), torch.from_numpy(feats).cuda())
    k = ops.conv_launch_kinds('cuda')
    ops.set_profiling('cuda', False)
    return k


if __name__ == '__main__':
    main(sys.argv[1])
