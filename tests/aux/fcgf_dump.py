"""Helper of tests/test_gpu_dense_conv.py: forward of the 3-D FCGF net on a seeded two-cloud batch (a full-size
3DMatch-shaped cloud + a small one: row counts that are no multiple of any block size), output features, the
intermediate tensors and the kernels that ran to an .npz.  Which kernel serves the same-stride C <= 64 layers is
fixed by the environment when the process starts (DGR_OS_LISTS), so every variant needs its own process."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepglobalregistration_amd import ops, synth   # noqa: E402


def main(out):
    a, b, _ = synth.synth_pair(3, n_raw=50000)
    _, ca, _ = ops.voxelize(a, 0.05, 0)
    _, cb, _ = ops.voxelize(b[:6000], 0.05, 1)
    coords = torch.cat([ca, cb])
    feats = torch.ones((len(coords), 1), dtype=torch.float32, device=coords.device)
    res = {'n': np.array([len(ca), len(cb)])}
    for name, ks, seed in (('k7', 7, 0), ('k3', 3, 5)):
        net = ops.NetHandle(synth.synth_state_dict(3, 1, 32, ks, seed), 3, 1, 32, ks, True)
        ops.set_profiling('cuda', True)
        F = net.forward(coords, feats)
        res[name + '_kinds'] = np.array(ops.conv_launch_kinds('cuda'))
        ops.set_profiling('cuda', False)
        res[name + '_F'] = F.cpu().numpy()
        for n in ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr'):
            res[f'{name}_{n}'] = net.intermediate(n)
    np.savez(out, **res)


if __name__ == '__main__':
    main(sys.argv[1])
