"""Seeded inputs shared by tests/test_gpu_split_f64.py (f64 oracle side) and tests/aux/split_f64_dump.py (device side)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import random_cloud_coords   # noqa: E402
from deepglobalregistration_amd import synth   # noqa: E402
from oracle import pipeline as opipe   # noqa: E402


def fullsize_case():
    """BASELINE configs[1]: one 50k-point pair at 5 cm; fragment 0 through the FCGF net, the 6-D net on the pair's
    correspondences (20 % ground-truth matches, the rest a fixed pseudo-random match: independent of the features)."""
    a, b, T = synth.synth_pair(0, n_raw=50000)
    ck = synth.synth_checkpoint(seed=0, voxel_size=0.05, feat_conv1_kernel_size=7)
    x0, c0, _ = opipe.preprocess(a, 0.05)
    x1, c1, _ = opipe.preprocess(b, 0.05)
    g = synth.gt_correspondences(x0, x1, T, 0.05, seed=0)
    idx1 = np.where(g >= 0, g, (np.arange(len(x0)) * 7919) % len(x1))
    c6, f6 = opipe.inlier_inputs(x0, x1, c0, c1, np.arange(len(x0)), idx1)
    return {'sd3': ck['state_dict'], 'sd6': ck['state_dict_inlier'], 'c0': c0, 'c6': np.asarray(c6, np.int32),
            'f6': np.asarray(f6, np.float32)}


def _spread_bn(sd, rng, layers):
    """batch-norm gammas of `layers` drawn log-uniformly from 1e-4 .. 1e4 per channel (folded into the kernels at load
    time: one per-layer weight scale has to serve all of them), the following norm's gamma compensating the level"""
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    for name in layers:
        g = sd[name + '.bn.weight']
        sd[name + '.bn.weight'] = (np.sign(g) * 10.0 ** rng.uniform(-4, 4, g.shape)).astype(np.float32)
    return sd


def wide_range_case():
    rng = np.random.default_rng(5)
    c0 = random_cloud_coords(rng, 1600, 12, 3)
    c1 = c0[:, 1:] + rng.integers(-2, 3, (len(c0), 3)).astype(np.int32)
    c6 = np.concatenate([c0, c1], axis=1).astype(np.int32)
    f6 = np.cos(rng.uniform(-3, 3, (len(c6), 6))).astype(np.float32)
    sd6 = _spread_bn(synth.synth_state_dict(6, 6, 1, 3, 21), rng, ['norm3', 'block3.norm1', 'norm4', 'block4.norm2', 'norm4_tr'])
    c3 = random_cloud_coords(rng, 3000, 24, 3)
    sd3 = _spread_bn(synth.synth_state_dict(3, 1, 32, 5, 22), rng, ['block1.norm1', 'norm2', 'block2.norm2', 'norm3', 'block4.norm1'])
    return {'sd6': sd6, 'c6': c6, 'f6': f6, 'sd3': sd3, 'c3': c3}


def adversarial_rows(rng, n, c):
    x = rng.uniform(-2, 2, (n, c)).astype(np.float32)
    x *= (10.0 ** ((np.arange(n) % 13) - 6.0))[:, None].astype(np.float32)          # row magnitudes 1e-6 .. 1e6
    x[:, 3::7] *= np.float32(1e-5)                                                 # small channels inside every row
    x[5] = 0                                                                       # a row of zeros
    x[6] = np.where(np.arange(c) % 2 == 1, np.float32(1e-41), np.float32(-3e-42))  # a denormal row
    mask = np.arange(c) != 17
    x[7, mask] *= np.float32(2.0 ** -20)                                           # one channel 2^20 above the rest
    x[8, mask] = 0                                                                 # ... and alone in its row
    x[9] = np.float32(3e38) * np.sign(x[9])                                        # close to the f32 limit
    return x


def adversarial_case():
    rng = np.random.default_rng(9)
    c0 = random_cloud_coords(rng, 1200, 10, 3)
    c1 = c0[:, 1:] + rng.integers(-1, 2, (len(c0), 3)).astype(np.int32)
    c6 = np.concatenate([c0, c1], axis=1).astype(np.int32)
    c3 = random_cloud_coords(rng, 2500, 20, 3)
    return {'sd6': synth.synth_state_dict(6, 6, 1, 3, 31), 'sd3': synth.synth_state_dict(3, 1, 32, 3, 32),
            'layer6': 10, 'layer3': 4,   # block4.conv1 (256 -> 256), block2.conv1 (64 -> 64)
            'c6': c6, 'x6': adversarial_rows(rng, len(c6), 256), 'c3': c3, 'x3': adversarial_rows(rng, len(c3), 64)}
