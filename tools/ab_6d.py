"""A/B of the 6-D net: default (split-operand wide layers: two f16 pieces; DGR_CONV_BF3=1: three bf16 pieces) vs
DGR_CONV_F32=1; dumps intermediates."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from helpers import random_cloud_coords
from deepglobalregistration_amd import ops, synth
rng = np.random.default_rng(7)
c0 = random_cloud_coords(rng, 1600, 12, 3)
c1 = c0[:, 1:] + rng.integers(-2, 3, (len(c0), 3)).astype(np.int32)
coords = np.concatenate([c0, c1], axis=1).astype(np.int32)
feats = np.cos(rng.uniform(-3, 3, (len(coords), 6))).astype(np.float32)
sd = synth.synth_state_dict(6, 6, 1, 3, 11)
net = ops.NetHandle(sd, 6, 6, 1, 3, False)
out = net.forward(torch.from_numpy(coords).cuda(), torch.from_numpy(feats).cuda()).cpu().numpy()
tag = 'f32' if os.environ.get('DGR_CONV_F32') else ('bf3' if os.environ.get('DGR_CONV_BF3') else 'f16x2')
d = {n: net.intermediate(n) for n in ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr')}
d['out'] = out
np.savez(f'gpurun_out/ab6d_{tag}.npz', **d)
st = net.layer_stats()
print(tag, [(s['pairs'], s['cin'], s['cout']) for s in st[6:12]])
