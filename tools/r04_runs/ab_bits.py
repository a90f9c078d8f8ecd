"""FCGF forward of one two-cloud batch, features written as raw bytes (A/B of library builds through DGR_HIP_LIB)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepglobalregistration_amd import ops, synth

a, b, _ = synth.synth_pair(5, n_raw=30000)
parts = [ops.voxelize(x, 0.05, i)[1] for i, x in enumerate((a, b))]
c = torch.cat(parts)
net = ops.NetHandle(synth.synth_state_dict(3, 1, 32, 7, 0), 3, 1, 32, 7, True)
F = net.forward(c, torch.ones(len(c), 1, device='cuda')).cpu().numpy()
F.tofile(sys.argv[1])
print(len(c), float(np.abs(F).max()))
