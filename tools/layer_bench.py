#!/usr/bin/env python
"""Per-layer timing of the sparse conv on the benchmark workload (one 50k-pt pair): for each conv of
the 6-D inlier net and of the 3-D FCGF net prints pairs, shape, GFLOP, MFMA-phase and reduce-phase
time, TFLOP/s and the implied GB/s of the gather + product-row traffic.  Kernel-tuning instrument."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepglobalregistration_amd import ops, synth
from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration

layers = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else None
ck = synth.synth_checkpoint(0)
dgr = DeepGlobalRegistration({'weights': ck}, torch.device('cuda'))
x0, x1, T = synth.synth_pair(0, 50000)
p0, c0, f0 = dgr.preprocess(x0); p1, c1, f1 = dgr.preprocess(x1)
F0 = dgr.fcgf_feature_extraction(f0, c0); F1 = dgr.fcgf_feature_extraction(f1, c1)
_, idx1 = dgr.fcgf_feature_matching(F0, F1)
gt = torch.from_numpy(synth.gt_correspondences(p0.cpu().numpy(), p1.cpu().numpy(), T, 0.05)).cuda()
# workload independent of the feature values (so that timing ablations of the conv kernel, which
# produce garbage features, still see the same kernel maps): non-GT rows get a fixed pseudo-random match
fallback = (torch.arange(len(gt), device=gt.device) * 7919) % len(p1)
idx1 = torch.where(gt >= 0, gt, fallback)
c6, f6 = ops.inlier_inputs(c0, p0, c1, p1, idx1, 'coords')
for name, net, args in (('6-D', dgr.inlier_model._handle(), (c6, f6)), ('3-D', dgr.fcgf_model._handle(), (c0, f0))):
    net.forward(*args)
    st = net.layer_stats()
    tot_g = tot_r = 0.0
    print(f'== {name} net: layer pairs Kne n_in n_out cin cout | GFLOP gemm_us reduce_us TF/s  gatherGB/s yGB/s')
    for li, s in enumerate(st):
        if layers is not None and li not in layers:
            continue
        try:
            g, r = net.rerun_layer(li, 5)
        except ValueError:   # conv1 fused with its neighbour search: not re-runnable in isolation
            print(f"{li:2d} {s['pairs']:8d} (fused with the neighbour search)")
            continue
        fl = 2.0 * s['pairs'] * s['cin'] * s['cout']
        gb = 4.0 * s['pairs'] * s['cin']
        yb = 4.0 * s['pairs'] * s['cout']
        tot_g += g; tot_r += r
        print(f"{li:2d} {s['pairs']:8d} {s['nonempty']:4d} {s['n_in']:6d} {s['n_out']:6d} {s['cin']:4d} {s['cout']:4d} | "
              f"{fl / 1e9:7.2f} {g * 1e3:8.1f} {r * 1e3:8.1f} {fl / (g * 1e-3) / 1e12:6.1f} {gb / (g * 1e-3) / 1e9:8.0f} {yb / (g * 1e-3) / 1e9:8.0f}")
    print(f'   total gemm {tot_g:.3f} ms, reduce {tot_r:.3f} ms')
