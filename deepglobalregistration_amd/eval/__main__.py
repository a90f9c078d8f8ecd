"""`python -m deepglobalregistration_amd.eval --threed_match_dir <root> --weights <ckpt>`: the 3DMatch
trajectory evaluation of scripts/test_3dmatch.py (success = RTE < 0.3 m and RRE < 15 deg by default) on
one MI355X.  Needs a real checkpoint and the benchmark files; neither is available offline."""
import argparse

import numpy as np
import torch

from . import ThreeDMatchTrajectory, evaluate


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threed_match_dir', required=True)
    ap.add_argument('--weights', required=True)
    ap.add_argument('--scenes', nargs='*', default=None)
    ap.add_argument('--success_rte_thresh', type=float, default=0.3)
    ap.add_argument('--success_rre_thresh', type=float, default=15.0)
    ap.add_argument('--no_icp', action='store_true')
    ap.add_argument('--out', default='3dmatch-stats_DeepGlobalRegistration.npz')
    args = ap.parse_args()
    from ..core.deep_global_registration import DeepGlobalRegistration
    dgr = DeepGlobalRegistration({'weights': args.weights, 'use_icp': not args.no_icp}, torch.device('cuda'))
    ds = ThreeDMatchTrajectory(args.threed_match_dir, args.scenes)
    stats, scene_means, _ = evaluate([dgr], ['DGR'], ds, args.success_rte_thresh, args.success_rre_thresh,
                                     summary_every=10)
    np.savez(args.out, stats=stats, names=['DGR'], scenes=ds.scenes, scene_means=scene_means)
    print('scene-wise mean [success, RTE, RRE]:')
    for s, v in zip(ds.scenes, scene_means[0]):
        print(f'  {s}: {v}')
    print('scene average:', scene_means[0].mean(0))


if __name__ == '__main__':
    main()
