"""`python -m deepglobalregistration_amd.eval --threed_match_dir <root> --weights <ckpt>`: the 3DMatch
trajectory evaluation of scripts/test_3dmatch.py (success = RTE < 0.3 m and RRE < 15 deg by default), or
`--kitti_dir <root>/dataset --drives 8 9 10` for scripts/test_kitti.py (RTE < 0.6 m, RRE < 5 deg, ground truth
refined by GPU ICP like the reference's cached poses), on one MI355X.  Needs a real checkpoint and the
benchmark files; neither is available offline."""
import argparse

import numpy as np
import torch

from . import KITTIOdometryPairs, ThreeDMatchTrajectory, evaluate, evaluate_kitti


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threed_match_dir')
    ap.add_argument('--kitti_dir', help='.../dataset with sequences/ and poses/')
    ap.add_argument('--drives', nargs='*', type=int, default=[8, 9, 10])
    ap.add_argument('--no_gt_icp', action='store_true', help='KITTI: raw odometry ground truth')
    ap.add_argument('--weights', required=True)
    ap.add_argument('--scenes', nargs='*', default=None)
    ap.add_argument('--success_rte_thresh', type=float, default=0.3)
    ap.add_argument('--success_rre_thresh', type=float, default=15.0)
    ap.add_argument('--no_icp', action='store_true')
    ap.add_argument('--out', default='3dmatch-stats_DeepGlobalRegistration.npz')
    args = ap.parse_args()
    from ..core.deep_global_registration import DeepGlobalRegistration
    dgr = DeepGlobalRegistration({'weights': args.weights, 'use_icp': not args.no_icp}, torch.device('cuda'))
    if args.kitti_dir:
        from .. import ops

        def refine(xyz0, xyz1, M):   # kitti_loader.py:139-158
            # the reference selects the 5 cm subsample of cloud 0 BEFORE moving it (sparse_quantize(xyz0 / 0.05) on the
            # untransformed scan, :142), applies M to the selected points (:147) and runs ICP from the identity with
            # max distance 0.2 m and 200 iterations (:150-152)
            s5, _, _ = ops.voxelize(xyz0, 0.05)
            d5, _, _ = ops.voxelize(xyz1, 0.05)
            Mt = torch.as_tensor(M, dtype=torch.float32, device=s5.device)
            s5 = s5 @ Mt[:3, :3].T + Mt[:3, 3]
            return ops.icp_point_to_point(s5, d5, 0.2, init=np.eye(4), max_iter=200)[0]
        ds = KITTIOdometryPairs(args.kitti_dir, args.drives, icp_refine=None if args.no_gt_icp else refine)
        stats, _ = evaluate_kitti(dgr, ds)
        np.savez('kitti-stats_DeepGlobalRegistration.npz' if args.out.startswith('3dmatch') else args.out, stats=stats)
        return
    if not args.threed_match_dir:
        ap.error('--threed_match_dir or --kitti_dir is required')
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
