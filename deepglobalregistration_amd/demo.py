"""`python -m deepglobalregistration_amd.demo --pcd0 a.ply --pcd1 b.ply --weights ckpt.pth`: the reference's
demo.py (:29-49) without the download and the Open3D viewer -- load two fragments (.ply / KITTI .bin / .npy /
.npz / .txt), register them on the GPU and print the 4x4 transformation (optionally save the aligned cloud).
`--synthetic` runs on a seeded synthetic pair with synthetic weights instead (no files needed)."""
import argparse

import numpy as np
import torch

from .eval.formats import load_cloud, write_ply


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pcd0')
    ap.add_argument('--pcd1')
    ap.add_argument('--weights')
    ap.add_argument('--no_icp', action='store_true')
    ap.add_argument('--out', help='write fragment 0 transformed into the frame of fragment 1 (.ply)')
    ap.add_argument('--synthetic', action='store_true')
    args = ap.parse_args()
    from .core.deep_global_registration import DeepGlobalRegistration
    if args.synthetic:
        from . import synth
        xyz0, xyz1, T_gt = synth.synth_pair(0, 50000)
        weights = synth.synth_checkpoint(0)
    else:
        if not (args.pcd0 and args.pcd1 and args.weights):
            ap.error('--pcd0, --pcd1 and --weights are required (or --synthetic)')
        xyz0, xyz1, T_gt = load_cloud(args.pcd0), load_cloud(args.pcd1), None
        weights = args.weights
    dgr = DeepGlobalRegistration({'weights': weights, 'use_icp': not args.no_icp}, torch.device('cuda'))
    T01 = dgr.register(np.asarray(xyz0, np.float64), np.asarray(xyz1, np.float64))
    np.set_printoptions(precision=6, suppress=True)
    print(T01)
    print(f'status: {dgr.last_status}; registration {dgr.reg_timer.diff * 1e3:.1f} ms '
          f'(features {dgr.feat_timer.diff * 1e3:.1f} ms)')
    if T_gt is not None:
        from .eval.metrics import rte_rre
        ok, rte, rre = rte_rre(T01, T_gt, 0.3, 15)
        print(f'synthetic ground truth: RTE {rte:.4f} m, RRE {rre:.3f} deg (untrained weights: expect the safeguard)')
    if args.out:
        write_ply(args.out, np.asarray(xyz0, np.float64) @ T01[:3, :3].T + T01[:3, 3])


if __name__ == '__main__':
    main()
