"""Which MinkowskiEngine conventions was a checkpoint written under?  (deepglobalregistration_amd/model/me_conventions.py)

    python tools/check_me_conventions.py --weights ckpt.pth --pcd0 a.ply --pcd1 b.ply
    python tools/check_me_conventions.py --synthetic          # dry run of the tool itself: no convention can stand out

Registers one real pair under the four combinations of {kernel offsets: first / last axis fastest} x {transposed
convolutions: same / mirrored kernel index} and prints, for each, what a trained network makes unmistakable: the number
of matches that are mutually consistent with the estimated pose, the summed confidence against the gate, and the share
of confident weights.  Under a wrong reading the FCGF features are noise (few geometrically consistent matches) and
the inlier network has nothing to be confident about; under the right one an overlapping pair passes the gate with a
large margin.  The default (first axis fastest, same index) is the library's reading of ME 0.5.4; the two keys go into
the runtime config (`me_kernel_order`, `me_transposed_mirrored`) if another combination wins.
"""
import argparse
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--weights')
    ap.add_argument('--pcd0')
    ap.add_argument('--pcd1')
    ap.add_argument('--synthetic', action='store_true')
    args = ap.parse_args()
    from deepglobalregistration_amd import ops, synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    from deepglobalregistration_amd.eval.formats import load_cloud
    from deepglobalregistration_amd.model import me_conventions as mc
    if args.synthetic:
        xyz0, xyz1, _ = synth.synth_pair(0, 20000)
        weights = synth.synth_checkpoint(0)
    else:
        if not (args.weights and args.pcd0 and args.pcd1):
            ap.error('--weights, --pcd0 and --pcd1 are required (or --synthetic)')
        xyz0, xyz1, weights = load_cloud(args.pcd0), load_cloud(args.pcd1), args.weights
        weights = torch.load(weights, map_location='cpu', weights_only=False)     # read once, convert four times
    rows = []
    for order, mirrored in itertools.product(mc.KERNEL_ORDERS, (False, True)):
        dgr = DeepGlobalRegistration({'weights': weights, 'use_icp': False, 'keep_intermediates': True, 'me_kernel_order': order,
                                      'me_transposed_mirrored': mirrored}, torch.device('cuda'))
        T = dgr.register(np.asarray(xyz0, np.float64), np.asarray(xyz1, np.float64))
        wsum, thr = dgr.last_wsum
        w = torch.sigmoid(dgr.last_logit.reshape(-1))
        p0, _, _ = dgr.preprocess(np.asarray(xyz0, np.float64))
        p1, _, _ = dgr.preprocess(np.asarray(xyz1, np.float64))
        moved = p0.double() @ torch.from_numpy(T[:3, :3]).to(p0.device).T + torch.from_numpy(T[:3, 3]).to(p0.device)
        resid = (moved - ops.gather_rows3(p1, dgr.last_corres_idx1).double()).norm(dim=1)
        rows.append({'kernel_order': order, 'transposed_mirrored': mirrored, 'status': dgr.last_status,
                     'wsum': wsum, 'gate': thr, 'confident_share': float((w > 0.5).float().mean()),
                     'matches_within_2_voxels_of_T': int((resid < 2 * dgr.voxel_size).sum()), 'matches': len(resid)})
        del dgr
        torch.cuda.empty_cache()
    print(f"{'kernel offsets':22s} {'transposed':10s} {'status':10s} {'wsum / gate':>18s} {'w > 0.5':>8s} {'matches consistent with T':>26s}")
    for r in rows:
        print(f"{r['kernel_order']:22s} {'mirrored' if r['transposed_mirrored'] else 'same':10s} {r['status']:10s} "
              f"{r['wsum']:9.1f} / {r['gate']:6.1f} {r['confident_share']:8.3f} {r['matches_within_2_voxels_of_T']:12d} / {r['matches']}")
    best = max(rows, key=lambda r: (r['matches_within_2_voxels_of_T'], r['wsum']))
    print(f"\nmost consistent: me_kernel_order = {best['kernel_order']!r}, me_transposed_mirrored = {best['transposed_mirrored']}"
          + ('   (synthetic weights: no reading can stand out; this run only exercises the tool)' if args.synthetic else ''))


if __name__ == '__main__':
    main()
