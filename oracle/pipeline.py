"""CPU restatement of `DeepGlobalRegistration.register()`: raw points -> 4x4 float64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows `core/deep_global_registration.py:134-217,238-324`
stage by stage, including the two Open3D steps: the safeguard RANSAC over the putative correspondences
(`:50-64, 302-315`) and the final point-to-point ICP (`:317-322`), both through `oracle/open3d_reg.py`
(restated Open3D 0.17 algorithms -- their ARITHMETIC is unpinned against Open3D, see that module).  The whole function is
pinned by runs of the reference's own `register()` (tests/golden/make_golden_register.py): the learned branch without
ICP (register_e2e.npz: intermediates exact / 1.5e-6, T 3e-7, bitwise given the same logits), and -- through the
functional Open3D stand-in that records what the reference's lines pass -- learned + ICP and gate-fails -> safeguard
RANSAC -> ICP (register_e2e_o3d.npz; tests/test_oracle_register_golden.py: the recorded arguments are the ones
restated below, T bitwise given the same matches / logits).
"""
import numpy as np
import torch

from . import me_semantics as me
from . import resunet, knn, registration, open3d_reg


def preprocess(xyz, voxel_size):
    """`preprocess`, deep_global_registration.py:134-161.
    xyz [M,3] (float64 for Open3D clouds) -> xyz [N,3] f32, coords [N,4] i32, feats [N,1] f32."""
    xyz = np.asarray(xyz)
    _, sel = me.sparse_quantize(xyz / voxel_size, return_index=True)
    xyz_sel = xyz[sel]
    coords = me.batched_coordinates([np.floor(xyz_sel / voxel_size).astype(np.int32)])
    feats = np.ones((len(sel), 1), np.float32)
    return xyz_sel.astype(np.float32), coords, feats


def inlier_inputs(xyz0, xyz1, coords0, coords1, idx0, idx1, feature_type='coords'):
    """6-D coordinates (`:261-262`) and input features (`:185-208`)."""
    idx0 = np.asarray(idx0).reshape(-1)
    idx1 = np.asarray(idx1).reshape(-1)
    coords6 = np.concatenate((coords0[idx0], coords1[idx1, 1:]), axis=1).astype(np.int32)
    if feature_type == 'ones':
        feats = np.ones((len(idx0), 1), np.float32)
    elif feature_type == 'coords':
        x0 = torch.from_numpy(np.ascontiguousarray(xyz0[idx0]))
        x1 = torch.from_numpy(np.ascontiguousarray(xyz1[idx1]))
        feats = torch.cat((torch.cos(x0), torch.cos(x1)), dim=1).numpy()
    else:
        raise TypeError('Undefined feature type')
    return coords6, feats


def confidence_gate(logit, clip_weight_thresh=0.05):
    """sigmoid / clip / sum, `:269-276`.  Returns weights [N,1] f32, wsum, threshold."""
    w = torch.sigmoid(torch.from_numpy(np.asarray(logit, np.float32).reshape(-1, 1)))
    if clip_weight_thresh > 0:
        w[w < clip_weight_thresh] = 0
    wsum = w.sum().item()
    return w.numpy(), wsum, max(200, len(w) * 0.05)


def register(ckpt, xyz0, xyz1, clip_weight_thresh=0.05, forced_logit_fn=None, idx1_fn=None, use_icp=True,
             safeguard=True, ransac_hypotheses=4000000, ransac_seed=0):
    """`register()` (`:238-324`) restated.  `ckpt` = {'config', 'state_dict', 'state_dict_inlier'} in the
    reference's checkpoint layout (core/trainer.py:527-549).  Returns a dict with every intermediate so that tests can
    compare stage by stage; `T` is the 4x4 float64 the reference returns.

    Harness hooks (None = the reference's behaviour): `idx1_fn(p0, p1, F0, F1, idx1) -> idx1` replaces matches after
    the search ran, `forced_logit_fn(x0[idx0], x1[idx1], logit) -> logit` replaces the logits after the inlier net ran
    (untrained weights give meaningless matches / confidences).  `ransac_hypotheses`: the reference hard-codes
    4 000 000 (`:61`); tests pass fewer, to both sides."""
    cfg = ckpt['config']
    voxel = cfg['voxel_size']
    out = {}
    p0, c0, f0 = preprocess(xyz0, voxel)
    p1, c1, f1 = preprocess(xyz1, voxel)
    out.update(xyz0=p0, xyz1=p1, coords0=c0, coords1=c1)
    ks = cfg.get('feat_conv1_kernel_size', cfg.get('conv1_kernel_size'))
    n_out = cfg.get('feat_model_n_out', cfg.get('model_n_out'))
    F0 = resunet.resunet_forward(ckpt['state_dict'], c0, f0, 3, ks, cfg['normalize_feature'])
    F1 = resunet.resunet_forward(ckpt['state_dict'], c1, f1, 3, ks, cfg['normalize_feature'])
    assert F0.shape[1] == n_out
    out.update(F0=F0, F1=F1)
    idx1 = knn.find_knn(F0, F1, nn_max_n=cfg.get('nn_max_n', 250)).reshape(-1)
    out['idx1_searched'] = idx1
    if idx1_fn is not None:
        idx1 = np.asarray(idx1_fn(p0, p1, F0, F1, idx1)).reshape(-1)
    idx0 = np.arange(len(idx1))
    out.update(idx0=idx0, idx1=idx1)
    ftype = cfg.get('inlier_feature_type', 'coords')
    coords6, feats6 = inlier_inputs(p0, p1, c0, c1, idx0, idx1, ftype)
    out.update(coords6=coords6, feats6=feats6)
    logit = resunet.resunet_forward(ckpt['state_dict_inlier'], coords6, feats6, 6,
                                    cfg['inlier_conv1_kernel_size'], False)
    out['logit_net'] = logit
    if forced_logit_fn is not None:
        logit = forced_logit_fn(p0[idx0], p1[idx1], logit)
    w, wsum, thr = confidence_gate(logit, clip_weight_thresh)
    out.update(logit=logit, weights=w, wsum=wsum, wsum_threshold=thr)
    T = np.identity(4)
    if wsum >= thr:                                     # :283-300
        R, t, stats = registration.global_registration(
            p0[idx0], p1[idx1], w, break_threshold_ratio=1e-4,
            quantization_size=2 * voxel)
        T[:3, :3] = R
        T[:3, 3] = t.reshape(3)
        out.update(R=R, t=t, stats=stats, confident=True, status='ok')
    else:                                               # :302-315
        out.update(confident=False, status='low_confidence')
        if safeguard:
            T, h, count, rmse = open3d_reg.ransac_correspondence(p0[idx0], p1[idx1], 2 * voxel, ransac_hypotheses,
                                                                 seed=ransac_seed)
            out.update(status='safeguard', ransac={'hypothesis': h, 'inliers': count, 'rmse': rmse})
    out['T_before_icp'] = T.copy()
    if use_icp:                                         # :317-322
        T, fitness, rmse, iters = open3d_reg.icp_point_to_point(p0, p1, 2 * voxel, init=T)
        out['icp'] = {'fitness': fitness, 'inlier_rmse': rmse, 'iterations': iters}
    out['T'] = T
    return out
