"""`DeepGlobalRegistration` with the reference's constructor / `register()` / stage-method
surface (core/deep_global_registration.py:67-324), executed on one MI355X by libdgr_hip.so.

Scope: steps 0-5 of `register()` -- voxelisation, FCGF features, feature
matching, 6-D inlier network, confidence gate, weighted Procrustes + robust refinement -- plus the two
Open3D steps around it, re-implemented on the GPU: the safeguard RANSAC from the
putative correspondences (:50-64, 302-315) when the confidence gate fails (an SVD failure leaves T = identity,
exactly like the reference's `except RuntimeError` branch :295-300), and the final point-to-point ICP (:317-322) when `use_icp` is set (the reference's default, :75).  The
`fcgf_feature_matching` safeguard variant (:31-46) is not implemented.
"""
import os

import numpy as np
import torch

from .. import _lib, ops
from ..model import load_model
from ..sparse import SparseTensor
from ..util.timer import Timer
from .knn import find_knn_gpu
from .registration import GlobalRegistration


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def _cfg_has(cfg, key):
    return (key in cfg) if isinstance(cfg, dict) else hasattr(cfg, key)


class DeepGlobalRegistration:
    def __init__(self, config, device=torch.device('cuda')):
        self.config = config
        self.clip_weight_thresh = _cfg_get(config, 'clip_weight_thresh', 0.05)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('DeepGlobalRegistration (MI355X build) runs on the GPU only')
        _lib.load()                      # fail loudly here if the HIP extension is missing
        self.safeguard_method = 'correspondence'
        self.use_icp = bool(_cfg_get(config, 'use_icp', True))   # reference: self.use_icp = True (:75)
        self.ransac_seed = int(_cfg_get(config, 'ransac_seed', 0))
        # the reference hard-codes RANSACConvergenceCriteria(4000000, ...) (:61); a config key so that parity tests can
        # give both sides a count the CPU oracle finishes
        self.ransac_max_iteration = int(_cfg_get(config, 'ransac_max_iteration', 4000000))
        # optional runtime key: keep the correspondences and logits of the last register() call (`last_corres_idx1`,
        # `last_logit`: device tensors that stay alive until the next call) for inspection -- tools/check_me_conventions.py
        # and the parity tests read them; off by default
        self.keep_intermediates = bool(_cfg_get(config, 'keep_intermediates', False))
        self.last_corres_idx1 = None
        self.last_logit = None
        self.last_wsum = None
        self.feat_timer = Timer()
        self.reg_timer = Timer()
        self.last_status = None
        self.last_stats = None
        self.last_icp = None

        weights = _cfg_get(config, 'weights')
        if isinstance(weights, (str, os.PathLike)):
            assert os.path.exists(weights), weights
            state = torch.load(weights, map_location='cpu', weights_only=False)
        elif isinstance(weights, dict):
            state = weights              # in-memory checkpoint (synthetic weights, tests)
        else:
            raise ValueError('config.weights must be a checkpoint path or a checkpoint dict')
        network_config = state['config']
        self.network_config = network_config
        self.inlier_feature_type = _cfg_get(network_config, 'inlier_feature_type', 'coords')
        self.voxel_size = _cfg_get(network_config, 'voxel_size')

        # FCGF network (:94-116); legacy checkpoints use un-prefixed keys (:104-112)
        if _cfg_has(network_config, 'feat_model'):
            name, n_out, ks = (_cfg_get(network_config, 'feat_model'), _cfg_get(network_config, 'feat_model_n_out'),
                               _cfg_get(network_config, 'feat_conv1_kernel_size'))
        else:
            name, n_out, ks = (_cfg_get(network_config, 'model'), _cfg_get(network_config, 'model_n_out'),
                               _cfg_get(network_config, 'conv1_kernel_size'))
        FCGFModel = load_model(name)
        if FCGFModel is None:
            raise NotImplementedError(f'feature model {name!r}: only ResUNetBN2C is implemented')
        self.fcgf_model = FCGFModel(1, n_out, bn_momentum=_cfg_get(network_config, 'bn_momentum', 0.05),
                                    conv1_kernel_size=ks,
                                    normalize_feature=_cfg_get(network_config, 'normalize_feature'))
        # optional runtime keys: the MinkowskiEngine conventions the checkpoint was written under (model/me_conventions.py)
        me_conv = {'kernel_order': _cfg_get(config, 'me_kernel_order', 'first_axis_fastest'),
                   'transposed_mirrored': bool(_cfg_get(config, 'me_transposed_mirrored', False))}
        self.fcgf_model.me_conventions = dict(me_conv)
        # optional runtime key `share_weights_with`: another DeepGlobalRegistration of the same checkpoint on this device
        # (one object per HIP stream / library context): its device-resident weights are used, not a second copy
        shared = _cfg_get(config, 'share_weights_with')
        if shared is not None:
            self.fcgf_model.share_weights(shared.fcgf_model)
        else:
            self.fcgf_model.load_state_dict(state['state_dict'])
        self.fcgf_model = self.fcgf_model.to(self.device).eval()

        # inlier network (:118-131)
        num_feats = 6 if self.inlier_feature_type == 'coords' else 1
        InlierModel = load_model(_cfg_get(network_config, 'inlier_model'))
        if InlierModel is None:
            raise NotImplementedError('inlier model: only ResUNetBN2C is implemented')
        self.inlier_model = InlierModel(num_feats, 1, bn_momentum=_cfg_get(network_config, 'bn_momentum', 0.05),
                                        conv1_kernel_size=_cfg_get(network_config, 'inlier_conv1_kernel_size'),
                                        normalize_feature=False, D=6)
        self.inlier_model.me_conventions = dict(me_conv)
        if shared is not None:
            self.inlier_model.share_weights(shared.inlier_model)
        else:
            self.inlier_model.load_state_dict(state['state_dict_inlier'])
        self.inlier_model = self.inlier_model.to(self.device).eval()
        self.nn_max_n = _cfg_get(network_config, 'nn_max_n', 250)

    # ---- stage methods, same names and argument order as the reference -------------------------
    def preprocess(self, pcd, batch_index=0):
        """Stage 0 (:134-161).  Returns xyz f32 [N,3] (device), coords i32 [N,4] (device),
        feats ones [N,1]."""
        if hasattr(pcd, 'points') and not isinstance(pcd, np.ndarray):   # o3d.geometry.PointCloud
            xyz = np.array(pcd.points)
        elif isinstance(pcd, np.ndarray):
            xyz = pcd
        elif torch.is_tensor(pcd):
            xyz = pcd
        else:
            raise Exception('Unrecognized pcd type')
        xyz_sel, coords, _ = ops.voxelize(xyz, self.voxel_size, batch_index, self.device)
        feats = torch.ones(len(xyz_sel), 1, device=self.device)
        return xyz_sel, coords, feats

    def fcgf_feature_extraction(self, feats, coords):
        """Step 1 (:163-169)."""
        sinput = SparseTensor(feats, coordinates=coords, device=self.device)
        return self.fcgf_model(sinput).F

    def fcgf_feature_matching(self, feats0, feats1):
        """Step 2 (:171-183)."""
        nns = find_knn_gpu(feats0, feats1, nn_max_n=self.nn_max_n, knn=1, return_distance=False)
        corres_idx0 = torch.arange(len(nns), device=self.device).long()
        corres_idx1 = nns.long().reshape(-1)
        return corres_idx0, corres_idx1

    def inlier_feature_generation(self, xyz0, xyz1, coords0, coords1, fcgf_feats0, fcgf_feats1,
                                  corres_idx0, corres_idx1):
        """Step 3 (:185-208).  corres_idx0 must be arange(N0), as produced by step 2."""
        assert len(corres_idx0) == len(corres_idx1)
        feat_type = self.inlier_feature_type
        assert feat_type in ['ones', 'feats', 'coords']
        if feat_type == 'feats':
            raise TypeError("inlier_feature_type 'feats' is inconsistent with the network input width "
                            'in the reference (deep_global_registration.py:119) and is not supported')
        _, feat = ops.inlier_inputs(coords0, xyz0, coords1, xyz1, corres_idx1, feat_type)
        return feat

    def inlier_prediction(self, inlier_feats, coords):
        """Step 4 (:210-217)."""
        sinput = SparseTensor(inlier_feats, coordinates=coords, device=self.device)
        return self.inlier_model(sinput).F

    def safeguard_registration(self, pcd0, pcd1, idx0, idx1, feats0, feats1, distance_threshold,
                               num_iterations):
        """Safeguard (:219-236): RANSAC over the putative correspondences.  `pcd0` / `pcd1` are the
        voxelised xyz tensors (the reference wraps them into Open3D clouds).  Like the reference's call
        (RANSACConvergenceCriteria(4000000, num_iterations) with the second argument clamped to
        confidence 1.0), all 4 000 000 hypotheses (`ransac_max_iteration`) are evaluated; `num_iterations` is accepted
        and unused."""
        if self.safeguard_method != 'correspondence':
            # :235.  The reference's other branch, 'fcgf_feature_matching' (:31-46), calls the pre-0.10
            # `o3d.registration` namespace, which does not exist in the pinned open3d==0.17.0: dead code there.
            raise ValueError('Undefined')
        idx0 = torch.as_tensor(idx0, device=self.device).long()
        X = pcd0 if len(idx0) == len(pcd0) and bool((idx0 == torch.arange(len(idx0), device=self.device)).all()) \
            else ops.gather_rows3(pcd0, idx0)
        Y = ops.gather_rows3(pcd1, torch.as_tensor(idx1, device=self.device).long())
        T, h, count, rmse = ops.ransac_correspondence(X, Y, distance_threshold, self.ransac_max_iteration,
                                                      seed=self.ransac_seed)
        self.last_stats = {'ransac_hypothesis': h, 'ransac_inliers': count, 'ransac_rmse': rmse}
        return T

    # ---- extension points of register(): identities here.  The test harness (tests/helpers.py::HarnessDGR) overrides
    #      them to replace matches / logits AFTER the search / the inlier net ran (untrained synthetic weights give
    #      meaningless matches and confidences); nothing in the product does.
    def _post_matching(self, xyz0, xyz1, corres_idx1):
        return corres_idx1

    def _post_inlier_prediction(self, xyz0, xyz1, corres_idx1, logit):
        return logit

    # ---- main entry ------------------------------------------------------------------------------
    def register(self, xyz0, xyz1, inlier_thr=0.00):
        """Main algorithm (:238-324).  Returns a 4x4 float64 numpy transformation."""
        self.reg_timer.tic()
        xyz0, coords0, feats0 = self.preprocess(xyz0)
        xyz1, coords1, feats1 = self.preprocess(xyz1)

        self.feat_timer.tic()
        fcgf_feats0 = self.fcgf_feature_extraction(feats0, coords0)
        fcgf_feats1 = self.fcgf_feature_extraction(feats1, coords1)
        self.feat_timer.toc()

        corres_idx0, corres_idx1 = self.fcgf_feature_matching(fcgf_feats0, fcgf_feats1)
        corres_idx1 = self._post_matching(xyz0, xyz1, corres_idx1)
        self.last_corres_idx1 = corres_idx1 if self.keep_intermediates else None

        inlier_coords, inlier_feats = ops.inlier_inputs(coords0, xyz0, coords1, xyz1, corres_idx1,
                                                        self.inlier_feature_type)
        logit = self.inlier_prediction(inlier_feats.contiguous(), coords=inlier_coords)
        logit = self._post_inlier_prediction(xyz0, xyz1, corres_idx1, logit)
        self.last_logit = logit if self.keep_intermediates else None
        weights, wsum = ops.sigmoid_clip_sum(logit, self.clip_weight_thresh)

        wsum_threshold = max(200, len(weights) * 0.05)
        self.last_wsum = (float(wsum), float(wsum_threshold))     # host values the gate needs anyway; the reference prints them (:279-281)
        T = np.identity(4)
        safeguard = wsum < wsum_threshold
        if not safeguard:
            try:
                rot, trans, opt_output = GlobalRegistration(xyz0, ops.gather_rows3(xyz1, corres_idx1),
                                                            weights=weights, break_threshold_ratio=1e-4,
                                                            quantization_size=2 * self.voxel_size,
                                                            verbose=False)
                T[0:3, 0:3] = rot.detach().cpu().numpy()
                T[0:3, 3] = trans.detach().cpu().numpy()
                self.last_status, self.last_stats = 'ok', opt_output
            except _lib.DgrError as e:
                # reference (:295-300): `except RuntimeError` around the SVD, "Will directly go to Safeguard" -- there
                # T simply stays the identity (the SVD branch never reaches the `else` below).  Only the SVD failure
                # is that case here: a workspace or HIP error is an infrastructure failure and must not be counted
                # as a registration failure.
                if e.code != _lib.DGR_ESVD:
                    raise
                self.last_status = 'svd_failed'
        else:
            # Case 1 (:302-315): safeguard RANSAC on the putative correspondences
            T = self.safeguard_registration(xyz0, xyz1, corres_idx0, corres_idx1, feats0, feats1,
                                            2 * self.voxel_size, num_iterations=80000)
            self.last_status = 'safeguard'
        self.reg_timer.toc()
        if self.use_icp:                       # :317-322
            T, fitness, rmse, iters = ops.icp_point_to_point(xyz0, xyz1, self.voxel_size * 2, init=T)
            self.last_icp = {'fitness': fitness, 'inlier_rmse': rmse, 'iterations': iters}
        return T

    # ---- batched throughput path (no reference counterpart) ---------------
    def register_batch(self, pairs, forced_logits=None, skip_refinement=False, safeguard=False, icp=False):
        """Registers a list of (xyz0, xyz1) pairs with ONE sparse tensor per network (pairs are
        distinguished by the batch column, the layout of ME.utils.batched_coordinates).  Returns
        T [n,4,4] float64, status [n] (0 ok / 1 low confidence / 2 SVD failed), stats [n,4]."""
        x0, c0, x1, c1, off0, off1 = [], [], [], [], [0], [0]
        for p, (a, b) in enumerate(pairs):
            xa, ca, _ = self.preprocess(a, batch_index=p)
            xb, cb, _ = self.preprocess(b, batch_index=p)
            x0.append(xa); c0.append(ca); x1.append(xb); c1.append(cb)
            off0.append(off0[-1] + len(xa)); off1.append(off1[-1] + len(xb))
        return self.register_voxelized(torch.cat(c0), torch.cat(x0), off0, torch.cat(c1), torch.cat(x1), off1,
                                       forced_logits=forced_logits, skip_refinement=skip_refinement,
                                       safeguard=safeguard, icp=icp)

    def register_collated(self, input_dict, **kw):
        """Registers every pair of a collated batch in the reference's data-loader layout
        (`CollationFunctionFactory.collate_pair_fn`, dataloader/base_loader.py:40-98): `sinput0_C` /
        `sinput1_C` int [N,4] batched coordinates (batch column first, `ME.utils.batched_coordinates`),
        `pcd0` / `pcd1` sequences of per-pair xyz [Ni,3] aligned with those rows, `len_batch` [[N0,N1],...].
        Returns T [n,4,4] float64, status [n], stats [n,4]."""
        len_batch = [(int(a), int(b)) for a, b in input_dict['len_batch']]
        off0, off1 = [0], [0]
        for n0, n1 in len_batch:
            off0.append(off0[-1] + n0)
            off1.append(off1[-1] + n1)
        c0 = torch.as_tensor(input_dict['sinput0_C']).to(self.device).int()
        c1 = torch.as_tensor(input_dict['sinput1_C']).to(self.device).int()
        x0 = torch.cat([torch.as_tensor(np.asarray(x)).float() for x in input_dict['pcd0']]).to(self.device)
        x1 = torch.cat([torch.as_tensor(np.asarray(x)).float() for x in input_dict['pcd1']]).to(self.device)
        if len(c0) != off0[-1] or len(c1) != off1[-1] or len(x0) != off0[-1] or len(x1) != off1[-1]:
            raise ValueError('len_batch does not match the concatenated coordinates / points')
        for p in range(len(len_batch)):   # the batch column must be the pair index of the row block
            if off0[p + 1] > off0[p] and (int(c0[off0[p], 0]) != p or int(c0[off0[p + 1] - 1, 0]) != p):
                raise ValueError('sinput0_C is not in batched_coordinates order')
        return self.register_voxelized(c0, x0, off0, c1, x1, off1, **kw)

    def register_voxelized(self, coords0, xyz0, off0, coords1, xyz1, off1, forced_logits=None,
                           skip_refinement=False, override_idx1=None, safeguard=False, icp=False):
        """Fused batched path (one `dgr_register_batch`).  With `safeguard`, pairs that fail the confidence
        gate (status 1) are re-estimated by the RANSAC safeguard over their putative correspondences and get
        status 3 -- like `register()` and the reference, a pair whose SVD failed (status 2) keeps T = identity
        (:295-300 never reaches the safeguard branch); with `icp`, every pair is finally refined by
        point-to-point ICP -- the two Open3D steps of `register()` (:302-322), both inside the same library call
        (dgr_params.safeguard / use_icp)."""
        T, status, stats = ops.register_batch(
            self.fcgf_model._handle(), self.inlier_model._handle(), coords0, xyz0, off0, coords1, xyz1, off1,
            self.voxel_size, clip_weight_thresh=self.clip_weight_thresh,
            inlier_feature_type=self.inlier_feature_type, break_threshold_ratio=1e-4,
            skip_refinement=skip_refinement, forced_logit=forced_logits, override_idx1=override_idx1,
            safeguard=safeguard, use_icp=icp, ransac_hypotheses=self.ransac_max_iteration,
            ransac_seed=self.ransac_seed)
        T = T.astype(np.float64)   # (already float64 when the safeguard / ICP stages ran: their results at full width)
        return T, status, stats
