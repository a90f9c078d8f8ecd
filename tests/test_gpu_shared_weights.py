"""One weight set per device for several contexts (dgr_net_share): a second `DeepGlobalRegistration` built with
`share_weights_with` runs in its own library context / HIP stream over the FIRST object's device-resident weights --
bitwise the same results, no second copy in HBM (the weight set reports two net objects), and the weights stay valid
when the object that loaded them is dropped first."""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
VOXEL = 0.05


def test_two_contexts_share_one_weight_set():
    from deepglobalregistration_amd import _lib, synth
    from deepglobalregistration_amd.core.deep_global_registration import DeepGlobalRegistration
    dev = torch.device('cuda:0')
    ck = synth.synth_checkpoint(seed=3, voxel_size=VOXEL, feat_conv1_kernel_size=5)
    x0, x1, _ = synth.synth_pair(3, n_raw=5000)
    a = DeepGlobalRegistration({'weights': ck, 'use_icp': False, 'keep_intermediates': True}, dev)
    Ta = a.register(x0, x1)
    la = a.last_logit.cpu().numpy().copy()
    ctx2 = _lib.new_ctx(dev)
    _lib.use_ctx(ctx2)
    try:
        with torch.cuda.stream(torch.cuda.Stream(dev)):
            b = DeepGlobalRegistration({'weights': ck, 'use_icp': False, 'keep_intermediates': True, 'share_weights_with': a}, dev)
            Tb = b.register(x0, x1)
            lb = b.last_logit.cpu().numpy().copy()
            torch.cuda.synchronize()
            assert b.inlier_model._handle().sharers == 2 and b.fcgf_model._handle().sharers == 2
            assert b.inlier_model._handle().param_bytes == a.inlier_model._handle().param_bytes > 5e8
            assert b.inlier_model._handle().handle.value != a.inlier_model._handle().handle.value   # own net object, same weights
            assert np.array_equal(la, lb) and np.array_equal(Ta, Tb)
            # the loader's handle goes away first: the sharer keeps the loader's models (and so the weights) alive
            del a
            gc.collect()
            Tc = b.register(x0, x1)
            assert np.array_equal(Tc, Tb)
    finally:
        _lib.use_ctx(None)
    with pytest.raises(ValueError):
        other = DeepGlobalRegistration({'weights': synth.synth_checkpoint(seed=3, voxel_size=VOXEL, feat_conv1_kernel_size=7),
                                        'use_icp': False}, dev)
        DeepGlobalRegistration({'weights': ck, 'use_icp': False, 'share_weights_with': other}, dev)
