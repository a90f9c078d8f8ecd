"""`dgr_net_create_device`: a state dict that is already in HBM (torch CUDA tensors, e.g. views of the RCCL broadcast
buffer of a multi-GPU start) is folded / split / tiled into the conv kernels' operand layouts by HIP kernels.  The weight
sets must be the ones `dgr_net_create` builds on the host from the same values: both networks' outputs are compared BIT
FOR BIT (3-D FCGF net with conv1 k = 7 and k = 5 -- every 3-D layout incl. the value-grid conv1 --, 6-D inlier net -- the
wide-layer pieces, the quad-major conv1, the f32 tiles)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _forward_both(state, D, cin, cout, ks, normalize, coords, feats):
    from deepglobalregistration_amd import ops
    dev = torch.device('cuda:0')
    host = ops.NetHandle({k: np.asarray(v) for k, v in state.items()}, D, cin, cout, ks, normalize, dev)
    on_dev = ops.NetHandle({k: torch.as_tensor(np.asarray(v)).to(dev) for k, v in state.items()}, D, cin, cout, ks, normalize, dev)
    assert not host.created_on_device and on_dev.created_on_device
    assert host.param_bytes == on_dev.param_bytes
    a = host.forward(coords, feats).cpu().numpy()
    b = on_dev.forward(coords, feats).cpu().numpy()
    return a, b


@pytest.mark.parametrize('ks', [7, 5, 3])
def test_fcgf_weights_prepared_on_device_are_bit_identical(ks):
    from deepglobalregistration_amd import ops, synth
    ck = synth.synth_checkpoint(seed=11, voxel_size=0.05, feat_conv1_kernel_size=ks, with_inlier=False)
    xyz, _, _ = synth.synth_pair(5, n_raw=6000)
    _, coords, _ = ops.voxelize(xyz, 0.05, 0, torch.device('cuda:0'))
    feats = torch.ones(len(coords), 1, device='cuda:0')
    a, b = _forward_both(ck['state_dict'], 3, 1, 32, ks, True, coords, feats)
    assert np.isfinite(a).all() and np.abs(a).max() > 0
    assert np.array_equal(a, b), np.abs(a - b).max()


def test_inlier_weights_prepared_on_device_are_bit_identical():
    from deepglobalregistration_amd import ops, synth
    ck = synth.synth_checkpoint(seed=12, voxel_size=0.05, feat_conv1_kernel_size=3)
    xa, xb, Tg = synth.synth_pair(6, n_raw=4000)
    dev = torch.device('cuda:0')
    p0, c0, _ = ops.voxelize(xa, 0.05, 0, dev)
    p1, c1, _ = ops.voxelize(xb, 0.05, 0, dev)
    idx1 = torch.from_numpy(np.random.default_rng(0).integers(0, len(p1), len(p0))).to(dev)
    coords6, feats6 = ops.inlier_inputs(c0, p0, c1, p1, idx1, 'coords')
    a, b = _forward_both(ck['state_dict_inlier'], 6, 6, 1, 3, False, coords6, feats6)
    assert np.isfinite(a).all() and np.abs(a).max() > 0
    assert np.array_equal(a, b), np.abs(a - b).max()


def test_device_entry_point_refuses_host_pointers():
    """`dgr_net_create_device` dereferences its tensors in kernels: a host pointer must be an error, not a GPU fault."""
    import ctypes as C
    from deepglobalregistration_amd import _lib, ops, synth
    ck = synth.synth_checkpoint(seed=11, voxel_size=0.05, feat_conv1_kernel_size=3, with_inlier=False)
    keep, descs = [], []
    for name, t in ck['state_dict'].items():
        if name.endswith('num_batches_tracked'):
            continue
        a = np.ascontiguousarray(np.asarray(t), dtype=np.float32)
        keep.append(a)
        descs.append(_lib.WeightDesc(name.encode(), a.ctypes.data, a.size))
    arr = (_lib.WeightDesc * len(descs))(*descs)
    h = C.c_void_p()
    with pytest.raises(ValueError, match='not a device pointer'):
        _lib.check(_lib.load().dgr_net_create_device(ops.get_ctx(torch.device('cuda:0')), 3, 1, 32, 3, 1, arr, len(descs), C.byref(h)))
