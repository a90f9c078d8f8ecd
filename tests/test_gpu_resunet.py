"""GPU parity: ResUNetBN2C forward (3-D FCGF and 6-D inlier net) against the CPU oracle.
Tolerance: f32 conv stack, |err| <= 1e-4 * max|activation| per compared tensor (the HIP path
folds batch norm into the kernels; its sums run in the oracle's order and are bit-reproducible)."""
import numpy as np
import pytest
import torch

from helpers import random_cloud_coords, rel_err
from oracle import resunet as oresunet

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _check(D, cin, cout, ks, normalize, coords, feats, seed):
    from deepglobalregistration_amd import ops, synth
    sd = synth.synth_state_dict(D, cin, cout, ks, seed)
    net = ops.NetHandle(sd, D, cin, cout, ks, normalize)
    out = net.forward(torch.from_numpy(coords).cuda(), torch.from_numpy(feats).cuda()).cpu().numpy()
    ref, inter = oresunet.resunet_forward(sd, coords, feats, D, ks, normalize, return_intermediates=True)
    for name in ('s1', 's2', 's4', 's8', 's4_tr', 's2_tr', 's1_tr'):
        got = net.intermediate(name)
        assert got.shape == inter[name].shape, name
        assert rel_err(got, np.maximum(inter[name], 0)) < TOL, (name, rel_err(got, inter[name]))
    assert out.shape == ref.shape
    assert rel_err(out, ref) < TOL, rel_err(out, ref)
    return net, out, ref


@pytest.mark.parametrize('ks', [7, 5, 3])
def test_fcgf_forward_matches_oracle(ks):
    rng = np.random.default_rng(ks)
    coords = np.concatenate([random_cloud_coords(rng, 3000, 24, 3, batch=b) for b in (0, 1)])
    feats = np.ones((len(coords), 1), np.float32)
    net, out, ref = _check(3, 1, 32, ks, True, coords, feats, seed=ks)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)
    stats = net.layer_stats()
    assert len(stats) == 23 and stats[0]['K'] == ks ** 3 and stats[-1]['K'] == 1


@pytest.mark.parametrize('n', [1, 2, 7, 40, 300])
def test_fcgf_forward_on_tiny_clouds_with_negative_coordinates(n):
    """Edge cases of the parity-class row lists of the transposed convs (conv_up.hip; round 6): clouds so small that
    most of the eight classes are empty and every workgroup is partial, coordinates on both sides of zero (the class of a
    row is the parity of coordinate / stride: floor-division semantics), two clouds of different size in one tensor."""
    rng = np.random.default_rng(100 + n)
    a = random_cloud_coords(rng, n, 6, 3, batch=0, surface=False)
    a[:, 1:] -= 9                                   # all three coordinates negative
    b = random_cloud_coords(rng, max(1, n // 2), 5, 3, batch=1, surface=False)
    b[:, 1] -= 3                                    # mixed signs
    coords = np.concatenate([a, b]).astype(np.int32)
    feats = np.ones((len(coords), 1), np.float32)
    net, out, ref = _check(3, 1, 32, 5, True, coords, feats, seed=31)
    from deepglobalregistration_amd import ops
    ops.set_profiling('cuda', True)
    net.forward(torch.from_numpy(coords).cuda(), torch.from_numpy(feats).cuda())
    kinds = ops.conv_launch_kinds('cuda')
    ops.set_profiling('cuda', False)
    assert sum(k.startswith('sparse_conv_up_f16x2') for k in kinds) == 3, kinds


def test_fcgf_forward_general_input_features():
    rng = np.random.default_rng(42)
    coords = random_cloud_coords(rng, 2500, 16, 3)
    feats = rng.standard_normal((len(coords), 3)).astype(np.float32)
    _check(3, 3, 16, 3, False, coords, feats, seed=5)


@pytest.mark.parametrize('n_out', [16, 32, 64])
def test_unit_norm_output_features(n_out):
    """`normalize_feature` (model/resunet.py:643-647): fused into the epilogue of the `final` conv for <= 32 output
    channels (16: half of the 32-channel block is masked), a pass of its own above."""
    rng = np.random.default_rng(44)
    coords = np.concatenate([random_cloud_coords(rng, 1500, 14, 3, batch=b) for b in (0, 1)])
    feats = np.ones((len(coords), 1), np.float32)
    net, out, ref = _check(3, 1, n_out, 5, True, coords, feats, seed=9)
    np.testing.assert_allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-5)


def test_3d_net_on_the_rule_major_path():
    """A 3-D net with more than 8 input channels does not take the fused conv1 / output-stationary route: all its
    layers run the rule-major two-phase kernels over kmap_search<3> / kmap_fill maps (conv.hip, kmap.hip)."""
    from deepglobalregistration_amd import ops
    rng = np.random.default_rng(43)
    coords = np.concatenate([random_cloud_coords(rng, 2000, 16, 3, batch=b) for b in (0, 1)])
    feats = rng.standard_normal((len(coords), 12)).astype(np.float32)
    _check(3, 12, 16, 3, False, coords, feats, seed=6)
    ops.set_profiling('cuda', True)
    from deepglobalregistration_amd import synth
    net = ops.NetHandle(synth.synth_state_dict(3, 12, 16, 3, 6), 3, 12, 16, 3, False)
    net.forward(torch.from_numpy(coords).cuda(), torch.from_numpy(feats).cuda())
    kinds = ops.conv_launch_kinds('cuda')
    ops.set_profiling('cuda', False)
    assert not any('sparse_conv_os' in k or 'conv1_grid' in k for k in kinds) and any('sparse_conv_mfma_v2' in k for k in kinds)


def test_inlier_net_6d_forward_matches_oracle():
    rng = np.random.default_rng(7)
    # 6-D rows built like the pipeline does: unique 3-D voxel + a second 3-D voxel
    c0 = random_cloud_coords(rng, 1600, 12, 3)
    c1 = c0[:, 1:] + rng.integers(-2, 3, (len(c0), 3)).astype(np.int32)
    coords = np.concatenate([c0, c1], axis=1).astype(np.int32)
    feats = np.cos(rng.uniform(-3, 3, (len(coords), 6))).astype(np.float32)
    _check(6, 6, 1, 3, False, coords, feats, seed=11)


def test_row_permutation_equivariance_full_size():
    """Size-independent property at the BASELINE size (50k-pt pair @5cm): permuting the input rows
    permutes the output rows (row alignment)."""
    from deepglobalregistration_amd import ops, synth
    xyz0, _, _ = synth.synth_pair(0, n_raw=50000)
    p, c, _ = ops.voxelize(xyz0, 0.05)
    sd = synth.synth_state_dict(3, 1, 32, 7, 0)
    net = ops.NetHandle(sd, 3, 1, 32, 7, True)
    ones = torch.ones(len(c), 1, device='cuda')
    F = net.forward(c, ones)
    perm = torch.randperm(len(c), device='cuda')
    Fp = net.forward(c[perm].contiguous(), ones)
    assert rel_err(Fp.cpu().numpy(), F[perm].cpu().numpy()) < TOL
    np.testing.assert_allclose(F.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)


def test_forward_rejects_bad_shapes():
    from deepglobalregistration_amd import ops, synth
    sd = synth.synth_state_dict(3, 1, 32, 3, 0)
    net = ops.NetHandle(sd, 3, 1, 32, 3, True)
    with pytest.raises(ValueError):
        net.forward(torch.zeros((4, 7), dtype=torch.int32).cuda(), torch.ones(4, 1).cuda())
    with pytest.raises(ValueError):
        net.forward(torch.zeros((0, 4), dtype=torch.int32).cuda(), torch.ones(0, 1).cuda())
    dup = torch.tensor([[0, 1, 1, 1], [0, 1, 1, 1]], dtype=torch.int32).cuda()
    with pytest.raises(ValueError):
        net.forward(dup, torch.ones(2, 1).cuda())
    bad = dict(sd)
    del bad['norm1.bn.weight']
    with pytest.raises(ValueError):
        ops.NetHandle(bad, 3, 1, 32, 3, True)


def test_forward_is_bit_reproducible():
    """The conv stack has no floating-point atomics (per-pair product rows + fixed-order segmented
    reduction), so two forwards of the same input are bitwise identical (the
    reference's CUDA path is not, because MinkowskiEngine scatters with atomicAdd)."""
    from deepglobalregistration_amd import ops, synth
    rng = np.random.default_rng(1)
    coords = torch.from_numpy(random_cloud_coords(rng, 6000, 30, 3)).cuda()
    ones = torch.ones(len(coords), 1, device='cuda')
    net = ops.NetHandle(synth.synth_state_dict(3, 1, 32, 7, 2), 3, 1, 32, 7, True)
    a = net.forward(coords, ones)
    b = net.forward(coords, ones)
    assert torch.equal(a, b)


@pytest.mark.parametrize('case', ['batch_index_70', 'far_apart', 'cin3_k5'])
def test_fused_conv1_fallback_and_variants(case):
    """FCGF conv1 is fused with its neighbour search over a dense voxel grid; batch indices >= 64 or a
    bounding box beyond the cell budget fall back (on the device, no host sync) to the hash-probe kernel.
    All variants must agree with the oracle."""
    rng = np.random.default_rng(13)
    a = random_cloud_coords(rng, 1500, 16, 3, batch=0)
    b = random_cloud_coords(rng, 1500, 16, 3, batch=1)
    cin, ks = 1, 7
    if case == 'batch_index_70':
        b[:, 0] = 70
    elif case == 'far_apart':            # 2 points 3e6 voxels apart: bbox volume >> 64 M cells
        b[:, 1:] += np.array([0, 3000000, 0], np.int32)
        b[:, 0] = 0
    else:
        cin, ks = 3, 5
    coords = np.concatenate([a, b])
    feats = np.ones((len(coords), 1), np.float32) if cin == 1 else rng.standard_normal((len(coords), cin)).astype(np.float32)
    _check(3, cin, 32, ks, True, coords, feats, seed=3)
