"""Known-answer tests pinning the restated MinkowskiEngine conventions (oracle side).
CPU only.  These are hand-computable cases."""
import numpy as np
import torch

from oracle import me_semantics as me
from oracle import resunet


def test_sparse_quantize_first_occurrence():
    xyz = np.array([[0.2, 0.2, 0.2], [1.7, 0.0, 0.0], [0.9, 0.9, 0.9], [-0.1, 0.0, 0.0],
                    [1.2, 0.3, 0.3], [-0.9, 0.5, 0.1]])
    c, sel = me.sparse_quantize(xyz, return_index=True)
    np.testing.assert_array_equal(sel, [0, 1, 3])          # rows 2,4,5 are repeats
    np.testing.assert_array_equal(c, [[0, 0, 0], [1, 0, 0], [-1, 0, 0]])
    assert c.dtype == np.int32


def test_batched_coordinates():
    b = me.batched_coordinates([np.array([[1, 2, 3]]), np.array([[4, 5, 6], [7, 8, 9]])])
    np.testing.assert_array_equal(b, [[0, 1, 2, 3], [1, 4, 5, 6], [1, 7, 8, 9]])
    assert b.dtype == np.int32


def test_kernel_offsets_first_dim_fastest():
    o = me.kernel_offsets(3, 3)
    assert o.shape == (27, 3)
    np.testing.assert_array_equal(o[0], [-1, -1, -1])
    np.testing.assert_array_equal(o[1], [0, -1, -1])       # x fastest
    np.testing.assert_array_equal(o[3], [-1, 0, -1])
    np.testing.assert_array_equal(o[13], [0, 0, 0])
    o6 = me.kernel_offsets(6, 3)
    assert o6.shape == (729, 6) and (o6[364] == 0).all()
    np.testing.assert_array_equal(o6[1], [0, -1, -1, -1, -1, -1])


def test_single_voxel_conv_uses_centre_weight():
    coords = np.array([[0, 5, -3, 2]], np.int32)
    k, i, o = me.kernel_map(coords, coords, 3, 3, 1)
    np.testing.assert_array_equal(k, [13])
    W = torch.arange(27 * 2 * 3, dtype=torch.float32).reshape(27, 2, 3)
    y = resunet.sparse_conv(torch.tensor([[1.0, 2.0]]), (k, i, o), W, 1)
    np.testing.assert_allclose(y.numpy(), (torch.tensor([[1.0, 2.0]]) @ W[13]).numpy())


def test_two_voxels_along_x_pin_offset_order():
    # out at x=0 sees the in voxel at x=1 through offset (+1,0,0) -> k = 13 + 1 = 14
    coords = np.array([[0, 0, 0, 0], [0, 1, 0, 0]], np.int32)
    k, i, o = me.kernel_map(coords, coords, 3, 3, 1)
    trip = set(zip(k.tolist(), i.tolist(), o.tolist()))
    assert trip == {(13, 0, 0), (13, 1, 1), (14, 1, 0), (12, 0, 1)}
    # along z the neighbour sits 9 offsets away
    coords = np.array([[0, 0, 0, 0], [0, 0, 0, 1]], np.int32)
    k, i, o = me.kernel_map(coords, coords, 3, 3, 1)
    assert (22, 1, 0) in set(zip(k.tolist(), i.tolist(), o.tolist()))


def test_stride2_floor_semantics_with_negative_coordinates():
    coords = np.array([[0, -1, 0, 0], [0, -2, 0, 0], [0, 1, 0, 0], [0, 0, 0, 0], [1, -1, 0, 0]], np.int32)
    c2 = me.stride_coords(coords, 2)
    np.testing.assert_array_equal(c2, [[0, -2, 0, 0], [0, 0, 0, 0], [1, -2, 0, 0]])
    # strided conv: in == out + delta * ts_in, delta in {-1,0,1}
    k, i, o = me.kernel_map(coords, c2, 3, 3, 1)
    trip = set(zip(k.tolist(), i.tolist(), o.tolist()))
    # out (0,-2,0,0): ins at x=-3(no) -2(yes,row1) -1(yes,row0)
    assert (13, 1, 0) in trip and (14, 0, 0) in trip
    # out (0,0,0,0): ins at x=-1 (row0, k=12), 0 (row3, k=13), 1 (row2, k=14)
    assert {(12, 0, 1), (13, 3, 1), (14, 2, 1)} <= trip
    # batch 1 never mixes with batch 0
    assert all(not (oo == 2 and ii != 4) for _, ii, oo in trip)


def test_transposed_map_is_swapped_forward_map():
    rng = np.random.default_rng(0)
    pts = np.unique(rng.integers(-6, 6, (60, 3)), axis=0)
    fine = np.concatenate([np.zeros((len(pts), 1), int), pts], 1).astype(np.int32)
    coarse = me.stride_coords(fine, 2)
    kf, i_f, o_c = me.kernel_map(fine, coarse, 3, 3, 1)
    kt, i_c, o_f = me.transposed_kernel_map(coarse, fine, 3, 3, 1)
    assert set(zip(kf.tolist(), i_f.tolist(), o_c.tolist())) == \
        set(zip(kt.tolist(), o_f.tolist(), i_c.tolist()))
    # round trip with one-hot kernels: up(down(x)) with centre-only weights
    W = torch.zeros(27, 1, 1)
    W[13] = 1
    x = torch.ones(len(fine), 1)
    down = resunet.sparse_conv(x, (kf, i_f, o_c), W, len(coarse))
    up = resunet.sparse_conv(down, (kt, i_c, o_f), W, len(fine))
    # only fine voxels that sit exactly on a coarse coordinate receive a value
    on_grid = np.all(fine[:, 1:] % 2 == 0, axis=1)
    assert np.all((up.numpy().reshape(-1) > 0) == on_grid)


def test_resunet_forward_shapes_and_row_alignment():
    from deepglobalregistration_amd import synth
    rng = np.random.default_rng(3)
    pts = np.unique(rng.integers(0, 12, (300, 3)), axis=0)
    coords = np.concatenate([np.zeros((len(pts), 1), int), pts], 1).astype(np.int32)
    sd = synth.synth_state_dict(3, 1, 32, 5, seed=7)
    F = resunet.resunet_forward(sd, coords, np.ones((len(pts), 1), np.float32), 3, 5, True)
    assert F.shape == (len(pts), 32)
    np.testing.assert_allclose(np.linalg.norm(F, axis=1), 1.0, atol=1e-5)
    # permuting the input rows permutes the output rows (row alignment)
    perm = rng.permutation(len(pts))
    Fp = resunet.resunet_forward(sd, coords[perm], np.ones((len(pts), 1), np.float32), 3, 5, True)
    np.testing.assert_allclose(Fp, F[perm], atol=2e-5)
