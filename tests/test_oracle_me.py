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


# ---- randomized properties of the restatement (hypothesis): internal consistency of the conventions --------
from hypothesis import given, settings, strategies as st


def _random_coords(seed, n, D, extent, batches):
    rng = np.random.default_rng(seed)
    c = np.concatenate([rng.integers(0, batches, (n, 1)), rng.integers(-extent, extent, (n, D))], axis=1).astype(np.int32)
    _, first = np.unique(c, axis=0, return_index=True)
    return c[np.sort(first)]


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), D=st.sampled_from([3, 6]), n=st.integers(1, 300), ts=st.sampled_from([1, 2, 4]))
def test_same_stride_maps_are_symmetric_and_contain_the_identity(seed, D, n, ts):
    c = _random_coords(seed, n, D, 4 if D == 6 else 8, 2)
    c[:, 1:] *= ts
    k, i, o = me.kernel_map(c, c, D, 3, ts)
    K = 3 ** D
    pairs = set(zip(k.tolist(), i.tolist(), o.tolist()))
    assert len(pairs) == len(k)                                     # no duplicates
    assert pairs == {(K - 1 - kk, oo, ii) for kk, ii, oo in pairs}  # (o,k)->i  <=>  (i,K-1-k)->o
    centre = [(kk, ii, oo) for kk, ii, oo in pairs if kk == K // 2]
    assert sorted(ii for _, ii, _ in centre) == list(range(len(c))) and all(ii == oo for _, ii, oo in centre)
    # the defining relation: in = out + delta_k * ts
    off = me.kernel_offsets(D, 3)
    np.testing.assert_array_equal(c[i][:, 1:], c[o][:, 1:] + off[k] * ts)
    np.testing.assert_array_equal(c[i][:, 0], c[o][:, 0])


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 10 ** 6), D=st.sampled_from([3, 6]), n=st.integers(1, 300))
def test_stride_maps_and_their_transpose(seed, D, n):
    fine = _random_coords(seed, n, D, 6, 2)
    coarse = me.stride_coords(fine, 2)
    assert (coarse[:, 1:] % 2 == 0).all() and len(np.unique(coarse, axis=0)) == len(coarse)
    np.testing.assert_array_equal(me.stride_coords(coarse, 2), coarse)          # idempotent
    # every fine row lies in exactly one coarse cell, and that cell exists
    cell = fine.copy(); cell[:, 1:] = (fine[:, 1:] // 2) * 2
    assert {tuple(r) for r in cell.tolist()} == {tuple(r) for r in coarse.tolist()}
    k, i, o = me.kernel_map(fine, coarse, D, 3, 1)                              # strided conv: fine -> coarse
    kt, it, ot = me.transposed_kernel_map(coarse, fine, D, 3, 1)                # transposed conv: coarse -> fine
    assert set(zip(k.tolist(), i.tolist(), o.tolist())) == set(zip(kt.tolist(), ot.tolist(), it.tolist()))
    # each fine row reaches its own cell through the offset 0 or +1 per axis (floor semantics), so every fine
    # row appears at least once as an input of the strided conv
    assert set(i.tolist()) == set(range(len(fine)))


def test_transposed_map_offsets_follow_the_parity_class_of_the_fine_row():
    """What conv_up.hip (round 6) builds on, held to the oracle's transposed kernel maps: a fine row f
    pairs with a coarse row c under offset delta only where c = f - delta ts lies on the coarse lattice, i.e. component d of
    delta is 0 exactly where coordinate d of f is an EVEN multiple of the fine stride and +-1 where it is odd -- so the
    set of offsets a fine row can use is a function of its parity class alone (2^(odd dims) of the 27), the classes
    partition the rows, and no pair of the oracle's map falls outside its class's offsets.  Negative coordinates and
    two tensor strides included."""
    rng = np.random.default_rng(7)
    for ts in (1, 2):
        fine = np.unique(rng.integers(-9, 9, (400, 3)) * ts, axis=0)
        fine = np.concatenate([np.zeros((len(fine), 1), np.int64), fine], axis=1).astype(np.int32)
        coarse = me.stride_coords(fine, 2 * ts)
        k, cin, fout = me.transposed_kernel_map(coarse, fine, 3, 3, ts)
        assert len(k) > len(fine)                       # every fine row has its parent, most have more
        offs = me.kernel_offsets(3, 3)
        odd = (fine[fout, 1:].astype(np.int64) // ts) & 1           # [P, 3] parity of the OUTPUT (fine) row of every pair
        np.testing.assert_array_equal(offs[k] != 0, odd == 1)       # delta_d != 0  <=>  coordinate d odd
        # the coarse row really is f - delta ts, and it is on the coarse lattice
        np.testing.assert_array_equal(coarse[cin, 1:], fine[fout, 1:] - offs[k] * ts)
        assert (coarse[:, 1:] % (2 * ts) == 0).all()
        # per class: the offsets used are a subset of the class's 2^(odd dims) offsets, and pairs per row <= that number
        cls = odd[:, 0] | (odd[:, 1] << 1) | (odd[:, 2] << 2)
        for c in range(8):
            allowed = [j for j in range(27) if all((offs[j, d] != 0) == bool((c >> d) & 1) for d in range(3))]
            assert len(allowed) == 1 << bin(c).count('1')
            assert set(k[cls == c].tolist()) <= set(allowed)
        rows, counts = np.unique(fout, return_counts=True)
        row_cls = ((fine[rows, 1].astype(np.int64) // ts) & 1) | ((((fine[rows, 2].astype(np.int64) // ts) & 1)) << 1) | ((((fine[rows, 3].astype(np.int64) // ts) & 1)) << 2)
        assert (counts <= (1 << np.array([bin(int(c)).count('1') for c in row_cls]))).all()
